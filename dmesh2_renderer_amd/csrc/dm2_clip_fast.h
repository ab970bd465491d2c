// dm2_clip_fast.h -- d(overlap area)/d(triangle corners) of a CCW triangle and a unit pixel WITHOUT building the
// polygon, for (pixel, face) pairs the forward has blended (dm2_backward_fast.hip).
//
// The reference differentiates its clipped polygon corner by corner (aa.h:276-294, :415-433).  Summed over the fan
// those partials telescope to one shoelace weight per corner (dm2_clip_seg.h), and the weights of the corners of ONE
// triangle edge only involve that edge's own in-pixel segment [S, E]:
//
//   * the part of triangle edge p0 -> p1 inside the pixel is t in [ts, te], ts = max(0, t-range of the x slab, of the
//     y slab), te = min(1, ...), from the reference's own crossing parameters t = (w - p0a) * ra (aa.h:250);
//   * S / E are the polygon corners at ts / te -- a crossing (coordinates as the reference computes them, aa.h:251-252:
//     the pixel line's constant and p0o + t * eo) or the end point itself -- and D = E - S;
//   * a crossing on a y = const line carries the weight 1/2 D.y on its computed x, one on an x = const line -1/2 D.x on
//     its computed y (its polygon neighbour on the far side lies on the same pixel line), pushed through the
//     reference's Jacobian of the crossing (aa.h:276-294: t, 1 - t and dt/dp = -(1 - t) / e, -t / e);
//   * an end point inside the pixel carries 1/2 (D.y, -D.x) from each of its two edges.
//
// Same polynomial in the same corner coordinates as the reference's, hence the same fp32 rounding of the corner
// coordinates (which dominates: 1e-4 of an entry at 1080p image coordinates) -- measured 2e-7 of max(1, largest
// entry) against the oracle on pairs in general position.  It is NOT the reference's result where the reference's
// polygon is not the geometric intersection: a triangle corner within rounding of a pixel boundary line (its closed
// interval tests and the end point tests then disagree about one crossing), an edge that passes within rounding of a
// pixel corner (the corner classification, aa.h:103-149, and the crossing validity tests disagree), an "iszero" edge
// (|e| < 1e-3, crossings with the parallel pixel lines are ignored, pyrenderer.py:14) that straddles such a line.
// Those pairs are recognised by distance tests with a margin of 16-32 ulp of the image coordinate (`tie`), take no
// gradient here and are handed to the exact segment clipper (dm2_clip_seg.h) by the caller: 1-5 % of the pairs of a
// 1080p frame.  tests/test_gpu_clippers.py (variant 4) holds every pair that is NOT flagged to the oracle's Jacobian
// on the reference's vectors, on random pairs and on the exact-tie stress sets.
#pragma once
#include "dm2_device_math.h"

namespace dm2 {

// margin of the tie tests: 2^-19 of the larger pixel coordinate = 16..32 ulp of an image coordinate there
constexpr float FAST_TIE_REL = 1.0f / 524288.0f;
constexpr float FAST_TIE_ISZERO = 1.5e-3f;          // both ends of an "iszero" edge lie within 1e-3 of a line it straddles

template <int TI, class Face>      // Face: anything with v[6], e[6], r[6], zmask (AAFace, or the backward's shorter LDS record)
__device__ __forceinline__ void fast_edge(const Face& f, float pxmin, float pxmax, float pymin, float pymax,
                                          float delta, float* g, bool& tie) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    // crossing parameters with the pixel lines x = pxmin (D), x = pxmax (B), y = pymin (A), y = pymax (C): aa.h:250
    const float dxl = pxmin - p0x, dxh = pxmax - p0x, dyl = pymin - p0y, dyh = pymax - p0y;
    const float tD = dxl * rx, tB = dxh * rx, tA = dyl * ry, tC = dyh * ry;
    const bool cx = tD < tB, cy = tA < tC;
    const float xlo = fminf(tD, tB), xhi = fmaxf(tD, tB), ylo = fminf(tA, tC), yhi = fmaxf(tA, tC);
    const float ts_ = fmaxf(fmaxf(xlo, ylo), 0.0f), te_ = fminf(fminf(xhi, yhi), 1.0f);
    const bool on = te_ > ts_;                                        // the edge has a piece inside the pixel
    // (an edge with no piece inside contributes through operands set to 0: the reciprocal of an axis-parallel edge is
    // infinite and its crossing parameters with it, and 0 * inf must not reach g)
    const float ts = on ? ts_ : 0.0f, te = on ? te_ : 1.0f;
    const bool sV = !(ts > 0.0f), eV = !(te < 1.0f);                  // that piece starts at p0 / ends at p1
    const bool sX = (xlo > ylo) && !sV, eX = (xhi < yhi) && !eV;      // ... at a crossing of an x = const line (else y = const)
    const float wxlo = cx ? pxmin : pxmax, wxhi = cx ? pxmax : pxmin;
    const float wylo = cy ? pymin : pymax, wyhi = cy ? pymax : pymin;
    // corner coordinates as the reference computes them (aa.h:251-252; an end point: the vertex itself)
    const float Sx = sX ? wxlo : p0x + ts * ex, Sy = (!sX && !sV) ? wylo : p0y + ts * ey;
    const float Ex = eV ? p1x : (eX ? wxhi : p0x + te * ex), Ey = eV ? p1y : (eX ? p0y + te * ey : wyhi);
    const float Dx = Ex - Sx, Dy = Ey - Sy;
    const float hx = on ? 0.5f * Dy : 0.0f, hy = on ? -(0.5f * Dx) : 0.0f;     // 1/2 (D.y, -D.x)
    // A corner at parameter t gives p0 (1 - t) V and p1 t V, with V = 1/2 (D.y, -D.x) for an end point (t = 0 or 1: identity
    // Jacobian), V = 1/2 D.y (1, -e.x / e.y) for a crossing of a y = const line (its x = p0x + t e.x moves, weight 1/2 D.y;
    // dt/dp0y = -(1 - t) / e.y, dt/dp1y = -t / e.y: aa.h:276-294 up to the rounding of one factor), V = -1/2 D.x (-e.y / e.x, 1)
    // for a crossing of an x = const line.
    // (kx, ky may be infinite for an axis-parallel edge: such an edge has no crossing with the lines parallel to it, so the
    // vectors built from them are never the ones selected below)
    const float kx = ex * ry, ky = ey * rx;
    const float vyy = -(kx * hx), vxx = -(ky * hy);
    const float Vsx = sV ? hx : (sX ? vxx : hx), Vsy = sV ? hy : (sX ? hy : vyy);
    const float Vex = eV ? hx : (eX ? vxx : hx), Vey = eV ? hy : (eX ? hy : vyy);
    const float oms = 1.0f - ts, ome = 1.0f - te;
    g[2 * TI] += oms * Vsx + ome * Vex; g[2 * TI + 1] += oms * Vsy + ome * Vey;
    g[2 * TJ] += ts * Vsx + te * Vex; g[2 * TJ + 1] += ts * Vsy + te * Vey;
    // ---- ties (see the header).  (a) the corner p0 within delta of a pixel line while inside the other slab widened by delta
    const float ax = fminf(fabsf(dxl), fabsf(dxh)), ay = fminf(fabsf(dyl), fabsf(dyh));
    const bool inx = (dxl <= delta) && (dxh >= -delta), iny = (dyl <= delta) && (dyh >= -delta);
    tie = tie || (!(ax >= delta) && iny) || (!(ay >= delta) && inx);
    // (b) an "iszero" edge (|e| < 1e-3 on an axis: the reference ignores its crossings with the pixel lines of that axis,
    // pyrenderer.py:14) one of whose ends is within 1.5e-3 of such a line: it may straddle it
    const uint32_t zx = (f.zmask >> (2 * TI)) & 1u, zy = (f.zmask >> (2 * TI + 1)) & 1u;
    if (zx | zy) {                                                     // (rare: one face in a few hundred)
        const float bx = fminf(fabsf(pxmin - p1x), fabsf(pxmax - p1x)), by = fminf(fabsf(pymin - p1y), fabsf(pymax - p1y));
        tie = tie || (zx && !(fminf(ax, bx) >= FAST_TIE_ISZERO)) || (zy && !(fminf(ay, by) >= FAST_TIE_ISZERO));
    }
    // (c) the edge's line within ~delta of a pixel corner: the crossings with the two pixel lines through that corner then have
    // (nearly) the same parameter.  (Whether that corner lies on the edge at all is not looked at: one pair in a thousand more.)
    const float emax = fmaxf(fabsf(ex), fabsf(ey));
    const float m3 = fminf(fminf(fabsf(tA - tB), fabsf(tB - tC)), fminf(fabsf(tC - tD), fabsf(tD - tA)));
    tie = tie || !(m3 * emax >= delta);
}

// g: [3][2] row-major, d(area)/d(aa_face_verts) of the pair.  tie: the pair needs the exact clipper instead (g is then
// meaningless and must not be used).
template <class Face>
__device__ __forceinline__ void fast_area_grad(const Face& f, float pxmin, float pxmax, float pymin, float pymax, float* g, bool& tie) {
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    const float delta = fmaxf(pxmax, pymax) * FAST_TIE_REL;
    tie = false;
    fast_edge<0>(f, pxmin, pxmax, pymin, pymax, delta, g, tie);
    fast_edge<1>(f, pxmin, pxmax, pymin, pymax, delta, g, tie);
    fast_edge<2>(f, pxmin, pxmax, pymin, pymax, delta, g, tie);
    // a non-finite entry (a degenerate face) never passes for a result
    tie = tie || !(fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2]) + fabsf(g[3]) + fabsf(g[4]) + fabsf(g[5]) < 3.0e38f);
}

}  // namespace dm2
