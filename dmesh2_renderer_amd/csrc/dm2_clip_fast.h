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
//     reference's Jacobian of the crossing (t, 1 - t, gt0, gt1 as written at aa.h:276-294);
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
constexpr float FAST_TIE_ISZERO = 1.5e-3f;          // with an "iszero" edge in the face: both ends of such an edge lie within 1e-3 of a line it straddles

template <int TI>
__device__ __forceinline__ void fast_edge(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                          float delta_v, float delta_c, float* g, bool& tie) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    // crossing parameters with the pixel lines x = pxmin (D), x = pxmax (B), y = pymin (A), y = pymax (C): aa.h:250
    const float dxl = pxmin - p0x, dxh = pxmax - p0x, dyl = pymin - p0y, dyh = pymax - p0y;
    const float tD = dxl * rx, tB = dxh * rx, tA = dyl * ry, tC = dyh * ry;
    const bool cx = tD < tB, cy = tA < tC;
    const float xlo = fminf(tD, tB), xhi = fmaxf(tD, tB), ylo = fminf(tA, tC), yhi = fmaxf(tA, tC);
    const float ts = fmaxf(fmaxf(xlo, ylo), 0.0f), te = fminf(fminf(xhi, yhi), 1.0f);
    const bool on = te > ts;                                          // the edge has a piece inside the pixel
    const bool sV = !(ts > 0.0f), eV = !(te < 1.0f);                  // that piece starts at p0 / ends at p1
    const bool sX = (xlo > ylo) && !sV, eX = (xhi < yhi) && !eV;      // ... at a crossing of an x = const line (else y = const)
    const float wxlo = cx ? pxmin : pxmax, wxhi = cx ? pxmax : pxmin;
    const float wylo = cy ? pymin : pymax, wyhi = cy ? pymax : pymin;
    // corner coordinates as the reference computes them (aa.h:251-252; an end point: the vertex itself)
    const float Sx = sX ? wxlo : p0x + ts * ex, Sy = (!sX && !sV) ? wylo : p0y + ts * ey;
    const float Ex = eV ? p1x : (eX ? wxhi : p0x + te * ex), Ey = eV ? p1y : (eX ? p0y + te * ey : wyhi);
    const float Dx = Ex - Sx, Dy = Ey - Sy;
    const float hx = on ? 0.5f * Dy : 0.0f, hy = on ? -(0.5f * Dx) : 0.0f;     // 1/2 (D.y, -D.x); 0: nothing below contributes
    // end points inside the pixel: identity Jacobian
    g[2 * TI] += sV ? hx : 0.0f; g[2 * TI + 1] += sV ? hy : 0.0f;
    g[2 * TJ] += eV ? hx : 0.0f; g[2 * TJ + 1] += eV ? hy : 0.0f;
    // crossings: computed coordinate m = p0o + t * eo on the line a = w; dm/dp0 = (1 - t [o], gt0 * eo [a]), dm/dp1 = (t, gt1 * eo)
    // (a corner that is not a crossing, or an edge with no piece inside, contributes through operands set to 0: the
    // reciprocal of an axis-parallel edge is infinite and its crossing parameters with it, and 0 * inf must not reach g)
    auto crossing = [&](bool isX, bool isV, float t_, float wx, float wy) {
        const bool c = on && !isV;
        const float w = isX ? wx : wy, p0a = isX ? p0x : p0y, p1a = isX ? p1x : p1y;
        const float ra = c ? (isX ? rx : ry) : 0.0f, t = c ? t_ : 0.0f;
        const float eo = isX ? ey : ex;
        const float gm = c ? (isX ? hy : hx) : 0.0f;
        const float gt0 = (w - p1a) * ra * ra, gt1 = (-w + p0a) * ra * ra;      // aa.h:276-294
        const float omt = 1.0f - t;
        const float a_o = omt * gm, a_a = (gt0 * eo) * gm, b_o = t * gm, b_a = (gt1 * eo) * gm;
        g[2 * TI] += isX ? a_a : a_o; g[2 * TI + 1] += isX ? a_o : a_a;
        g[2 * TJ] += isX ? b_a : b_o; g[2 * TJ + 1] += isX ? b_o : b_a;
    };
    crossing(sX, sV, ts, wxlo, wylo);
    crossing(eX, eV, te, wxhi, wyhi);
    // ties: the corner p0 within delta_v of one of the four pixel lines; the edge's line within ~delta_c of a pixel corner
    // (there the crossings with the two lines through that corner have the same parameter; the distance along the edge)
    const float m2 = fminf(fminf(fabsf(dxl), fabsf(dxh)), fminf(fabsf(dyl), fabsf(dyh)));
    const float m3 = fminf(fminf(fabsf(tA - tB), fabsf(tB - tC)), fminf(fabsf(tC - tD), fabsf(tD - tA)));
    const float emax = fmaxf(fabsf(ex), fabsf(ey));
    tie = tie || !(m2 >= delta_v) || !(m3 * emax >= delta_c);         // (NaN -- 0 * inf of a degenerate edge -- counts as a tie)
}

// g: [3][2] row-major, d(area)/d(aa_face_verts) of the pair.  tie: the pair needs the exact clipper instead (g is then
// meaningless and must not be used).
__device__ __forceinline__ void fast_area_grad(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax, float* g, bool& tie) {
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    const float delta_c = fmaxf(pxmax, pymax) * FAST_TIE_REL;
    const float delta_v = (f.zmask & 0x3Fu) ? fmaxf(delta_c, FAST_TIE_ISZERO) : delta_c;
    tie = false;
    fast_edge<0>(f, pxmin, pxmax, pymin, pymax, delta_v, delta_c, g, tie);
    fast_edge<1>(f, pxmin, pxmax, pymin, pymax, delta_v, delta_c, g, tie);
    fast_edge<2>(f, pxmin, pxmax, pymin, pymax, delta_v, delta_c, g, tie);
    // a non-finite entry (a degenerate face) never passes for a result
    tie = tie || !(fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2]) + fabsf(g[3]) + fabsf(g[4]) + fabsf(g[5]) < 3.0e38f);
}

}  // namespace dm2
