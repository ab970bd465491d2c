// dm2_clip_seg.h -- overlap area of a CCW triangle and a unit pixel AND its Jacobian w.r.t. the triangle's
// corners, for (pixel, face) pairs the FORWARD has already accepted (backward pass, gfx950).
//
// The reference builds the clipped polygon in a vertex table, fan-triangulates it and differentiates every
// fan triangle through the Jacobians of its three corners (aa.h:151-441).  The backward kernels only meet
// pairs whose forward clip returned "no error, area > 0" (the forward's blend masks say so), which makes a
// much shorter formulation possible -- no table, no walk, no corner codes:
//
//   * Per triangle edge the part inside the pixel is ONE segment [start, end]; start / end is either the
//     edge's end point (inside the pixel) or its crossing with a pixel edge.  The crossings, their validity
//     tests and the sort by t are the reference's expressions (aa.h:230-258, :308-313), evaluated on the
//     same operands as in the forward: the decisions are bit-for-bit the forward's, so the error paths
//     E00-E03 cannot occur here.
//   * Gradient.  Summed over the fan, the reference's per-triangle partials telescope to the shoelace form
//     dA/dc = 1/2 (y_next - y_prev, x_prev - x_next) per polygon corner c.  Pixel corners have a zero
//     Jacobian.  A crossing on a y = const pixel edge moves only in x, and its neighbour on the pixel
//     boundary has the same y, so its shoelace weight is 1/2 (end.y - start.y) of ITS OWN segment; likewise
//     1/2 (start.x - end.x) for a crossing on an x = const edge.  A triangle corner inside the pixel joins two
//     segments: weight 1/2 (end_next.y - start_prev.y, start_prev.x - end_next.x), Jacobian identity.  The
//     crossing Jacobians are aa.h:276-294 (t, gt0, gt1 as written there).  The entries the reference
//     multiplies with the weight's other component are (1 - t) + gt0*e and t + gt1*e, zero up to rounding
//     (the crossing does not leave its pixel edge); they are not issued.  Same polynomial as the reference's
//     in the same corner coordinates, regrouped: agrees with the oracle to ~4e-7 absolute (tests).
//   * Area.  Shoelace in pixel-local coordinates (the subtraction of the pixel origin is exact): 1/2 cross per
//     segment plus the pixel-boundary pieces between an exit crossing and the next entry crossing, which are
//     a function G of the perimeter coordinate.  Agrees with the reference's fan sum to 1 ulp of the pixel
//     area (6e-8); the backward only needs alpha to that accuracy (the blend decision is the forward's).
//
// tests/test_gpu_clippers.py runs this function on the reference-produced vectors (tests/golden/aa_pairs.npz)
// through dm2_debug_aa_overlap.
#pragma once
#include "dm2_device_math.h"

namespace dm2 {

struct EdgeSeg {
    bool has;           // the edge has a piece inside the pixel
    bool sX, eX;        // start / end is a crossing (otherwise the edge's own end point p_i / p_{i+1})
    float sx, sy, ex, ey;
    float ps, pe;       // perimeter coordinate of start / end (meaningful for crossings only)
};

// perimeter coordinate of a point on pixel edge `pedge` (0: y = ymin, 1: x = xmax, 2: y = ymax, 3: x = xmin), CCW
// from corner (xmin, ymin); lx, ly are pixel-local
__device__ __forceinline__ float perimeter_coord(int pedge, float lx, float ly) {
    const float along = (pedge & 1) ? ly : lx;
    const float fwd = (pedge & 2) ? 1.0f - along : along;
    return (float)pedge + fwd;
}
// shoelace weight of the pixel boundary from perimeter coordinate 0 to s (local origin at (xmin, ymin)): only the
// x = xmax and y = ymax edges contribute
__device__ __forceinline__ float perimeter_area(float s) { return 0.5f * fminf(fmaxf(s - 1.0f, 0.0f), 2.0f); }

template <int TI>
__device__ __forceinline__ void seg_edge(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax, EdgeSeg& S, float* g) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool e_vertical = (f.zmask >> (2 * TI)) & 1u, e_horizontal = (f.zmask >> (2 * TI + 1)) & 1u;
    const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
    // crossings with the pixel edges 0: y=pymin, 1: x=pxmax, 2: y=pymax, 3: x=pxmin (aa.h:230-258)
    const float tA = (pymin - p0y) * ry, xA = p0x + tA * ex;
    const float tB = (pxmax - p0x) * rx, yB = p0y + tB * ey;
    const float tC = (pymax - p0y) * ry, xC = p0x + tC * ex;
    const float tD = (pxmin - p0x) * rx, yD = p0y + tD * ey;
    const bool vA = (tA >= 0) && (tA <= 1) && (xA >= pxmin) && (xA <= pxmax) && !e_horizontal;
    const bool vB = (tB >= 0) && (tB <= 1) && (yB >= pymin) && (yB <= pymax) && !e_vertical;
    const bool vC = (tC >= 0) && (tC <= 1) && (xC >= pxmin) && (xC <= pxmax) && !e_horizontal;
    const bool vD = (tD >= 0) && (tD <= 1) && (yD >= pymin) && (yD <= pymax) && !e_vertical;
    const bool any = vA || vB || vC || vD;
    // first valid crossing in pixel-edge order and the last one (the same one when there is only one)
    const float x0 = vA ? xA : (vB ? pxmax : (vC ? xC : pxmin));
    const float y0 = vA ? pymin : (vB ? yB : (vC ? pymax : yD));
    const float t0 = vA ? tA : (vB ? tB : (vC ? tC : tD));
    const int pe0 = vA ? 0 : (vB ? 1 : (vC ? 2 : 3));
    const float x1 = vD ? pxmin : (vC ? xC : (vB ? pxmax : xA));
    const float y1 = vD ? yD : (vC ? pymax : (vB ? yB : pymin));
    const float t1 = vD ? tD : (vC ? tC : (vB ? tB : tA));
    const int pe1 = vD ? 3 : (vC ? 2 : (vB ? 1 : 0));
    const bool two = pe0 != pe1;                                       // (with `any`) two valid crossings: the forward excluded > 2
    const bool sw = t0 > t1;                                           // aa.h:308-313 (equal when there is one crossing)
    // [start, end]: entry crossing (or p0 when it is inside), exit crossing (or p1)
    const bool sX = any && (two || !p0in), eX = any && (two || p0in);
    S.has = any || p0in;                                               // no crossing: both end points inside, or the edge misses the pixel
    S.sX = sX; S.eX = eX;
    const float cs_x = sw ? x1 : x0, cs_y = sw ? y1 : y0, cs_t = sw ? t1 : t0;
    const float ce_x = sw ? x0 : x1, ce_y = sw ? y0 : y1, ce_t = sw ? t0 : t1;
    const int cs_pe = sw ? pe1 : pe0, ce_pe = sw ? pe0 : pe1;
    S.sx = sX ? cs_x : p0x; S.sy = sX ? cs_y : p0y;
    S.ex = eX ? ce_x : p1x; S.ey = eX ? ce_y : p1y;
    S.ps = perimeter_coord(cs_pe, S.sx - pxmin, S.sy - pymin);
    S.pe = perimeter_coord(ce_pe, S.ex - pxmin, S.ey - pymin);
    // shoelace weights of a crossing of this segment on a y = const / x = const pixel edge
    const float wH = 0.5f * (S.ey - S.sy), wV = 0.5f * (S.sx - S.ex);
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;                      // d/d(p0x, p0y), d/d(p1x, p1y)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const bool on = k == 0 ? sX : eX;
        const int pe = k == 0 ? cs_pe : ce_pe;
        const float t = k == 0 ? cs_t : ce_t;
        const bool isH = (pe & 1) == 0;
        const float w = isH ? ((pe & 2) ? pymax : pymin) : ((pe & 2) ? pxmin : pxmax);     // the pixel edge's constant
        const float p0a = isH ? p0y : p0x, p1a = isH ? p1y : p1x, ra = isH ? ry : rx, eo = isH ? ex : ey;
        const float wt = isH ? wH : wV;
        const float gt0 = (w - p1a) * ra * ra, gt1 = (-w + p0a) * ra * ra;                // aa.h:276-279
        const float omt = (float)(1.0 - (double)t);
        const float u0 = omt * wt, u1 = (gt0 * eo) * wt, v0 = t * wt, v1 = (gt1 * eo) * wt;
        // H: the crossing moves in x: d(x)/d(p0) = (omt, gt0*ex), d(x)/d(p1) = (t, gt1*ex);  V: in y, components swapped
        a0 += on ? (isH ? u0 : u1) : 0.f; a1 += on ? (isH ? u1 : u0) : 0.f;
        b0 += on ? (isH ? v0 : v1) : 0.f; b1 += on ? (isH ? v1 : v0) : 0.f;
    }
    g[2 * TI] += a0; g[2 * TI + 1] += a1; g[2 * TJ] += b0; g[2 * TJ + 1] += b1;
}

// Area and d(area)/d(corners) of pixel [pxmin,pxmax]x[pymin,pymax] and face f, for a pair whose reference clip
// (aa.h:446-504) returns no error and a positive area.  g: [3][2] row-major.
__device__ __forceinline__ void seg_area_grad(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                              float pix_area, float& area, float* g) {
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    EdgeSeg S0, S1, S2;
    seg_edge<0>(f, pxmin, pxmax, pymin, pymax, S0, g);
    seg_edge<1>(f, pxmin, pxmax, pymin, pymax, S1, g);
    seg_edge<2>(f, pxmin, pxmax, pymin, pymax, S2, g);
    // triangle corners inside the pixel: p1 joins segments 0 and 1, p2 joins 1 and 2, p0 joins 2 and 0
    {
        const bool in1 = S0.has && !S0.eX, in2 = S1.has && !S1.eX, in0 = S2.has && !S2.eX;
        g[2] += in1 ? 0.5f * (S1.ey - S0.sy) : 0.f; g[3] += in1 ? 0.5f * (S0.sx - S1.ex) : 0.f;
        g[4] += in2 ? 0.5f * (S2.ey - S1.sy) : 0.f; g[5] += in2 ? 0.5f * (S1.sx - S2.ex) : 0.f;
        g[0] += in0 ? 0.5f * (S0.ey - S2.sy) : 0.f; g[1] += in0 ? 0.5f * (S2.sx - S0.ex) : 0.f;
    }
    if (!(S0.has || S1.has || S2.has)) { area = pix_area; return; }   // no edge reaches the pixel: it lies inside (aa.h:493-496)
    // shoelace: segments in pixel-local coordinates + the pixel boundary from every exit crossing to the next entry crossing
    float a = 0.f;
    const float gs0 = perimeter_area(S0.ps), gs1 = perimeter_area(S1.ps), gs2 = perimeter_area(S2.ps);
    auto piece = [&](const EdgeSeg& S, const EdgeSeg& N1, float gN1, const EdgeSeg& N2, float gN2, float gSelf) {
        const float ax = S.sx - pxmin, ay = S.sy - pymin, bx = S.ex - pxmin, by = S.ey - pymin;
        const float seg = 0.5f * (ax * by - bx * ay);
        const float nps = N1.has ? N1.ps : (N2.has ? N2.ps : S.ps);
        const float ngs = N1.has ? gN1 : (N2.has ? gN2 : gSelf);
        float walk = ngs - perimeter_area(S.pe);
        walk += (nps < S.pe) ? 1.0f : 0.0f;                            // wrapped past corner (xmin, ymin)
        a += S.has ? seg + (S.eX ? walk : 0.f) : 0.f;
    };
    piece(S0, S1, gs1, S2, gs2, gs0);
    piece(S1, S2, gs2, S0, gs0, gs1);
    piece(S2, S0, gs0, S1, gs1, gs2);
    area = a;
}

}  // namespace dm2
