// dm2_clip_seg.h -- overlap area of a CCW triangle and a unit pixel AND its Jacobian w.r.t. the triangle's
// corners, for (pixel, face) pairs the FORWARD has already accepted (backward pass, gfx950).
//
// The reference builds the clipped polygon in a vertex table, fan-triangulates it and differentiates every
// fan triangle through the Jacobians of its three corners (aa.h:151-441).  The backward kernels only meet
// pairs whose forward clip returned "no error, area > 0" (the forward's blend masks say so).  This file
// rebuilds exactly the reference's polygon -- the same corners in the same cyclic order, also where exact
// ties make that polygon geometrically inconsistent (with fp32 image coordinates a triangle corner lies
// EXACTLY on a pixel boundary about once in 10^4, and the reference then returns whatever its rules give) --
// but without a table, a walk loop or corner codes:
//
//   * Per triangle edge the reference emits at most two corners with a non-zero Jacobian: [entry crossing,
//     exit crossing], [entry crossing, end point p1], [exit crossing] or [p1] (aa.h:304-357), decided by its
//     crossing-validity and point-in-pixel tests (aa.h:230-258, :308-313).  Those tests are evaluated here by
//     the same expressions on the same operands as in the forward, so the decisions are the forward's bit for
//     bit and the error paths E00-E03 cannot occur.
//   * Behind an exit crossing on pixel edge e the reference appends the pixel corners e+1, e+2, ... while they
//     are inside all three half planes (aa.h:359-379, the classification of aa.h:103-149): a count of
//     consecutive set bits of the 4-bit corner mask, no loop.
//   * So every emitted corner knows its polygon neighbours in closed form: within the edge, the first / last
//     walked pixel corner, or the first corner of the next edge that emits anything (the last one of the
//     previous edge).  Gradient: summed over the fan, the reference's per-triangle partials (aa.h:415-433)
//     telescope to the shoelace form dA/dc = 1/2 (y_next - y_prev, x_prev - x_next) per corner, pushed through
//     the corner's Jacobian (identity for p1, aa.h:276-294 for a crossing, zero for a pixel corner).  Same
//     polynomial in the same corner coordinates, regrouped: agrees with the oracle to ~1e-6 absolute.
//   * Area: shoelace over the same cyclic sequence in pixel-local coordinates (the subtraction of the pixel
//     origin is exact); the pixel-corner chain contributes 1/2 per traversed x = xmax or y = ymax edge.
//     Agrees with the reference's fan sum to 2 ulp of the pixel area; enough for alpha up to ~0.9 (the blend decision
//     itself is the forward's, never re-taken).  For nearly opaque faces the caller asks for `exact_area`: the fan sum
//     itself over the same corners, bit-identical to the forward's area.
//
// tests/test_gpu_clippers.py runs this function on the reference-produced vectors (tests/golden/aa_pairs.npz,
// aa_error_pairs.npz), on random pairs and on exact-tie stress sets through dm2_debug_aa_overlap.
#pragma once
#include <type_traits>

#include "dm2_clip_area.h"
#include "dm2_device_math.h"

namespace dm2 {

struct EdgeSeg {
    bool has;           // the edge emits at least one corner
    bool sX, eX;        // first emitted corner is an (entry) crossing / last emitted non-pixel corner is an (exit) crossing
    float sx, sy;       // the entry crossing (meaningful when sX)
    float ex, ey;       // the last emitted non-pixel corner: exit crossing (eX) or the edge's end point p1
    float st, et;       // crossing parameters
    int spe, epe;       // pixel edges of the crossings (0: y = ymin, 1: x = xmax, 2: y = ymax, 3: x = xmin)
    int k;              // pixel corners appended behind the exit crossing
    float ax, ay;       // first emitted corner
    float lx, ly;       // last emitted corner (a pixel corner when k > 0)
    float wx, wy;       // first appended pixel corner (meaningful when k > 0)
};

template <int TI>
__device__ __forceinline__ void seg_edge(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax, uint32_t inside, EdgeSeg& S) {
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
    // emitted: two crossings [a, b]; one crossing and p0 outside [c, p1]; one crossing and p0 inside [c]; none, both inside [p1]
    const bool sX = any && (two || !p0in), eX = any && (two || p0in);
    S.has = any || p0in;                                               // no crossing: both end points inside, or the edge misses the pixel
    S.sX = sX; S.eX = eX;
    S.sx = sw ? x1 : x0; S.sy = sw ? y1 : y0; S.st = sw ? t1 : t0; S.spe = sw ? pe1 : pe0;
    const float cx = sw ? x0 : x1, cy = sw ? y0 : y1;
    S.et = sw ? t0 : t1; S.epe = sw ? pe0 : pe1;
    S.ex = eX ? cx : p1x; S.ey = eX ? cy : p1y;
    S.ax = sX ? S.sx : S.ex; S.ay = sX ? S.sy : S.ey;
    // pixel corners behind the exit crossing: corner (epe + 1) & 3, ... while inside (the caller excluded inside == 0xF)
    const int c1 = (S.epe + 1) & 3;
    const uint32_t rot = ((inside | (inside << 4)) >> c1) & 0xFu;
    const int k = (rot & 1u) ? ((rot & 2u) ? ((rot & 4u) ? 3 : 2) : 1) : 0;
    S.k = (S.has && eX) ? k : 0;
    const int lc = (S.epe + S.k) & 3;
    S.wx = (c1 == 1 || c1 == 2) ? pxmax : pxmin; S.wy = (c1 & 2) ? pymax : pymin;
    const float lcx = (lc == 1 || lc == 2) ? pxmax : pxmin, lcy = (lc & 2) ? pymax : pymin;
    S.lx = S.k > 0 ? lcx : S.ex; S.ly = S.k > 0 ? lcy : S.ey;
}

// d(area) of one crossing of edge TI with shoelace weight (gax, gay): aa.h:276-294 (t, gt0, gt1 as written there)
template <int TI>
__device__ __forceinline__ void seg_push_crossing(const AAFace& f, bool on, int pe, float t, float pxmin, float pxmax, float pymin,
                                                  float pymax, float gax, float gay, float* g) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool isH = (pe & 1) == 0;                                    // crossing of a y = const pixel edge: x was computed
    const float w = isH ? ((pe & 2) ? pymax : pymin) : ((pe & 2) ? pxmin : pxmax);       // the pixel edge's constant
    const float p0a = isH ? p0y : p0x, p1a = isH ? p1y : p1x, ra = isH ? ry : rx;
    const float ea = isH ? ey : ex, eo = isH ? ex : ey;               // edge component along the crossing axis / the other one
    const float gm = isH ? gax : gay, gr = isH ? gay : gax;           // weight of the computed coordinate / of the fixed one
    const float gt0 = (w - p1a) * ra * ra, gt1 = (-w + p0a) * ra * ra;
    const float omt = 1.0f - t;                                        // == (float)(1.0 - (double)t)
    // computed coordinate m = p0o + t * eo: dm/dp0 = (omt [other axis], gt0 * eo [crossing axis]), dm/dp1 = (t, gt1 * eo);
    // fixed coordinate:  d/dp0 = ((1 - t) + gt0 * ea) on the crossing axis (zero up to rounding), d/dp1 = (t + gt1 * ea)
    const float a_o = omt * gm, a_a = (gt0 * eo) * gm + (omt + gt0 * ea) * gr;
    const float b_o = t * gm, b_a = (gt1 * eo) * gm + (t + gt1 * ea) * gr;
    // H: other axis = x, crossing axis = y
    g[2 * TI] += on ? (isH ? a_o : a_a) : 0.f; g[2 * TI + 1] += on ? (isH ? a_a : a_o) : 0.f;
    g[2 * TJ] += on ? (isH ? b_o : b_a) : 0.f; g[2 * TJ + 1] += on ? (isH ? b_a : b_o) : 0.f;
}

// 1/2 * (shoelace terms of the pixel-corner chain c1 -> c1+1 -> ... -> c1+k-1): only the transitions along x = xmax
// (corner 1 -> 2) and y = ymax (2 -> 3) have a non-zero cross product in pixel-local coordinates
__device__ __forceinline__ float corner_chain_area(int c1, int k) {
    auto G = [](float u) { return 0.5f * (fminf(fmaxf(u - 1.0f, 0.0f), 2.0f) + fminf(fmaxf(u - 5.0f, 0.0f), 2.0f)); };
    return k > 0 ? G((float)(c1 + k - 1)) - G((float)c1) : 0.0f;
}

// Area and d(area)/d(corners) of pixel [pxmin,pxmax]x[pymin,pymax] and face f, for a pair whose reference clip
// (aa.h:446-504) returns no error and a positive area.  g: [3][2] row-major.
// exact_area (per lane): also run the reference's fan sum over the rebuilt corners -- the same corners in the same order
// through the forward's own fan_push (dm2_clip_area.h), hence the forward's area TO THE BIT instead of the shoelace's 2 ulp.
// The backward needs that where alpha is close to 1 (an ulp of alpha is ulp / (1 - alpha) of the replayed T); the pass is
// skipped by a wave none of whose lanes asks for it.
__device__ __forceinline__ void seg_area_grad(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                              float pix_area, float& area, float* g, bool exact_area = false) {
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    uint32_t inside;
    classify_pixel(f, pxmin, pxmax, pymin, pymax, inside);            // aa.h:103-149 (the forward passed its all-outside test)
    if (inside == 0xFu) { area = pix_area; return; }                  // aa.h:493-496: not clipped at all, zero Jacobian
    EdgeSeg S0, S1, S2;
    seg_edge<0>(f, pxmin, pxmax, pymin, pymax, inside, S0);
    seg_edge<1>(f, pxmin, pxmax, pymin, pymax, inside, S1);
    seg_edge<2>(f, pxmin, pxmax, pymin, pymax, inside, S2);
    // previous / next edge that emits anything (itself when it is the only one)
    auto sel = [](bool c1, float a, bool c2, float b, float self) { return c1 ? a : (c2 ? b : self); };
    // last corner before edge i's first, first corner behind edge i's last
    const float P0x = sel(S2.has, S2.lx, S1.has, S1.lx, S0.lx), P0y = sel(S2.has, S2.ly, S1.has, S1.ly, S0.ly);
    const float P1x = sel(S0.has, S0.lx, S2.has, S2.lx, S1.lx), P1y = sel(S0.has, S0.ly, S2.has, S2.ly, S1.ly);
    const float P2x = sel(S1.has, S1.lx, S0.has, S0.lx, S2.lx), P2y = sel(S1.has, S1.ly, S0.has, S0.ly, S2.ly);
    const float N0x = sel(S1.has, S1.ax, S2.has, S2.ax, S0.ax), N0y = sel(S1.has, S1.ay, S2.has, S2.ay, S0.ay);
    const float N1x = sel(S2.has, S2.ax, S0.has, S0.ax, S1.ax), N1y = sel(S2.has, S2.ay, S0.has, S0.ay, S1.ay);
    const float N2x = sel(S0.has, S0.ax, S1.has, S1.ax, S2.ax), N2y = sel(S0.has, S0.ay, S1.has, S1.ay, S2.ay);
    float a2 = 0.f;                                                   // twice the area
    // exact_area: the reference's fan sum in its emission order (aa.h:304-379): per edge [entry crossing] [exit crossing |
    // end point] [pixel corners], streamed through the forward's fan_push while the edge's corners are at hand
    const bool do_fan = __any(exact_area);                            // wave-uniform
    FanState F;
    F.fx = F.fy = F.px = F.py = 0.f; F.area = 0.f; F.cnt = 0; F.err = false;
    auto edge = [&](auto ti, const EdgeSeg& S, float Px, float Py, float Nx, float Ny) {
        constexpr int TI = decltype(ti)::value;
        constexpr int TJ = (TI + 1) % 3;
        if (do_fan) {
            fan_push(F, exact_area && S.has && S.sX, S.sx, S.sy);
            fan_push(F, exact_area && S.has, S.ex, S.ey);
#pragma unroll
            for (int pvi = 0; pvi < 3; pvi++) {                       // at most three: inside != 0xF
                const int cur = (S.epe + 1 + pvi) & 3;
                fan_push(F, exact_area && (pvi < S.k), (cur == 1 || cur == 2) ? pxmax : pxmin, (cur >= 2) ? pymax : pymin);
            }
        }
        // first emitted corner when it is the entry crossing: neighbours = (last corner before this edge, the second corner)
        seg_push_crossing<TI>(f, S.has && S.sX, S.spe, S.st, pxmin, pxmax, pymin, pymax, 0.5f * (S.ey - Py), 0.5f * (Px - S.ex), g);
        // last emitted non-pixel corner: previous = the entry crossing or the last corner before this edge; next = the first
        // appended pixel corner or the first corner of the next edge
        const float qx = S.sX ? S.sx : Px, qy = S.sX ? S.sy : Py;
        const float nx = S.k > 0 ? S.wx : Nx, ny = S.k > 0 ? S.wy : Ny;
        const float gax = 0.5f * (ny - qy), gay = 0.5f * (qx - nx);
        seg_push_crossing<TI>(f, S.has && S.eX, S.epe, S.et, pxmin, pxmax, pymin, pymax, gax, gay, g);
        const bool vert = S.has && !S.eX;                             // the edge's end point p1: identity Jacobian
        g[2 * TJ] += vert ? gax : 0.f; g[2 * TJ + 1] += vert ? gay : 0.f;
        // shoelace, pixel-local: (entry, last non-pixel corner), (that, first pixel corner), the chain, (last corner, next edge's first)
        const float slx = S.sx - pxmin, sly = S.sy - pymin, elx = S.ex - pxmin, ely = S.ey - pymin;
        const float wlx = S.wx - pxmin, wly = S.wy - pymin, llx = S.lx - pxmin, lly = S.ly - pymin;
        const float nlx = Nx - pxmin, nly = Ny - pymin;
        float t2 = S.sX ? (slx * ely - elx * sly) : 0.f;
        t2 += S.k > 0 ? (elx * wly - wlx * ely) + 2.0f * corner_chain_area((S.epe + 1) & 3, S.k) : 0.f;
        t2 += llx * nly - nlx * lly;
        a2 += S.has ? t2 : 0.f;
    };
    edge(std::integral_constant<int, 0>{}, S0, P0x, P0y, N0x, N0y);
    edge(std::integral_constant<int, 1>{}, S1, P1x, P1y, N1x, N1y);
    edge(std::integral_constant<int, 2>{}, S2, P2x, P2y, N2x, N2y);
    area = 0.5f * a2;
    if (do_fan && exact_area) area = F.area;
}

}  // namespace dm2
