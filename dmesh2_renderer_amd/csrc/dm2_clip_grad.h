// dm2_clip_grad.h -- overlap area of a CCW triangle and a unit pixel AND its Jacobian
// w.r.t. the triangle's corners (aa.h:151-441), for the backward pass on gfx950.
//
// Construction of the clipped polygon is the straight-line code of dm2_clip_area.h (same
// corners, same order, same fan sum -> the AREA is bit-identical to the forward's, which
// the backward relies on to replay the forward's decisions).  Corners are additionally
// written to a per-lane, lane-strided LDS table with a 4-bit kind code each.
//
// Gradient: the reference differentiates its fan sum triangle by triangle
// (aa.h:415-433), i.e. every polygon corner c_i receives d(fan triangle)/d(c_i) from up
// to all fan triangles it belongs to, each pushed through the corner's Jacobian J_i
// separately.  Summed over the fan these partials telescope to the shoelace form
//       dA/dc_i = 1/2 * ( y_{i+1} - y_{i-1},  x_{i-1} - x_{i+1} )      (cyclic),
// so here each corner is pushed through J_i ONCE with that vector.  This is the same
// polynomial in the same inputs, regrouped: results differ from the reference's order of
// fp32 additions at the 1e-7 relative level (the gradients are accumulated with atomics
// afterwards, whose order is not reproducible either); DESIGN.md lists it as the one
// place where the backward departs from the reference's operation order.
#pragma once
#include "dm2_clip_area.h"
#include "dm2_clip_lds.h"
#include "dm2_device_math.h"

namespace dm2 {

struct TabState {
    FanState fan;
    uint64_t codes;
};

// DM2_CLIP_DEFERRED_AREA (default): the corners only go to the LDS table here; the fan area (aa.h:404-413, same
// triangles in the same order) is summed by the loop that walks the table for the gradient anyway.  The streaming
// fan (fan_push: cross product, double-literal product and seven selects at each of the 18 emit sites) is the
// forward kernel's way, which has no table.
#ifndef DM2_CLIP_DEFERRED_AREA
#define DM2_CLIP_DEFERRED_AREA 1
#endif
__device__ __forceinline__ void tab_push(TabState& S, bool en, float x, float y, uint32_t code, float* polyx, float* polyy) {
    const int slot = S.fan.cnt;
    if (en && slot < MAX_POLY) {
        polyx[slot * POLY_STRIDE] = x; polyy[slot * POLY_STRIDE] = y;
        S.codes |= (uint64_t)code << (4 * slot);
    }
#if DM2_CLIP_DEFERRED_AREA
    S.fan.err = S.fan.err || (en && slot >= MAX_POLY);                 // aa.h:45-48 -> error 5
    S.fan.cnt += en ? 1 : 0;
#else
    fan_push(S.fan, en, x, y);
#endif
}

template <int TI>
__device__ __forceinline__ void clip_edge_tab(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                              uint32_t inside, TabState& S, float* polyx, float* polyy) {
    constexpr int TJ = (TI + 1) % 3;
    constexpr uint32_t ecode = (uint32_t)TI << 2;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool e_vertical = (f.zmask >> (2 * TI)) & 1u, e_horizontal = (f.zmask >> (2 * TI + 1)) & 1u;
    const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
    const bool p1in = (p1x >= pxmin) && (p1x <= pxmax) && (p1y >= pymin) && (p1y <= pymax);
    const float tA = (pymin - p0y) * ry, xA = p0x + tA * ex;
    const float tB = (pxmax - p0x) * rx, yB = p0y + tB * ey;
    const float tC = (pymax - p0y) * ry, xC = p0x + tC * ex;
    const float tD = (pxmin - p0x) * rx, yD = p0y + tD * ey;
    const bool vA = (tA >= 0) && (tA <= 1) && (xA >= pxmin) && (xA <= pxmax) && !e_horizontal;
    const bool vB = (tB >= 0) && (tB <= 1) && (yB >= pymin) && (yB <= pymax) && !e_vertical;
    const bool vC = (tC >= 0) && (tC <= 1) && (xC >= pxmin) && (xC <= pxmax) && !e_horizontal;
    const bool vD = (tD >= 0) && (tD <= 1) && (yD >= pymin) && (yD <= pymax) && !e_vertical;
    const bool e00 = (vA && ((xA == pxmin) || (xA == pxmax))) || (vB && ((yB == pymin) || (yB == pymax))) ||
                     (vC && ((xC == pxmin) || (xC == pxmax))) || (vD && ((yD == pymin) || (yD == pymax)));
    const int n = (int)vA + (int)vB + (int)vC + (int)vD;
    const float x0 = vA ? xA : (vB ? pxmax : (vC ? xC : pxmin));
    const float y0 = vA ? pymin : (vB ? yB : (vC ? pymax : yD));
    const float t0 = vA ? tA : (vB ? tB : (vC ? tC : tD));
    const int pe0 = vA ? 0 : (vB ? 1 : (vC ? 2 : 3));
    const float x1 = vD ? pxmin : (vC ? xC : pxmax);
    const float y1 = vD ? yD : (vC ? pymax : yB);
    const float t1 = vD ? tD : (vC ? tC : tB);
    const int pe1 = vD ? 3 : (vC ? 2 : 1);
    const uint32_t k0 = (pe0 & 1) ? PK_XV : PK_XH, k1 = (pe1 & 1) ? PK_XV : PK_XH;
    const bool two = (n == 2), one = (n == 1), none = (n == 0);
    const bool sw = two && (t0 > t1);
    S.fan.err = S.fan.err || e00 || (n > 2) || (one && (p0in == p1in)) || (none && (p0in != p1in));
    const bool en1 = two || one || (none && p0in && p1in);
    const float ax = two ? (sw ? x1 : x0) : (one ? x0 : p1x);
    const float ay = two ? (sw ? y1 : y0) : (one ? y0 : p1y);
    const uint32_t ac = (two ? (sw ? k1 : k0) : (one ? k0 : PK_TRIV)) | ecode;
    tab_push(S, en1 && !S.fan.err, ax, ay, ac, polyx, polyy);
    const bool en2 = two || (one && !p0in && p1in);
    const float bx = two ? (sw ? x0 : x1) : p1x;
    const float by = two ? (sw ? y0 : y1) : p1y;
    const uint32_t bc = (two ? (sw ? k0 : k1) : PK_TRIV) | ecode;
    tab_push(S, en2 && !S.fan.err, bx, by, bc, polyx, polyy);
    const bool walk = two || (one && p0in && !p1in);
    const int final_pe = two ? (sw ? pe0 : pe1) : pe0;
    bool go = walk && !S.fan.err;
#pragma unroll
    for (int pvi = 0; pvi < 4; pvi++) {
        const int cur = (final_pe + 1 + pvi) & 3;
        go = go && ((inside >> cur) & 1u);
        const float cx = (cur == 1 || cur == 2) ? pxmax : pxmin;
        const float cy = (cur >= 2) ? pymax : pymin;
        tab_push(S, go, cx, cy, PK_CORNER, polyx, polyy);
    }
}

// Push dA/dc = (gax, gay) of one polygon corner through its Jacobian (aa.h:67-86, :276-294).
// Straight-line: edge data selected from registers by the corner's edge index.
__device__ __forceinline__ void corner_grad_sel(const AAFace& f, uint32_t code, float x, float y, float gax, float gay, float* g) {
    const uint32_t kind = code & 3u;
    const int ti = (int)(code >> 2);
    const int tj = ti == 2 ? 0 : ti + 1;
    const float ex = ti == 0 ? f.e[0] : (ti == 1 ? f.e[2] : f.e[4]);
    const float ey = ti == 0 ? f.e[1] : (ti == 1 ? f.e[3] : f.e[5]);
    const bool xh = kind == PK_XH;
    // along-axis quantities: axis0 = y for a crossing of a y=const pixel edge, x otherwise
    const float p0a = xh ? (ti == 0 ? f.v[1] : (ti == 1 ? f.v[3] : f.v[5])) : (ti == 0 ? f.v[0] : (ti == 1 ? f.v[2] : f.v[4]));
    const float p1a = xh ? (tj == 0 ? f.v[1] : (tj == 1 ? f.v[3] : f.v[5])) : (tj == 0 ? f.v[0] : (tj == 1 ? f.v[2] : f.v[4]));
    const float ra = xh ? (ti == 0 ? f.r[1] : (ti == 1 ? f.r[3] : f.r[5])) : (ti == 0 ? f.r[0] : (ti == 1 ? f.r[2] : f.r[4]));
    const float ia = xh ? y : x;
    const float t = (ia - p0a) * ra;
    const float gt0 = (ia - p1a) * ra * ra;
    const float gt1 = (-ia + p0a) * ra * ra;
    const float omt = (float)(1.0 - (double)t);
    // crossing Jacobians, see dm2_clip_lds.h corner_grad for the derivation of the two layouts
    const float a0_h = omt * gax, a1_h = (gt0 * ex) * gax + one_minus_t_plus(t, gt0 * ey) * gay;
    const float b0_h = t * gax, b1_h = (gt1 * ex) * gax + (t + (gt1 * ey)) * gay;
    const float a0_v = one_minus_t_plus(t, gt0 * ex) * gax + (gt0 * ey) * gay, a1_v = omt * gay;
    const float b0_v = (t + (gt1 * ex)) * gax + (gt1 * ey) * gay, b1_v = t * gay;
    const bool cross = kind >= PK_XH;
    const bool triv = kind == PK_TRIV;
    const float a0 = cross ? (xh ? a0_h : a0_v) : 0.f, a1 = cross ? (xh ? a1_h : a1_v) : 0.f;
    const float b0 = cross ? (xh ? b0_h : b0_v) : (triv ? gax : 0.f), b1 = cross ? (xh ? b1_h : b1_v) : (triv ? gay : 0.f);
    const int rowa = cross ? ti : -1;                 // triangle vertex: only the (0, I) block acts (row tj)
    const int rowb = (cross || triv) ? tj : -1;       // pixel corner: nothing
    rows_add(g, rowa, a0, a1);
    rows_add(g, rowb, b0, b1);
}

// ---- compact corner table ---------------------------------------------------------------------------------
// A polygon corner is a pixel corner, a triangle vertex or the crossing of a triangle edge with a pixel edge;
// only a crossing carries a computed coordinate, and only one (the other is the pixel edge's constant).  So the
// LDS table holds ONE float per corner and a 6-bit code (kind | edge << 2 | aux << 4; aux = pixel edge of a
// crossing / index of a pixel corner); the walk rebuilds (x, y) -- the same values the reference stores.  Half
// the LDS of the (x, y) table: what lets a fourth block of the backward kernel live on a CU.
#ifndef DM2_CLIP_COMPACT_TABLE
#define DM2_CLIP_COMPACT_TABLE 1
#endif
struct CTab { int cnt; bool err; uint64_t codes; };

__device__ __forceinline__ void ctab_push(CTab& S, bool en, float val, uint32_t code6, float* polyv) {
    const int slot = S.cnt;
    if (en && slot < MAX_POLY) {
        polyv[slot * POLY_STRIDE] = val;
        S.codes |= (uint64_t)code6 << (6 * slot);
    }
    S.err = S.err || (en && slot >= MAX_POLY);                         // aa.h:45-48 -> error 5
    S.cnt += en ? 1 : 0;
}

template <int TI>
__device__ __forceinline__ void clip_edge_ctab(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                               uint32_t inside, CTab& S, float* polyv) {
    constexpr int TJ = (TI + 1) % 3;
    constexpr uint32_t ecode = (uint32_t)TI << 2;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool e_vertical = (f.zmask >> (2 * TI)) & 1u, e_horizontal = (f.zmask >> (2 * TI + 1)) & 1u;
    const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
    const bool p1in = (p1x >= pxmin) && (p1x <= pxmax) && (p1y >= pymin) && (p1y <= pymax);
    const float tA = (pymin - p0y) * ry, xA = p0x + tA * ex;
    const float tB = (pxmax - p0x) * rx, yB = p0y + tB * ey;
    const float tC = (pymax - p0y) * ry, xC = p0x + tC * ex;
    const float tD = (pxmin - p0x) * rx, yD = p0y + tD * ey;
    const bool vA = (tA >= 0) && (tA <= 1) && (xA >= pxmin) && (xA <= pxmax) && !e_horizontal;
    const bool vB = (tB >= 0) && (tB <= 1) && (yB >= pymin) && (yB <= pymax) && !e_vertical;
    const bool vC = (tC >= 0) && (tC <= 1) && (xC >= pxmin) && (xC <= pxmax) && !e_horizontal;
    const bool vD = (tD >= 0) && (tD <= 1) && (yD >= pymin) && (yD <= pymax) && !e_vertical;
    const bool e00 = (vA && ((xA == pxmin) || (xA == pxmax))) || (vB && ((yB == pymin) || (yB == pymax))) ||
                     (vC && ((xC == pxmin) || (xC == pxmax))) || (vD && ((yD == pymin) || (yD == pymax)));
    const int n = (int)vA + (int)vB + (int)vC + (int)vD;
    // first valid crossing in pixel-edge order and the last one: the coordinate that was computed, its t, its pixel edge
    const float w0 = vA ? xA : (vB ? yB : (vC ? xC : yD));
    const float t0 = vA ? tA : (vB ? tB : (vC ? tC : tD));
    const int pe0 = vA ? 0 : (vB ? 1 : (vC ? 2 : 3));
    const float w1 = vD ? yD : (vC ? xC : yB);
    const float t1 = vD ? tD : (vC ? tC : tB);
    const int pe1 = vD ? 3 : (vC ? 2 : 1);
    const bool two = (n == 2), one = (n == 1), none = (n == 0);
    const bool sw = two && (t0 > t1);
    S.err = S.err || e00 || (n > 2) || (one && (p0in == p1in)) || (none && (p0in != p1in));
    // corners of this edge, in order: [crossing a] [crossing b | end point p1] [pixel corners ...]
    const bool en1 = two || one || (none && p0in && p1in);
    const int pea = (two && sw) ? pe1 : pe0;
    const float wa = (two && sw) ? w1 : w0;
    const uint32_t ca = (two || one) ? (((pea & 1) ? PK_XV : PK_XH) | ecode | ((uint32_t)pea << 4)) : (PK_TRIV | ecode);
    ctab_push(S, en1 && !S.err, wa, ca, polyv);
    const bool en2 = two || (one && !p0in && p1in);
    const int peb = sw ? pe0 : pe1;
    const float wb = sw ? w0 : w1;
    const uint32_t cb = two ? (((peb & 1) ? PK_XV : PK_XH) | ecode | ((uint32_t)peb << 4)) : (PK_TRIV | ecode);
    ctab_push(S, en2 && !S.err, wb, cb, polyv);
    // pixel corners inside the triangle, counter-clockwise from the edge's exit (aa.h:359-379)
    const bool walk = two || (one && p0in && !p1in);
    const int final_pe = two ? (sw ? pe0 : pe1) : pe0;
    bool go = walk && !S.err;
#pragma unroll
    for (int pvi = 0; pvi < 4; pvi++) {
        const int cur = (final_pe + 1 + pvi) & 3;
        go = go && ((inside >> cur) & 1u);
        if (__ballot(go) == 0ull) break;          // wave-uniform: no lane walks on (see dm2_clip_area.h)
        ctab_push(S, go, 0.f, PK_CORNER | ((uint32_t)cur << 4), polyv);
    }
}

// (x, y) of a table corner
__device__ __forceinline__ void ctab_xy(const AAFace& f, uint32_t code6, float val, float pxmin, float pxmax, float pymin, float pymax,
                                        float& x, float& y) {
    const uint32_t kind = code6 & 3u, aux = code6 >> 4;
    const int tj = (int)((code6 >> 2) & 3u) == 2 ? 0 : (int)((code6 >> 2) & 3u) + 1;
    const float vx = tj == 0 ? f.v[0] : (tj == 1 ? f.v[2] : f.v[4]);
    const float vy = tj == 0 ? f.v[1] : (tj == 1 ? f.v[3] : f.v[5]);
    const float ex_ = (aux == 1u || aux == 2u) ? pxmax : pxmin;          // pixel corner aux / x of the x = const pixel edge aux (1 or 3)
    const float ey_ = (aux >= 2u) ? pymax : pymin;                       // pixel corner aux / y of the y = const pixel edge aux (0 or 2)
    x = kind == PK_XH ? val : (kind == PK_TRIV ? vx : ex_);
    y = kind == PK_XV ? val : (kind == PK_TRIV ? vy : ey_);
}

// aa.h:151-441 with Jacobian for a pixel that passed classify_pixel (dm2_clip_area.h) with corner mask `inside`.
// Returns non-zero on any reference error; area / g valid when 0.
__device__ __forceinline__ int clip_area_grad_classified(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                         uint32_t inside, float pix_area, float* polyx, float* polyy,
                                                         float& area, float* g);

// aa.h:446-504 with Jacobian.  Returns non-zero on any reference error; area / g valid when 0.
__device__ __forceinline__ int tri_pix_overlap_area_grad(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                         float pix_area, float* polyx, float* polyy, float& area, float* g) {
    area = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    if ((pxmax < f.bb[0]) || (pxmin > f.bb[1]) || (pymax < f.bb[2]) || (pymin > f.bb[3])) return 0;
    uint32_t inside;
    if (!classify_pixel(f, pxmin, pxmax, pymin, pymax, inside)) return 0;
    return clip_area_grad_classified(f, pxmin, pxmax, pymin, pymax, inside, pix_area, polyx, polyy, area, g);
}

__device__ __forceinline__ int clip_area_grad_classified(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                         uint32_t inside, float pix_area, float* polyx, float* polyy,
                                                         float& area, float* g) {
    area = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = 0.f;
    if (inside == 0xF) { area = pix_area; return 0; }                 // aa.h:493-496: zero Jacobian
#ifdef DM2_ABLATE_CLIP
    area = 0.5f * pix_area; return 0;
#endif
#if DM2_CLIP_COMPACT_TABLE
    (void)polyy;
    CTab S; S.cnt = 0; S.err = false; S.codes = 0;
    clip_edge_ctab<0>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx);
    clip_edge_ctab<1>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx);
    clip_edge_ctab<2>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx);
    if (S.err) return 1;
    const int cnt = S.cnt;
    if (cnt < 3) return 0;                        // no fan triangle: area 0 (the caller skips it)
    // one walk over the corners: shoelace gradient of corner i, and the reference's fan triangle (c_0, c_{i-1}, c_i)
    float fx, fy, xm, ym;
    ctab_xy(f, (uint32_t)(S.codes & 63u), polyx[0], pxmin, pxmax, pymin, pymax, fx, fy);
    ctab_xy(f, (uint32_t)((S.codes >> (6 * (cnt - 1))) & 63u), polyx[(cnt - 1) * POLY_STRIDE], pxmin, pxmax, pymin, pymax, xm, ym);
    float xc = fx, yc = fy;                                                           // c_i   (c_0 == first)
    float fan_area = 0.f;
    bool e04 = false;
#pragma unroll 1
    for (int i = 0; i < cnt; i++) {
        const int nxt = (i + 1 == cnt) ? 0 : i + 1;
        float xn, yn;
        ctab_xy(f, (uint32_t)((S.codes >> (6 * nxt)) & 63u), polyx[nxt * POLY_STRIDE], pxmin, pxmax, pymin, pymax, xn, yn);
        const uint32_t code = (uint32_t)((S.codes >> (6 * i)) & 15u);
        corner_grad_sel(f, code, xc, yc, 0.5f * (yn - ym), 0.5f * (xm - xn), g);
        if (i >= 2) {
            const float cr = (xm - fx) * (yc - fy) - (xc - fx) * (ym - fy);
            const float s_area = (float)(0.5 * (double)cr);                           // aa.h:93
            e04 = e04 || (s_area < 0);                                                // E04
            fan_area = fan_area + s_area;
        }
        xm = xc; ym = yc; xc = xn; yc = yn;
    }
    if (e04) return 1;
    if (fan_area > pix_area) return 6;
    area = fan_area;
    if (area == 0.0f) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = 0.f;
    }
    return 0;
#else
    TabState S;
    S.fan.fx = S.fan.fy = S.fan.px = S.fan.py = 0.f; S.fan.area = 0.f; S.fan.cnt = 0; S.fan.err = false; S.codes = 0;
    clip_edge_tab<0>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx, polyy);
    clip_edge_tab<1>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx, polyy);
    clip_edge_tab<2>(f, pxmin, pxmax, pymin, pymax, inside, S, polyx, polyy);
    if (S.fan.err) return 1;
#if DM2_CLIP_DEFERRED_AREA
    const int cnt = S.fan.cnt;
    if (cnt < 3) return 0;                        // no fan triangle: area 0 (the caller skips it)
    // one walk over the corners: shoelace gradient of corner i, and the reference's fan triangle (c_0, c_{i-1}, c_i)
    const float fx = polyx[0], fy = polyy[0];
    float xm = polyx[(cnt - 1) * POLY_STRIDE], ym = polyy[(cnt - 1) * POLY_STRIDE];   // c_{i-1}
    float xc = fx, yc = fy;                                                           // c_i   (c_0 == first)
    float fan_area = 0.f;
    bool e04 = false;
#pragma unroll 1
    for (int i = 0; i < cnt; i++) {
        const int nxt = (i + 1 == cnt) ? 0 : i + 1;
        const float xn = polyx[nxt * POLY_STRIDE], yn = polyy[nxt * POLY_STRIDE];
        const uint32_t code = (uint32_t)((S.codes >> (4 * i)) & 15u);
        corner_grad_sel(f, code, xc, yc, 0.5f * (yn - ym), 0.5f * (xm - xn), g);
        if (i >= 2) {
            const float cr = (xm - fx) * (yc - fy) - (xc - fx) * (ym - fy);
            const float s_area = (float)(0.5 * (double)cr);                           // aa.h:93
            e04 = e04 || (s_area < 0);                                                // E04
            fan_area = fan_area + s_area;
        }
        xm = xc; ym = yc; xc = xn; yc = yn;
    }
    if (e04) return 1;
    if (fan_area > pix_area) return 6;
    area = fan_area;
    if (area == 0.0f) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = 0.f;
    }
    return 0;
#else
    if (S.fan.area > pix_area) return 6;
    area = S.fan.area;
    const int cnt = S.fan.cnt;
    if (cnt < 3 || area == 0.0f) return 0;        // no fan triangle: zero Jacobian (and the caller skips area 0)
    // shoelace gradient, corner by corner
    float xm = polyx[(cnt - 1) * POLY_STRIDE], ym = polyy[(cnt - 1) * POLY_STRIDE];   // c_{i-1}
    float xc = S.fan.fx, yc = S.fan.fy;                                               // c_i   (c_0 == first)
#pragma unroll 1
    for (int i = 0; i < cnt; i++) {
        const int nxt = (i + 1 == cnt) ? 0 : i + 1;
        const float xn = polyx[nxt * POLY_STRIDE], yn = polyy[nxt * POLY_STRIDE];
        const uint32_t code = (uint32_t)((S.codes >> (4 * i)) & 15u);
        corner_grad_sel(f, code, xc, yc, 0.5f * (yn - ym), 0.5f * (xm - xn), g);
        xm = xc; ym = yc; xc = xn; yc = yn;
    }
    return 0;
#endif
#endif
}

}  // namespace dm2
