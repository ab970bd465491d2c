// dm2_clip_area.h -- overlap AREA of a CCW triangle and a unit pixel, branch-free.
//
// Forward-pass variant of aa.h:151-441 (no Jacobians).  Measured on gfx950 the polygon
// clip was 70 % of the forward's pair phase when written with per-lane branches: every
// lane of a wave clips a different (pixel,face) pair, so each small `if` became an
// exec-mask save/restore and the wave executed the union of all paths anyway.  Here the
// whole clip is straight-line code: all three triangle edges unrolled, every potential
// polygon corner ("emit site") evaluated under a predicate with v_cndmask selects, and
// the fan triangulation streamed through (first, previous) registers.
//
// The floating-point operations that reach the result are exactly the reference's, in
// its order: crossings t / iaxis1 (aa.h:250-252), validity tests (:258), the sort by t
// (:308-313), corner order (:361-379) and the fan sum of 0.5*cross with the double
// literal (:93, :404-413).  Error codes collapse to "non-zero" (callers test != 0 only).
#pragma once
#include "dm2_device_math.h"

namespace dm2 {

struct FanState {
    float fx, fy;       // first corner
    float px, py;       // previous corner
    float area;
    int cnt;
    bool err;
};

// Append corner (x,y) if `en` (aa.h:33-65), accounting the fan triangle (first, prev, (x,y))
// from the third corner on (aa.h:404-413).  Straight-line.
__device__ __forceinline__ void fan_push(FanState& S, bool en, float x, float y) {
    const bool over = en && (S.cnt >= MAX_POLY);                       // aa.h:45-48 -> error 5
    S.err = S.err || over;
    const bool tri = en && (S.cnt >= 2);
    const float cr = (S.px - S.fx) * (y - S.fy) - (x - S.fx) * (S.py - S.fy);
    const float s_area = (float)(0.5 * (double)cr);                    // aa.h:93
    S.err = S.err || (tri && (s_area < 0));                            // E04
    const float na = S.area + s_area;
    S.area = tri ? na : S.area;
    const bool isfirst = en && (S.cnt == 0);
    S.fx = isfirst ? x : S.fx; S.fy = isfirst ? y : S.fy;
    const bool setprev = en && (S.cnt >= 1);
    S.px = setprev ? x : S.px; S.py = setprev ? y : S.py;
    S.cnt += en ? 1 : 0;
}

template <int TI>
__device__ __forceinline__ void clip_edge_area(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                               uint32_t inside, FanState& S) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1], p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1], rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool e_vertical = (f.zmask >> (2 * TI)) & 1u, e_horizontal = (f.zmask >> (2 * TI + 1)) & 1u;
    const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
    const bool p1in = (p1x >= pxmin) && (p1x <= pxmax) && (p1y >= pymin) && (p1y <= pymax);
    // crossings with the pixel edges 0: y=pymin, 1: x=pxmax, 2: y=pymax, 3: x=pxmin (aa.h:230-258)
    const float tA = (pymin - p0y) * ry, xA = p0x + tA * ex;
    const float tB = (pxmax - p0x) * rx, yB = p0y + tB * ey;
    const float tC = (pymax - p0y) * ry, xC = p0x + tC * ex;
    const float tD = (pxmin - p0x) * rx, yD = p0y + tD * ey;
    const bool vA = (tA >= 0) && (tA <= 1) && (xA >= pxmin) && (xA <= pxmax) && !e_horizontal;
    const bool vB = (tB >= 0) && (tB <= 1) && (yB >= pymin) && (yB <= pymax) && !e_vertical;
    const bool vC = (tC >= 0) && (tC <= 1) && (xC >= pxmin) && (xC <= pxmax) && !e_horizontal;
    const bool vD = (tD >= 0) && (tD <= 1) && (yD >= pymin) && (yD <= pymax) && !e_vertical;
    const bool e00 = (vA && ((xA == pxmin) || (xA == pxmax))) || (vB && ((yB == pymin) || (yB == pymax))) ||
                     (vC && ((xC == pxmin) || (xC == pxmax))) || (vD && ((yD == pymin) || (yD == pymax)));   // aa.h:263-266
    const int n = (int)vA + (int)vB + (int)vC + (int)vD;
    // first valid crossing in pixel-edge order, and the last one (== second when n == 2)
    const float x0 = vA ? xA : (vB ? pxmax : (vC ? xC : pxmin));
    const float y0 = vA ? pymin : (vB ? yB : (vC ? pymax : yD));
    const float t0 = vA ? tA : (vB ? tB : (vC ? tC : tD));
    const int pe0 = vA ? 0 : (vB ? 1 : (vC ? 2 : 3));
    const float x1 = vD ? pxmin : (vC ? xC : pxmax);
    const float y1 = vD ? yD : (vC ? pymax : yB);
    const float t1 = vD ? tD : (vC ? tC : tB);
    const int pe1 = vD ? 3 : (vC ? 2 : 1);
    const bool two = (n == 2), one = (n == 1), none = (n == 0);
    const bool sw = two && (t0 > t1);                                                       // aa.h:308-313
    // E01 (n > 2), E02 (one crossing but both / neither end point inside), E03 (none, one end point inside)
    S.err = S.err || e00 || (n > 2) || (one && (p0in == p1in)) || (none && (p0in != p1in));
    // corners of this edge, in order: [crossing a] [crossing b | end point p1] [pixel corners ...]
    const bool en1 = two || one || (none && p0in && p1in);
    const float ax = two ? (sw ? x1 : x0) : (one ? x0 : p1x);
    const float ay = two ? (sw ? y1 : y0) : (one ? y0 : p1y);
    fan_push(S, en1 && !S.err, ax, ay);
    const bool en2 = two || (one && !p0in && p1in);
    const float bx = two ? (sw ? x0 : x1) : p1x;
    const float by = two ? (sw ? y0 : y1) : p1y;
    fan_push(S, en2 && !S.err, bx, by);
    // pixel corners inside the triangle, counter-clockwise from the edge's exit (aa.h:359-379)
    const bool walk = two || (one && p0in && !p1in);
    const int final_pe = two ? (sw ? pe0 : pe1) : pe0;
    bool go = walk && !S.err;
#pragma unroll
    for (int pvi = 0; pvi < 4; pvi++) {
        const int cur = (final_pe + 1 + pvi) & 3;
        go = go && ((inside >> cur) & 1u);
        // `go` only ever falls: once no lane of the wave walks on, the remaining corner sites are dead for all of
        // them (a wave-uniform skip; two or three consecutive inside corners are rare with small triangles)
        if (__ballot(go) == 0ull) break;
        const float cx = (cur == 1 || cur == 2) ? pxmax : pxmin;
        const float cy = (cur >= 2) ? pymax : pymin;
        fan_push(S, go, cx, cy);
    }
}

// aa.h:103-149: per pixel corner, is it inside all three half planes (bit i of `inside`, corners in the
// order (xmin,ymin) (xmax,ymin) (xmax,ymax) (xmin,ymax)).  Returns false when some edge has all four
// corners on its outer side -- the reference then reports area 0.
__device__ __forceinline__ bool classify_pixel(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax, uint32_t& inside) {
    inside = 0xF;
    bool outside = false;
#pragma unroll
    for (int ti = 0; ti < 3; ti++) {
        const float nx = f.n[2 * ti], ny = f.n[2 * ti + 1], c = f.c[ti];
        const bool i0 = (pxmin * nx) + (pymin * ny) - c >= 0;
        const bool i1 = (pxmax * nx) + (pymin * ny) - c >= 0;
        const bool i2 = (pxmax * nx) + (pymax * ny) - c >= 0;
        const bool i3 = (pxmin * nx) + (pymax * ny) - c >= 0;
        outside = outside || !(i0 || i1 || i2 || i3);
        inside &= (uint32_t)i0 | ((uint32_t)i1 << 1) | ((uint32_t)i2 << 2) | ((uint32_t)i3 << 3);
    }
    return !outside;
}

// aa.h:151-441 for a pixel that passed classify_pixel.  Returns non-zero on any reference error.
__device__ __forceinline__ int clip_area_classified(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                    uint32_t inside, float pix_area, float& area) {
    area = 0.f;
    if (inside == 0xF) { area = pix_area; return 0; }
#ifdef DM2_ABLATE_CLIP   // diagnostic only: price of the polygon clip
    area = 0.5f * pix_area; return 0;
#endif
    FanState S;
    S.fx = S.fy = S.px = S.py = 0.f; S.area = 0.f; S.cnt = 0; S.err = false;
    clip_edge_area<0>(f, pxmin, pxmax, pymin, pymax, inside, S);
    clip_edge_area<1>(f, pxmin, pxmax, pymin, pymax, inside, S);
    clip_edge_area<2>(f, pxmin, pxmax, pymin, pymax, inside, S);
    if (S.err) return 1;
    if (S.area > pix_area) return 6;                                                                    // E05
    area = S.area;
    return 0;
}

// aa.h:446-504, area only.  Returns non-zero on any reference error; area valid when 0.
__device__ __forceinline__ int tri_pix_overlap_area_only(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                         float pix_area, float& area) {
    area = 0.f;
    if ((pxmax < f.bb[0]) || (pxmin > f.bb[1]) || (pymax < f.bb[2]) || (pymin > f.bb[3])) return 0;   // aa.h:96-101
    uint32_t inside;
    if (!classify_pixel(f, pxmin, pxmax, pymin, pymax, inside)) return 0;
    return clip_area_classified(f, pxmin, pxmax, pymin, pymax, inside, pix_area, area);
}

}  // namespace dm2
