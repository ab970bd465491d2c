// dm2_clip_lds.h -- triangle/pixel clipper with a per-lane polygon table in LDS.
//
// Same function as tri_pix_overlap_area<> in dm2_device_math.h (aa.h:446-504), built
// for the dense pair kernels where every lane clips a DIFFERENT (pixel,face) pair:
//   * polygon corners go into a per-lane, lane-strided LDS table (x,y only); what kind
//     of corner each one is (pixel corner / triangle vertex / edge crossing, which
//     triangle edge) is a 4-bit code in a register.  The fan loop (aa.h:404-434) then
//     runs polygon_size-2 iterations per lane -- typically 1-4 -- instead of the 18
//     inlined "maybe append + account" sites of the register-streaming version.
//   * Jacobians of edge crossings are re-derived inside the fan loop from the corner
//     itself: the crossing's parameter t is recomputed by the SAME expression from the
//     same operands (the pixel-edge coordinate is stored exactly in the corner), so it
//     is bit-identical to the value the reference keeps in its table.
//   * Products with the structural zeros / ones of the Jacobians (aa.h:280-294, zero2,
//     eye2) are not issued.  For finite coordinates this changes at most the sign of an
//     exact zero; the CPU oracle keeps the literal form.
// Results (area, error/no-error, gradient) are otherwise bit-identical to the oracle.
#pragma once
#include "dm2_device_math.h"

namespace dm2 {

constexpr int POLY_STRIDE = TILE_PIX;           // table is [MAX_POLY][TILE_PIX], lane-strided

// corner codes: bits 0-1 kind, bits 2-3 triangle edge
constexpr uint32_t PK_CORNER = 0, PK_TRIV = 1, PK_XH = 2 /* crossing of a y=const pixel edge */, PK_XV = 3;

struct ClipOut { float area; float g[6]; };

__device__ __forceinline__ void rows_add(float* g, int row, float a0, float a1) {
    g[0] = row == 0 ? g[0] + a0 : g[0]; g[1] = row == 0 ? g[1] + a1 : g[1];
    g[2] = row == 1 ? g[2] + a0 : g[2]; g[3] = row == 1 ? g[3] + a1 : g[3];
    g[4] = row == 2 ? g[4] + a0 : g[4]; g[5] = row == 2 ? g[5] + a1 : g[5];
}

// aa.h:67-86 for one polygon corner (x,y) of kind `code`, d(area)/d(corner) = (gax, gay).
__device__ __forceinline__ void corner_grad(const AAFace& f, uint32_t code, float x, float y, float gax, float gay, float* g) {
    const uint32_t kind = code & 3u;
    if (kind == PK_CORNER) return;                       // Jacobian (0,0): contributes +-0 only
    const int ti = (int)(code >> 2);
    const int tj = ti == 2 ? 0 : ti + 1;
    if (kind == PK_TRIV) { rows_add(g, tj, gax, gay); return; }      // Jacobian (0, I)
    const float ex = f.e[2 * ti], ey = f.e[2 * ti + 1];
    float a0, a1, b0, b1;
    if (kind == PK_XH) {                                 // axis0 = y: iaxis0 = y (pixel edge), t from y
        const float p0y = f.v[2 * ti + 1], p1y = f.v[2 * tj + 1], ry = f.r[2 * ti + 1];
        const float t = (y - p0y) * ry;
        const float gt0 = (y - p1y) * ry * ry;
        const float gt1 = (-y + p0y) * ry * ry;
        const float omt = (float)(1.0 - (double)t);
        a0 = omt * gax;                                          // g0 = [omt, 0; gt0*ex, (1-t)+gt0*ey]
        a1 = (gt0 * ex) * gax + one_minus_t_plus(t, gt0 * ey) * gay;
        b0 = t * gax;                                            // g1 = [t, 0; gt1*ex, t+gt1*ey]
        b1 = (gt1 * ex) * gax + (t + (gt1 * ey)) * gay;
    } else {                                             // axis0 = x
        const float p0x = f.v[2 * ti], p1x = f.v[2 * tj], rx = f.r[2 * ti];
        const float t = (x - p0x) * rx;
        const float gt0 = (x - p1x) * rx * rx;
        const float gt1 = (-x + p0x) * rx * rx;
        const float omt = (float)(1.0 - (double)t);
        a0 = one_minus_t_plus(t, gt0 * ex) * gax + (gt0 * ey) * gay;     // g0 = [(1-t)+gt0*ex, gt0*ey; 0, omt]
        a1 = omt * gay;
        b0 = (t + (gt1 * ex)) * gax + (gt1 * ey) * gay;                  // g1 = [t+gt1*ex, gt1*ey; 0, t]
        b1 = t * gay;
    }
    rows_add(g, ti, a0, a1);
    rows_add(g, tj, b0, b1);
}

// polyx/polyy: this lane's column of the LDS table (element v at [v * POLY_STRIDE]).
// Returns non-zero on any reference error code.  area / g valid only when 0 is returned.
template <bool GRAD>
__device__ __forceinline__ int tri_pix_overlap_area_lds(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                        float pix_area, float* polyx, float* polyy, float& area, float* g) {
    area = 0.f;
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = 0.f;
    }
    if ((pxmax < f.bb[0]) || (pxmin > f.bb[1]) || (pymax < f.bb[2]) || (pymin > f.bb[3])) return 0;   // aa.h:96-101
    uint32_t inside = 0xF;
    bool outside = false;
#pragma unroll
    for (int ti = 0; ti < 3; ti++) {                                                                    // aa.h:103-149
        const float nx = f.n[2 * ti], ny = f.n[2 * ti + 1], c = f.c[ti];
        const bool i0 = (pxmin * nx) + (pymin * ny) - c >= 0;
        const bool i1 = (pxmax * nx) + (pymin * ny) - c >= 0;
        const bool i2 = (pxmax * nx) + (pymax * ny) - c >= 0;
        const bool i3 = (pxmin * nx) + (pymax * ny) - c >= 0;
        outside = outside || !(i0 || i1 || i2 || i3);
        inside &= (uint32_t)i0 | ((uint32_t)i1 << 1) | ((uint32_t)i2 << 2) | ((uint32_t)i3 << 3);
    }
    if (outside) return 0;
    if (inside == 0xF) { area = pix_area; return 0; }
#ifdef DM2_ABLATE_CLIP   // diagnostic only: price of the polygon clip
    area = 0.5f * pix_area; return 0;
#endif

    int cnt = 0;
    uint64_t codes = 0;
    bool err = false;
    auto push = [&](float x, float y, uint32_t code) {
        if (cnt >= MAX_POLY) { err = true; return; }                   // aa.h:45-48
        polyx[cnt * POLY_STRIDE] = x; polyy[cnt * POLY_STRIDE] = y;
        codes |= (uint64_t)code << (4 * cnt);
        cnt++;
    };

#pragma unroll 1
    for (int ti = 0; ti < 3 && !err; ti++) {                                                            // aa.h:206-401
        const int tj = ti == 2 ? 0 : ti + 1;
        const float p0x = f.v[2 * ti], p0y = f.v[2 * ti + 1], p1x = f.v[2 * tj], p1y = f.v[2 * tj + 1];
        const float ex = f.e[2 * ti], ey = f.e[2 * ti + 1], rx = f.r[2 * ti], ry = f.r[2 * ti + 1];
        const bool e_vertical = (f.zmask >> (2 * ti)) & 1u, e_horizontal = (f.zmask >> (2 * ti + 1)) & 1u;
        const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
        const bool p1in = (p1x >= pxmin) && (p1x <= pxmax) && (p1y >= pymin) && (p1y <= pymax);
        // crossings with the 4 pixel edges: 0: y=pymin, 1: x=pxmax, 2: y=pymax, 3: x=pxmin
        const float tA = (pymin - p0y) * ry, xA = p0x + tA * ex;
        const float tB = (pxmax - p0x) * rx, yB = p0y + tB * ey;
        const float tC = (pymax - p0y) * ry, xC = p0x + tC * ex;
        const float tD = (pxmin - p0x) * rx, yD = p0y + tD * ey;
        const bool vA = (tA >= 0) && (tA <= 1) && (xA >= pxmin) && (xA <= pxmax) && !e_horizontal;
        const bool vB = (tB >= 0) && (tB <= 1) && (yB >= pymin) && (yB <= pymax) && !e_vertical;
        const bool vC = (tC >= 0) && (tC <= 1) && (xC >= pxmin) && (xC <= pxmax) && !e_horizontal;
        const bool vD = (tD >= 0) && (tD <= 1) && (yD >= pymin) && (yD <= pymax) && !e_vertical;
        // E00: a crossing exactly on a pixel corner (aa.h:263-266)
        if ((vA && ((xA == pxmin) || (xA == pxmax))) || (vB && ((yB == pymin) || (yB == pymax))) ||
            (vC && ((xC == pxmin) || (xC == pxmax))) || (vD && ((yD == pymin) || (yD == pymax)))) { err = true; break; }
        const int n = (int)vA + (int)vB + (int)vC + (int)vD;
        if (n > 2) { err = true; break; }                                                               // E01
        // first / second valid crossing in pixel-edge order
        float x0, y0, t0; int pe0; uint32_t k0;
        if (vA) { x0 = xA; y0 = pymin; t0 = tA; pe0 = 0; k0 = PK_XH; }
        else if (vB) { x0 = pxmax; y0 = yB; t0 = tB; pe0 = 1; k0 = PK_XV; }
        else if (vC) { x0 = xC; y0 = pymax; t0 = tC; pe0 = 2; k0 = PK_XH; }
        else { x0 = pxmin; y0 = yD; t0 = tD; pe0 = 3; k0 = PK_XV; }
        const uint32_t ecode = (uint32_t)ti << 2;
        int final_pe = -1;
        if (n == 2) {
            float x1, y1, t1; int pe1; uint32_t k1;
            if (vD) { x1 = pxmin; y1 = yD; t1 = tD; pe1 = 3; k1 = PK_XV; }
            else if (vC) { x1 = xC; y1 = pymax; t1 = tC; pe1 = 2; k1 = PK_XH; }
            else { x1 = pxmax; y1 = yB; t1 = tB; pe1 = 1; k1 = PK_XV; }
            const bool sw = t0 > t1;                                                                    // aa.h:308-313
            push(sw ? x1 : x0, sw ? y1 : y0, (sw ? k1 : k0) | ecode);
            push(sw ? x0 : x1, sw ? y0 : y1, (sw ? k0 : k1) | ecode);
            final_pe = sw ? pe0 : pe1;
        } else if (n == 1) {
            push(x0, y0, k0 | ecode);
            if (!p0in && p1in) push(p1x, p1y, PK_TRIV | ecode);
            else if (p0in && !p1in) final_pe = pe0;
            else { err = true; break; }                                                                 // E02
        } else {
            if (p0in && p1in) push(p1x, p1y, PK_TRIV | ecode);
            else if (!p0in && !p1in) { /* edge misses the pixel */ }
            else { err = true; break; }                                                                 // E03
        }
        if (final_pe != -1) {                                                                           // aa.h:359-379
            const int start = (final_pe + 1) & 3;
#pragma unroll 1
            for (int pvi = 0; pvi < 4; pvi++) {
                const int cur = (start + pvi) & 3;
                if (!((inside >> cur) & 1u)) break;
                push((cur == 1 || cur == 2) ? pxmax : pxmin, (cur >= 2) ? pymax : pymin, PK_CORNER);
            }
        }
    }
    if (err) return 1;

    // fan triangulation from corner 0 (aa.h:404-434)
    float acc_area = 0.f;
    if (cnt >= 3) {
        const float ax = polyx[0], ay = polyy[0];
        const uint32_t ca = (uint32_t)(codes & 15u);
        float bx = polyx[POLY_STRIDE], by = polyy[POLY_STRIDE];
        uint32_t cb = (uint32_t)((codes >> 4) & 15u);
#pragma unroll 1
        for (int si = 0; si < cnt - 2; si++) {
            const float cx = polyx[(si + 2) * POLY_STRIDE], cy = polyy[(si + 2) * POLY_STRIDE];
            const uint32_t cc = (uint32_t)((codes >> (4 * (si + 2))) & 15u);
            const float cr = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay);
            const float s_area = (float)(0.5 * (double)cr);                                             // aa.h:93
            if (s_area < 0) return 5;                                                                   // E04
            acc_area += s_area;
            if (GRAD) {
                corner_grad(f, ca, ax, ay, 0.5f * (by - cy), 0.5f * (-bx + cx), g);
                corner_grad(f, cb, bx, by, 0.5f * (cy - ay), 0.5f * (-cx + ax), g);
                corner_grad(f, cc, cx, cy, 0.5f * (ay - by), 0.5f * (-ax + bx), g);
            }
            bx = cx; by = cy; cb = cc;
        }
    }
    if (acc_area > pix_area) return 6;                                                                  // E05
    area = acc_area;
    return 0;
}

}  // namespace dm2
