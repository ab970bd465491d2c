// dm2_device_math.h -- per-(pixel,face) math of the rasterizer, gfx950 device code.
//
// Written for CDNA4: everything lives in VGPRs (no runtime-indexed local arrays,
// hence no scratch), the polygon clipper is a streaming triangle fan instead of
// the reference's vertex table, and the caller keeps the per-face tables in LDS.
// Results are bit-identical to an unfused (-ffp-contract=off) evaluation of the
// reference formulas; the mixed float/double promotions the reference's literals
// imply are spelled out.  Reference lines are cited per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dm2 {

constexpr int TILE = 16;                 // config.h:4-5
constexpr int TILE_PIX = TILE * TILE;    // auxiliary.h:11 BLOCK_SIZE
constexpr float T_EPS = 0.0001f;         // auxiliary.h:9
constexpr int MAX_POLY = 10;             // aa.h:11

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
__device__ __forceinline__ f3 operator*(float b, f3 a) { return {b * a.x, b * a.y, b * a.z}; }
__device__ __forceinline__ f3 operator/(f3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---- primary ray of a pixel, computed instead of read (DM2_FLAG_ANALYTIC_RAYS).  Operation order of the reference's
// Renderer._init_rays (__init__.py:198-237): pixel centre (x + 0.5) / W * 2 - 1 -> (ndc_x, ndc_y, -1, 1), row vector times
// inv(proj)^T, times inv(mv)^T (4-term sums in index order: the reference leaves the order to its BLAS), NO perspective
// divide, origin = inv(mv)[:3, 3], direction = (target - origin) / (|.| + 1e-6).  cam: inv(mv) (16) then inv(proj) (16).
__device__ __forceinline__ void analytic_ray(const float* __restrict__ cam, float x_abs, float y_abs, float Wf, float Hf, f3& ro, f3& rd) {
    const float* imv = cam;
    const float* ipr = cam + 16;
    const float h[4] = {((x_abs + 0.5f) / Wf * 2.0f) - 1.0f, ((y_abs + 0.5f) / Hf * 2.0f) - 1.0f, -1.0f, 1.0f};
    float v[4], w[3];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = ((h[0] * ipr[4 * j] + h[1] * ipr[4 * j + 1]) + h[2] * ipr[4 * j + 2]) + h[3] * ipr[4 * j + 3];
#pragma unroll
    for (int j = 0; j < 3; j++) w[j] = ((v[0] * imv[4 * j] + v[1] * imv[4 * j + 1]) + v[2] * imv[4 * j + 2]) + v[3] * imv[4 * j + 3];
    ro = {imv[3], imv[7], imv[11]};
    const f3 dd = {w[0] - ro.x, w[1] - ro.y, w[2] - ro.z};
    const float len = sqrtf((dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z) + 1e-6f;
    rd = {dd.x / len, dd.y / len, dd.z / len};
}

// ---- Moeller-Trumbore without inside test (auxiliary.h:212-243) -------------
__device__ __forceinline__ bool ray_tri_intersection(f3 ro, f3 rd, f3 p0, f3 p1, f3 p2, f3& tuv) {
    f3 T = ro - p0, E1 = p1 - p0, E2 = p2 - p0;
    f3 P = cross(rd, E2), Q = cross(T, E1);
    float denom = dot(P, E1);
    if (denom == 0.0f) return false;
    float inv_denom = 1.0f / denom;
    tuv.x = dot(Q, E2) * inv_denom;
    tuv.y = dot(P, T) * inv_denom;
    tuv.z = dot(Q, rd) * inv_denom;
    return true;
}

// ---- auxiliary.h:245-290.  As written the "dv" outputs are grad(t), not grad(v);
// corrected=true computes the true grad(v) (opt-in flag DM2_FLAG_CORRECTED_DV).
// 1 / x to <= 1 ulp without the IEEE division's ~19 instructions (v_rcp_f32 + one Newton step): for gradient terms, which owe
// the reference 1e-5, not its bits
__device__ __forceinline__ float rcp_refined(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

template <bool FAST_RCP = false>
__device__ __forceinline__ void ray_tri_intersection_grad(f3 ro, f3 rd, f3 p0, f3 p1, f3 p2, bool corrected,
                                                          f3& du_dp0, f3& du_dp1, f3& du_dp2,
                                                          f3& dv_dp0, f3& dv_dp1, f3& dv_dp2) {
    f3 T = ro - p0, E1 = p1 - p0, E2 = p2 - p0;
    f3 dxE2 = cross(rd, E2);
    float denom_sqrt = dot(dxE2, E1);
    float denom = denom_sqrt * denom_sqrt;
    float denom_inv = FAST_RCP ? rcp_refined(denom) : 1.0f / denom;          // the reference clamps denom AFTER this (dead clamp)
    float v0 = dot(dxE2, T);
    float v1 = denom_sqrt;
    f3 E1xd = cross(E1, rd);
    f3 du_dE1 = ((-1.0f * dxE2) * v0) * denom_inv;
    f3 du_dE2 = (cross(T, rd) * v1 - v0 * E1xd) * denom_inv;
    f3 du_dT = (dxE2 * v1) * denom_inv;
    f3 dv_dE1, dv_dE2, dv_dT;
    if (!corrected) {
        float v2 = dot(cross(T, E1), E2);
        dv_dE1 = ((cross(E2, T) * v1) - (v2 * dxE2)) * denom_inv;
        dv_dE2 = ((cross(T, E1) * v1) - (v2 * E1xd)) * denom_inv;
        dv_dT = (cross(E1, E2) * v1) * denom_inv;
    } else {
        float N = dot(cross(T, E1), rd);
        dv_dE1 = ((cross(rd, T) * v1) - (N * dxE2)) * denom_inv;
        dv_dE2 = ((-N) * E1xd) * denom_inv;
        dv_dT = (E1xd * v1) * denom_inv;
    }
    du_dp0 = -du_dE1 - du_dE2 - du_dT;
    dv_dp0 = -dv_dE1 - dv_dE2 - dv_dT;
    du_dp1 = du_dE1; dv_dp1 = dv_dE1;
    du_dp2 = du_dE2; dv_dp2 = dv_dE2;
}

// ---- auxiliary.h:292-329 ------------------------------------------------------
__device__ __forceinline__ void clamp_bary_uv(float u, float v, float& u_c, float& v_c, int& code) {
    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) { u_c = u; v_c = v; code = 0; }
    else if (u <= 0.0f && v <= 0.0f) { u_c = 0.0f; v_c = 0.0f; code = 1; }
    else if ((u >= 1.0f && v <= 0.0f) || (v >= 0.0f && v <= u - 1.0f)) { u_c = 1.0f; v_c = 0.0f; code = 2; }
    else if ((u <= 0.0f && v >= 1.0f) || (u >= 0.0f && v >= u + 1.0f)) { u_c = 0.0f; v_c = 1.0f; code = 3; }
    else if (u <= 0.0f && v <= 1.0f && v >= 0.0f) { u_c = 0.0f; v_c = v; code = 4; }
    else if (u <= 1.0f && u >= 0.0f && v <= 0.0f) { u_c = u; v_c = 0.0f; code = 5; }
    else { u_c = (1.0f + u - v) * 0.5f; v_c = (1.0f - u + v) * 0.5f; code = 6; }
}

// ---- auxiliary.h:331-357 ------------------------------------------------------
__device__ __forceinline__ void clamp_bary_uv_grad(int code, float& duc_du, float& duc_dv, float& dvc_du, float& dvc_dv) {
    dvc_du = 0.0f; duc_dv = 0.0f;
    if (code == 0) { duc_du = 1.0f; dvc_dv = 1.0f; }
    else if (code == 1 || code == 2 || code == 3) { duc_du = 0.0f; dvc_dv = 0.0f; }
    else if (code == 4) { duc_du = 0.0f; dvc_dv = 1.0f; }
    else if (code == 5) { duc_du = 1.0f; dvc_dv = 0.0f; }
    else { duc_du = 0.5f; dvc_du = -0.5f; duc_dv = -0.5f; dvc_dv = 0.5f; }
}

// coverage mix, forward.cu:375-378 / backward.cu:313-316: the 1.0 / 0.0 literals
// make the sum a double expression rounded once to float.
__device__ __forceinline__ float mix_coverage(int code, float ratio, float temp) {
    if (code == 0) return (float)(1.0 * (double)(1.0f - temp) + (double)(ratio * temp));
    return (float)(0.0 * (double)(1.0f - temp) + (double)(ratio * temp));
}

// float -> int the way CUDA's cvt.rzi.s32.f32 does (NaN -> 0, saturating)
__device__ __forceinline__ int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}

// ---- auxiliary.h:72-92: half-open tile rect of a triangle inside the patch ----
__device__ __forceinline__ void patch_rect_from_tri(uint32_t pmx, uint32_t pmy, float p0x, float p0y, float p1x, float p1y,
                                                    float p2x, float p2y, uint32_t gx, uint32_t gy,
                                                    uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1) {
    float min_x = fminf(fminf(p0x, p1x), p2x) - (float)pmx;
    float min_y = fminf(fminf(p0y, p1y), p2y) - (float)pmy;
    float max_x = fmaxf(fmaxf(p0x, p1x), p2x) - (float)pmx;
    float max_y = fmaxf(fmaxf(p0y, p1y), p2y) - (float)pmy;
    int ix0 = f2i_sat(floorf(min_x / TILE)), iy0 = f2i_sat(floorf(min_y / TILE));
    int ix1 = f2i_sat(ceilf(max_x / TILE)), iy1 = f2i_sat(ceilf(max_y / TILE));
    x0 = min(gx, (uint32_t)max(0, ix0)); y0 = min(gy, (uint32_t)max(0, iy0));
    x1 = min(gx, (uint32_t)max(0, ix1)); y1 = min(gy, (uint32_t)max(0, iy1));
}

// =============================================================================
// AA: overlap area of a CCW triangle with a unit pixel, + d(area)/d(tri verts)
// (aa.h:15-504).  Per-face tables as the caller staged them (LDS or registers).
// =============================================================================
struct AAFace {    // (member order: what the default backward reads -- v, e, r, zmask -- fills the record's first five 16-byte parts)
    float v[6];     // aa_face_verts        [3][2]
    float e[6];     // aa_face_edges        [3][2]
    float r[6];     // aa_face_edges_recip  [3][2]
    uint32_t zmask; // bit (2*i+k) = aa_face_edges_iszero[i][k]
    float c[3];     // aa_face_edges_normal_c
    float n[6];     // aa_face_edges_normal [3][2]
    float bb[4];    // txmin, txmax, tymin, tymax  (aa_face_verts.min/max over corners, forward.cu:480-481)
};
static_assert(sizeof(AAFace) == 128, "AAFace: 32 dwords");

struct PolyVert {   // one vertex of the clipped polygon and its Jacobians w.r.t. the
    float x, y;     // two end points (i0, i0+1 mod 3) of triangle edge `idx` (-1: pixel corner)
    int idx;
    float g0[4], g1[4];
};

template <bool GRAD>
struct PolyAcc {
    PolyVert first, prev;
    int cnt;
    float area;
    float g[6];     // d(area)/d(v[3][2]), row-major
    bool err;
};

__device__ __forceinline__ float one_minus_t_plus(float t, float prod) {   // aa.h:286,289: "(1.0 - t) + (g*e)"
    return (float)((1.0 - (double)t) + (double)prod);
}

__device__ __forceinline__ void grad_rows_add(float* g, int row, float a0, float a1) {
    float s0 = g[0] + a0, s1 = g[1] + a1, s2 = g[2] + a0, s3 = g[3] + a1, s4 = g[4] + a0, s5 = g[5] + a1;
    g[0] = row == 0 ? s0 : g[0]; g[1] = row == 0 ? s1 : g[1];
    g[2] = row == 1 ? s2 : g[2]; g[3] = row == 1 ? s3 : g[3];
    g[4] = row == 2 ? s4 : g[4]; g[5] = row == 2 ? s5 : g[5];
}

// aa.h:67-86.  idx == -1 (pixel corner): the reference's write to row -1 is out of
// bounds and adds an all-zero product; it is dropped, the "+= 0" into row 0 stays.
__device__ __forceinline__ void update_grad(float* g, const PolyVert& v, float gax, float gay) {
    float a0 = v.g0[0] * gax + v.g0[1] * gay;
    float a1 = v.g0[2] * gax + v.g0[3] * gay;
    float b0 = v.g1[0] * gax + v.g1[1] * gay;
    float b1 = v.g1[2] * gax + v.g1[3] * gay;
    int i0 = v.idx;
    int i1 = (i0 == 2) ? 0 : i0 + 1;
    grad_rows_add(g, i0, a0, a1);
    grad_rows_add(g, i1, b0, b1);
}

// Append a polygon vertex (aa.h:33-65) and, from the third vertex on, account the fan
// triangle (first, prev, v) exactly as the reference's area loop does (aa.h:404-434).
template <bool GRAD>
__device__ __forceinline__ void poly_emit(PolyAcc<GRAD>& A, const PolyVert& v) {
    if (A.cnt >= MAX_POLY) { A.err = true; return; }
    if (A.cnt == 0) A.first = v;
    else if (A.cnt >= 2) {
        const PolyVert& ip0 = A.first; const PolyVert& ip1 = A.prev; const PolyVert& ip2 = v;
        float cr = (ip1.x - ip0.x) * (ip2.y - ip0.y) - (ip2.x - ip0.x) * (ip1.y - ip0.y);
        float s_area = (float)(0.5 * (double)cr);                       // aa.h:93
        if (s_area < 0) A.err = true;                                    // E04
        A.area += s_area;
        if (GRAD) {
            update_grad(A.g, ip0, 0.5f * (ip1.y - ip2.y), 0.5f * (-ip1.x + ip2.x));
            update_grad(A.g, ip1, 0.5f * (ip2.y - ip0.y), 0.5f * (-ip2.x + ip0.x));
            update_grad(A.g, ip2, 0.5f * (ip0.y - ip1.y), 0.5f * (-ip0.x + ip1.x));
        }
    }
    if (A.cnt >= 1) A.prev = v;
    A.cnt++;
}

// One triangle edge TI against the 4 pixel edges (aa.h:206-401).
template <int TI, bool GRAD>
__device__ __forceinline__ void clip_tri_edge(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                              uint32_t inside_mask, PolyAcc<GRAD>& A) {
    constexpr int TJ = (TI + 1) % 3;
    const float p0x = f.v[2 * TI], p0y = f.v[2 * TI + 1];
    const float p1x = f.v[2 * TJ], p1y = f.v[2 * TJ + 1];
    const float ex = f.e[2 * TI], ey = f.e[2 * TI + 1];
    const float rx = f.r[2 * TI], ry = f.r[2 * TI + 1];
    const bool e_vertical = (f.zmask >> (2 * TI)) & 1u;        // edges_iszero[ti][0]
    const bool e_horizontal = (f.zmask >> (2 * TI + 1)) & 1u;  // edges_iszero[ti][1]
    const bool p0in = (p0x >= pxmin) && (p0x <= pxmax) && (p0y >= pymin) && (p0y <= pymax);
    const bool p1in = (p1x >= pxmin) && (p1x <= pxmax) && (p1y >= pymin) && (p1y <= pymax);

    PolyVert s0, s1;
    float t0 = 0.f, t1 = 0.f;
    int pe0 = -1, pe1 = -1, n = 0;
    s0.x = s0.y = s1.x = s1.y = 0.f; s0.idx = s1.idx = TI;
#pragma unroll
    for (int k = 0; k < 4; k++) { s0.g0[k] = s0.g1[k] = s1.g0[k] = s1.g1[k] = 0.f; }

#pragma unroll
    for (int pi = 0; pi < 4; pi++) {
        const bool horiz = (pi == 0) || (pi == 2);      // pixel edge y = const -> intersect along y
        const float iaxis0 = pi == 0 ? pymin : (pi == 1 ? pxmax : (pi == 2 ? pymax : pxmin));
        const bool parallel = horiz ? e_horizontal : e_vertical;
        const float pmin1 = horiz ? pxmin : pymin, pmax1 = horiz ? pxmax : pymax;
        const float p0a0 = horiz ? p0y : p0x, p0a1 = horiz ? p0x : p0y;
        const float p1a0 = horiz ? p1y : p1x;
        const float ra0 = horiz ? ry : rx;
        const float ea1 = horiz ? ex : ey;
        const float t = (iaxis0 - p0a0) * ra0;
        const float iaxis1 = p0a1 + t * ea1;
        const bool valid = (t >= 0) && (t <= 1) && (iaxis1 >= pmin1) && (iaxis1 <= pmax1) && (!parallel);
        if (valid) {
            if ((iaxis1 == pmin1) || (iaxis1 == pmax1)) A.err = true;          // E00
            PolyVert nv;
            nv.x = horiz ? iaxis1 : iaxis0;
            nv.y = horiz ? iaxis0 : iaxis1;
            nv.idx = TI;
            if (GRAD) {
                const float gt0 = (iaxis0 - p1a0) * ra0 * ra0;
                const float gt1 = (-iaxis0 + p0a0) * ra0 * ra0;
                const float g0x = horiz ? 0.0f : gt0, g0y = horiz ? gt0 : 0.0f;   // grad_t_p0[2]
                const float g1x = horiz ? 0.0f : gt1, g1y = horiz ? gt1 : 0.0f;   // grad_t_p1[2]
                nv.g0[0] = one_minus_t_plus(t, g0x * ex);
                nv.g0[1] = g0x * ey;
                nv.g0[2] = g0y * ex;
                nv.g0[3] = one_minus_t_plus(t, g0y * ey);
                nv.g1[0] = t + (g1x * ex);
                nv.g1[1] = g1x * ey;
                nv.g1[2] = g1y * ex;
                nv.g1[3] = t + (g1y * ey);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) { nv.g0[k] = 0.f; nv.g1[k] = 0.f; }
            }
            if (n == 0) { s0 = nv; t0 = t; pe0 = pi; }
            else if (n == 1) { s1 = nv; t1 = t; pe1 = pi; }
            n++;
        }
    }
    if (n > 2) { A.err = true; return; }                                          // E01

    PolyVert tp1;                                   // the edge's end point, Jacobian (0, I)
    tp1.x = p1x; tp1.y = p1y; tp1.idx = TI;
    tp1.g0[0] = tp1.g0[1] = tp1.g0[2] = tp1.g0[3] = 0.f;
    tp1.g1[0] = 1.f; tp1.g1[1] = 0.f; tp1.g1[2] = 0.f; tp1.g1[3] = 1.f;

    int final_pe = -1;
    if (n == 2) {
        const bool sw = t0 > t1;
        poly_emit<GRAD>(A, sw ? s1 : s0);
        poly_emit<GRAD>(A, sw ? s0 : s1);
        final_pe = sw ? pe0 : pe1;
    } else if (n == 1) {
        poly_emit<GRAD>(A, s0);
        if (!p0in && p1in) poly_emit<GRAD>(A, tp1);
        else if (p0in && !p1in) final_pe = pe0;
        else { A.err = true; return; }                                            // E02
    } else {
        if (p0in && p1in) poly_emit<GRAD>(A, tp1);
        else if (!p0in && !p1in) { /* edge misses the pixel */ }
        else { A.err = true; return; }                                            // E03
    }
    if (final_pe != -1) {                            // walk pixel corners inside the triangle (aa.h:359-379)
        const int start = (final_pe + 1) & 3;
#pragma unroll 1
        for (int pvi = 0; pvi < 4; pvi++) {
            const int cur = (start + pvi) & 3;
            if (!((inside_mask >> cur) & 1u)) break;
            PolyVert cv;
            cv.x = (cur == 1 || cur == 2) ? pxmax : pxmin;
            cv.y = (cur >= 2) ? pymax : pymin;
            cv.idx = -1;
#pragma unroll
            for (int k = 0; k < 4; k++) { cv.g0[k] = 0.f; cv.g1[k] = 0.f; }
            poly_emit<GRAD>(A, cv);
        }
    }
}

// aa.h:446-504.  Returns non-zero on any of the reference's error codes 1..6
// (callers only test != 0).  area/g are meaningful only when the return is 0.
template <bool GRAD>
__device__ __forceinline__ int tri_pix_overlap_area(const AAFace& f, float pxmin, float pxmax, float pymin, float pymax,
                                                    float pix_area, float& area, float* g /*[6] or null*/) {
    area = 0.f;
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = 0.f;
    }
    if ((pxmax < f.bb[0]) || (pxmin > f.bb[1]) || (pymax < f.bb[2]) || (pymin > f.bb[3])) return 0;   // aa.h:96-101
    // corner-in-half-plane tests (aa.h:103-149); corners: (min,min) (max,min) (max,max) (min,max)
    uint32_t inside = 0xF;
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
    if (outside) return 0;
    if (inside == 0xF) { area = pix_area; return 0; }

    PolyAcc<GRAD> A;
    A.cnt = 0; A.area = 0.f; A.err = false;
#pragma unroll
    for (int k = 0; k < 6; k++) A.g[k] = 0.f;
    A.first.x = A.first.y = A.prev.x = A.prev.y = 0.f; A.first.idx = A.prev.idx = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { A.first.g0[k] = A.first.g1[k] = A.prev.g0[k] = A.prev.g1[k] = 0.f; }

    clip_tri_edge<0, GRAD>(f, pxmin, pxmax, pymin, pymax, inside, A);
    if (!A.err) clip_tri_edge<1, GRAD>(f, pxmin, pxmax, pymin, pymax, inside, A);
    if (!A.err) clip_tri_edge<2, GRAD>(f, pxmin, pxmax, pymin, pymax, inside, A);
    if (A.err) return 1;
    if (A.area > pix_area) return 6;                                             // E05
    area = A.area;
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] = A.g[k];
    }
    return 0;
}

}  // namespace dm2
