// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// CPU restatement of the reference's per-pixel math (SonSang/dmesh2_renderer),
// used only as the parity oracle by tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py.  Nothing under dmesh2_renderer_amd/ includes,
// links or calls this file.
//
// Parity status: the reference ships no tests or golden outputs for this path
// and its native code cannot be built here (needs nvcc + CUB + glm).  The AA
// clipper below is pinned against vectors produced by the reference's own
// Python AA oracle (tests/golden/aa_pairs.npz <- pyrenderer.py:207-425); the
// remaining functions follow the cited reference lines and are cross-checked
// by fp64 finite differences of this restatement ("parity unpinned" by
// reference-owned vectors for everything except the AA area/gradient and the
// Python host prep).
//
// Every function is templated on the scalar type R: R=float is the oracle
// (arithmetic order and mixed-precision promotions as in the reference, build
// with -ffp-contract=off), R=double is used for finite-difference validation.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace orc {

constexpr int BLOCK_X = 16;          // config.h:4
constexpr int BLOCK_Y = 16;          // config.h:5
constexpr int BLOCK_SIZE = BLOCK_X * BLOCK_Y;   // auxiliary.h:11
constexpr float T_EPS = 0.0001f;     // auxiliary.h:9
constexpr int MAX_NUM_POLYGONS = 10; // aa.h:11

template <class R> struct V3 { R x, y, z; };

// float3 operators, evaluation order of cuda_math.h:738 (a-b), :1009/:1013
// (vector*scalar), :1273 (vector/scalar), :1524-1527 (dot), :1696-1699 (cross)
template <class R> inline V3<R> vsub(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline V3<R> vadd(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline V3<R> vneg(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> inline V3<R> vmul(V3<R> a, R b) { return {a.x * b, a.y * b, a.z * b}; }
template <class R> inline V3<R> smul(R b, V3<R> a) { return {b * a.x, b * a.y, b * a.z}; }
template <class R> inline V3<R> vdiv(V3<R> a, R b) { return {a.x / b, a.y / b, a.z / b}; }
template <class R> inline R vdot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> inline V3<R> vcross(V3<R> a, V3<R> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---- mixed-precision helpers (double literals in float expressions) -------
// aa.h:286,289  "(1.0 - t) + (g * e)":  1.0 is a double literal, so the
// subtraction and the addition run in double; g*e is a float product that is
// promoted afterwards; the store rounds once to float.
inline float one_minus_t_plus(float t, float prod) {
    return (float)((1.0 - (double)t) + (double)prod);
}
inline double one_minus_t_plus(double t, double prod) { return (1.0 - t) + prod; }
// aa.h:93  "0.5 * (float expr)"
inline float half_of(float e) { return (float)(0.5 * (double)e); }
inline double half_of(double e) { return 0.5 * e; }
// forward.cu:376 / backward.cu:314: "1.0 * (1.0f - temp) + ratio * temp"
inline float mix_inside(float ratio, float temp) {
    return (float)(1.0 * (double)(1.0f - temp) + (double)(ratio * temp));
}
inline double mix_inside(double ratio, double temp) { return (1.0 - temp) + ratio * temp; }
// forward.cu:378 / backward.cu:316: "0.0 * (1.0f - temp) + ratio * temp"
inline float mix_outside(float ratio, float temp) {
    return (float)(0.0 * (double)(1.0f - temp) + (double)(ratio * temp));
}
inline double mix_outside(double ratio, double temp) { return ratio * temp; }

// ---- auxiliary.h:212-243  Moeller-Trumbore, no inside test ----------------
template <class R>
inline bool ray_tri_intersection(V3<R> ro, V3<R> rd, V3<R> p0, V3<R> p1, V3<R> p2, V3<R>& tuv) {
    V3<R> T = vsub(ro, p0);
    V3<R> E1 = vsub(p1, p0);
    V3<R> E2 = vsub(p2, p0);
    V3<R> P = vcross(rd, E2);
    V3<R> Q = vcross(T, E1);
    R denom = vdot(P, E1);
    if (denom == (R)0) return false;
    R inv_denom = (R)1 / denom;
    tuv.x = vdot(Q, E2) * inv_denom;
    tuv.y = vdot(P, T) * inv_denom;
    tuv.z = vdot(Q, rd) * inv_denom;
    return true;
}

// ---- auxiliary.h:245-290 --------------------------------------------------
// As written, the "dv" outputs are the gradient of the ray parameter t
// (v2 = dot(cross(T,E1),E2), :272) rather than of barycentric v -- SURVEY §8a
// a18.  corrected=true swaps in the true gradient of v = dot(cross(T,E1),d)/denom
// (used only to validate the rest of the chain by finite differences).
template <class R>
inline void ray_tri_intersection_grad(V3<R> ro, V3<R> rd, V3<R> p0, V3<R> p1, V3<R> p2,
                                      V3<R>& du_dp0, V3<R>& du_dp1, V3<R>& du_dp2,
                                      V3<R>& dv_dp0, V3<R>& dv_dp1, V3<R>& dv_dp2,
                                      bool corrected = false) {
    V3<R> T = vsub(ro, p0);
    V3<R> E1 = vsub(p1, p0);
    V3<R> E2 = vsub(p2, p0);
    R denom_sqrt = vdot(vcross(rd, E2), E1);
    R denom = denom_sqrt * denom_sqrt;
    R denom_inv = (R)1 / denom;          // computed BEFORE the clamp (:264-265): clamp is dead
    R v0 = vdot(vcross(rd, E2), T);
    R v1 = denom_sqrt;
    R v2 = vdot(vcross(T, E1), E2);

    V3<R> du_dE1 = vmul(vmul(smul((R)-1, vcross(rd, E2)), v0), denom_inv);
    V3<R> du_dE2 = vmul(vsub(vmul(vcross(T, rd), v1), smul(v0, vcross(E1, rd))), denom_inv);
    V3<R> du_dT = vmul(vmul(vcross(rd, E2), v1), denom_inv);

    V3<R> dv_dE1, dv_dE2, dv_dT;
    if (!corrected) {
        dv_dE1 = vmul(vsub(vmul(vcross(E2, T), v1), smul(v2, vcross(rd, E2))), denom_inv);
        dv_dE2 = vmul(vsub(vmul(vcross(T, E1), v1), smul(v2, vcross(E1, rd))), denom_inv);
        dv_dT = vmul(vmul(vcross(E1, E2), v1), denom_inv);
    } else {
        // v = N/D, N = dot(cross(T,E1), d) = dot(E1, cross(d,T)) = dot(T, cross(E1,d)), D = denom_sqrt
        R N = vdot(vcross(T, E1), rd);
        // dN/dE1 = cross(d,T); dN/dT = cross(E1,d); dN/dE2 = 0
        // dD/dE1 = cross(d,E2); dD/dE2 = cross(E1,d)
        dv_dE1 = vmul(vsub(vmul(vcross(rd, T), v1), smul(N, vcross(rd, E2))), denom_inv);
        dv_dE2 = vmul(smul(-N, vcross(E1, rd)), denom_inv);
        dv_dT = vmul(vmul(vcross(E1, rd), v1), denom_inv);
    }
    du_dp0 = vsub(vsub(vneg(du_dE1), du_dE2), du_dT);
    dv_dp0 = vsub(vsub(vneg(dv_dE1), dv_dE2), dv_dT);
    du_dp1 = du_dE1; dv_dp1 = dv_dE1;
    du_dp2 = du_dE2; dv_dp2 = dv_dE2;
}

// ---- auxiliary.h:292-329 --------------------------------------------------
template <class R>
inline void clamp_bary_uv(R u, R v, R& u_c, R& v_c, int& code) {
    if (u >= 0 && v >= 0 && u + v <= 1) { u_c = u; v_c = v; code = 0; }
    else if (u <= 0 && v <= 0) { u_c = 0; v_c = 0; code = 1; }
    else if ((u >= 1 && v <= 0) || (v >= 0 && v <= u - 1)) { u_c = 1; v_c = 0; code = 2; }
    else if ((u <= 0 && v >= 1) || (u >= 0 && v >= u + 1)) { u_c = 0; v_c = 1; code = 3; }
    else if (u <= 0 && v <= 1 && v >= 0) { u_c = 0; v_c = v; code = 4; }
    else if (u <= 1 && u >= 0 && v <= 0) { u_c = u; v_c = 0; code = 5; }
    else { u_c = ((R)1 + u - v) * (R)0.5; v_c = ((R)1 - u + v) * (R)0.5; code = 6; }
}

// ---- auxiliary.h:331-357 --------------------------------------------------
template <class R>
inline void clamp_bary_uv_grad(int code, R& duc_du, R& duc_dv, R& dvc_du, R& dvc_dv) {
    dvc_du = 0; duc_dv = 0;
    if (code == 0) { duc_du = 1; dvc_dv = 1; }
    else if (code == 1 || code == 2 || code == 3) { duc_du = 0; dvc_dv = 0; }
    else if (code == 4) { duc_du = 0; dvc_dv = 1; }
    else if (code == 5) { duc_du = 1; dvc_dv = 0; }
    else { duc_du = (R)0.5; dvc_du = (R)-0.5; duc_dv = (R)-0.5; dvc_dv = (R)0.5; }
}

// float -> int as CUDA's cvt.rzi.s32.f32 does it: NaN -> 0, saturating.
inline int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return std::numeric_limits<int>::max();
    if (x <= -2147483648.0f) return std::numeric_limits<int>::min();
    return (int)x;
}

// ---- auxiliary.h:72-92 ----------------------------------------------------
// patch_min is a uint2 in the reference; "float - uint" converts the uint to
// float first.
inline void patch_rect_from_tri(uint32_t pmx, uint32_t pmy, const float p0[2], const float p1[2],
                                const float p2[2], uint32_t gx, uint32_t gy,
                                uint32_t rmin[2], uint32_t rmax[2]) {
    float min_x = fminf(fminf(p0[0], p1[0]), p2[0]) - (float)pmx;
    float min_y = fminf(fminf(p0[1], p1[1]), p2[1]) - (float)pmy;
    float max_x = fmaxf(fmaxf(p0[0], p1[0]), p2[0]) - (float)pmx;
    float max_y = fmaxf(fmaxf(p0[1], p1[1]), p2[1]) - (float)pmy;
    int min_x_tile = f2i_sat(floorf(min_x / BLOCK_X));
    int min_y_tile = f2i_sat(floorf(min_y / BLOCK_Y));
    int max_x_tile = f2i_sat(ceilf(max_x / BLOCK_X));
    int max_y_tile = f2i_sat(ceilf(max_y / BLOCK_Y));
    auto clampu = [](uint32_t g, int v) -> uint32_t {
        uint32_t m = (uint32_t)(v > 0 ? v : 0);
        return g < m ? g : m;
    };
    rmin[0] = clampu(gx, min_x_tile); rmin[1] = clampu(gy, min_y_tile);
    rmax[0] = clampu(gx, max_x_tile); rmax[1] = clampu(gy, max_y_tile);
}

// ---- renderer.cu:396-411 --------------------------------------------------
inline uint32_t get_higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// ---- aa.h -----------------------------------------------------------------
// Per-(batch,face) AA tables, as 6/6/6/6/6/3 contiguous values
// (verts, edges, iszero, recip, normal: [3][2]; normal_c: [3]).
template <class R>
struct AATri {
    const R* verts; const R* edges; const uint8_t* iszero; const R* recip; const R* normal; const R* normal_c;
};

// aa.h:15-21
template <class R>
inline bool is_vert_inside_triangle_edge(const R* vert, const R* n, R c) {
    return (vert[0] * n[0]) + (vert[1] * n[1]) - c >= 0;
}

// aa.h:67-86; ip_tri_edge_idx == -1 (pixel corner): the reference indexes
// grad_tri_verts[-1] with an all-zero Jacobian (out-of-bounds "+= 0"); the
// write to row -1 is dropped here, the "+= 0*ga" into row 0 is kept so NaN/inf
// propagate as they would.
template <class R>
inline void update_grad_tri_verts(const R* g0, const R* g1, const R* ga, int idx, R (*grad)[2]) {
    R a0 = g0[0] * ga[0] + g0[1] * ga[1];
    R a1 = g0[2] * ga[0] + g0[3] * ga[1];
    R b0 = g1[0] * ga[0] + g1[1] * ga[1];
    R b1 = g1[2] * ga[0] + g1[3] * ga[1];
    int i0 = idx;
    int i1 = (idx + 1) % 3;
    if (i0 >= 0) { grad[i0][0] += a0; grad[i0][1] += a1; }
    grad[i1][0] += b0; grad[i1][1] += b1;
}

// aa.h:151-441.  Returns 0 ok, 1..6 error (README E00..E05).
template <class R>
inline int sub_tri_pix_overlap_area(const AATri<R>& t, R pxmin, R pxmax, R pymin, R pymax, R pix_area,
                                    const R (*pix_verts)[2], const bool* pix_verts_is_inside,
                                    R* area, R (*grad_tri_verts)[2]) {
    R polygon[MAX_NUM_POLYGONS][2];
    int polygon_tri_edge_idx[MAX_NUM_POLYGONS];
    R polygon_grad_ip_p0[MAX_NUM_POLYGONS][4];
    R polygon_grad_ip_p1[MAX_NUM_POLYGONS][4];
    int num_polygons = 0;
    const R zero2[4] = {0, 0, 0, 0};
    const R eye2[4] = {1, 0, 0, 1};

    auto add_polygon = [&](const R* p, int eidx, const R* g0, const R* g1) -> bool {
        if (num_polygons >= MAX_NUM_POLYGONS) return false;          // aa.h:45-48
        polygon[num_polygons][0] = p[0]; polygon[num_polygons][1] = p[1];
        for (int k = 0; k < 4; k++) { polygon_grad_ip_p0[num_polygons][k] = g0[k]; polygon_grad_ip_p1[num_polygons][k] = g1[k]; }
        polygon_tri_edge_idx[num_polygons] = eidx;
        num_polygons++;
        return true;
    };
    auto inside_pixel = [&](const R* v) {                             // aa.h:23-31
        return (v[0] >= pxmin) && (v[0] <= pxmax) && (v[1] >= pymin) && (v[1] <= pymax);
    };

    for (int ti = 0; ti < 3; ti++) {
        const R* tri_p0 = t.verts + 2 * ti;
        const R* tri_p1 = t.verts + 2 * ((ti + 1) % 3);
        const R* tri_edge = t.edges + 2 * ti;
        const uint8_t* tri_edge_iszero = t.iszero + 2 * ti;
        const R* tri_edge_recip = t.recip + 2 * ti;
        bool is_tri_edge_horizontal = tri_edge_iszero[1] != 0;
        bool is_tri_edge_vertical = tri_edge_iszero[0] != 0;
        bool is_tri_p0_inside = inside_pixel(tri_p0);
        bool is_tri_p1_inside = inside_pixel(tri_p1);

        R inter_point[4][2]; R inter_t[4]; int inter_pedge_idx[4];
        R inter_grad_p0[4][4]; R inter_grad_p1[4][4];
        int num_intersections = 0;

        for (int pi = 0; pi < 4; pi++) {
            bool is_pedge_horizontal = ((pi == 0) || (pi == 2));
            bool is_pedge_vertical = !is_pedge_horizontal;
            bool is_tedge_parallel = (is_tri_edge_horizontal && is_pedge_horizontal) || (is_tri_edge_vertical && is_pedge_vertical);
            int axis0; R pmin1, pmax1;
            if (is_pedge_horizontal) { axis0 = 1; pmin1 = pxmin; pmax1 = pxmax; }
            else { axis0 = 0; pmin1 = pymin; pmax1 = pymax; }
            int axis1 = 1 - axis0;

            R iaxis0 = pix_verts[pi][axis0];
            R tt = (iaxis0 - tri_p0[axis0]) * tri_edge_recip[axis0];
            R iaxis1 = tri_p0[axis1] + tt * tri_edge[axis1];

            bool is_t_valid = ((tt >= 0) && (tt <= 1) && (iaxis1 >= pmin1) && (iaxis1 <= pmax1) && (!is_tedge_parallel));
            if (!is_t_valid) continue;
            bool is_ipoint_pixvert = ((iaxis1 == pmin1) || (iaxis1 == pmax1));
            if (is_ipoint_pixvert) return 1;                          // E00

            inter_point[num_intersections][axis0] = iaxis0;
            inter_point[num_intersections][axis1] = iaxis1;
            inter_t[num_intersections] = tt;
            inter_pedge_idx[num_intersections] = pi;

            R grad_t_p0[2] = {0, 0};
            R grad_t_p1[2] = {0, 0};
            grad_t_p0[axis0] = (iaxis0 - tri_p1[axis0]) * tri_edge_recip[axis0] * tri_edge_recip[axis0];
            grad_t_p1[axis0] = (-iaxis0 + tri_p0[axis0]) * tri_edge_recip[axis0] * tri_edge_recip[axis0];

            inter_grad_p0[num_intersections][0] = one_minus_t_plus(tt, (R)(grad_t_p0[0] * tri_edge[0]));
            inter_grad_p0[num_intersections][1] = grad_t_p0[0] * tri_edge[1];
            inter_grad_p0[num_intersections][2] = grad_t_p0[1] * tri_edge[0];
            inter_grad_p0[num_intersections][3] = one_minus_t_plus(tt, (R)(grad_t_p0[1] * tri_edge[1]));

            inter_grad_p1[num_intersections][0] = tt + (grad_t_p1[0] * tri_edge[0]);
            inter_grad_p1[num_intersections][1] = grad_t_p1[0] * tri_edge[1];
            inter_grad_p1[num_intersections][2] = grad_t_p1[1] * tri_edge[0];
            inter_grad_p1[num_intersections][3] = tt + (grad_t_p1[1] * tri_edge[1]);
            num_intersections++;
        }
        if (num_intersections > 2) return 2;                         // E01

        if (num_intersections > 0) {
            int final_pedge_id = -1;
            if (num_intersections == 2) {
                int si[2] = {0, 1};
                if (inter_t[0] > inter_t[1]) { si[0] = 1; si[1] = 0; }
                bool ok = add_polygon(inter_point[si[0]], ti, inter_grad_p0[si[0]], inter_grad_p1[si[0]]);
                ok &= add_polygon(inter_point[si[1]], ti, inter_grad_p0[si[1]], inter_grad_p1[si[1]]);
                if (!ok) return 5;
                final_pedge_id = inter_pedge_idx[si[1]];
            } else {
                if (!add_polygon(inter_point[0], ti, inter_grad_p0[0], inter_grad_p1[0])) return 5;
                if (!is_tri_p0_inside && is_tri_p1_inside) {
                    if (!add_polygon(tri_p1, ti, zero2, eye2)) return 5;
                } else if (is_tri_p0_inside && !is_tri_p1_inside) {
                    final_pedge_id = inter_pedge_idx[0];
                } else {
                    return 3;                                         // E02
                }
            }
            if (final_pedge_id != -1) {
                int start_pvert_id = (final_pedge_id + 1) % 4;
                for (int pvi = 0; pvi < 4; pvi++) {
                    int cur = (start_pvert_id + pvi) % 4;
                    if (pix_verts_is_inside[cur]) {
                        if (!add_polygon(pix_verts[cur], -1, zero2, zero2)) return 5;
                    } else break;
                }
            }
        } else {
            if (is_tri_p0_inside && is_tri_p1_inside) {
                if (!add_polygon(tri_p1, ti, zero2, eye2)) return 5;
            } else if (!is_tri_p0_inside && !is_tri_p1_inside) {
                continue;
            } else {
                return 4;                                             // E03
            }
        }
    }

    int num_subtris = num_polygons - 2;
    for (int si = 0; si < num_subtris; si++) {
        const R* ip0 = polygon[0];
        const R* ip1 = polygon[si + 1];
        const R* ip2 = polygon[si + 2];
        R s_area = half_of((R)((ip1[0] - ip0[0]) * (ip2[1] - ip0[1]) - (ip2[0] - ip0[0]) * (ip1[1] - ip0[1])));
        if (s_area < 0) return 5;                                     // E04
        *area += s_area;
        R ga0[2] = {(R)0.5 * (ip1[1] - ip2[1]), (R)0.5 * (-ip1[0] + ip2[0])};
        R ga1[2] = {(R)0.5 * (ip2[1] - ip0[1]), (R)0.5 * (-ip2[0] + ip0[0])};
        R ga2[2] = {(R)0.5 * (ip0[1] - ip1[1]), (R)0.5 * (-ip0[0] + ip1[0])};
        update_grad_tri_verts(polygon_grad_ip_p0[0], polygon_grad_ip_p1[0], ga0, polygon_tri_edge_idx[0], grad_tri_verts);
        update_grad_tri_verts(polygon_grad_ip_p0[si + 1], polygon_grad_ip_p1[si + 1], ga1, polygon_tri_edge_idx[si + 1], grad_tri_verts);
        update_grad_tri_verts(polygon_grad_ip_p0[si + 2], polygon_grad_ip_p1[si + 2], ga2, polygon_tri_edge_idx[si + 2], grad_tri_verts);
    }
    if (*area > pix_area) return 6;                                   // E05
    return 0;
}

// aa.h:446-504.  area and grad must be zero on entry.
template <class R>
inline int tri_pix_overlap_area(const AATri<R>& t, R txmin, R txmax, R tymin, R tymax,
                                R pxmin, R pxmax, R pymin, R pymax, R pix_area,
                                R* area, R (*grad_tri_verts)[2]) {
    if ((pxmax < txmin) || (pxmin > txmax) || (pymax < tymin) || (pymin > tymax)) return 0;   // aa.h:96-101
    R pix_verts[4][2] = {{pxmin, pymin}, {pxmax, pymin}, {pxmax, pymax}, {pxmin, pymax}};
    bool inside[4] = {true, true, true, true};
    for (int ti = 0; ti < 3; ti++) {                                  // aa.h:103-149
        bool every_out = true;
        for (int pvi = 0; pvi < 4; pvi++) {
            bool in = is_vert_inside_triangle_edge(pix_verts[pvi], t.normal + 2 * ti, t.normal_c[ti]);
            every_out = every_out && (!in);
            inside[pvi] = inside[pvi] && in;
        }
        if (every_out) return 0;
    }
    if (inside[0] && inside[1] && inside[2] && inside[3]) { *area = pix_area; return 0; }
    return sub_tri_pix_overlap_area(t, pxmin, pxmax, pymin, pymax, pix_area, pix_verts, inside, area, grad_tri_verts);
}

// ---- auxiliary.h:382-431 --------------------------------------------------
template <class R>
inline V3<R> tet_face_outward_normal(const R* verts, const int* faces, const int* tets, int face_idx, int tet_idx) {
    auto vert = [&](int i) { return V3<R>{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]}; };
    V3<R> p0 = vert(faces[3 * face_idx]), p1 = vert(faces[3 * face_idx + 1]), p2 = vert(faces[3 * face_idx + 2]);
    V3<R> n = vcross(vsub(p1, p0), vsub(p2, p0));
    R n_norm = std::sqrt(vdot(n, n));
    n_norm = n_norm > (R)0.0001f ? n_norm : (R)0.0001f;
    n = vdiv(n, n_norm);
    V3<R> q0 = vert(tets[4 * tet_idx]), q1 = vert(tets[4 * tet_idx + 1]), q2 = vert(tets[4 * tet_idx + 2]), q3 = vert(tets[4 * tet_idx + 3]);
    V3<R> c = vmul(vadd(vadd(vadd(q0, q1), q2), q3), (R)0.25);
    V3<R> d = vsub(c, p0);
    if (vdot(n, d) > 0) n = vneg(n);
    return n;
}

}  // namespace orc
