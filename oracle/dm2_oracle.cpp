// TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see dm2_oracle_math.hpp header).
//
// CPU restatement of the reference pipeline behind `_C.render_forward_cuda`,
// `_C.render_backward_cuda` and `_C.generate_render_layers_cuda`
// (render.cu:28-476 -> cuda_impl/renderer.cu -> forward.cu / backward.cu).
// C ABI for ctypes (oracle/cpu.py).  Build: oracle/Makefile
// (g++ -O2 -ffp-contract=off -fopenmp).
#include "dm2_oracle_math.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

// ===========================================================================
// Binning: preprocessFaceCUDA (forward.cu:16-108) -> InclusiveSum
// (renderer.cu:165-171) -> duplicateWithKeys (:415-465) -> stable radix sort on
// bits [0, 32+getHigherMsb(B*tiles)) (:199-207) -> identifyTileRanges (:470-492)
// ===========================================================================
struct Binning {
    int B, F, W, H, gx, gy;
    std::vector<float> depths, min_depths, max_depths;   // (B*F); culled faces stay 0 (reference: unwritten)
    std::vector<uint32_t> tiles_touched, face_offsets;   // (B*F)
    std::vector<uint64_t> keys;                          // sorted (R)
    std::vector<uint32_t> face_list;                     // sorted (R)
    std::vector<uint32_t> ranges;                        // (B*gx*gy*2)
    int64_t num_rendered;
};

Binning* do_binning(int B, int P, int F, int W, int H, const int* patch_min, const int* faces,
                    const float* verts_ndc, const float* verts_image, int key_min_depth) {
    Binning* bn = new Binning();
    bn->B = B; bn->F = F; bn->W = W; bn->H = H;
    const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    bn->gx = gx; bn->gy = gy;
    const int64_t BF = (int64_t)B * F;
    bn->depths.assign(BF, 0.f); bn->min_depths.assign(BF, 0.f); bn->max_depths.assign(BF, 0.f);
    bn->tiles_touched.assign(BF, 0); bn->face_offsets.assign(BF, 0);
    std::vector<uint32_t> rect(BF * 4, 0);

    for (int64_t idx = 0; idx < BF; idx++) {
        int b = (int)(idx / F), f = (int)(idx % F);
        uint32_t pmx = (uint32_t)patch_min[2 * b], pmy = (uint32_t)patch_min[2 * b + 1];
        float max_z = 0, min_z = 0, depth = 0;
        float vi[3][2];
        for (int i = 0; i < 3; i++) {
            int v = faces[3 * f + i];
            float z = verts_ndc[((int64_t)b * P + v) * 3 + 2];
            if (i == 0) { max_z = z; min_z = z; }
            else { max_z = fmaxf(max_z, z); min_z = fminf(min_z, z); }
            depth += z;
            vi[i][0] = verts_image[((int64_t)b * P + v) * 2];
            vi[i][1] = verts_image[((int64_t)b * P + v) * 2 + 1];
        }
        depth = depth / 3.0f;
        if (max_z < -1.0f || min_z > 1.0f) continue;                    // forward.cu:71
        uint32_t rmin[2], rmax[2];
        patch_rect_from_tri(pmx, pmy, vi[0], vi[1], vi[2], gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;  // forward.cu:88
        bn->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
        auto to01 = [](float z) { float d = (z + 1.0f) * 0.5f; if (d < 0.0f) d = 0.0f; if (d > 1.0f) d = 1.0f; return d; };
        bn->depths[idx] = to01(depth);
        bn->min_depths[idx] = to01(min_z);
        bn->max_depths[idx] = to01(max_z);
        rect[4 * idx] = rmin[0]; rect[4 * idx + 1] = rmin[1]; rect[4 * idx + 2] = rmax[0]; rect[4 * idx + 3] = rmax[1];
    }
    uint32_t run = 0;
    for (int64_t i = 0; i < BF; i++) { run += bn->tiles_touched[i]; bn->face_offsets[i] = run; }
    const int64_t R = BF > 0 ? bn->face_offsets[BF - 1] : 0;
    bn->num_rendered = R;

    std::vector<uint64_t> keys_unsorted(R);
    std::vector<uint32_t> vals_unsorted(R);
    const float* kd = key_min_depth ? bn->min_depths.data() : bn->depths.data();   // renderer.cu:192 vs :603
    const uint32_t grid_size = gx * gy;
    for (int64_t idx = 0; idx < BF; idx++) {
        if (bn->tiles_touched[idx] == 0) continue;
        int b = (int)(idx / F), f = (int)(idx % F);
        uint32_t off = (idx == 0) ? 0 : bn->face_offsets[idx - 1];
        uint32_t dbits; std::memcpy(&dbits, &kd[idx], 4);
        for (uint32_t y = rect[4 * idx + 1]; y < rect[4 * idx + 3]; y++)
            for (uint32_t x = rect[4 * idx]; x < rect[4 * idx + 2]; x++) {
                uint64_t key = (uint64_t)(y * gx + x) + (uint64_t)grid_size * b;
                key <<= 32; key |= dbits;
                keys_unsorted[off] = key; vals_unsorted[off] = (uint32_t)f; off++;
            }
    }
    const uint32_t bit = get_higher_msb((uint32_t)B * gx * gy);
    const int end_bit = 32 + (int)bit;
    const uint64_t mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
    std::vector<uint32_t> perm(R);
    for (int64_t i = 0; i < R; i++) perm[i] = (uint32_t)i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t c) {
        return (keys_unsorted[a] & mask) < (keys_unsorted[c] & mask);
    });
    bn->keys.resize(R); bn->face_list.resize(R);
    for (int64_t i = 0; i < R; i++) { bn->keys[i] = keys_unsorted[perm[i]]; bn->face_list[i] = vals_unsorted[perm[i]]; }

    bn->ranges.assign((size_t)B * grid_size * 2, 0);                    // renderer.cu:211
    for (int64_t i = 0; i < R; i++) {                                   // renderer.cu:470-492
        uint32_t cur = (uint32_t)(bn->keys[i] >> 32);
        if (i == 0) bn->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(bn->keys[i - 1] >> 32);
            if (cur != prev) { bn->ranges[2 * prev + 1] = (uint32_t)i; bn->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) bn->ranges[2 * cur + 1] = (uint32_t)R;
    }
    return bn;
}

// ===========================================================================
// Render forward / backward
// ===========================================================================
template <class R>
struct RenderArgs {
    int B, P, F, W, H, K;
    R temp;
    const int* patch_min;       // (B,2)
    const R* background;        // (3)
    const R* verts;             // (P,3)
    const int* faces;           // (F,3)
    const R* verts_color;       // (P,3)
    const R* faces_opacity;     // (F)
    const R* verts_ndc;         // (B,P,3)
    const R* faces_intense;     // (B,F)
    const R* aa_verts; const R* aa_edges; const uint8_t* aa_iszero; const R* aa_recip; const R* aa_normal; const R* aa_normal_c;
    const R* ray_o; const R* ray_d;   // (B,H,W,3)
    const uint32_t* ranges; const uint32_t* face_list;
};

template <class R>
struct FaceCtx {       // what the reference stages in shared memory per list entry (forward.cu:262-303)
    int face_id; int vid[3];
    V3<R> v[3]; R col[3][3]; R dep[3]; R opacity, intense;
    R txmin, txmax, tymin, tymax; AATri<R> aa;
};

template <class R>
inline void load_face(const RenderArgs<R>& a, int b, int face_id, FaceCtx<R>& fc) {
    fc.face_id = face_id;
    for (int i = 0; i < 3; i++) {
        int v = a.faces[3 * face_id + i];
        fc.vid[i] = v;
        fc.v[i] = {a.verts[3 * v], a.verts[3 * v + 1], a.verts[3 * v + 2]};
        for (int ch = 0; ch < 3; ch++) fc.col[i][ch] = a.verts_color[3 * v + ch];
        fc.dep[i] = a.verts_ndc[((int64_t)b * a.P + v) * 3 + 2];
    }
    fc.opacity = a.faces_opacity[face_id];
    fc.intense = a.faces_intense[(int64_t)b * a.F + face_id];
    int64_t bf = (int64_t)b * a.F + face_id;
    fc.aa = {a.aa_verts + bf * 6, a.aa_edges + bf * 6, a.aa_iszero + bf * 6, a.aa_recip + bf * 6, a.aa_normal + bf * 6, a.aa_normal_c + bf * 3};
    // aa_face_verts.min(2)/.max(2)  (forward.cu:480-481, backward.cu:589-590)
    const R* tv = fc.aa.verts;
    fc.txmin = std::min(std::min(tv[0], tv[2]), tv[4]); fc.txmax = std::max(std::max(tv[0], tv[2]), tv[4]);
    fc.tymin = std::min(std::min(tv[1], tv[3]), tv[5]); fc.tymax = std::max(std::max(tv[1], tv[3]), tv[5]);
}

// FORWARD::renderCUDA<3>, forward.cu:139-432, one pixel at a time.
template <class R>
void render_forward(const RenderArgs<R>& a, R* out_color, R* out_depth, R* final_T, R* final_prev_T,
                    uint32_t* n_contrib, R* buf_oarea, int* buf_tri_id, int* buf_tri_cnt, R* buf_doarea,
                    int nthreads) {
    const int gx = (a.W + BLOCK_X - 1) / BLOCK_X, gy = (a.H + BLOCK_Y - 1) / BLOCK_Y;
    const int K = a.K;
    const int64_t ntiles = (int64_t)a.B * gx * gy;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t tile = 0; tile < ntiles; tile++) {
        const int b = (int)(tile / (gx * gy));
        const int ty = (int)((tile % (gx * gy)) / gx), tx = (int)(tile % gx);
        const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
        const uint32_t pmx = (uint32_t)a.patch_min[2 * b], pmy = (uint32_t)a.patch_min[2 * b + 1];
        std::vector<FaceCtx<R>> fcs(r1 - r0);
        for (uint32_t i = r0; i < r1; i++) load_face(a, b, (int)a.face_list[i], fcs[i - r0]);
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            const uint32_t pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
            if ((int)pxi >= a.W || (int)pyi >= a.H) continue;
            const int64_t pix = ((int64_t)b * a.H + pyi) * a.W + pxi;
            const V3<R> ro = {a.ray_o[3 * pix], a.ray_o[3 * pix + 1], a.ray_o[3 * pix + 2]};
            const V3<R> rd = {a.ray_d[3 * pix], a.ray_d[3 * pix + 1], a.ray_d[3 * pix + 2]};
            R pT = 1, T = 1;
            uint32_t contributor = 0, last_contributor = 0;
            R C[3] = {0, 0, 0}, D = 0;
            int cnt = 0;
            const R pxmin = (R)(float)(pxi + pmx), pxmax = pxmin + 1;
            const R pymin = (R)(float)(pyi + pmy), pymax = pymin + 1;
            const R pix_area = 1;
            for (uint32_t j = 0; j < r1 - r0; j++) {
                contributor++;
                const FaceCtx<R>& fc = fcs[j];
                R oarea = 0;
                R g[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                if (a.temp > 0) {
                    int err = tri_pix_overlap_area(fc.aa, fc.txmin, fc.txmax, fc.tymin, fc.tymax,
                                                   pxmin, pxmax, pymin, pymax, pix_area, &oarea, g);
                    if (err != 0 || oarea == 0) continue;
                }
                R ratio = oarea / pix_area;
                if (a.temp > 0 && cnt < K) {                            // forward.cu:344-352
                    buf_oarea[pix * K + cnt] = oarea;
                    buf_tri_id[pix * K + cnt] = fc.face_id;
                    for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 2; jj++)
                        buf_doarea[((pix * K + cnt) * 3 + ii) * 2 + jj] = g[ii][jj];
                    cnt++;
                }
                V3<R> tuv = {0, 0, 0};
                if (!ray_tri_intersection(ro, rd, fc.v[0], fc.v[1], fc.v[2], tuv)) continue;
                R iuc, ivc; int code;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                R i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                ratio = (code == 0) ? mix_inside(ratio, a.temp) : mix_outside(ratio, a.temp);
                if (ratio == 0) continue;
                R iC[3], iD;
                for (int ch = 0; ch < 3; ch++) {
                    iC[ch] = i0 * fc.col[0][ch] + i1 * fc.col[1][ch] + i2 * fc.col[2][ch];
                    iC[ch] = iC[ch] * fc.intense;
                }
                iD = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                R alpha = fc.opacity * ratio;
                R test_T = T * (1 - alpha);
                for (int ch = 0; ch < 3; ch++) C[ch] += iC[ch] * alpha * T;
                D += iD * alpha * T;
                pT = T; T = test_T;
                last_contributor = contributor;
                if (T < (R)T_EPS) break;
            }
            final_prev_T[pix] = pT; final_T[pix] = T; n_contrib[pix] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[3 * pix + ch] = C[ch] + T * a.background[ch];
            out_depth[pix] = D + T * (R)1.0f;
            buf_tri_cnt[pix] = cnt;
        }
    }
}

template <class R> inline void acc(R* p, R v, bool atomic) {
    if (atomic) {
#pragma omp atomic
        *p += v;
    } else *p += v;
}

// BACKWARD::renderCUDA<3>, backward.cu:17-532.  corrected_dv: see
// ray_tri_intersection_grad.  With nthreads==1 the accumulation order is fixed
// (tile-major, row-major pixels, back-to-front) -> deterministic.
template <class R>
void render_backward(const RenderArgs<R>& a, const R* dL_dcolor, const R* dL_ddepth,
                     const R* final_T, const R* final_prev_T, const uint32_t* n_contrib,
                     const R* buf_oarea, const int* buf_tri_id, const int* buf_tri_cnt, const R* buf_doarea,
                     R* dL_dverts, R* dL_dverts_color, R* dL_dfaces_opacity, R* dL_dverts_ndc,
                     R* dL_dfaces_intense, R* dL_daa_face_verts, int corrected_dv, int nthreads) {
    const int gx = (a.W + BLOCK_X - 1) / BLOCK_X, gy = (a.H + BLOCK_Y - 1) / BLOCK_Y;
    const int K = a.K;
    const int64_t ntiles = (int64_t)a.B * gx * gy;
    const bool atomic = nthreads > 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t tile = 0; tile < ntiles; tile++) {
        const int b = (int)(tile / (gx * gy));
        const int ty = (int)((tile % (gx * gy)) / gx), tx = (int)(tile % gx);
        const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
        const uint32_t pmx = (uint32_t)a.patch_min[2 * b], pmy = (uint32_t)a.patch_min[2 * b + 1];
        const uint32_t n = r1 - r0;
        std::vector<FaceCtx<R>> fcs(n);   // fcs[j] = entry r1-1-j  (back to front, backward.cu:171)
        for (uint32_t j = 0; j < n; j++) load_face(a, b, (int)a.face_list[r1 - 1 - j], fcs[j]);
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            const uint32_t pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
            if ((int)pxi >= a.W || (int)pyi >= a.H) continue;
            const int64_t pix = ((int64_t)b * a.H + pyi) * a.W + pxi;
            const V3<R> ro = {a.ray_o[3 * pix], a.ray_o[3 * pix + 1], a.ray_o[3 * pix + 2]};
            const V3<R> rd = {a.ray_d[3 * pix], a.ray_d[3 * pix + 1], a.ray_d[3 * pix + 2]};
            const R T_final = final_T[pix], prev_T_final = final_prev_T[pix];
            R T = prev_T_final;
            bool T_first_pass = true;
            uint32_t contributor = n;
            const uint32_t last_contributor = n_contrib[pix];
            R accum_rec[3] = {0, 0, 0}, accum_recd = 0;
            const R dLc[3] = {dL_dcolor[3 * pix], dL_dcolor[3 * pix + 1], dL_dcolor[3 * pix + 2]};
            const R dLd = dL_ddepth[pix];
            int ptr = buf_tri_cnt[pix];
            R last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0;
            const R pxmin = (R)(float)(pxi + pmx), pxmax = pxmin + 1;
            const R pymin = (R)(float)(pyi + pmy), pymax = pymin + 1;
            const R pix_area = 1;
            for (uint32_t j = 0; j < n; j++) {
                contributor--;
                if (contributor >= last_contributor) continue;
                const FaceCtx<R>& fc = fcs[j];
                R oarea = 0;
                R dg[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                int err = 0;
                bool need = false;
                if (a.temp > 0) {                                       // backward.cu:241-274
                    if (ptr > 0) {
                        if (buf_tri_id[pix * K + ptr - 1] == fc.face_id) {
                            oarea = buf_oarea[pix * K + ptr - 1];
                            for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 2; jj++)
                                dg[ii][jj] = buf_doarea[((pix * K + ptr - 1) * 3 + ii) * 2 + jj];
                            ptr--;
                        } else need = (ptr == K);
                    } else need = !(K > 0);
                }
                if (need)
                    err = tri_pix_overlap_area(fc.aa, fc.txmin, fc.txmax, fc.tymin, fc.tymax,
                                               pxmin, pxmax, pymin, pymax, pix_area, &oarea, dg);
                if (a.temp > 0) { if (err != 0 || oarea == 0) continue; }
                R ratio = oarea / pix_area;
                V3<R> tuv = {0, 0, 0};
                if (!ray_tri_intersection(ro, rd, fc.v[0], fc.v[1], fc.v[2], tuv)) continue;
                R iuc, ivc; int code;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                R i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                ratio = (code == 0) ? mix_inside(ratio, a.temp) : mix_outside(ratio, a.temp);
                if (ratio == 0) continue;
                R iC[3], iD;
                for (int ch = 0; ch < 3; ch++) {
                    iC[ch] = i0 * fc.col[0][ch] + i1 * fc.col[1][ch] + i2 * fc.col[2][ch];
                    iC[ch] = iC[ch] * fc.intense;
                }
                iD = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                R alpha = fc.opacity * ratio;
                if (!T_first_pass) T = T / ((R)1 - alpha);
                T_first_pass = false;

                R dL_dic[3], dL_did, dL_dalpha = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const R c = iC[ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + ((R)1 - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dic[ch] = dLc[ch] * alpha * T;
                    dL_dalpha += (c - accum_rec[ch]) * dLc[ch];
                }
                {
                    const R c = iD;
                    accum_recd = last_alpha * last_depth + ((R)1 - last_alpha) * accum_recd;
                    last_depth = c;
                    dL_did = dLd * alpha * T;
                    dL_dalpha += (c - accum_recd) * dLd;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                R bg_dot = 0, bd_dot = 0;
                for (int ch = 0; ch < 3; ch++) bg_dot += a.background[ch] * dLc[ch];
                bd_dot = (R)((double)bd_dot + 1.0 * (double)dLd);       // backward.cu:394
                if (alpha == (R)1) {
                    dL_dalpha += (-prev_T_final) * bg_dot;
                    dL_dalpha += (-prev_T_final) * bd_dot;
                } else {
                    dL_dalpha += (-T_final / ((R)1 - alpha)) * bg_dot;
                    dL_dalpha += (-T_final / ((R)1 - alpha)) * bd_dot;
                }
                R dL_dfop = dL_dalpha * ratio;
                R dL_dratio = (dL_dalpha * fc.opacity) * a.temp;
                R dL_doarea = dL_dratio / pix_area;
                R dL_daa[3][2];
                for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 2; jj++) dL_daa[ii][jj] = dL_doarea * dg[ii][jj];

                R dL_di0 = 0, dL_di1 = 0, dL_di2 = 0;
                R dvc0[3] = {0, 0, 0}, dvc1[3] = {0, 0, 0}, dvc2[3] = {0, 0, 0};
                R dvd0 = 0, dvd1 = 0, dvd2 = 0, dL_dfint = 0;
                for (int ch = 0; ch < 3; ch++) {
                    dL_di0 += fc.col[0][ch] * dL_dic[ch] * fc.intense;
                    dL_di1 += fc.col[1][ch] * dL_dic[ch] * fc.intense;
                    dL_di2 += fc.col[2][ch] * dL_dic[ch] * fc.intense;
                    dvc0[ch] += i0 * dL_dic[ch] * fc.intense;
                    dvc1[ch] += i1 * dL_dic[ch] * fc.intense;
                    dvc2[ch] += i2 * dL_dic[ch] * fc.intense;
                    dL_dfint += (i0 * fc.col[0][ch] + i1 * fc.col[1][ch] + i2 * fc.col[2][ch]) * dL_dic[ch];
                }
                dL_di0 += fc.dep[0] * dL_did; dL_di1 += fc.dep[1] * dL_did; dL_di2 += fc.dep[2] * dL_did;
                dvd0 += i0 * dL_did; dvd1 += i1 * dL_did; dvd2 += i2 * dL_did;

                const R di0_diuc = -1, di0_divc = -1, di1_diuc = 1, di1_divc = 0, di2_diuc = 0, di2_divc = 1;
                R diuc_diu, diuc_div, divc_diu, divc_div;
                clamp_bary_uv_grad(code, diuc_diu, diuc_div, divc_diu, divc_div);
                R di0_diu = di0_diuc * diuc_diu + di0_divc * divc_diu;
                R di0_div = di0_diuc * diuc_div + di0_divc * divc_div;
                R di1_diu = di1_diuc * diuc_diu + di1_divc * divc_diu;
                R di1_div = di1_diuc * diuc_div + di1_divc * divc_div;
                R di2_diu = di2_diuc * diuc_diu + di2_divc * divc_diu;
                R di2_div = di2_diuc * diuc_div + di2_divc * divc_div;
                R dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
                R dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;
                V3<R> du0, du1, du2, dv0, dv1, dv2;
                ray_tri_intersection_grad(ro, rd, fc.v[0], fc.v[1], fc.v[2], du0, du1, du2, dv0, dv1, dv2, corrected_dv != 0);
                V3<R> dp0 = vadd(smul(dL_diu, du0), smul(dL_div, dv0));
                V3<R> dp1 = vadd(smul(dL_diu, du1), smul(dL_div, dv1));
                V3<R> dp2 = vadd(smul(dL_diu, du2), smul(dL_div, dv2));

                const V3<R> dps[3] = {dp0, dp1, dp2};
                const R* dvcs[3] = {dvc0, dvc1, dvc2};
                const R dvds[3] = {dvd0, dvd1, dvd2};
                for (int k = 0; k < 3; k++) {
                    int v = fc.vid[k];
                    acc(&dL_dverts[3 * v], dps[k].x, atomic); acc(&dL_dverts[3 * v + 1], dps[k].y, atomic); acc(&dL_dverts[3 * v + 2], dps[k].z, atomic);
                }
                for (int ch = 0; ch < 3; ch++) for (int k = 0; k < 3; k++) acc(&dL_dverts_color[3 * fc.vid[k] + ch], dvcs[k][ch], atomic);
                for (int k = 0; k < 3; k++) acc(&dL_dverts_ndc[((int64_t)b * a.P + fc.vid[k]) * 3 + 2], dvds[k], atomic);
                acc(&dL_dfaces_opacity[fc.face_id], dL_dfop, atomic);
                acc(&dL_dfaces_intense[(int64_t)b * a.F + fc.face_id], dL_dfint, atomic);
                for (int ii = 0; ii < 3; ii++) for (int jj = 0; jj < 2; jj++)
                    acc(&dL_daa_face_verts[(((int64_t)b * a.F + fc.face_id) * 3 + ii) * 2 + jj], dL_daa[ii][jj], atomic);
            }
        }
    }
}

// ===========================================================================
// LayeredRenderer: firstIntersectCUDA (forward.cu:538-709) +
// generateRenderLayersCUDA (forward.cu:744-1000).  The reference's unguarded
// out-of-image writes (forward.cu:584-585) are not reproduced: only in-image
// pixels are visited.
// ===========================================================================
void render_layers(int B, int P, int F, int T, int W, int H, const float* verts, const int* faces, const int* tets,
                   const int* face_tets, const int* tet_faces, const int* face_exist,
                   const float* ray_o, const float* ray_d, const Binning& bn, int L,
                   int* first_face, int* first_tet, int* layers, int* layers_cnt, int nthreads) {
    (void)P;
    const int gx = bn.gx, gy = bn.gy;
    const int64_t ntiles = (int64_t)B * gx * gy;
    auto vert = [&](int i) { return V3<float>{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]}; };
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t tile = 0; tile < ntiles; tile++) {
        const int b = (int)(tile / (gx * gy));
        const int ty = (int)((tile % (gx * gy)) / gx), tx = (int)(tile % gx);
        const uint32_t r0 = bn.ranges[2 * tile], r1 = bn.ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
            if (pxi >= W || pyi >= H) continue;
            const int64_t pix = ((int64_t)b * H + pyi) * W + pxi;
            const V3<float> ro = {ray_o[3 * pix], ray_o[3 * pix + 1], ray_o[3 * pix + 2]};
            const V3<float> rd = {ray_d[3 * pix], ray_d[3 * pix + 1], ray_d[3 * pix + 2]};
            int ff = -1, ft = -1;
            float min_T = -1.0f, min_T_max_depth = -1.0f;
            for (uint32_t i = r0; i < r1; i++) {
                int f = (int)bn.face_list[i];
                int64_t fb = (int64_t)b * F + f;
                if (min_T >= 0.0f && bn.min_depths[fb] > min_T_max_depth) break;      // forward.cu:648-651
                V3<float> tuv;
                if (!ray_tri_intersection(ro, rd, vert(faces[3 * f]), vert(faces[3 * f + 1]), vert(faces[3 * f + 2]), tuv)) continue;
                bool hit = (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
                if (!hit) continue;
                if (min_T < 0.0f || tuv.x < min_T) { min_T = tuv.x; min_T_max_depth = bn.max_depths[fb]; ff = f; }
            }
            if (ff >= 0) {
                for (int i = 0; i < 2; i++) {
                    int tet = face_tets[2 * ff + i];
                    if (tet < 0) continue;
                    V3<float> n = tet_face_outward_normal(verts, faces, tets, ff, tet);
                    if (vdot(n, rd) < 0.0f) ft = tet;
                }
            }
            first_face[pix] = ff; first_tet[pix] = ft;

            // ---- tet walk ----
            bool done = (ff == -1 || ft == -1);
            int curr_face = ff, curr_tet = ft, ndone = 0;
            int steps = 0;      // a walk cannot cross more than T tets; guards against numeric cycles
            while (!done) {
                if (++steps > T + 1) break;
                if (face_exist[curr_face]) {
                    if (ndone < L) layers[pix * L + ndone] = curr_face;   // reference writes unguarded (L==0 would overflow)
                    ndone++;
                    if (ndone >= L) done = true;
                }
                int next_face = -1, next_tet = -1;
                if (curr_tet == -1) done = true;
                if (!done) {
                    int others[4]; int cnt = 0;
                    for (int i = 0; i < 4; i++) {
                        int tf = tet_faces[4 * curr_tet + i];
                        if (tf == curr_face) continue;
                        others[cnt++] = tf;                               // cnt can reach 4 (reference overflows [3])
                    }
                    if (cnt != 3) done = true;
                    V3<float> ncur = tet_face_outward_normal(verts, faces, tets, curr_face, curr_tet);
                    if (vdot(ncur, rd) >= 0.0f) done = true;
                    int ncand = 0;
                    for (int i = 0; i < std::min(cnt, 3); i++) {
                        int of = others[i];
                        V3<float> tuv;
                        if (!ray_tri_intersection(ro, rd, vert(faces[3 * of]), vert(faces[3 * of + 1]), vert(faces[3 * of + 2]), tuv)) continue;
                        bool hit = (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
                        V3<float> no = tet_face_outward_normal(verts, faces, tets, of, curr_tet);
                        if (hit && vdot(no, rd) > 0.0f) { next_face = of; ncand++; }
                    }
                    if (ncand != 1) done = true;
                    else {
                        for (int i = 0; i < 2; i++) {
                            int pt = face_tets[2 * next_face + i];
                            if (pt == curr_tet) continue;
                            next_tet = pt; break;
                        }
                    }
                    curr_face = next_face; curr_tet = next_tet;
                }
            }
            layers_cnt[pix] = ndone;
        }
    }
}

template <class R>
RenderArgs<R> make_args(int B, int P, int F, int W, int H, int K, double temp, const int* patch_min,
                        const R* background, const R* verts, const int* faces, const R* verts_color,
                        const R* faces_opacity, const R* verts_ndc, const R* faces_intense,
                        const R* aa_verts, const R* aa_edges, const uint8_t* aa_iszero, const R* aa_recip,
                        const R* aa_normal, const R* aa_normal_c, const R* ray_o, const R* ray_d,
                        const uint32_t* ranges, const uint32_t* face_list) {
    RenderArgs<R> a;
    a.B = B; a.P = P; a.F = F; a.W = W; a.H = H; a.K = K; a.temp = (R)temp; a.patch_min = patch_min;
    a.background = background; a.verts = verts; a.faces = faces; a.verts_color = verts_color;
    a.faces_opacity = faces_opacity; a.verts_ndc = verts_ndc; a.faces_intense = faces_intense;
    a.aa_verts = aa_verts; a.aa_edges = aa_edges; a.aa_iszero = aa_iszero; a.aa_recip = aa_recip;
    a.aa_normal = aa_normal; a.aa_normal_c = aa_normal_c; a.ray_o = ray_o; a.ray_d = ray_d;
    a.ranges = ranges; a.face_list = face_list;
    return a;
}

}  // namespace

// ===========================================================================
// C ABI (ctypes)
// ===========================================================================
extern "C" {

int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void* orc_binning_create(int B, int P, int F, int W, int H, const int* patch_min, const int* faces,
                         const float* verts_ndc, const float* verts_image, int key_min_depth) {
    return do_binning(B, P, F, W, H, patch_min, faces, verts_ndc, verts_image, key_min_depth);
}
int64_t orc_binning_num_rendered(void* h) { return ((Binning*)h)->num_rendered; }
void orc_binning_copy(void* h, float* depths, float* min_depths, float* max_depths, uint32_t* tiles_touched,
                      uint64_t* keys, uint32_t* face_list, uint32_t* ranges) {
    Binning* b = (Binning*)h;
    auto cp = [](void* d, const void* s, size_t n) { if (d && n) std::memcpy(d, s, n); };
    cp(depths, b->depths.data(), b->depths.size() * 4);
    cp(min_depths, b->min_depths.data(), b->min_depths.size() * 4);
    cp(max_depths, b->max_depths.data(), b->max_depths.size() * 4);
    cp(tiles_touched, b->tiles_touched.data(), b->tiles_touched.size() * 4);
    cp(keys, b->keys.data(), b->keys.size() * 8);
    cp(face_list, b->face_list.data(), b->face_list.size() * 4);
    cp(ranges, b->ranges.data(), b->ranges.size() * 4);
}
void orc_binning_free(void* h) { delete (Binning*)h; }

#define ORC_RENDER_API(SUF, R)                                                                              \
    void orc_render_forward_##SUF(int B, int P, int F, int W, int H, int K, double temp, const int* patch_min, \
        const R* background, const R* verts, const int* faces, const R* verts_color, const R* faces_opacity, \
        const R* verts_ndc, const R* faces_intense, const R* aa_verts, const R* aa_edges,                   \
        const uint8_t* aa_iszero, const R* aa_recip, const R* aa_normal, const R* aa_normal_c,              \
        const R* ray_o, const R* ray_d, const uint32_t* ranges, const uint32_t* face_list,                  \
        R* out_color, R* out_depth, R* final_T, R* final_prev_T, uint32_t* n_contrib,                       \
        R* buf_oarea, int* buf_tri_id, int* buf_tri_cnt, R* buf_doarea, int nthreads) {                     \
        auto a = make_args<R>(B, P, F, W, H, K, temp, patch_min, background, verts, faces, verts_color,     \
            faces_opacity, verts_ndc, faces_intense, aa_verts, aa_edges, aa_iszero, aa_recip, aa_normal,    \
            aa_normal_c, ray_o, ray_d, ranges, face_list);                                                  \
        render_forward<R>(a, out_color, out_depth, final_T, final_prev_T, n_contrib, buf_oarea, buf_tri_id, \
            buf_tri_cnt, buf_doarea, nthreads);                                                             \
    }                                                                                                       \
    void orc_render_backward_##SUF(int B, int P, int F, int W, int H, int K, double temp, const int* patch_min, \
        const R* background, const R* verts, const int* faces, const R* verts_color, const R* faces_opacity, \
        const R* verts_ndc, const R* faces_intense, const R* aa_verts, const R* aa_edges,                   \
        const uint8_t* aa_iszero, const R* aa_recip, const R* aa_normal, const R* aa_normal_c,              \
        const R* ray_o, const R* ray_d, const uint32_t* ranges, const uint32_t* face_list,                  \
        const R* dL_dcolor, const R* dL_ddepth, const R* final_T, const R* final_prev_T,                    \
        const uint32_t* n_contrib, const R* buf_oarea, const int* buf_tri_id, const int* buf_tri_cnt,       \
        const R* buf_doarea, R* dL_dverts, R* dL_dverts_color, R* dL_dfaces_opacity, R* dL_dverts_ndc,      \
        R* dL_dfaces_intense, R* dL_daa_face_verts, int corrected_dv, int nthreads) {                       \
        auto a = make_args<R>(B, P, F, W, H, K, temp, patch_min, background, verts, faces, verts_color,     \
            faces_opacity, verts_ndc, faces_intense, aa_verts, aa_edges, aa_iszero, aa_recip, aa_normal,    \
            aa_normal_c, ray_o, ray_d, ranges, face_list);                                                  \
        render_backward<R>(a, dL_dcolor, dL_ddepth, final_T, final_prev_T, n_contrib, buf_oarea, buf_tri_id, \
            buf_tri_cnt, buf_doarea, dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc,          \
            dL_dfaces_intense, dL_daa_face_verts, corrected_dv, nthreads);                                  \
    }                                                                                                       \
    int orc_aa_overlap_##SUF(const R* tv, const R* te, const uint8_t* tz, const R* tr, const R* tn,         \
        const R* tc, R pxmin, R pymin, R* area, R* grad) {                                                  \
        AATri<R> t = {tv, te, tz, tr, tn, tc};                                                              \
        R txmin = std::min(std::min(tv[0], tv[2]), tv[4]), txmax = std::max(std::max(tv[0], tv[2]), tv[4]); \
        R tymin = std::min(std::min(tv[1], tv[3]), tv[5]), tymax = std::max(std::max(tv[1], tv[3]), tv[5]); \
        R g[3][2] = {{0, 0}, {0, 0}, {0, 0}};                                                               \
        *area = 0;                                                                                          \
        int e = tri_pix_overlap_area<R>(t, txmin, txmax, tymin, tymax, pxmin, pxmin + 1, pymin, pymin + 1, (R)1, area, g); \
        for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) grad[2 * i + j] = g[i][j];                  \
        return e;                                                                                           \
    }                                                                                                       \
    int orc_ray_tri_##SUF(const R* ro, const R* rd, const R* p, R* tuv, R* grads, int corrected) {          \
        V3<R> o = {ro[0], ro[1], ro[2]}, d = {rd[0], rd[1], rd[2]};                                         \
        V3<R> p0 = {p[0], p[1], p[2]}, p1 = {p[3], p[4], p[5]}, p2 = {p[6], p[7], p[8]};                    \
        V3<R> t = {0, 0, 0};                                                                                \
        bool ok = ray_tri_intersection<R>(o, d, p0, p1, p2, t);                                             \
        tuv[0] = t.x; tuv[1] = t.y; tuv[2] = t.z;                                                           \
        V3<R> g[6];                                                                                         \
        ray_tri_intersection_grad<R>(o, d, p0, p1, p2, g[0], g[1], g[2], g[3], g[4], g[5], corrected != 0); \
        for (int i = 0; i < 6; i++) { grads[3 * i] = g[i].x; grads[3 * i + 1] = g[i].y; grads[3 * i + 2] = g[i].z; } \
        return ok ? 1 : 0;                                                                                  \
    }                                                                                                       \
    int orc_clamp_bary_##SUF(R u, R v, R* out) {                                                            \
        int code; clamp_bary_uv<R>(u, v, out[0], out[1], code);                                             \
        clamp_bary_uv_grad<R>(code, out[2], out[3], out[4], out[5]);                                        \
        return code;                                                                                        \
    }

ORC_RENDER_API(f32, float)
ORC_RENDER_API(f64, double)

void orc_patch_rect(uint32_t pmx, uint32_t pmy, const float* p, uint32_t gx, uint32_t gy, uint32_t* out) {
    patch_rect_from_tri(pmx, pmy, p, p + 2, p + 4, gx, gy, out, out + 2);
}
uint32_t orc_higher_msb(uint32_t n) { return get_higher_msb(n); }

void orc_render_layers(int B, int P, int F, int T, int W, int H, const float* verts, const int* faces,
                       const int* tets, const int* face_tets, const int* tet_faces, const int* face_exist,
                       const float* ray_o, const float* ray_d, void* binning, int L,
                       int* first_face, int* first_tet, int* layers, int* layers_cnt, int nthreads) {
    render_layers(B, P, F, T, W, H, verts, faces, tets, face_tets, tet_faces, face_exist, ray_o, ray_d,
                  *(Binning*)binning, L, first_face, first_tet, layers, layers_cnt, nthreads);
}

}  // extern "C"
