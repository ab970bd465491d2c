// dm2_backward_point.hip -- backward composite for aa_temperature == 0 (point-sampled coverage).
//
// Same results as k_render_backward (dm2_backward.hip; BACKWARD::renderCUDA<3>, backward.cu:17-532) up to
// fp32 summation order of the scattered gradients.  With temperature 0 the reference applies no bounding-box
// test (backward.cu:241-244): every face of a tile's list is intersected with all 256 pixel rays, and a face
// contributes to a pixel only where the clamped barycentrics say "inside" (coverage 1, else 0).  That test is
// cheap and dense; the gradient chain behind it is expensive and sparse (about 4 hits per pixel out of
// ~250 faces per tile at 1080p / 1 M faces).  Per chunk of 48 staged faces (walked back to front):
//
//   B1  wave w = pixels [64w, 64w+64), ray in registers: for every face of the chunk the intersection +
//       inside test, one ballot per (face, wave) -> 64-bit hit mask in LDS.  All lanes busy, no divergence.
//       When the forward of this frame was dm2_forward_point.hip, the masks it stored are simply loaded.
//   scan over the 192 (face, wave) hit counts -> every hit gets a slot, face-major.
// then per round of 256 hits:
//   B2  lane s: locate its (face, pixel) from the hit masks, recompute the intersection, barycentrics,
//       colour / depth / alpha -> record in LDS
//   C   pixel p: replay its records back to front (backward.cu:340-405)
//   D   lane s: chain rule (backward.cu:408-488), DPP pre-reduction over the lanes of one face, ds_add_f32
// and per chunk a flush with (entry,component) global atomics (AA gradients are identically zero here).
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_dpp.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_BP_CHUNK
#define DM2_BP_CHUNK 48      // A/B at cfg4, temperature 0, MI355X: 64 faces / 3 blocks per CU 1.57 ms, 48 / 3: 1.55,
#endif                       // 48 / 4 (128 VGPRs, 36 B/lane scratch, 36.4 KB LDS): 1.43, 40 / 4: 1.43, 32 / 4: 1.50
#ifndef DM2_BP_BLOCKS
#define DM2_BP_BLOCKS 4
#endif
constexpr int BP_CHUNK = DM2_BP_CHUNK;       // one mask bit per staged face, one scan thread per (face, wave)
static_assert(BP_CHUNK <= 64 && BP_CHUNK * 4 <= TILE_PIX, "chunk");
constexpr int BP_SLOTS = BP_CHUNK * 4;
constexpr int BP_ACC = 32;
constexpr int P_DV = 0, P_DC = 9, P_DZ = 18, P_OP = 21, P_IN = 22, P_N = 23, P_FLAG = 31;
constexpr uint32_t PB_BLEND = 1u, PB_ACTIVE = 2u;

struct __attribute__((aligned(16))) BpPair { float alpha, c0, c1, c2, depth; uint32_t flags; float T, dL_dalpha; };
static_assert(sizeof(BpPair) == 32, "BpPair");

// index of the n-th (0-based) set bit of m; n < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long lowmask = (w == 64) ? ~0ull : ((1ull << w) - 1ull);
        const int c = __popcll((m >> pos) & lowmask);
        if (n >= c) { n -= c; pos += w; }
    }
    return pos;
}

__global__ void __launch_bounds__(TILE_PIX, DM2_BP_BLOCKS)
k_render_backward_point(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                        ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                        float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                        float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                        float* __restrict__ dL_dfaces_intense, const uint64_t* __restrict__ hit_masks,
                        const uint32_t* __restrict__ hit_valid) {
    __shared__ FaceRec recs[BP_CHUNK];
    __shared__ float acc[BP_CHUNK * BP_ACC];
    __shared__ BpPair s_pair[TILE_PIX];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ unsigned long long s_hit[BP_CHUNK * 4];         // [face][wave]: which of the wave's 64 pixels the face blends into
    __shared__ int s_base[BP_CHUNK * 4 + 1];                   // exclusive scan of the hit counts, face-major
    __shared__ int s_wave[4];
    __shared__ unsigned long long s_mask[TILE_PIX];            // per pixel: faces of the chunk with a record in this round
    __shared__ uint32_t s_max_lc;

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    s_mask[tid] = 0;
    const int lx = tid & 15, ly = tid >> 4;
    const uint32_t px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const bool corrected = (d.flags & DM2_FLAG_CORRECTED_DV) != 0;

    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    float T_final = 0.f, prev_T_final = 0.f;
    uint32_t last_contributor = 0;
    float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f;
    if (inside) {
        pixel_ray(d, b, pix, px + (uint32_t)d.patch_min[2 * b], py + (uint32_t)d.patch_min[2 * b + 1], d.full_W, d.full_H, ro, rd);
        s_ray[tid * 6] = ro.x; s_ray[tid * 6 + 1] = ro.y; s_ray[tid * 6 + 2] = ro.z;
        s_ray[tid * 6 + 3] = rd.x; s_ray[tid * 6 + 4] = rd.y; s_ray[tid * 6 + 5] = rd.z;
        T_final = is.final_T[pix]; prev_T_final = is.final_prev_T[pix];
        last_contributor = is.n_contrib[pix];
        dLc0 = dL_dcolor[3 * pix]; dLc1 = dL_dcolor[3 * pix + 1]; dLc2 = dL_dcolor[3 * pix + 2];
        dLd = dL_ddepth[pix];
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];

    if (tid == 0) s_max_lc = 0;
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_lc, last_contributor);
    __syncthreads();
    const int total = (int)min(s_max_lc, range.y - range.x);       // entries behind every pixel's last contributor are dead

    const float temp = d.aa_temperature;                           // == 0 (the launcher dispatches on it)
    const float pix_area = 1.0f;
    const bool use_masks = hit_masks && hit_valid && (hit_valid[0] == 1u);   // written by the matching forward?
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];

    float T = prev_T_final;
    bool T_first_pass = true;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    for (int base = 0; base < total; base += BP_CHUNK) {
        __syncthreads();                                            // previous chunk flushed, LDS reusable
        const int n = min(BP_CHUNK, total - base);
        // recs[j] = entry (total-1) - (base+j): back to front (backward.cu:171)
        if (tid < n) stage_face(is.face_recs, (int64_t)b * d.F + face_list[range.x + (uint32_t)(total - 1 - base - tid)], recs[tid]);
        for (int k = tid; k < n * BP_ACC; k += TILE_PIX) acc[k] = 0.f;
        __syncthreads();

        // ---- phase B1: which pixels does each face blend into --------------------------------
        if (use_masks) {
            // the forward (dm2_forward_point.hip) left one ballot per (list entry, wave): a set bit means the entry
            // blended into that pixel there, which implies that it lies before the pixel's last contributor
            if (tid < n * 4) {
                const int64_t entry = (int64_t)range.x + (uint32_t)(total - 1 - base - (tid >> 2));
                s_hit[tid] = hit_masks[entry * 4 + (tid & 3)];
            }
        } else {
            for (int j = 0; j < n; j++) {
                const FaceRec& fc = recs[j];
                const uint32_t e = (uint32_t)(total - 1 - base - j);                  // 0-based position in the list
                bool hit = inside && (e < last_contributor);                          // backward.cu:219-221
                const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                f3 tuv = {0, 0, 0};
                const bool ok = ray_tri_intersection(ro, rd, p0, p1, p2, tuv);
                float iuc, ivc; int code;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                const float ratio = mix_coverage(code, 0.0f / pix_area, temp);        // 1 inside, 0 outside at temperature 0
                hit = hit && ok && (ratio != 0.0f);
                const unsigned long long bal = __ballot(hit);
                if (lane == 0) s_hit[j * 4 + wid] = bal;
            }
        }
        __syncthreads();
        int S;
        {
            const int cnt = (tid < n * 4) ? __popcll(s_hit[tid]) : 0;             // thread = (face, wave), face-major
            const int ex = block_exclusive_scan(cnt, s_wave, S);
            if (tid < BP_SLOTS) s_base[tid] = ex;
            if (tid == TILE_PIX - 1) s_base[BP_SLOTS] = S;
        }
        __syncthreads();

        for (int r0 = 0; r0 < S; r0 += TILE_PIX) {
            // ---- phase B2: one hit per lane ---------------------------------------------------
            const int s = r0 + tid;
            const bool have = s < S;
            int j = 0, q = 0;
            float i0 = 0.f, i1 = 0.f, i2 = 0.f, ratio = 0.f, alpha = 0.f;
            int code = 0;
            bool blend = false;
            if (have) {
                int lo = 0, hi = BP_SLOTS;                                        // s_base[lo] <= s < s_base[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_base[mid] <= s) lo = mid; else hi = mid;
                }
                j = lo >> 2;
                q = ((lo & 3) << 6) + nth_set_bit(s_hit[lo], s - s_base[lo]);
                const FaceRec& fc = recs[j];
                const f3 qo = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                const f3 qd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                BpPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0; out.T = 0.f; out.dL_dalpha = 0.f;
                f3 tuv = {0, 0, 0};
                if (ray_tri_intersection(qo, qd, p0, p1, p2, tuv)) {
                    float iuc, ivc;
                    clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                    i0 = 1 - iuc - ivc; i1 = iuc; i2 = ivc;
                    ratio = mix_coverage(code, 0.0f / pix_area, temp);
                    if (ratio != 0.0f) {
                        float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                        float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                        float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                        out.c0 = c0 * fc.intense; out.c1 = c1 * fc.intense; out.c2 = c2 * fc.intense;
                        out.depth = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                        alpha = fc.opacity * ratio;
                        out.alpha = alpha;
                        out.flags = PB_BLEND;
                        blend = true;
                    }
                }
                s_pair[tid] = out;
                if (blend) atomicOr(&s_mask[q], 1ull << j);
            }
            __syncthreads();

            // ---- phase C: per-pixel back-to-front replay ------------------------------------
            {
                unsigned long long m = s_mask[tid];
                s_mask[tid] = 0;
                while (m) {                                                       // ascending face = back to front
                    const int jj = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    // slot of (face jj, this pixel): hits before (jj, this wave) + hits of lower pixels of this wave
                    const int t = jj * 4 + wid;
                    const int slot = s_base[t] + __popcll(s_hit[t] & ((1ull << lane) - 1ull));
                    BpPair& pr = s_pair[slot - r0];
                    const float a = pr.alpha, iC0 = pr.c0, iC1 = pr.c1, iC2 = pr.c2, iD = pr.depth;
                    if (!T_first_pass) T = T / (1.f - a);                         // backward.cu:340-348
                    T_first_pass = false;
                    float dL_dalpha = 0.0f;
                    accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = iC0;
                    dL_dalpha += (iC0 - accum_rec0) * dLc0;
                    accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = iC1;
                    dL_dalpha += (iC1 - accum_rec1) * dLc1;
                    accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = iC2;
                    dL_dalpha += (iC2 - accum_rec2) * dLc2;
                    accum_recd = last_alpha * last_depth + (1.f - last_alpha) * accum_recd; last_depth = iD;
                    dL_dalpha += (iD - accum_recd) * dLd;
                    dL_dalpha *= T;
                    last_alpha = a;
                    float bg_dot = 0.f;
                    bg_dot += bg0 * dLc0; bg_dot += bg1 * dLc1; bg_dot += bg2 * dLc2;
                    const float bd_dot = (float)(0.0 + 1.0 * (double)dLd);        // backward.cu:394
                    if (a == 1.0f) {
                        dL_dalpha += (-prev_T_final) * bg_dot;
                        dL_dalpha += (-prev_T_final) * bd_dot;
                    } else {
                        dL_dalpha += (-T_final / (1.f - a)) * bg_dot;
                        dL_dalpha += (-T_final / (1.f - a)) * bd_dot;
                    }
                    pr.T = T; pr.dL_dalpha = dL_dalpha; pr.flags = PB_BLEND | PB_ACTIVE;
                    // phase D needs this pixel's loss gradients, not the colours any more: hand them over in place
                    pr.c0 = dLc0; pr.c1 = dLc1; pr.c2 = dLc2; pr.depth = dLd;
                }
            }
            __syncthreads();

            // ---- phase D: chain rule + per-entry accumulation (see dm2_backward_queue.hip) ----
            const int jkey = have ? j : -1;
            const int l16 = tid & 15;
            // NB: every DPP read must execute with all lanes enabled, hence the unconditional reads and `&`, `|`.
            const int k1 = dpp_shr_i<1>(jkey), k2 = dpp_shr_i<2>(jkey), k4 = dpp_shr_i<4>(jkey), k8 = dpp_shr_i<8>(jkey);
            const int kn = dpp_shl_i<1>(jkey);
            const bool s1 = (l16 >= 1) & (k1 == jkey);
            const bool s2 = (l16 >= 2) & (k2 == jkey);
            const bool s4 = (l16 >= 4) & (k4 == jkey);
            const bool s8 = (l16 >= 8) & (k8 == jkey);
            BpPair pr; pr.flags = 0; pr.T = 0.f; pr.dL_dalpha = 0.f; pr.c0 = pr.c1 = pr.c2 = pr.depth = 0.f;
            if (have && blend) pr = s_pair[tid];
            const bool active = (pr.flags & PB_ACTIVE) != 0;
            float nact = active ? 1.f : 0.f;
            seg_scan16(nact, s1, s2, s4, s8);
            const bool emit = ((l16 == 15) | (kn != jkey)) & (jkey >= 0) & (nact > 0.f);
            float* const arow = acc + j * BP_ACC;
            const FaceRec& fcD = recs[j];
            float dL_diu = 0.f, dL_div = 0.f;
            {   // group 1: vertex colours, NDC depth, intensity, opacity
                float g1[14];
#pragma unroll
                for (int c = 0; c < 14; c++) g1[c] = 0.f;
                if (active) {
                    const float Tq = pr.T, dL_dalpha = pr.dL_dalpha;
                    const float qc0 = pr.c0, qc1 = pr.c1, qc2 = pr.c2, qd = pr.depth;   // dL/dcolour, dL/ddepth of the pixel
                    const float intense = fcD.intense;
                    const float dics[3] = {qc0 * alpha * Tq, qc1 * alpha * Tq, qc2 * alpha * Tq};
                    const float did = qd * alpha * Tq;
                    g1[12] = dL_dalpha * ratio;
                    float dL_di0 = 0.f, dL_di1 = 0.f, dL_di2 = 0.f, dL_dfint = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dL_di0 += fcD.col[ch] * dics[ch] * intense;
                        dL_di1 += fcD.col[3 + ch] * dics[ch] * intense;
                        dL_di2 += fcD.col[6 + ch] * dics[ch] * intense;
                        g1[ch] = 0.f + i0 * dics[ch] * intense;
                        g1[3 + ch] = 0.f + i1 * dics[ch] * intense;
                        g1[6 + ch] = 0.f + i2 * dics[ch] * intense;
                        dL_dfint += (i0 * fcD.col[ch] + i1 * fcD.col[3 + ch] + i2 * fcD.col[6 + ch]) * dics[ch];
                    }
                    g1[13] = dL_dfint;
                    dL_di0 += fcD.dep[0] * did; dL_di1 += fcD.dep[1] * did; dL_di2 += fcD.dep[2] * did;
                    g1[9] = 0.f + i0 * did; g1[10] = 0.f + i1 * did; g1[11] = 0.f + i2 * did;
                    float diuc_diu, diuc_div, divc_diu, divc_div;
                    clamp_bary_uv_grad(code, diuc_diu, diuc_div, divc_diu, divc_div);
                    const float di0_diu = -1.f * diuc_diu + -1.f * divc_diu, di0_div = -1.f * diuc_div + -1.f * divc_div;
                    const float di1_diu = 1.f * diuc_diu + 0.f * divc_diu, di1_div = 1.f * diuc_div + 0.f * divc_div;
                    const float di2_diu = 0.f * diuc_diu + 1.f * divc_diu, di2_div = 0.f * diuc_div + 1.f * divc_div;
                    dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
                    dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;
                }
#pragma unroll
                for (int c = 0; c < 14; c++) seg_scan16(g1[c], s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 12; c++) atomicAdd(arow + P_DC + c, g1[c]);      // P_DC..+8 and P_DZ..+2 are contiguous
                    atomicAdd(arow + P_OP, g1[12]);
                    atomicAdd(arow + P_IN, g1[13]);
                    arow[P_FLAG] = 1.0f;
                }
            }
            {   // group 2: world-space corners through the ray/triangle intersection
                float g3[9];
#pragma unroll
                for (int c = 0; c < 9; c++) g3[c] = 0.f;
                if (active) {
                    const f3 qo = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                    const f3 qd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                    const f3 p0 = {fcD.v[0], fcD.v[1], fcD.v[2]}, p1 = {fcD.v[3], fcD.v[4], fcD.v[5]}, p2 = {fcD.v[6], fcD.v[7], fcD.v[8]};
                    f3 du0, du1, du2, dv0, dv1, dv2;
                    ray_tri_intersection_grad(qo, qd, p0, p1, p2, corrected, du0, du1, du2, dv0, dv1, dv2);
                    const f3 dp0 = dL_diu * du0 + dL_div * dv0;
                    const f3 dp1 = dL_diu * du1 + dL_div * dv1;
                    const f3 dp2 = dL_diu * du2 + dL_div * dv2;
                    g3[0] = dp0.x; g3[1] = dp0.y; g3[2] = dp0.z;
                    g3[3] = dp1.x; g3[4] = dp1.y; g3[5] = dp1.z;
                    g3[6] = dp2.x; g3[7] = dp2.y; g3[8] = dp2.z;
                }
#pragma unroll
                for (int c = 0; c < 9; c++) seg_scan16(g3[c], s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 9; c++) atomicAdd(arow + P_DV + c, g3[c]);
                }
            }
            __syncthreads();      // single-buffered records: D(r) must finish before B2(r+1) overwrites them (and before the flush)
        }

        // ---- flush: lane = (entry, component); 8 entries per pass.  dL/daa_face_verts is identically zero
        // at temperature 0 (dL_dratio = dL_dalpha * opacity * temp, backward.cu:410) and stays at its memset value.
        const int comp = tid & 31;
        if (comp < P_N) {
            for (int e = tid >> 5; e < n; e += TILE_PIX / 32) {
                const float* a = acc + e * BP_ACC;
                if (a[P_FLAG] == 0.f) continue;
                const FaceRec& fc = recs[e];
                const float val = a[comp];
                float* dst;
                if (comp < P_DC) dst = dL_dverts + 3 * (int64_t)fc.vid[comp / 3] + (comp % 3);
                else if (comp < P_DZ) dst = dL_dverts_color + 3 * (int64_t)fc.vid[(comp - P_DC) / 3] + ((comp - P_DC) % 3);
                else if (comp < P_OP) dst = dL_dverts_ndc + ((int64_t)b * d.P + fc.vid[comp - P_DZ]) * 3 + 2;
                else if (comp == P_OP) dst = dL_dfaces_opacity + fc.face_id;
                else dst = dL_dfaces_intense + (int64_t)b * d.F + fc.face_id;
                atomicAdd(dst, val);
            }
        }
    }
}

void launch_render_backward_point(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                  const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                  float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                  const uint64_t* hit_masks, const uint32_t* hit_valid, hipStream_t st) {
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    StageTimer tm(ST_BWD, st);
    hipLaunchKernelGGL(k_render_backward_point, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                       dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, hit_masks, hit_valid);
}

}  // namespace dm2
