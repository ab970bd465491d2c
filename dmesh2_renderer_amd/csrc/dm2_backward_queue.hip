// dm2_backward_queue.hip -- backward composite, dense pairs with survivor compaction.
//
// Same results as k_render_backward (dm2_backward.hip; BACKWARD::renderCUDA<3>,
// backward.cu:17-532) up to fp32 summation order of the scattered gradients.
// Per chunk of staged faces (walked back to front):
//   A   stage, exact pixel rectangle per face, block scan -> pair index k
//   B1  one pair per lane: corner / half-plane classification only (aa.h:103-149);
//       survivors compacted in pair order into an LDS queue (as dm2_forward_queue.hip)
// then per round of 256 survivors:
//   B2  lane s: AA area + Jacobian, Moeller-Trumbore, clamp, coverage, alpha, interpolated
//       colour / depth of survivor s -> small record in LDS; Jacobian / barycentrics stay
//       in registers
//   C   pixel p: replay its records of the round back to front (transmittance recovery,
//       dL/dalpha recurrence, backward.cu:340-405) -> T, dL/dalpha into the record
//   D   lane s: chain rule (backward.cu:408-488), DPP pre-reduction over the lanes of one
//       face, ds_add_f32 into the per-(tile,entry) accumulators
// and per chunk a flush with (entry,component) global atomics.
#include <hip/hip_runtime.h>

#include "dm2_clip_grad.h"
#include "dm2_device_math.h"
#include "dm2_dpp.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_BQ_CHUNK
#define DM2_BQ_CHUNK 24      // 4 blocks/CU need <= 40 KB LDS (24 faces: 38.8 KB, 29 is the most that fits) and <= 128 VGPRs.
                             // A/B at cfg4 on MI355X, 4 blocks/CU: 20: 2.23 ms, 22: 2.14, 24: 2.13, 26: 2.15, 28: 2.14;
                             // 3 blocks/CU (153 VGPRs, (x,y) corner table, 30 faces): 2.29
#endif
#ifndef DM2_BQ_BLOCKS
#define DM2_BQ_BLOCKS 4       // resident blocks per CU the register budget is set for (128 VGPRs, 12 B/lane of scratch)
#endif
#ifndef DM2_BQ_PAIRCAP
#define DM2_BQ_PAIRCAP 512
#endif
#ifndef DM2_BQ_TAILMIN
#define DM2_BQ_TAILMIN 256    // a last round with fewer survivors than this is cut off and its faces staged again.  With 256
                              // a chunk never runs a partial second round (7 barriers for a few dozen lanes of work): cfg4
                              // backward 2.64 ms (no cut), 2.53 (64), 2.40 (96..192), 2.38 (256, chunk 30)
#endif
constexpr int BQ_CHUNK = DM2_BQ_CHUNK;
constexpr int BQ_PAIRCAP = DM2_BQ_PAIRCAP;
constexpr int BQ_TAILMIN = DM2_BQ_TAILMIN;
constexpr int BQ_QCAP = ((BQ_PAIRCAP + 3) / 4 + 63) & ~63;     // queue region of one wave
static_assert(BQ_CHUNK <= 64, "one mask bit per staged face");
static_assert(BQ_PAIRCAP >= TILE_PIX, "a single face may own 256 pairs");

constexpr int BQ_ACC = 32;
constexpr int Q_DV = 0, Q_DC = 9, Q_DZ = 18, Q_OP = 21, Q_IN = 22, Q_AA = 23, Q_N = 29, Q_FLAG = 31;
constexpr uint32_t QB_BLEND = 1u, QB_ACTIVE = 2u;

struct __attribute__((aligned(16))) BqPair { float alpha, c0, c1, c2, depth; uint32_t flags; float T, dL_dalpha; };
static_assert(sizeof(BqPair) == 32, "BqPair");

__global__ void __launch_bounds__(TILE_PIX, DM2_BQ_BLOCKS)
k_render_backward_queue(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                        ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                        float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                        float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                        float* __restrict__ dL_dfaces_intense, float* __restrict__ dL_daa_face_verts,
                        const uint32_t* __restrict__ hit_valid STAMP_PARAM) {
    if (hit_valid && hit_valid[0] == 2u) return;      // the forward left blend masks: dm2_backward_mask.hip does this frame
    __shared__ FaceRec recs[BQ_CHUNK];
    __shared__ float acc[BQ_CHUNK * BQ_ACC];
    __shared__ BqPair s_pair[TILE_PIX];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ int s_off[BQ_CHUNK + 1];
    __shared__ uint32_t s_rect[BQ_CHUNK];
    __shared__ int s_wave[4];
    __shared__ int s_wtot[4];                                  // survivors per wave
    __shared__ int s_inv[17];
    __shared__ uint16_t s_slot[BQ_PAIRCAP];                    // per pair: survivors before it within its wave's range
    __shared__ uint32_t s_queue[4 * BQ_QCAP];                  // survivors: q | face << 8 | corner mask << 14
    __shared__ unsigned long long s_mask[TILE_PIX];            // per pixel: faces of the chunk that blend into it in this round
    __shared__ float s_polyx[MAX_POLY * POLY_STRIDE];
#if !DM2_CLIP_COMPACT_TABLE
    __shared__ float s_polyy[MAX_POLY * POLY_STRIDE];
#else
    float* const s_polyy = nullptr;
#endif
    __shared__ uint32_t s_max_lc;

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    STAMP_DECL
    fill_inv_table(s_inv);
    s_mask[tid] = 0;
    const int lx = tid & 15, ly = tid >> 4;
    const int X0 = blockIdx.x * TILE, Y0 = blockIdx.y * TILE;
    const uint32_t px = X0 + lx, py = Y0 + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const int X0a = X0 + (int)pmx, Y0a = Y0 + (int)pmy;
    const int xlim = min(TILE - 1, d.W - 1 - X0), ylim = min(TILE - 1, d.H - 1 - Y0);
    const bool corrected = (d.flags & DM2_FLAG_CORRECTED_DV) != 0;

    float T_final = 0.f, prev_T_final = 0.f;
    uint32_t last_contributor = 0;
    float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f;
    if (inside) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s_ray[tid * 6 + k] = d.image_ray_o[3 * pix + k];
            s_ray[tid * 6 + 3 + k] = d.image_ray_d[3 * pix + k];
        }
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

    const float temp = d.aa_temperature;
    const bool use_aa = temp > 0.0f;
    const float pix_area = 1.0f;
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];

    float T = prev_T_final;
    bool T_first_pass = true;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    STAMP(0)
    int n = 0;
    for (int base = 0; base < total; base += n) {
        __syncthreads();                                            // previous chunk flushed, LDS reusable
        STAMP(1)
        // ---- phase A ----------------------------------------------------------------------
        n = min(BQ_CHUNK, total - base);
        const bool last_chunk = base + n >= total;
        int cnt = 0;
        if (tid < n) {
            // recs[j] = entry (total-1) - (base+j): back to front (backward.cu:171)
            stage_face(d, b, (int)face_list[range.x + (uint32_t)(total - 1 - base - tid)], recs[tid]);
            uint32_t rect;
            cnt = face_pixel_rect(recs[tid].aa.bb, use_aa, X0a, Y0a, xlim, ylim, rect);
            s_rect[tid] = rect;
        }
        STAMP(2)
        for (int k = tid; k < n * BQ_ACC; k += TILE_PIX) acc[k] = 0.f;
        // (scanning in wave 0 alone -- the staging lanes all live there -- saves a barrier and measured 5 % SLOWER
        // here, 2.96 vs 2.81 ms at cfg4; the forward kernel does use it, at no difference)
        int tot;
        const int ex = block_exclusive_scan(cnt, s_wave, tot);
        if (tid < n) s_off[tid] = ex;
        if (tid == n) s_off[n] = tot;
        __syncthreads();
        if (tot > BQ_PAIRCAP) {                                     // cut 1: faces [0, n) hold at most PAIRCAP pairs
            n = find_face(s_off, n, BQ_PAIRCAP);                    // >= 1: a face owns at most 256 pairs
            tot = s_off[n];
        }
        STAMP(3)

        // ---- phase B1: classify, compact survivors in pair order ---------------------------
        const int Q = (((tot + 3) >> 2) + 63) & ~63;               // pairs per wave, whole rounds of 64
        int wcount = 0;
        for (int r = 0; r < Q; r += 64) {
            const int k = wid * Q + r + lane;
            bool surv = false;
            uint32_t entry = 0;
            if (k < tot) {
                const int j = find_face(s_off, n, k);
                int qx, qy;
                pair_xy(s_rect[j], k - s_off[j], s_inv, qx, qy);
                uint32_t cmask = 0xF;
                surv = true;
                if (use_aa) {
                    const float pxmin = (float)(uint32_t)(X0a + qx), pymin = (float)(uint32_t)(Y0a + qy);
                    // (the rectangle already is the exact set of pixels that pass the bbox test, aa.h:96-101)
                    surv = classify_pixel(recs[j].aa, pxmin, pxmin + 1, pymin, pymin + 1, cmask);
                }
                entry = (uint32_t)(qy * TILE + qx) | ((uint32_t)j << 8) | (cmask << 14);
            }
            const unsigned long long bal = __ballot(surv);
            const int before = wcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (k < tot) s_slot[k] = (uint16_t)before;
            if (surv) s_queue[wid * BQ_QCAP + before] = entry;
            wcount += __popcll(bal);
        }
        if (lane == 0) s_wtot[wid] = wcount;
        __syncthreads();
        STAMP(4)
        const int wb1 = s_wtot[0], wb2 = wb1 + s_wtot[1], wb3 = wb2 + s_wtot[2];
        int S = wb3 + s_wtot[3];
        // global survivor prefix at pair k (k <= tot)
        auto surv_before = [&](int k) -> int {
            if (k >= tot) return S;
            const int w = (k >= Q) + (k >= 2 * Q) + (k >= 3 * Q);
            return (w == 0 ? 0 : (w == 1 ? wb1 : (w == 2 ? wb2 : wb3))) + (int)s_slot[k];
        };
        if (BQ_TAILMIN > 0 && !last_chunk && S > TILE_PIX && (S & (TILE_PIX - 1)) != 0 && (S & (TILE_PIX - 1)) < BQ_TAILMIN) {
            // cut 2: drop a nearly empty last round; its faces are staged again by the next chunk
            const int target = S & ~(TILE_PIX - 1);
            int lo = 1, hi = n;                                     // surv_before(off[1]) <= 256 <= target < S = surv_before(off[n])
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (surv_before(s_off[mid]) <= target) lo = mid; else hi = mid;
            }
            const int S2 = surv_before(s_off[lo]);
            n = lo; tot = s_off[lo]; S = S2;
        }

        for (int r0 = 0; r0 < S; r0 += TILE_PIX) {
            // ---- phase B2 -----------------------------------------------------------------
            const int s = r0 + tid;
            const bool have = s < S;
            int j = 0, q = 0;
            float dg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float i0 = 0.f, i1 = 0.f, i2 = 0.f, ratio = 0.f, alpha = 0.f;
            int code = 0;
            bool blend = false;
            if (have) {
                const int w = (s >= wb1) + (s >= wb2) + (s >= wb3);
                const int wb = (w == 0 ? 0 : (w == 1 ? wb1 : (w == 2 ? wb2 : wb3)));
                const uint32_t entry = s_queue[w * BQ_QCAP + (s - wb)];
                q = (int)(entry & 255u); j = (int)((entry >> 8) & 63u);
                const uint32_t cmask = entry >> 14;
                const FaceRec& fc = recs[j];
                const float pxmin = (float)(uint32_t)(X0a + (q & 15)), pxmax = pxmin + 1;
                const float pymin = (float)(uint32_t)(Y0a + (q >> 4)), pymax = pymin + 1;
                float oarea = 0.f;
                bool live = true;
                if (use_aa) {
#ifdef DM2_BWD_FAN_GRAD     // reference's per-fan-triangle accumulation order of the Jacobian (slower; kept for A/B)
                    (void)cmask;
                    const int err = tri_pix_overlap_area_lds<true>(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, s_polyx + tid, s_polyy + tid, oarea, dg);
#else
                    const int err = clip_area_grad_classified(fc.aa, pxmin, pxmax, pymin, pymax, cmask, pix_area, s_polyx + tid, s_polyy + tid, oarea, dg);
#endif
                    live = !((err != 0) || (oarea == 0.0f));
                }
                BqPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0; out.T = 0.f; out.dL_dalpha = 0.f;
                if (live) {
                    ratio = oarea / pix_area;
                    const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                    const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                    const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                    f3 tuv = {0, 0, 0};
                    if (ray_tri_intersection(ro, rd, p0, p1, p2, tuv)) {
                        float iuc, ivc;
                        clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                        i0 = 1 - iuc - ivc; i1 = iuc; i2 = ivc;
                        ratio = mix_coverage(code, ratio, temp);
                        if (ratio != 0.0f) {
                            float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                            float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                            float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                            out.c0 = c0 * fc.intense; out.c1 = c1 * fc.intense; out.c2 = c2 * fc.intense;
                            out.depth = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                            alpha = fc.opacity * ratio;
                            out.alpha = alpha;
                            out.flags = QB_BLEND;
                            blend = true;
                        }
                    }
                }
                s_pair[tid] = out;
                if (blend) atomicOr(&s_mask[q], 1ull << j);
            }
            STAMP(5)
            __syncthreads();

            // ---- phase C: per-pixel back-to-front replay ------------------------------------
            {
                unsigned long long m = s_mask[tid];
                s_mask[tid] = 0;
                while (m) {                                                       // ascending face = back to front
                    const int jj = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint32_t e = (uint32_t)(total - 1 - base - jj);        // 0-based position in the list
                    if (e >= last_contributor) continue;                          // backward.cu:219-221
                    // a mask bit is only set by a record of this round for this pixel
                    const int kk = pixel_pair(s_rect[jj], s_off[jj], lx, ly);
                    BqPair& pr = s_pair[surv_before(kk) - r0];
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
                    pr.T = T; pr.dL_dalpha = dL_dalpha; pr.flags = QB_BLEND | QB_ACTIVE;
                    // phase D needs this pixel's loss gradients, not the colours any more: hand them over in place
                    pr.c0 = dLc0; pr.c1 = dLc1; pr.c2 = dLc2; pr.depth = dLd;
                }
            }
            STAMP(6)
            __syncthreads();
            STAMP(8)

            // ---- phase D: chain rule + per-entry accumulation -------------------------------
            // Survivors are face-major, so the lanes of one face are neighbours.  Every partial is summed over
            // the run of equal faces inside its row of 16 lanes with DPP shifts (pure VALU), then only the last
            // lane of each run touches LDS: ~8 instead of up to 64 lane-atomics per ds_add_f32 (the LDS atomic
            // unit was the bottleneck of this phase).  The 29 partials are produced and retired in three groups
            // (colour/depth/opacity, AA corners, world-space corners) to keep the live register set small: all 29
            // at once needed 168 VGPRs + 4 spilled and ran 2.82 ms at cfg4, grouped 155 VGPRs, no spill, 2.64 ms.
            const int jkey = have ? j : -1;
            const int l16 = tid & 15;
            // NB: every DPP read must execute with all lanes enabled (a lane disabled by a short-circuit
            // `&&` reads as 0 for its neighbours), hence the unconditional reads first and `&`, `|` below.
            const int k1 = dpp_shr_i<1>(jkey), k2 = dpp_shr_i<2>(jkey), k4 = dpp_shr_i<4>(jkey), k8 = dpp_shr_i<8>(jkey);
            const int kn = dpp_shl_i<1>(jkey);
            const bool s1 = (l16 >= 1) & (k1 == jkey);
            const bool s2 = (l16 >= 2) & (k2 == jkey);
            const bool s4 = (l16 >= 4) & (k4 == jkey);
            const bool s8 = (l16 >= 8) & (k8 == jkey);
            BqPair pr; pr.flags = 0; pr.T = 0.f; pr.dL_dalpha = 0.f; pr.c0 = pr.c1 = pr.c2 = pr.depth = 0.f;
            if (have && blend) pr = s_pair[tid];
            const bool active = (pr.flags & QB_ACTIVE) != 0;
            float nact = active ? 1.f : 0.f;
            seg_scan16(nact, s1, s2, s4, s8);
            const bool emit = ((l16 == 15) | (kn != jkey)) & (jkey >= 0) & (nact > 0.f);
            float* const arow = acc + j * BQ_ACC;
            const FaceRec& fcD = recs[j];
            float dL_diu = 0.f, dL_div = 0.f, dL_doarea = 0.f;
            {   // group 1: vertex colours, NDC depth, intensity, opacity
                float g1[14];
#pragma unroll
                for (int c = 0; c < 14; c++) g1[c] = 0.f;
                if (active) {
                    const float Tq = pr.T, dL_dalpha = pr.dL_dalpha;
                    const float qc0 = pr.c0, qc1 = pr.c1, qc2 = pr.c2, qd = pr.depth;   // dL/dcolour, dL/ddepth of the pixel (written by phase C)
                    const float intense = fcD.intense, opacity = fcD.opacity;
                    const float dics[3] = {qc0 * alpha * Tq, qc1 * alpha * Tq, qc2 * alpha * Tq};
                    const float did = qd * alpha * Tq;
                    g1[12] = dL_dalpha * ratio;
                    const float dL_dratio = (dL_dalpha * opacity) * temp;
                    dL_doarea = dL_dratio / pix_area;
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
                    for (int c = 0; c < 12; c++) atomicAdd(arow + Q_DC + c, g1[c]);      // Q_DC..Q_DC+8, Q_DZ..Q_DZ+2 are contiguous
                    atomicAdd(arow + Q_OP, g1[12]);
                    atomicAdd(arow + Q_IN, g1[13]);
                    arow[Q_FLAG] = 1.0f;
                }
            }
            {   // group 2: AA corners
                float g2[6];
#pragma unroll
                for (int c = 0; c < 6; c++) g2[c] = dL_doarea * dg[c];                   // dL_doarea is 0 on inactive lanes
#pragma unroll
                for (int c = 0; c < 6; c++) seg_scan16(g2[c], s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 6; c++) atomicAdd(arow + Q_AA + c, g2[c]);
                }
            }
            {   // group 3: world-space corners through the ray/triangle intersection
                float g3[9];
#pragma unroll
                for (int c = 0; c < 9; c++) g3[c] = 0.f;
                if (active) {
                    const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                    const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                    const f3 p0 = {fcD.v[0], fcD.v[1], fcD.v[2]}, p1 = {fcD.v[3], fcD.v[4], fcD.v[5]}, p2 = {fcD.v[6], fcD.v[7], fcD.v[8]};
                    f3 du0, du1, du2, dv0, dv1, dv2;
                    ray_tri_intersection_grad(ro, rd, p0, p1, p2, corrected, du0, du1, du2, dv0, dv1, dv2);
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
                    for (int c = 0; c < 9; c++) atomicAdd(arow + Q_DV + c, g3[c]);
                }
            }
            STAMP(9)
            __syncthreads();      // single-buffered records: D(r) must finish before B2(r+1) overwrites them (and before the flush)
        }
        STAMP(10)

        // ---- flush: lane = (entry, component); 8 entries per pass --------------------------
        const int comp = tid & 31;
        if (comp < Q_N) {
            for (int e = tid >> 5; e < n; e += TILE_PIX / 32) {
                const float* a = acc + e * BQ_ACC;
                if (a[Q_FLAG] == 0.f) continue;
                const FaceRec& fc = recs[e];
                const float val = a[comp];
                float* dst;
                if (comp < Q_DC) dst = dL_dverts + 3 * (int64_t)fc.vid[comp / 3] + (comp % 3);
                else if (comp < Q_DZ) dst = dL_dverts_color + 3 * (int64_t)fc.vid[(comp - Q_DC) / 3] + ((comp - Q_DC) % 3);
                else if (comp < Q_OP) dst = dL_dverts_ndc + ((int64_t)b * d.P + fc.vid[comp - Q_DZ]) * 3 + 2;
                else if (comp == Q_OP) dst = dL_dfaces_opacity + fc.face_id;
                else if (comp == Q_IN) dst = dL_dfaces_intense + (int64_t)b * d.F + fc.face_id;
                else dst = dL_daa_face_verts + ((int64_t)b * d.F + fc.face_id) * 6 + (comp - Q_AA);
#ifdef DM2_ABLATE_FLUSH       // diagnostic only (wrong results): what do the global atomics of the flush cost
                if (val == 123456.789f) atomicAdd(dst, val);
#else
                atomicAdd(dst, val);
#endif
            }
        }
        STAMP(11)
    }
    STAMP_FLUSH
}

void launch_render_backward_queue(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                  const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                  float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                  float* dL_daa_face_verts, const uint32_t* hit_valid, hipStream_t st) {
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    hipLaunchKernelGGL(k_render_backward_queue, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                       dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                       hit_valid STAMP_ARG(1));
}

}  // namespace dm2
