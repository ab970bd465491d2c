// dm2_forward_dense.hip -- forward composite, dense (pixel,face)-pair formulation.
//
// Same results as k_render_forward (dm2_forward.hip; FORWARD::renderCUDA<3>,
// forward.cu:139-432), different work distribution (see dm2_pairs.h): per chunk of
// staged faces, phase B evaluates the expensive per-pair quantities (AA overlap
// area, Moeller-Trumbore, barycentric clamp, coverage mix, interpolated colour /
// depth) with one pair per lane, phase C lets every pixel blend its own pairs in
// list order.  One barrier per batch of 256 pairs (pair records double buffered).
#include <hip/hip_runtime.h>

#include "dm2_clip_area.h"
#include "dm2_clip_lds.h"
#include "dm2_device_math.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_FD_CHUNK
#define DM2_FD_CHUNK 48   // A/B on MI355X at cfg4: 32: 1.23 ms, 48: 1.15, 64: 1.34, 96: 1.31, 128: 1.72 (LDS-limited occupancy)
#endif
constexpr int FD_CHUNK = DM2_FD_CHUNK;
constexpr uint32_t PF_REC = 1u;      // AA overlap found (the reference takes an AA record here)
constexpr uint32_t PF_BLEND = 2u;    // the face blends into the pixel

struct __attribute__((aligned(8))) FwdPair { float alpha, c0, c1, c2, depth; uint32_t flags; };

__global__ void __launch_bounds__(TILE_PIX)
k_render_forward_dense(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, float* __restrict__ out_color, float* __restrict__ out_depth,
                       int32_t* __restrict__ out_tri_cnt STAMP_PARAM) {
    __shared__ FaceRec recs[FD_CHUNK];
    __shared__ FwdPair s_pair[2][TILE_PIX];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ int s_off[FD_CHUNK + 1];
    __shared__ int s_jlo[FD_CHUNK + 1];                  // first face of every 256-pair batch of the chunk
    __shared__ uint32_t s_rect[FD_CHUNK];
    __shared__ int s_kb[FD_CHUNK];                       // pair index of the face's (virtual) tile pixel (0,0): off - y0*w - x0
    __shared__ int s_wave[4];
    __shared__ int s_inv[17];
    __shared__ unsigned long long s_mask[2][TILE_PIX];   // per pixel: faces of the current batch that produced a pair for it
    __shared__ uint32_t s_ovf[2][TILE_PIX];              // per pixel: it also has pairs of faces beyond the 64 mask bits
#ifdef DM2_FWD_LDS_CLIP
    __shared__ float s_polyx[MAX_POLY * POLY_STRIDE];
    __shared__ float s_polyy[MAX_POLY * POLY_STRIDE];
#endif

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x;
    STAMP_DECL
    fill_inv_table(s_inv);
    s_mask[0][tid] = 0; s_mask[1][tid] = 0; s_ovf[0][tid] = 0; s_ovf[1][tid] = 0;
    const int lx = tid & 15, ly = tid >> 4;
    const int X0 = blockIdx.x * TILE, Y0 = blockIdx.y * TILE;
    const uint32_t px = X0 + lx, py = Y0 + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const int X0a = X0 + (int)pmx, Y0a = Y0 + (int)pmy;
    const int xlim = min(TILE - 1, d.W - 1 - X0), ylim = min(TILE - 1, d.H - 1 - Y0);

    if (inside) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s_ray[tid * 6 + k] = d.image_ray_o[3 * pix + k];
            s_ray[tid * 6 + 3 + k] = d.image_ray_d[3 * pix + k];
        }
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const float temp = d.aa_temperature;
    const bool use_aa = temp > 0.0f;
    const float pix_area = 1.0f;
    const int K = d.K;

    bool done = !inside;
    float pT = 1.0f, T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    int rec_cnt = 0;

    STAMP(0)
    int n = 0;
    for (int base = 0; base < total; base += n) {
        if (__syncthreads_count(done) == TILE_PIX) break;          // forward.cu:258-260; also fences LDS reuse
        STAMP(1)
        n = min(FD_CHUNK, total - base);
        const bool last_chunk = base + n >= total;
        int cnt = 0;
        if (tid < n) {
            stage_face(d, b, (int)face_list[range.x + base + tid], recs[tid]);
            uint32_t rect;
            cnt = face_pixel_rect(recs[tid].aa.bb, use_aa, X0a, Y0a, xlim, ylim, rect);
            s_rect[tid] = rect;
        }
        STAMP(2)
        int tot;
        const int ex = block_exclusive_scan(cnt, s_wave, tot);
        int nb = (tot + TILE_PIX - 1) / TILE_PIX;
        if (tid < n) {
            s_off[tid] = ex; note_batch_starts(s_jlo, tid, ex, cnt);
            const uint32_t r = s_rect[tid];
            s_kb[tid] = ex - (int)((r >> 4) & 15u) * ((int)((r >> 8) & 15u) + 1) - (int)(r & 15u);
        }
        if (tid == n) { s_off[n] = tot; s_jlo[nb] = n; }
        __syncthreads();
        cut_to_full_batches(s_off, s_jlo, last_chunk, n, tot, nb);
        STAMP(3)

        // ---- phase B: one pair per lane ------------------------------------------------
        auto eval_batch = [&](int bi) {
            const int k = bi * TILE_PIX + tid;
            if (k >= tot) return;
            const int jlo_b = s_jlo[bi];                               // first face of this batch (block uniform)
            const int j = find_face_in(s_off, jlo_b, min(s_jlo[bi + 1] + 1, n), k);
            const uint32_t rect = s_rect[j];
            int qx, qy;
            pair_xy(rect, k - s_off[j], s_inv, qx, qy);
            const int q = qy * TILE + qx;
            const FaceRec& fc = recs[j];
            const float pxmin = (float)(uint32_t)(X0a + qx), pxmax = pxmin + 1;
            const float pymin = (float)(uint32_t)(Y0a + qy), pymax = pymin + 1;
            FwdPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0;
            float oarea = 0.f;
            bool live = true;
            if (use_aa) {
#if defined(DM2_FWD_LDS_CLIP)
                const int err = tri_pix_overlap_area_lds<false>(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, s_polyx + tid, s_polyy + tid, oarea, nullptr);
#elif defined(DM2_FWD_REG_CLIP)
                const int err = tri_pix_overlap_area<false>(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea, nullptr);
#else
                const int err = tri_pix_overlap_area_only(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea);
#endif
                live = !((err != 0) || (oarea == 0.0f));
                if (live) out.flags |= PF_REC;
            }
            if (live) {
                float ratio = oarea / pix_area;
                const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                f3 tuv = {0, 0, 0};
                if (ray_tri_intersection(ro, rd, p0, p1, p2, tuv)) {
                    float iuc, ivc; int code;
                    clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                    const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                    ratio = mix_coverage(code, ratio, temp);
                    if (ratio != 0.0f) {
                        float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                        float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                        float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                        out.c0 = c0 * fc.intense; out.c1 = c1 * fc.intense; out.c2 = c2 * fc.intense;
                        out.depth = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                        out.alpha = fc.opacity * ratio;
                        out.flags |= PF_BLEND;
                    }
                }
            }
            s_pair[bi & 1][tid] = out;
            if (out.flags) {
                // tell pixel q which faces of this batch it has to look at (bit = face - first face of batch)
                const int bit = j - jlo_b;
                if (bit < 64) atomicOr(&s_mask[bi & 1][q], 1ull << bit);
                else s_ovf[bi & 1][q] = 1;
            }
        };

        if (nb > 0) eval_batch(0);
        STAMP(4)
        for (int bi = 0; bi < nb; bi++) {
            __syncthreads();
            STAMP(5)
            // ---- phase C: ordered blend of this pixel's pairs of batch bi ---------------
            {
                const int k0 = bi * TILE_PIX, k1 = min(k0 + TILE_PIX, tot);
                unsigned long long m = s_mask[bi & 1][tid];
                s_mask[bi & 1][tid] = 0;                                   // ready for batch bi+2
                const bool ovf = s_ovf[bi & 1][tid] != 0;
                s_ovf[bi & 1][tid] = 0;
                const int jlo = s_jlo[bi];
                auto blend_pair = [&](int kk, int j) -> bool {            // returns true when the pixel terminates
                    const FwdPair pr = s_pair[bi & 1][kk];
                    if ((pr.flags & PF_REC) && rec_cnt < K) rec_cnt++;       // forward.cu:344-352
                    if (!(pr.flags & PF_BLEND)) return false;
                    const float alpha = pr.alpha;
                    const float test_T = T * (1 - alpha);
                    C0 += pr.c0 * alpha * T; C1 += pr.c1 * alpha * T; C2 += pr.c2 * alpha * T;
                    D += pr.depth * alpha * T;
                    pT = T; T = test_T;
                    last_contributor = (uint32_t)(base + j + 1);
                    return T < T_EPS;
                };
                // a mask bit is only ever set by a pair of this batch that covers this pixel, so the pair's
                // slot follows from the face's row pitch without any rectangle or range test
                const int lin = lx - k0;
                while (m && !done) {
                    const int j = jlo + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int w = (int)((s_rect[j] >> 8) & 15u) + 1;
                    if (blend_pair(s_kb[j] + ly * w + lin, j)) done = true;
                }
                if (ovf) {                                                 // > 64 faces in one batch: plain walk of the rest
                    const int jhi = find_face(s_off, n, k1 - 1);
                    for (int j = jlo + 64; j <= jhi && !done; j++) {
                        const int o = s_off[j];
                        if (s_off[j + 1] == o) continue;
                        const int k = pixel_pair(s_rect[j], o, lx, ly);
                        if (k < k0 || k >= k1) continue;
                        if (blend_pair(k - k0, j)) done = true;
                    }
                }
            }
            STAMP(6)
            if (bi + 1 < nb) eval_batch(bi + 1);
            STAMP(4)
        }
    }

    if (inside) {
        is.final_prev_T[pix] = pT;
        is.final_T[pix] = T;
        is.n_contrib[pix] = last_contributor;
        out_color[3 * pix] = C0 + T * d.background[0];
        out_color[3 * pix + 1] = C1 + T * d.background[1];
        out_color[3 * pix + 2] = C2 + T * d.background[2];
        out_depth[pix] = D + T * 1.0f;
        if (out_tri_cnt) out_tri_cnt[pix] = rec_cnt;
    }
    STAMP(7)
    STAMP_FLUSH
}

void launch_render_forward_dense(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, hipStream_t st) {
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    StageTimer tm(ST_FWD, st);
    hipLaunchKernelGGL(k_render_forward_dense, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth, out_tri_cnt STAMP_ARG(0));
}

}  // namespace dm2
