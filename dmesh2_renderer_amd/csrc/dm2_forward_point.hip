// dm2_forward_point.hip -- forward composite for aa_temperature == 0 (point-sampled coverage).
//
// Same results as k_render_forward (dm2_forward.hip; FORWARD::renderCUDA<3>, forward.cu:139-432): with temperature
// 0 the reference applies no bounding-box test (forward.cu:314), every face of a tile's list is intersected with
// all 256 pixel rays and blends where the clamped barycentrics say "inside".  This is that per-pixel walk, written
// so that all lanes of a wave stay in lock-step through the intersection test of a face (no per-lane `continue` /
// `break` in front of it), which makes one more product free: a ballot per (list entry, wave) of the pixels the
// entry blended into.  The backward (dm2_backward_point.hip) reads these 64-bit masks instead of repeating the
// dense intersection test, which is two thirds of its time.
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int FP_CHUNK = 128;

__global__ void __launch_bounds__(TILE_PIX)
k_render_forward_point(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, float* __restrict__ out_color, float* __restrict__ out_depth,
                       int32_t* __restrict__ out_tri_cnt, uint64_t* __restrict__ hit_masks, uint32_t* __restrict__ hit_valid) {
    __shared__ FaceRec recs[FP_CHUNK];

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t lx = tid & 15, ly = tid >> 4;
    const uint32_t px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) hit_valid[0] = 1u;   // the masks of this launch are current

    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) {
        pixel_ray(d, b, pix, px + (uint32_t)d.patch_min[2 * b], py + (uint32_t)d.patch_min[2 * b + 1], d.full_W, d.full_H, ro, rd);
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    bool done = !inside;
    float pT = 1.0f, T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;

    for (int base = 0; base < total; base += FP_CHUNK) {
        if (__syncthreads_count(done) == TILE_PIX) break;          // forward.cu:258-260 (also guards LDS reuse)
        const int n = min(FP_CHUNK, total - base);
        if (tid < n) stage_face(is.face_recs, (int64_t)b * d.F + face_list[range.x + base + tid], recs[tid]);
        __syncthreads();

        for (int j = 0; j < n; j++) {
            contributor++;
            const FaceRec& fc = recs[j];
            const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
            f3 tuv = {0, 0, 0};
            const bool ok = ray_tri_intersection(ro, rd, p0, p1, p2, tuv);
            // At temperature 0 the coverage is 1 where the clamped barycentrics say "inside" (clamp_bary_uv's first region,
            // auxiliary.h:294: code 0, u_c = u, v_c = v) and 0 everywhere else (forward.cu:375-381: mix_coverage(code, 0, 0)):
            // the other six regions and the double-precision mix need not be evaluated to know that.
            const float iuc = tuv.y, ivc = tuv.z;
            const float ratio = 1.0f;
            const bool hit = !done && ok && (iuc >= 0.0f) && (ivc >= 0.0f) && (iuc + ivc <= 1.0f);
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) hit_masks[((int64_t)range.x + base + j) * 4 + wid] = bal;
            if (hit) {
                const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                c0 = c0 * fc.intense; c1 = c1 * fc.intense; c2 = c2 * fc.intense;
                const float iD = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                const float alpha = fc.opacity * ratio;
                const float test_T = T * (1 - alpha);
                C0 += c0 * alpha * T; C1 += c1 * alpha * T; C2 += c2 * alpha * T;
                D += iD * alpha * T;
                pT = T; T = test_T;
                last_contributor = contributor;
                if (T < T_EPS) done = true;
            }
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
        if (out_tri_cnt) out_tri_cnt[pix] = 0;                     // no AA records at temperature 0 (K is forced to 0)
    }
}

void launch_render_forward_point(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, uint64_t* hit_masks,
                                 uint32_t* hit_valid, hipStream_t st) {
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    StageTimer tm(ST_FWD, st);
    hipLaunchKernelGGL(k_render_forward_point, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth,
                       out_tri_cnt, hit_masks, hit_valid);
}

}  // namespace dm2
