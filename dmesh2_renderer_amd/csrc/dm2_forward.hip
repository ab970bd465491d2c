// dm2_forward.hip -- per-pixel front-to-back composite with analytic AA coverage
// (FORWARD::renderCUDA<3>, forward.cu:139-432) for gfx950.
//
// One 256-thread workgroup per 16x16 tile; each of the 4 waves owns a 16x4 pixel
// strip.  The tile's sorted face list is consumed in chunks of CHUNK entries;
// every entry is gathered once into a 240-B LDS record (dm2_stage.h) and read
// back by the pixel loop with wave-uniform ds_reads.  The forward needs the
// overlap AREA only (gradients are recomputed by the backward), so the clipper
// is instantiated without its Jacobian bookkeeping.
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int FWD_CHUNK = 128;

__global__ void __launch_bounds__(TILE_PIX)
k_render_forward(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                 ImageState is, float* __restrict__ out_color, float* __restrict__ out_depth,
                 int32_t* __restrict__ out_tri_cnt) {
    __shared__ FaceRec recs[FWD_CHUNK];

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x;
    // lane -> pixel: wave w covers rows 4w..4w+3 of the tile, 16 pixels per row
    const uint32_t lx = tid & 15, ly = tid >> 4;
    const uint32_t px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];

    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) {
        pixel_ray(d, b, pix, px + pmx, py + pmy, d.full_W, d.full_H, ro, rd);
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    const float temp = d.aa_temperature;
    const float pxmin = (float)(px + pmx), pxmax = pxmin + 1;
    const float pymin = (float)(py + pmy), pymax = pymin + 1;
    const float pix_area = 1.0f;

    bool done = !inside;
    float pT = 1.0f, T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    int rec_cnt = 0;

    for (int base = 0; base < total; base += FWD_CHUNK) {
        if (__syncthreads_count(done) == TILE_PIX) break;          // forward.cu:258-260 (also guards LDS reuse)
        const int n = min(FWD_CHUNK, total - base);
        if (tid < n) stage_face(is.face_recs, (int64_t)b * d.F + face_list[range.x + base + tid], recs[tid]);
        __syncthreads();

        for (int j = 0; !done && j < n; j++) {
            contributor++;
            const FaceRec& fc = recs[j];
            float oarea = 0.f;
            if (temp > 0.0f) {
                const int err = tri_pix_overlap_area<false>(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea, nullptr);
                if ((err != 0) || (oarea == 0.0f)) continue;
                if (rec_cnt < d.K) rec_cnt++;                       // forward.cu:344-352: a record is taken here
            }
            float ratio = oarea / pix_area;
            const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
            f3 tuv = {0, 0, 0};
            if (!ray_tri_intersection(ro, rd, p0, p1, p2, tuv)) continue;
            float iuc, ivc; int code;
            clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
            const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
            ratio = mix_coverage(code, ratio, temp);
            if (ratio == 0.0f) continue;
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
            if (T < T_EPS) { done = true; break; }
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
}

// Candidate pairs per list entry from which phase B2 of the dense forward goes class by class (dm2_forward_queue.hip): 17 at
// 1080p / 1 M faces (slower with classes), 49 at 256 x 256 / 2 k and at depth complexity 60 (14-17 % faster).
constexpr float FQ_CLASSES_FROM = 32.0f;

int launch_render_forward(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                          float* out_color, float* out_depth, int32_t* out_tri_cnt, const BinningState& bs, bool use_pool,
                          float pairs_per_entry, hipStream_t st) {
    uint64_t* const hit_masks = bs.hit_masks; uint32_t* const hit_valid = bs.hit_valid;
    if (!(d.flags & DM2_FLAG_LEGACY_KERNELS)) {
        if (d.aa_temperature > 0.0f) {
            // no backward will follow: no blend masks (hit_valid stays 0 from the binning, so a backward that comes
            // anyway takes the mask-free walk)
            const bool masks = !(d.flags & DM2_FLAG_NO_BACKWARD) && hit_masks && hit_valid;
            const bool pool = masks && use_pool && bs.pool && bs.pool_cap > 0;
            launch_render_forward_queue(d, ranges, face_list, is, out_color, out_depth, out_tri_cnt, masks ? hit_masks : nullptr,
                                        masks ? hit_valid : nullptr, pool ? bs.pool : nullptr, pool ? bs.pool_cap : 0, bs.hit_base,
                                        pairs_per_entry >= FQ_CLASSES_FROM, st);
            return pool ? DM2_FWD_POOL : (masks ? DM2_FWD_MASKS : DM2_FWD_NONE);
        }
        // aa_temperature == 0: the reference applies no bbox test (forward.cu:314), every face of a tile's list meets
        // all 256 pixels: the per-pixel walk is the dense formulation there.  dm2_forward_point.hip is that walk with
        // one extra product: the per-(entry, wave) hit masks the backward would otherwise have to recompute.
        // -DDM2_POINT_PER_PIXEL: the plain walk below instead (A/B).
#ifndef DM2_POINT_PER_PIXEL
        if (hit_masks && hit_valid) {
            launch_render_forward_point(d, ranges, face_list, is, out_color, out_depth, out_tri_cnt, hit_masks, hit_valid, st);
            return DM2_FWD_POINT;
        }
#endif
    }
    // (hit_valid was reset by the binning of this forward, dm2_binning.hip: no masks from this path)
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    StageTimer tm(ST_FWD, st);
    hipLaunchKernelGGL(k_render_forward, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth, out_tri_cnt);
    return DM2_FWD_NONE;
}

}  // namespace dm2
