// dm2_backward.hip -- visibility / colour / depth gradients of the composite
// (BACKWARD::renderCUDA<3>, backward.cu:17-532) for gfx950.
//
// Same tiling as the forward: one 256-thread workgroup per 16x16 tile, the
// tile's face list walked back to front in chunks staged in LDS.  Two things
// differ from the reference by design:
//   * AA records are not stored by the forward; the overlap area and its
//     Jacobian are recomputed here (the reference's own K=0 path,
//     backward.cu:264-272).  Recomputed values are bit-identical to recorded
//     ones, so results only differ in the reference's record-stack desync
//     corner case.
//   * The reference issues 29 global fp32 atomics per (pixel,face).  All 256
//     pixels of a tile walk the same entry in lock-step, so the 29 partials are
//     first summed per (tile,entry) in LDS (ds_add_f32) and flushed once per
//     chunk with (entry,component)-shaped global atomics.
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int BWD_CHUNK = 128;
constexpr int ACC_STRIDE = 32;     // floats per entry accumulator
// accumulator slots
constexpr int A_DV = 0;            // 9: dL/dverts of the 3 corners
constexpr int A_DC = 9;            // 9: dL/dverts_color
constexpr int A_DZ = 18;           // 3: dL/dverts_ndc[...,2]
constexpr int A_OP = 21;           // dL/dfaces_opacity
constexpr int A_IN = 22;           // dL/dfaces_intense
constexpr int A_AA = 23;           // 6: dL/daa_face_verts
constexpr int A_N = 29;
constexpr int A_FLAG = 31;

__device__ __forceinline__ void lds_add(float* p, float v) { atomicAdd(p, v); }   // ds_add_f32, no return

__global__ void __launch_bounds__(TILE_PIX)
k_render_backward(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                  ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                  float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                  float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                  float* __restrict__ dL_dfaces_intense, float* __restrict__ dL_daa_face_verts,
                  const uint32_t* __restrict__ skip_if_masks) {
    // launched behind the mask-driven kernels when the caller did not know what the forward left (DM2_FWD_UNKNOWN): one of
    // them did the work when the forward left masks
    if (skip_if_masks && skip_if_masks[0] >= 2u) return;
    __shared__ FaceRec recs[BWD_CHUNK];
    __shared__ float acc[BWD_CHUNK * ACC_STRIDE];
    __shared__ uint32_t s_max_lc;

    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x;
    const uint32_t lx = tid & 15, ly = tid >> 4;
    const uint32_t px = blockIdx.x * TILE + lx, py = blockIdx.y * TILE + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const bool corrected = (d.flags & DM2_FLAG_CORRECTED_DV) != 0;

    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    float T_final = 0.f, prev_T_final = 0.f;
    uint32_t last_contributor = 0;
    float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f;
    if (inside) {
        pixel_ray(d, b, pix, px + pmx, py + pmy, d.full_W, d.full_H, ro, rd);
        T_final = is.final_T[pix]; prev_T_final = is.final_prev_T[pix];
        last_contributor = is.n_contrib[pix];
        dLc0 = dL_dcolor[3 * pix]; dLc1 = dL_dcolor[3 * pix + 1]; dLc2 = dL_dcolor[3 * pix + 2];
        dLd = dL_ddepth[pix];
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];

    // Entries behind every pixel's last contributor cannot contribute
    // (backward.cu:219-221): start the walk at the tile's deepest contributor.
    if (tid == 0) s_max_lc = 0;
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_lc, last_contributor);
    __syncthreads();
    const int total = (int)min(s_max_lc, range.y - range.x);

    const float temp = d.aa_temperature;
    const float pxmin = (float)(px + pmx), pxmax = pxmin + 1;
    const float pymin = (float)(py + pmy), pymax = pymin + 1;
    const float pix_area = 1.0f;
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];

    float T = prev_T_final;
    bool T_first_pass = true;
    uint32_t contributor = (uint32_t)total;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    for (int base = 0; base < total; base += BWD_CHUNK) {
        __syncthreads();                                            // previous chunk flushed
        const int n = min(BWD_CHUNK, total - base);
        // recs[j] = entry (total-1) - (base+j): back to front (backward.cu:171)
        if (tid < n) stage_face(is.face_recs, (int64_t)b * d.F + face_list[range.x + (uint32_t)(total - 1 - base - tid)], recs[tid]);
        for (int k = tid; k < n * ACC_STRIDE; k += TILE_PIX) acc[k] = 0.f;
        __syncthreads();

        if (inside) {
            for (int j = 0; j < n; j++) {
                contributor--;
                if (contributor >= last_contributor) continue;
                const FaceRec& fc = recs[j];
                float oarea = 0.f;
                float dg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (temp > 0.0f) {
                    const int err = tri_pix_overlap_area<true>(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea, dg);
                    if ((err != 0) || (oarea == 0.0f)) continue;
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
                const float intense = fc.intense, opacity = fc.opacity;
                float iC0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                float iC1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                float iC2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                iC0 = iC0 * intense; iC1 = iC1 * intense; iC2 = iC2 * intense;
                const float iD = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                const float alpha = opacity * ratio;

                if (!T_first_pass) T = T / (1.f - alpha);           // backward.cu:340-348
                T_first_pass = false;

                float dL_dalpha = 0.0f;
                accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = iC0;
                const float dic0 = dLc0 * alpha * T; dL_dalpha += (iC0 - accum_rec0) * dLc0;
                accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = iC1;
                const float dic1 = dLc1 * alpha * T; dL_dalpha += (iC1 - accum_rec1) * dLc1;
                accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = iC2;
                const float dic2 = dLc2 * alpha * T; dL_dalpha += (iC2 - accum_rec2) * dLc2;
                accum_recd = last_alpha * last_depth + (1.f - last_alpha) * accum_recd; last_depth = iD;
                const float did = dLd * alpha * T; dL_dalpha += (iD - accum_recd) * dLd;
                dL_dalpha *= T;
                last_alpha = alpha;

                float bg_dot = 0.f;
                bg_dot += bg0 * dLc0; bg_dot += bg1 * dLc1; bg_dot += bg2 * dLc2;
                const float bd_dot = (float)(0.0 + 1.0 * (double)dLd);          // backward.cu:394
                if (alpha == 1.0f) {
                    dL_dalpha += (-prev_T_final) * bg_dot;
                    dL_dalpha += (-prev_T_final) * bd_dot;
                } else {
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bd_dot;
                }
                const float dL_dfop = dL_dalpha * ratio;
                const float dL_dratio = (dL_dalpha * opacity) * temp;
                const float dL_doarea = dL_dratio / pix_area;

                float dL_di0 = 0.f, dL_di1 = 0.f, dL_di2 = 0.f, dL_dfint = 0.f;
                const float dics[3] = {dic0, dic1, dic2};
                float dvc[9];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    dL_di0 += fc.col[ch] * dics[ch] * intense;
                    dL_di1 += fc.col[3 + ch] * dics[ch] * intense;
                    dL_di2 += fc.col[6 + ch] * dics[ch] * intense;
                    dvc[ch] = 0.f + i0 * dics[ch] * intense;
                    dvc[3 + ch] = 0.f + i1 * dics[ch] * intense;
                    dvc[6 + ch] = 0.f + i2 * dics[ch] * intense;
                    dL_dfint += (i0 * fc.col[ch] + i1 * fc.col[3 + ch] + i2 * fc.col[6 + ch]) * dics[ch];
                }
                dL_di0 += fc.dep[0] * did; dL_di1 += fc.dep[1] * did; dL_di2 += fc.dep[2] * did;
                const float dvd0 = 0.f + i0 * did, dvd1 = 0.f + i1 * did, dvd2 = 0.f + i2 * did;

                float diuc_diu, diuc_div, divc_diu, divc_div;
                clamp_bary_uv_grad(code, diuc_diu, diuc_div, divc_diu, divc_div);
                const float di0_diu = -1.f * diuc_diu + -1.f * divc_diu, di0_div = -1.f * diuc_div + -1.f * divc_div;
                const float di1_diu = 1.f * diuc_diu + 0.f * divc_diu, di1_div = 1.f * diuc_div + 0.f * divc_div;
                const float di2_diu = 0.f * diuc_diu + 1.f * divc_diu, di2_div = 0.f * diuc_div + 1.f * divc_div;
                const float dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
                const float dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;
                f3 du0, du1, du2, dv0, dv1, dv2;
                ray_tri_intersection_grad(ro, rd, p0, p1, p2, corrected, du0, du1, du2, dv0, dv1, dv2);
                const f3 dp0 = dL_diu * du0 + dL_div * dv0;
                const f3 dp1 = dL_diu * du1 + dL_div * dv1;
                const f3 dp2 = dL_diu * du2 + dL_div * dv2;

                float* a = acc + j * ACC_STRIDE;
                lds_add(a + A_DV + 0, dp0.x); lds_add(a + A_DV + 1, dp0.y); lds_add(a + A_DV + 2, dp0.z);
                lds_add(a + A_DV + 3, dp1.x); lds_add(a + A_DV + 4, dp1.y); lds_add(a + A_DV + 5, dp1.z);
                lds_add(a + A_DV + 6, dp2.x); lds_add(a + A_DV + 7, dp2.y); lds_add(a + A_DV + 8, dp2.z);
#pragma unroll
                for (int k = 0; k < 9; k++) lds_add(a + A_DC + k, dvc[k]);
                lds_add(a + A_DZ + 0, dvd0); lds_add(a + A_DZ + 1, dvd1); lds_add(a + A_DZ + 2, dvd2);
                lds_add(a + A_OP, dL_dfop);
                lds_add(a + A_IN, dL_dfint);
#pragma unroll
                for (int k = 0; k < 6; k++) lds_add(a + A_AA + k, dL_doarea * dg[k]);
                a[A_FLAG] = 1.0f;
            }
        }
        __syncthreads();

        // flush: lane = (entry, component); 8 entries per pass
        const int comp = tid & 31;
        if (comp < A_N) {
            for (int e = tid >> 5; e < n; e += TILE_PIX / 32) {
                const float* a = acc + e * ACC_STRIDE;
                if (a[A_FLAG] == 0.f) continue;
                const FaceRec& fc = recs[e];
                const float val = a[comp];
                float* dst;
                if (comp < A_DC) dst = dL_dverts + 3 * (int64_t)fc.vid[comp / 3] + (comp % 3);
                else if (comp < A_DZ) dst = dL_dverts_color + 3 * (int64_t)fc.vid[(comp - A_DC) / 3] + ((comp - A_DC) % 3);
                else if (comp < A_OP) dst = dL_dverts_ndc + ((int64_t)b * d.P + fc.vid[comp - A_DZ]) * 3 + 2;
                else if (comp == A_OP) dst = dL_dfaces_opacity + fc.face_id;
                else if (comp == A_IN) dst = dL_dfaces_intense + (int64_t)b * d.F + fc.face_id;
                else if (d.flags & DM2_FLAG_AA_GRAD_TO_VERTS) {                  // corner -> the vertex the CCW reorder took it from
                    const int c = (comp - A_AA) >> 1;
                    const bool flip = (fc.aa.zmask >> 8) & 1u;
                    dst = dL_daa_face_verts + ((int64_t)b * d.P + fc.vid[c == 0 ? 0 : (flip ? 3 - c : c)]) * 2 + ((comp - A_AA) & 1);
                }
                else dst = dL_daa_face_verts + ((int64_t)b * d.F + fc.face_id) * 6 + (comp - A_AA);
                atomicAdd(dst, val);
            }
        }
    }
}

void launch_render_backward(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                            const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                            float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                            float* dL_daa_face_verts, const BinningState& bs, int fwd_mode, TieEntry* tie_queue, int64_t tie_cap,
                            hipStream_t st) {
    const uint64_t* const hit_masks = bs.hit_masks; const uint32_t* const hit_valid = bs.hit_valid;
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    const bool fast_only = !(d.flags & DM2_FLAG_LEGACY_KERNELS) && d.aa_temperature > 0.0f && hit_masks && hit_valid && fwd_mode == DM2_FWD_POOL &&
                           bs.pool && bs.pool_cap > 0 && tie_queue && tie_cap > 0;
    if (fast_only) {      // (times its two kernels as two stages)
        launch_render_backward_fast(d, ranges, face_list, is, dL_dcolor, dL_ddepth, dL_dverts, dL_dverts_color, dL_dfaces_opacity,
                                    dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, bs, tie_queue, tie_cap, false, st);
        return;
    }
    StageTimer tm(ST_BWD, st);
    if (!(d.flags & DM2_FLAG_LEGACY_KERNELS)) {
        // aa_temperature == 0: no bbox test in the reference (backward.cu:241-244), every face of a tile's list meets
        // all 256 pixels; the pair enumeration has nothing to prune there -> dm2_backward_point.hip (dense
        // intersection test per wave, compacted hits for the gradient chain).  -DDM2_POINT_PER_PIXEL: the
        // reference-shaped per-pixel walk below instead (A/B).
#ifndef DM2_POINT_PER_PIXEL
        if (!(d.aa_temperature > 0.0f)) {
            if (fwd_mode == DM2_FWD_POINT && hit_masks && hit_valid) {
                // the masks of this frame's dm2_forward_point.hip drive the same kernel as at temperature > 0 (coverage 1, no AA terms)
                launch_render_backward_fast(d, ranges, face_list, is, dL_dcolor, dL_ddepth, dL_dverts, dL_dverts_color, dL_dfaces_opacity,
                                            dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, bs, nullptr, 0, false, st);
                return;
            }
            // (what the forward left is not known, or it left nothing: dm2_backward_point.hip looks at hit_valid itself and
            // repeats the dense intersection test when the masks are not this frame's)
            launch_render_backward_point(d, ranges, face_list, is, dL_dcolor, dL_ddepth, dL_dverts, dL_dverts_color,
                                         dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, hit_masks, hit_valid, st);
            return;
        }
#endif
        if (d.aa_temperature > 0.0f && hit_masks && hit_valid && fwd_mode != DM2_FWD_NONE) {
            // What the forward left decides the kernel: masks + pool -> dm2_backward_fast.hip, masks -> dm2_backward_mask.hip,
            // nothing -> the per-pixel walk.  A caller that says DM2_FWD_UNKNOWN gets all of them, each looking at hit_valid
            // on the device (no host read-back) and all but one returning at once.
            const bool unknown = fwd_mode == DM2_FWD_UNKNOWN;
            const bool pool_ok = bs.pool && bs.pool_cap > 0 && tie_queue && tie_cap > 0;
            if ((fwd_mode == DM2_FWD_POOL || unknown) && pool_ok)
                launch_render_backward_fast(d, ranges, face_list, is, dL_dcolor, dL_ddepth, dL_dverts, dL_dverts_color, dL_dfaces_opacity,
                                            dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, bs, tie_queue, tie_cap, unknown, st);
            if (fwd_mode == DM2_FWD_MASKS || unknown)
                launch_render_backward_mask(d, ranges, face_list, is, dL_dcolor, dL_ddepth, dL_dverts, dL_dverts_color,
                                            dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, hit_masks, hit_valid, st);
            if (unknown)
                hipLaunchKernelGGL(k_render_backward, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                                   dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, hit_valid);
            return;
        }
    }
    hipLaunchKernelGGL(k_render_backward, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                       dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                       (const uint32_t*)nullptr);
}

}  // namespace dm2
