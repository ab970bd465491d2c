// dm2_binning.hip -- face -> 16x16 tile binning for gfx950.
//
//   k_preprocess   forward.cu:16-108   one lane per (view,face): depth keys, cull, tile rect
//   rocPRIM scan   renderer.cu:165-171 inclusive sum of tiles_touched
//   k_emit_keys    renderer.cu:415-465 (tile | depth bits) keys + face ids, emission order
//                                      (view,face) ascending, then y, then x
//   rocPRIM sort   renderer.cu:199-207 stable LSD radix sort on bits [0, 32+msb(B*tiles))
//   k_tile_ranges  renderer.cu:470-492 [start,end) of every tile in the sorted list
#include <hip/hip_runtime.h>

#include <cstring>   // rocprim's texture_cache_iterator.hpp calls host memset without including it
#include <rocprim/rocprim.hpp>

#include "dm2_device_math.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

// PACK: also write the face's packed record (dm2_stage.h) for the composite kernels -- only for faces that reach a
// tile list.  `d` is read only then.
template <bool PACK>
__global__ void __launch_bounds__(256)
k_preprocess(int B, int P, int F, uint32_t gx, uint32_t gy, const int32_t* __restrict__ patch_min,
             const int32_t* __restrict__ faces, const float* __restrict__ verts_ndc,
             const float* __restrict__ verts_image, FaceState fs, dm2_render_desc d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * F) return;
    const int b = (int)(idx / F), f = (int)(idx % F);
    const uint32_t pmx = patch_min ? (uint32_t)patch_min[2 * b] : 0u, pmy = patch_min ? (uint32_t)patch_min[2 * b + 1] : 0u;
    const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
    const float* ndc = verts_ndc + (int64_t)b * P * 3;
    const float* img = verts_image + (int64_t)b * P * 2;
    const float z0 = ndc[3 * v0 + 2], z1 = ndc[3 * v1 + 2], z2 = ndc[3 * v2 + 2];
    const float2 i0 = *reinterpret_cast<const float2*>(img + 2 * v0);
    const float2 i1 = *reinterpret_cast<const float2*>(img + 2 * v1);
    const float2 i2 = *reinterpret_cast<const float2*>(img + 2 * v2);
    float max_z = fmaxf(fmaxf(z0, z1), z2);
    float min_z = fminf(fminf(z0, z1), z2);
    float depth = ((0.0f + z0) + z1) + z2;
    depth = depth / 3.0f;

    uint32_t touched = 0, lo = 0, hi = 0;
    float d01 = 0.f, dmin = 0.f, dmax = 0.f;
    if (!(max_z < -1.0f || min_z > 1.0f)) {                               // forward.cu:71
        uint32_t x0, y0, x1, y1;
        patch_rect_from_tri(pmx, pmy, i0.x, i0.y, i1.x, i1.y, i2.x, i2.y, gx, gy, x0, y0, x1, y1);
        touched = (y1 - y0) * (x1 - x0);                                   // forward.cu:88,93
        if (touched != 0) {
            auto to01 = [](float z) { float d = (z + 1.0f) * 0.5f; if (d < 0.0f) d = 0.0f; if (d > 1.0f) d = 1.0f; return d; };
            d01 = to01(depth); dmin = to01(min_z); dmax = to01(max_z);
            lo = x0 | (y0 << 16); hi = x1 | (y1 << 16);
        }
    }
    fs.tiles_touched[idx] = touched;
    fs.depths[idx] = d01; fs.min_depths[idx] = dmin; fs.max_depths[idx] = dmax;
    fs.rect_lo[idx] = lo; fs.rect_hi[idx] = hi;
    if (PACK && touched != 0) {
        FaceRec r;
        pack_face(d, b, f, r);
        r.pad[0] = 0.f;
        const uint4* src = reinterpret_cast<const uint4*>(&r);
        uint4* dst = fs.recs + idx * FACE_REC_U4;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(FaceRec) / 16); k++) dst[k] = src[k];
    }
}

__global__ void __launch_bounds__(256)
k_emit_keys(int B, int F, uint32_t gx, uint32_t gy, const float* __restrict__ key_depth, FaceState fs,
            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * F) return;
    if (fs.tiles_touched[idx] == 0) return;
    const int b = (int)(idx / F), f = (int)(idx % F);
    uint32_t off = (idx == 0) ? 0u : fs.face_offsets[idx - 1];
    const uint32_t lo = fs.rect_lo[idx], hi = fs.rect_hi[idx];
    const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, x1 = hi & 0xFFFFu, y1 = hi >> 16;
    const uint32_t dbits = __float_as_uint(key_depth[idx]);
    const uint64_t tile_base = (uint64_t)gx * gy * (uint64_t)b;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) {
            const uint64_t key = ((tile_base + (uint64_t)(y * gx + x)) << 32) | dbits;
            keys[off] = key; vals[off] = (uint32_t)f; off++;
        }
}

__global__ void __launch_bounds__(256)
k_tile_ranges(int64_t L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges, uint32_t* __restrict__ hit_valid) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    if (idx == 0) *hit_valid = 0u;          // new lists: the hit masks of an earlier point-sampled forward are stale
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) { ranges[prev].y = (uint32_t)idx; ranges[cur].x = (uint32_t)idx; }
    }
    if (idx == L - 1) ranges[cur].y = (uint32_t)L;
}

// renderer.cu:396-411 (bit count so that tile ids < 2^bit)
static uint32_t higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

size_t scan_temp_bytes(int64_t BF) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, p, p, (size_t)(BF > 0 ? BF : 1), rocprim::plus<uint32_t>());
    return bytes;
}

unsigned sort_end_bit(int64_t Tn) {
    const unsigned e = 32u + higher_msb((uint32_t)Tn);
    return e > 64u ? 64u : e;
}

size_t sort_temp_bytes(int64_t R, int64_t Tn) {
    size_t bytes = 0;
    uint64_t* k = nullptr; uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(R > 0 ? R : 1), 0u, sort_end_bit(Tn));
    return bytes;
}

hipError_t launch_preprocess_scan(int B, int P, int F, int W, int H, const int32_t* patch_min, const int32_t* faces,
                                  const float* verts_ndc, const float* verts_image, FaceState fs, const dm2_render_desc* pack,
                                  hipStream_t st) {
    const int64_t BF = (int64_t)B * F;
    if (BF == 0) return hipSuccess;
    const uint32_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int blocks = (int)((BF + 255) / 256);
    StageTimer tm(ST_PREP, st);
    if (pack && fs.recs)
        hipLaunchKernelGGL(k_preprocess<true>, dim3(blocks), dim3(256), 0, st, B, P, F, gx, gy, patch_min, faces, verts_ndc, verts_image, fs, *pack);
    else
        hipLaunchKernelGGL(k_preprocess<false>, dim3(blocks), dim3(256), 0, st, B, P, F, gx, gy, patch_min, faces, verts_ndc, verts_image, fs,
                           dm2_render_desc{});
    size_t bytes = fs.scan_temp_bytes;
    return rocprim::inclusive_scan(fs.scan_temp, bytes, fs.tiles_touched, fs.face_offsets, (size_t)BF, rocprim::plus<uint32_t>(), st);
}

hipError_t launch_bin_sort(int B, int F, int W, int H, int64_t R, const float* key_depth, FaceState fs, BinningState bs,
                           uint2* ranges, hipStream_t st) {
    const uint32_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int64_t Tn = (int64_t)B * gx * gy;
    hipError_t e = hipMemsetAsync(ranges, 0, (size_t)Tn * sizeof(uint2), st);           // renderer.cu:211
    if (e != hipSuccess || R <= 0) return e;
    const int64_t BF = (int64_t)B * F;
    {
        StageTimer tm(ST_EMIT, st);
        hipLaunchKernelGGL(k_emit_keys, dim3((int)((BF + 255) / 256)), dim3(256), 0, st, B, F, gx, gy, key_depth, fs,
                           bs.keys_unsorted, bs.face_list_unsorted);
    }
    {
        StageTimer tm(ST_SORT, st);
        size_t bytes = bs.sort_temp_bytes;
        e = rocprim::radix_sort_pairs(bs.sort_temp, bytes, bs.keys_unsorted, bs.keys, bs.face_list_unsorted, bs.face_list,
                                      (size_t)R, 0u, sort_end_bit(Tn), st);
        if (e != hipSuccess) return e;
    }
    StageTimer tm(ST_RANGES, st);
    hipLaunchKernelGGL(k_tile_ranges, dim3((int)((R + 255) / 256)), dim3(256), 0, st, R, bs.keys, ranges, bs.hit_valid);
    return hipSuccess;
}

}  // namespace dm2
