// dm2_binning.hip -- face -> 16x16 tile binning for gfx950.
//
// What the reference builds (renderer.cu:165-219): per tile, the list of faces whose patch rectangle touches it,
// ordered by depth key, ties in emission order ((view,face) ascending) -- through one global stable radix sort of
// (tile | depth bits) keys.  Here the lists are bucketed first and sorted per tile:
//
//   plan  k_preprocess   forward.cu:16-108   one lane per (view,face): depth keys, cull, tile rect, packed record;
//                                            one atomic per touched tile -> entries per tile, and (faces of up to
//                                            four tiles) the entry's place in the tile's segment
//         k_tile_scan                        exclusive scan of the tile counts: list starts, num_rendered (= the
//                                            reference's scan of tiles_touched, renderer.cu:165-171), longest list
//   run   k_bin_scatter  renderer.cu:415-465 (depth bits | face id) keys into the tile's segment, any order (no
//                                            atomics but for faces of more than four tiles)
//         k_tile_sort                        one block per tile orders the segment: by counting (up to 512 entries), a
//                                            bitonic network in LDS (2048) or in global memory (32768); the key is unique,
//                                            and ascending (depth bits, face id) IS the stable radix order; writes face
//                                            ids + ranges
//
// 1M faces / 1.6M entries at 1080p: 0.06 ms against 0.24 ms for emit + six radix passes + range detection.
// Lists beyond TILE_SORT_MAX entries (a plan result) and DM2_FLAG_LEGACY_KERNELS take the reference's route:
//   rocPRIM scan   renderer.cu:165-171 inclusive sum of tiles_touched
//   k_emit_keys    renderer.cu:415-465 (tile | depth bits) keys + face ids, emission order
//   rocPRIM sort   renderer.cu:199-207 stable LSD radix sort on bits [0, 32+msb(B*tiles))
//   k_tile_ranges  renderer.cu:470-492 [start,end) of every tile in the sorted list
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>   // rocprim's texture_cache_iterator.hpp calls host memset without including it
#include <rocprim/rocprim.hpp>

#include "dm2_device_math.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

// Clears n 32-bit words.  A kernel of the library's own instead of hipMemsetAsync: the runtime's fill is a blit with a
// barrier packet on either side (kernel-trace gaps of 10 us behind a compute kernel, 4 us in front of the next one);
// a plain dispatch follows and is followed back to back.
__global__ void __launch_bounds__(256) k_zero_words(uint32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static void launch_zero_words(void* p, int64_t n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n);
}

// One count per active lane into cnt[STRIDE * t].  RANK: also returns a place per lane, as its own atomicAdd(.., 1) would.
// A global atomic costs the same ~60 ps per LANE whether or not lanes collide (1.6 M of them: 0.08 ms), so lanes of a wave
// that hit the same tile -- neighbouring faces of a mesh, a tet lattice -- are grouped first: the group's first lane adds the
// group's size, the others take the consecutive places.  Grouping costs a few scalar instructions per distinct tile; a
// probe skips it where hardly any lanes share (a triangle soup).  Call with the whole wave (act = lane has a tile).
// Two halves when the place is wanted: `issue` fires the atomic and returns at once (TilePlace holds the pending return
// value), `finish` reads it -- so that a caller's atomics are all in flight together and their latency hides behind
// whatever it does in between.
struct TilePlace { uint32_t base; int leader; uint32_t offset; };
// do the lanes of the wave share tiles? (the probe: how many have the first active lane's tile)
__device__ __forceinline__ bool wave_tiles_shared(uint32_t t, bool act) {
    const unsigned long long all = __ballot(act);
    if (all == 0) return false;
    const int l0 = __ffsll((long long)all) - 1;
    const unsigned long long m0 = __ballot(act && t == (uint32_t)__builtin_amdgcn_readlane((int)t, l0));
    return __popcll(m0) * 8 > __popcll(all);
}
// grouped: the (wave-uniform) answer of wave_tiles_shared for these lanes, or for related tiles of the same lanes
template <bool RANK, int STRIDE>
__device__ __forceinline__ TilePlace wave_count_tiles_issue(uint32_t* cnt, uint32_t t, bool act, bool grouped) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    TilePlace p; p.base = 0u; p.leader = lane; p.offset = 0u;
    const unsigned long long all = __ballot(act);
    if (all == 0) return p;
    if (!grouped) {                                                       // hardly any sharing: every lane for itself
        if (act) { if (RANK) p.base = atomicAdd(cnt + (int64_t)STRIDE * t, 1u); else atomicAdd(cnt + (int64_t)STRIDE * t, 1u); }
        return p;
    }
    unsigned long long todo = all, mine = 0;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)t, l);
        const unsigned long long m = __ballot(act && t == tl);
        if (act && t == tl) mine = m;
        todo &= ~m;
    }
    p.leader = act ? __ffsll((long long)mine) - 1 : lane;
    if (act && lane == p.leader) {
        if (RANK) p.base = atomicAdd(cnt + (int64_t)STRIDE * t, (uint32_t)__popcll(mine));
        else atomicAdd(cnt + (int64_t)STRIDE * t, (uint32_t)__popcll(mine));
    }
    p.offset = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    return p;
}
// the lane's place: the group's base (held by its first lane; the lane itself when it counted alone) + its rank in the group.
// Call with the whole wave.
__device__ __forceinline__ uint32_t wave_count_tiles_finish(const TilePlace& p) {
    return (uint32_t)__shfl((int)p.base, p.leader) + p.offset;
}
template <bool RANK, int STRIDE>
__device__ __forceinline__ uint32_t wave_count_tiles(uint32_t* cnt, uint32_t t, bool act) {
    const TilePlace p = wave_count_tiles_issue<RANK, STRIDE>(cnt, t, act, wave_tiles_shared(t, act));
    return RANK ? wave_count_tiles_finish(p) : 0u;
}

// The plan kernel's two forms
enum { PRE_LAYERS = 0,      // depth keys, cull, tile rectangles and counts (LayeredRenderer: no records, no pair bound)
       PRE_FUSED = 1 };     // + the pair bound + the packed face records
// (view, face) items per lane, 256 lanes apart, their operands fetched together.  One: with two the fused form needs 142 VGPRs
// (3 waves per SIMD), and culling two per lane at twice the occupancy with the records packed by a kernel of their own
// behind the scan is slower on every input tried (four 512 x 512 windows of 1080p cameras: plan 0.33 against 0.26 ms;
// a 1/8 band of a 1080p frame 0.117 against 0.093 ms; the whole frame 0.173 against 0.136 ms).
__host__ __device__ constexpr int pre_items(int) { return 1; }

// A wave's records leave through LDS, a 128-B half at a time, so that every store instruction writes whole 128-B lines (8
// lanes per line).  A lane storing its own record 16 B per instruction leaves 64 partially written lines per instruction
// to the L2, which evicts part of them before the rest of the line arrives: WRITE_SIZE 386 MB for 296 MB of stores at cfg 4,
// the plan 0.175 -> 0.133 ms without them.  Whole waves (every lane of the wave calls, `has`: the lane holds a record);
// idx: the lane's record index, consecutive over the wave; sw: 8 KB of LDS of the wave's own.
__device__ __forceinline__ void store_records(const FaceRec& r, bool has, unsigned long long tmask, int64_t idx, uint4* __restrict__ recs, uint4* sw) {
    const uint4* src = reinterpret_cast<const uint4*>(&r);
    const int lane = (int)(threadIdx.x & 63);
    uint4* dst = recs + (idx - lane) * FACE_REC_U4;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (has) {
#pragma unroll
            for (int k = 0; k < 8; k++) sw[lane * 8 + (k ^ (lane & 7))] = src[half * 8 + k];    // (swizzled: 8 lanes cover the 32 banks)
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int rr = j * 8 + (lane >> 3), c = lane & 7;
            if ((tmask >> rr) & 1ull) dst[(int64_t)rr * FACE_REC_U4 + half * 8 + c] = sw[rr * 8 + (c ^ (rr & 7))];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// PACK: also write the face's packed record (dm2_stage.h) for the composite kernels -- only for faces that reach a
// tile list.  `d` is read only then.  (A/B at cfg4: packing in a kernel of its own behind the plan's read-back, so that
// it runs while the host sizes and enqueues the run step, costs more than it hides -- the binning part alone is bound by
// its atomics, 0.08 ms, which here disappear behind the record traffic: 0.16 ms fused against 0.08 + 0.13 ms split.)
#ifndef DM2_PRE_WAVES
#define DM2_PRE_WAVES 1
#endif
template <int MODE>
__global__ void __launch_bounds__(256, DM2_PRE_WAVES)
k_preprocess(int B, int P, int F, uint32_t gx, uint32_t gy, uint32_t block_stride, const int32_t* __restrict__ patch_min,
             const int32_t* __restrict__ faces, const float* __restrict__ verts_ndc,
             const float* __restrict__ verts_image, FaceState fs, dm2_render_desc d) {
    // pre_items(MODE) (view, face) items per lane, 256 lanes apart: their vertex ids, then their image corners, fetched for
    // all items before the first is looked at.  No lane leaves early (the records are stored by whole waves): invalid items
    // (behind the last one) read the last item's operands and count as culled.
    constexpr bool PACK = MODE == PRE_FUSED;
    constexpr int DM2_PRE_ITEMS = pre_items(MODE);
    const int64_t BF = (int64_t)B * F;
    int64_t idx_[DM2_PRE_ITEMS]; int vid_[DM2_PRE_ITEMS][3]; float2 img_[DM2_PRE_ITEMS][3];
#pragma unroll
    for (int it = 0; it < DM2_PRE_ITEMS; it++) {
        // Workgroups that run side by side take chunks of items that lie far apart (block b: chunk b * stride mod #chunks,
        // the stride coprime to the count): where the face order follows space -- a mesh, a tet lattice -- neighbouring chunks
        // bin into the same few tiles, and their atomics on those tiles' counters queue up behind each other (they are
        // device-scope, carried out behind the per-XCD L2s: the cfg 4 triangles sorted by tile planned in 0.197 ms against
        // 0.135 ms in random order).  A soup is indifferent to the order of its chunks.
        const uint32_t chunk = (uint32_t)(((uint64_t)blockIdx.x * block_stride) % gridDim.x);
        idx_[it] = ((int64_t)chunk * DM2_PRE_ITEMS + it) * 256 + threadIdx.x;
        const int64_t ic = idx_[it] < BF ? idx_[it] : BF - 1;
        const int f = (int)(ic % F);
        vid_[it][0] = faces[3 * f]; vid_[it][1] = faces[3 * f + 1]; vid_[it][2] = faces[3 * f + 2];
    }
#pragma unroll
    for (int it = 0; it < DM2_PRE_ITEMS; it++) {
        const int64_t ic = idx_[it] < BF ? idx_[it] : BF - 1;
        const float* img = verts_image + (ic / F) * P * 2;
#pragma unroll
        for (int k = 0; k < 3; k++) img_[it][k] = *reinterpret_cast<const float2*>(img + 2 * vid_[it][k]);
    }
#pragma unroll
    for (int it = 0; it < DM2_PRE_ITEMS; it++) {
    const bool valid = idx_[it] < BF;
    if (!__ballot(valid)) break;                                              // (wave-uniform)
    const int64_t idx = valid ? idx_[it] : BF - 1;
    const int b = (int)(idx / F), f = (int)(idx % F);
    const uint32_t pmx = patch_min ? (uint32_t)patch_min[2 * b] : 0u, pmy = patch_min ? (uint32_t)patch_min[2 * b + 1] : 0u;
    const int v0 = vid_[it][0], v1 = vid_[it][1], v2 = vid_[it][2];
    const float* ndc = verts_ndc + (int64_t)b * P * 3;
    const float2 i0 = img_[it][0], i1 = img_[it][1], i2 = img_[it][2];

    // The tile rectangle first (forward.cu:77-88), the depth cull (forward.cu:71) only for faces that have one: a face has
    // to pass both, and most (view, face) items of a window or of a rank's band fail this one -- their NDC z is never fetched.
    uint32_t touched = 0, lo = 0, hi = 0;
    float d01 = 0.f, dmin = 0.f, dmax = 0.f;
    {
        uint32_t x0, y0, x1, y1;
        patch_rect_from_tri(pmx, pmy, i0.x, i0.y, i1.x, i1.y, i2.x, i2.y, gx, gy, x0, y0, x1, y1);
        touched = valid ? (y1 - y0) * (x1 - x0) : 0u;                      // forward.cu:88,93
        if (touched != 0) {
            const float z0 = ndc[3 * v0 + 2], z1 = ndc[3 * v1 + 2], z2 = ndc[3 * v2 + 2];
            const float max_z = fmaxf(fmaxf(z0, z1), z2);
            const float min_z = fminf(fminf(z0, z1), z2);
            float depth = ((0.0f + z0) + z1) + z2;
            depth = depth / 3.0f;
            if (max_z < -1.0f || min_z > 1.0f) touched = 0;               // forward.cu:71
            else {
                auto to01 = [](float z) { float d = (z + 1.0f) * 0.5f; if (d < 0.0f) d = 0.0f; if (d > 1.0f) d = 1.0f; return d; };
                d01 = to01(depth); dmin = to01(min_z); dmax = to01(max_z);
                lo = x0 | (y0 << 16); hi = x1 | (y1 << 16);
            }
        }
    }
    const bool small = touched != 0 && touched <= 4;
    const bool any_small = __ballot(small) != 0ull;
    TilePlace place[4];
    {
        // entries per tile (the lists' sizes).  A face with at most four tiles -- nearly all of a fine mesh -- takes its
        // place inside each tile's segment right here (the atomic's return value) and keeps it for k_bin_scatter, which
        // then needs no second round of atomics; larger faces are counted apart and placed behind them.
        // (wave-uniform control flow from here to the end of the block: wave_count_tiles is a wave-wide operation)
        const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, x1 = hi & 0xFFFFu;
        const uint32_t tb = (uint32_t)((int64_t)gx * gy * b);
        const bool big = touched > 4;
        if (any_small) {
            const uint32_t w = x1 - x0;                                       // entry k (rect order): tile (x0 + k % w, y0 + k / w)
            const bool grouped = wave_tiles_shared(tb + (y0 * gx + x0), small);   // one probe (the faces' first tiles) for all four
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {                                // four atomics in flight; their places are read at the very end
                const uint32_t ky = w == 1 ? k : (w == 2 ? k >> 1 : (w == 3 ? (k == 3 ? 1u : 0u) : 0u)), kx = k - ky * w;
                const bool act = small && k < touched;
                place[k] = wave_count_tiles_issue<true, 1>(fs.tile_cnt, act ? tb + ((y0 + ky) * gx + x0 + kx) : 0u, act, grouped);
            }
        }
        if (__ballot(big)) {
            uint32_t cx = x0, cy = y0;
            for (uint32_t k = 0; ; k++) {
                const bool act = big && k < touched;
                if (!__ballot(act)) break;
                wave_count_tiles<false, 1>(fs.tile_cnt_big, act ? tb + (cy * gx + cx) : 0u, act);
                if (++cx == x1) { cx = x0; cy++; }
            }
        }
    }
    if (MODE != PRE_LAYERS) {
        // Upper bound of the (pixel, face) pairs the forward composite will enumerate for this face: the patch pixels whose
        // unit square passes the clipper's bounding-box test (aa.h:96-101; dm2_pairs.h face_pixel_rect, tile by tile).  The
        // sum over the faces sizes the forward's pair pool (dm2_state.h), read back with num_rendered.
        unsigned long long cand = 0;
        if (touched != 0) {
            const float bx0 = fminf(fminf(i0.x, i1.x), i2.x), bx1 = fmaxf(fmaxf(i0.x, i1.x), i2.x);
            const float by0 = fminf(fminf(i0.y, i1.y), i2.y), by1 = fmaxf(fmaxf(i0.y, i1.y), i2.y);
            if (bx0 == bx0 && bx1 == bx1 && by0 == by0 && by1 == by1) {
                const float Wm = (float)(gx * TILE), Hm = (float)(gy * TILE);          // (>= the patch: an upper bound is all that is asked)
                const float lo_x = fminf(fmaxf(ceilf(bx0) - 1.0f - (float)pmx, 0.0f), Wm), hi_x = fminf(fmaxf(floorf(bx1) - (float)pmx, -1.0f), Wm - 1.0f);
                const float lo_y = fminf(fmaxf(ceilf(by0) - 1.0f - (float)pmy, 0.0f), Hm), hi_y = fminf(fmaxf(floorf(by1) - (float)pmy, -1.0f), Hm - 1.0f);
                const float w = hi_x - lo_x + 1.0f, h = hi_y - lo_y + 1.0f;
                if (w > 0.0f && h > 0.0f) cand = (unsigned long long)w * (unsigned long long)h;
            }
        }
        // one atomic per wave (the active lanes are a prefix of the wave; a rectangle has at most 2^40 pixels: two limbs
        // whose wave sums fit 32 bits, summed with the DPP scan)
        const int sa = wave_inclusive_scan((int)(cand & 0xFFFFFFull)), sb = wave_inclusive_scan((int)(cand >> 24));
        const int last = 63 - __clzll((long long)__ballot(true));
        if ((int)(threadIdx.x & 63) == last) {
            const unsigned long long tot = (unsigned long long)(uint32_t)sa + ((unsigned long long)(uint32_t)sb << 24);
            if (tot) atomicAdd(fs.pair_part + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (PAIR_PARTS - 1)), tot);
        }
    }
    if (valid) fs.tiles_touched[idx] = touched;
    if (touched != 0) {      // (as the reference, forward.cu:71-72,88-89: a culled face leaves these unwritten; their readers look at tiles_touched first)
        fs.depths[idx] = d01; fs.min_depths[idx] = dmin; fs.max_depths[idx] = dmax;
        fs.rect_lo[idx] = lo; fs.rect_hi[idx] = hi;
    }
    if (PACK) {
        __shared__ uint4 s_t[PACK ? 4 : 1][PACK ? 64 * 8 : 1];
        const unsigned long long tmask = __ballot(touched != 0);
        if (tmask != 0ull) {                                               // (wave-uniform)
            FaceRec r;
            if (touched != 0) { pack_face(d, b, f, i1, r); r.pad[0] = 0.f; }
            store_records(r, touched != 0, tmask, idx_[it], fs.recs, s_t[threadIdx.x >> 6]);   // (the unclamped index: every lane derives the wave's first record from its own)
        }
    }
    if (any_small) {                                                       // (wave-uniform: the shuffles need the whole wave)
        const uint32_t r0 = wave_count_tiles_finish(place[0]), r1 = wave_count_tiles_finish(place[1]);
        const uint32_t r2 = wave_count_tiles_finish(place[2]), r3 = wave_count_tiles_finish(place[3]);
        if (small) fs.tile_rank[idx] = make_uint4(r0, r1, r2, r3);
    }
    __builtin_amdgcn_sched_barrier(0);                                      // (one item after the other: interleaved they need 142 VGPRs, 3 waves per SIMD)
    }   // items
}

// Exclusive scan of the tile counts, their sum and maximum.  One block; 8192 tiles (a 1080p frame) per round: coalesced
// loads into LDS, thread i scans the run of 8 consecutive counts [8i, 8i + 8) there, coalesced stores.  (A thread that
// reads its run straight from global memory touches 32 cache lines per wave instruction -- 14 us on the one CU this
// kernel runs on, against 4.)
__global__ void __launch_bounds__(1024)
k_tile_scan(int64_t Tn, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cnt_big, uint32_t* __restrict__ start,
            const unsigned long long* __restrict__ pair_part, uint32_t* __restrict__ meta, uint32_t* host_meta, uint32_t host_seq,
            uint2* __restrict__ ranges_to_clear) {
    constexpr int RUN = 8, ROUND = 1024 * RUN;
    __shared__ uint32_t s_c[ROUND + ROUND / 32];          // (+1 dword per 32: the runs of the 64 lanes start in different banks)
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_max;
    __shared__ unsigned long long s_pairs, s_total64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { s_max = 0; s_pairs = 0ull; s_total64 = 0ull; }
    unsigned long long pair_sum = pair_part[tid];          // (requested first: it travels while the counts are scanned)
    uint32_t carry = 0, mx = 0;
    for (int64_t base = 0; base < Tn; base += ROUND) {
#pragma unroll
        for (int k = 0; k < RUN; k++) {
            const int i = k * 1024 + tid;
            const int64_t t = base + i;
            s_c[i + (i >> 5)] = t < Tn ? cnt[t] + cnt_big[t] : 0u;
        }
        __syncthreads();
        uint32_t c8[RUN], sum = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) { const int i = tid * RUN + k; c8[k] = s_c[i + (i >> 5)]; sum += c8[k]; mx = mx > c8[k] ? mx : c8[k]; }
        const uint32_t inc = (uint32_t)wave_inclusive_scan((int)sum);
        if (lane == 63) s_w[wid] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < 16; w++) { const uint32_t v = s_w[w]; if (w < wid) before += v; total += v; }
        uint32_t run = carry + before + inc - sum;
#pragma unroll
        for (int k = 0; k < RUN; k++) { const int i = tid * RUN + k; s_c[i + (i >> 5)] = run; run += c8[k]; }
        carry += total;
        // (the pair count in 64 bits, on its own path: a round's 32-bit total may itself have wrapped)
        unsigned long long s64 = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) s64 += c8[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s64 += __shfl_xor(s64, o);
        if (lane == 0 && s64) atomicAdd(&s_total64, s64);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RUN; k++) {
            const int i = k * 1024 + tid;
            const int64_t t = base + i;
            if (t < Tn) {
                start[t] = s_c[i + (i >> 5)];
                if (ranges_to_clear) ranges_to_clear[t] = make_uint2(0u, 0u);     // (the run step's fill cursors / renderer.cu:211)
            }
        }
        __syncthreads();                                   // s_c and s_w are rewritten by the next round
    }
    if (mx) atomicMax(&s_max, mx);
    {   // the pair bound: one partial sum per thread (PAIR_PARTS == blockDim.x), summed over the block
        static_assert(PAIR_PARTS == 1024, "one partial sum per thread of k_tile_scan");
        unsigned long long v = pair_sum;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && v) atomicAdd(&s_pairs, v);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t longest = s_total64 > 0x7FFFFFFFull ? 0xFFFFFFFFu : s_max;     // the list positions are 32-bit (as the reference's int num_rendered): notice a wrap
        const unsigned long long pairs = s_pairs;
        meta[0] = carry; meta[1] = longest; meta[2] = (uint32_t)pairs; meta[3] = (uint32_t)(pairs >> 32);
        if (host_meta) {
            // the numbers the host waits for, straight into its mapped memory, then the sequence word it polls: no copy
            // command, no event -- the host learns them microseconds after this store instead of ~30 us later
            host_meta[0] = carry; host_meta[1] = longest; host_meta[2] = (uint32_t)pairs; host_meta[3] = (uint32_t)(pairs >> 32);
            __hip_atomic_store(&host_meta[4], host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Block -> tile order of the composite kernels (dm2_pairs.h tile_of_block): XCD x (blocks x, x + 8, ...) keeps its contiguous
// band of ceil(Tn / 8) tiles and takes them by list length.  The hardware hands the next block to whichever slot frees up, so
// a kernel ends when its unluckiest slot does; in index order the last round starts tiles of every length (6 % above the mean
// load in a model of 1080p / 1 M faces: 8 rounds of 1024 resident blocks, lists of 195 +- 19 entries;
// tools/tile_order_sim.py), with the short lists at the end the slots run dry together.  Against that stands the locality of
// index order: neighbouring tiles share faces, whose records and gradient rows then meet in the XCD's L2.  Two modes, chosen by
// the number of rounds (A/B of the three orders on one MI355X, step ms, profiles/r03_ab_tile_order.txt):
//   TILE_ORDER_ALL   the whole band longest list first             cfg 4: 1.968 (index order) -> 1.934, point-sampled 1.532 ->
//                    (up to 16 rounds: the tail is what counts)     1.502, B = 4: 1.349 -> 1.330, cfg 2: 0.310 -> 0.305;
//                                                                   cfg 5 (32 rounds): 6.29 -> 6.33, its backward + 3 %
//   TILE_ORDER_TAIL  index order, but the quarter of the band with  cfg 5: 6.29 -> 6.25 (cfg 4: 1.946)
//                    the shortest lists last, longest first
// A counting sort over 256 length classes; inside a class the order is whatever the atomics give (it is only an order of
// execution).  One block per XCD band; cnt == nullptr: index order.  Runs in the plan-to-run gap when the image scratch is at
// hand.
enum { TILE_ORDER_INDEX = 0, TILE_ORDER_TAIL = 1, TILE_ORDER_ALL = 2 };
#ifndef DM2_TILE_ORDER
#define DM2_TILE_ORDER -1       // (A/B builds: force one of the modes)
#endif
static int tile_order_mode(int64_t Tn) {
    if (DM2_TILE_ORDER >= 0) return DM2_TILE_ORDER;
    const char* e = getenv("DM2_TILE_ORDER_MODE");                   // (tests: every mode on one small frame)
    if (e && *e >= '0' && *e <= '2') return *e - '0';
    return Tn <= 16 * 1024 ? TILE_ORDER_ALL : TILE_ORDER_TAIL;      // (1024 resident workgroups at 4 per CU)
}
__global__ void __launch_bounds__(1024)
k_tile_order(int64_t Tn, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cnt_big, const uint32_t* __restrict__ meta,
             uint32_t* __restrict__ order, int mode) {
    __shared__ uint32_t s_h[257];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_k;
    const int64_t per = (Tn + 7) / 8, t0 = (int64_t)blockIdx.x * per;
    const int64_t n = t0 >= Tn ? 0 : (Tn - t0 < per ? Tn - t0 : per);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int64_t i = n + tid; i < per; i += 1024) order[t0 + i] = 0xFFFFFFFFu;        // (blocks behind the band's last tile leave at once)
    if (!cnt || mode == TILE_ORDER_INDEX) { for (int64_t i = tid; i < n; i += 1024) order[t0 + i] = (uint32_t)(t0 + i); return; }
    const uint32_t mx = meta[1] ? meta[1] : 1u;                                        // the longest list (k_tile_scan)
    auto cls = [&](int64_t t) { const uint64_t c = (uint64_t)cnt[t] + cnt_big[t]; const uint64_t q = c * 255u / mx; return 255u - (uint32_t)(q > 255u ? 255u : q); };
    if (tid < 257) s_h[tid] = 0;
    if (tid == 0) s_k = 255u;
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) atomicAdd(&s_h[cls(t0 + i)], 1u);
    __syncthreads();
    {   // exclusive scan of the 256 class sizes (class 0 = the longest lists): s_h[k] = tiles in front of class k, longest first
        const uint32_t v = tid < 256 ? s_h[tid] : 0u;
        const uint32_t inc = (uint32_t)wave_inclusive_scan((int)v);
        if (tid < 256 && lane == 63) s_w[wid] = inc;
        __syncthreads();
        if (tid < 256) {
            uint32_t before = 0;
            for (int w = 0; w < wid; w++) before += s_w[w];
            s_h[tid] = before + inc - v;
            // the head: classes 0..K, K the first class with which three quarters of the band are in
            if (mode == TILE_ORDER_TAIL && (uint64_t)(before + inc) * 4u >= (uint64_t)n * 3u) atomicMin(&s_k, (uint32_t)tid);
        }
    }
    __syncthreads();
    const uint32_t K = s_k;                                                            // (TILE_ORDER_ALL: no head, everything by class)
    __syncthreads();                                                                   // (s_w is reused below)
    uint32_t carry = 0;                                                                // head tiles placed so far
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const uint32_t c = i < n ? cls(t0 + i) : 0xFFFFFFFFu;
        const bool head = i < n && mode == TILE_ORDER_TAIL && c <= K;
        const unsigned long long m = __ballot(head);
        if (lane == 0) s_w[wid] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < 16; w++) { const uint32_t v = s_w[w]; if (w < wid) before += v; total += v; }
        if (head) order[t0 + carry + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(t0 + i);
        else if (i < n) order[t0 + atomicAdd(&s_h[c], 1u)] = (uint32_t)(t0 + i);       // (the tail's classes start behind the head)
        carry += total;
        __syncthreads();
    }
}

// (depth bits | face id) of every list entry into its tile's segment [tile_start, tile_start + count): entries of faces
// with at most four tiles at the place the plan gave them, entries of larger faces behind those, wherever the cursor
// (ranges[t].y, zeroed) puts them -- k_tile_sort orders the segment.
__global__ void __launch_bounds__(256)
k_bin_scatter(int B, int F, uint32_t gx, uint32_t gy, uint32_t block_stride, const float* __restrict__ key_depth, FaceState fs,
              uint2* __restrict__ ranges, uint64_t* __restrict__ keys) {
    // (the plan's chunk order: the places it handed out inside a tile's segment rise with the order in which it took the chunks,
    // and taking them in the same order here keeps the 8-byte stores into a segment next to each other in time: 0.028 against 0.043 ms)
    const int64_t idx0 = (int64_t)(((uint64_t)blockIdx.x * block_stride) % gridDim.x) * blockDim.x + threadIdx.x;
    const bool valid = idx0 < (int64_t)B * F;
    const int64_t idx = valid ? idx0 : 0;
    const uint32_t touched = valid ? fs.tiles_touched[idx] : 0u;
    if (!__ballot(touched != 0)) return;
    const int b = (int)(idx / F), f = (int)(idx % F);
    const uint32_t lo = fs.rect_lo[idx], hi = fs.rect_hi[idx];
    const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, x1 = hi & 0xFFFFu;
    const uint64_t key = ((uint64_t)__float_as_uint(key_depth[idx]) << 32) | (uint32_t)f;
    const int64_t tile_base = (int64_t)gx * gy * b;
    if (touched != 0 && touched <= 4) {
        const uint4 r4 = fs.tile_rank[idx];
        const uint32_t rk[4] = {r4.x, r4.y, r4.z, r4.w};
        const uint32_t w = x1 - x0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t ky = w == 1 ? k : (w == 2 ? k >> 1 : (w == 3 ? (k == 3 ? 1u : 0u) : 0u)), kx = k - ky * w;
            if (k < touched) keys[fs.tile_start[tile_base + ((y0 + ky) * gx + x0 + kx)] + rk[k]] = key;
        }
    }
    const bool big = touched > 4;                 // (0 for the lanes past the end: they stay in the wave for the ballots)
    if (__ballot(big)) {
        uint32_t cx = x0, cy = y0;
        for (uint32_t k = 0; ; k++) {
            const bool act = big && k < touched;
            if (!__ballot(act)) break;
            const uint32_t t = act ? (uint32_t)tile_base + (cy * gx + cx) : 0u;
            const uint32_t place = wave_count_tiles<true, 2>(&ranges[0].y, t, act);
            if (act) keys[fs.tile_start[t] + fs.tile_cnt[t] + place] = key;
            if (++cx == x1) { cx = x0; cy++; }
        }
    }
}

// One block per tile: sort the tile's segment of (depth bits | face id) keys ascending, leave the face ids in face_list and
// [start, end) in ranges (0,0 for an empty tile, as the reference's zero-filled ranges, renderer.cu:211).
// The network is the bitonic sorter in its all-ascending form (a merge of size k: mirror exchange i <-> i ^ (k - 1), then
// half-cleaners i <-> i ^ j, j = k/4 .. 1; the smaller key always goes to the lower index), so a list that is not a power
// of two long is padded with +inf -- real padding in LDS, virtual (skip exchanges whose upper index is past the end) in
// global memory.
constexpr int TILE_SORT_LDS = 2048;
constexpr int TILE_SORT_RANK = 512;     // lists up to here are ordered by counting, longer ones by the network
// pair p of a step: mirror step of a merge of size 2^lk (lk > 0), or half-cleaner at distance 2^lj (lk == 0)
__device__ __forceinline__ void sort_pair_indices(int p, int lk, int lj, int& i, int& j) {
    if (lk) { const int blk = p >> (lk - 1), off = p & ((1 << (lk - 1)) - 1); i = (blk << lk) + off; j = (blk << lk) + ((1 << lk) - 1 - off); }
    else { const int blk = p >> lj, off = p & ((1 << lj) - 1); i = (blk << (lj + 1)) + off; j = i + (1 << lj); }
}
// The steps of the network in order: for lk = 1 .. LN: mirror(lk), then half-cleaners lj = lk - 2 .. 0.
// Pair p is always handled by thread p & 255, so a step whose pairs stay inside 128 consecutive keys (mirror with lk <= 7,
// half-cleaner with lj <= 5) only touches keys of the thread's own wave: such steps need no workgroup barrier between
// them, the wave's LDS operations execute in order (a fence keeps the compiler from reordering them).
__device__ __forceinline__ void sort_step_sync(bool wave_local_next, bool wave_local_this) {
    if (wave_local_next && wave_local_this) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    else __syncthreads();
}
__global__ void __launch_bounds__(256)
k_tile_sort(int64_t Tn, uint32_t R, const uint32_t* __restrict__ tile_start, uint64_t* __restrict__ keys,
            uint32_t* __restrict__ face_list, uint2* __restrict__ ranges, uint32_t* __restrict__ hit_valid) {
    __shared__ __attribute__((aligned(16))) uint64_t s_key[TILE_SORT_LDS];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s0 = tile_start[t], e0 = (t + 1 < Tn) ? tile_start[t + 1] : R;
    const int n = (int)(e0 - s0);
    if (t == 0 && tid < 4) hit_valid[tid] = 0u;    // new lists: the blend masks of an earlier forward are stale; pool and tie queue empty
    if (tid == 0) ranges[t] = n ? make_uint2(s0, e0) : make_uint2(0u, 0u);
    if (n == 0) return;
    uint64_t* const seg = keys + s0;
    int LN = 0;
    while ((1 << LN) < n) LN++;
    const int N = 1 << LN, half = N >> 1;
    if (n <= TILE_SORT_RANK) {
        // short list (the usual case): every key counts the keys below it -- they are all different, so that is its place.
        // n broadcast reads (two keys each) and compares per thread, nothing waits for anything: cheaper than the network's
        // ~LN^2 / 2 dependent LDS round trips up to a few hundred entries.
        const int n2 = (n + 1) & ~1;
        for (int i = tid; i < n2; i += 256) s_key[i] = i < n ? seg[i] : ~0ull;
        __syncthreads();
        const uint64_t k0 = tid < n ? s_key[tid] : 0ull, k1 = tid + 256 < n ? s_key[tid + 256] : 0ull;
        int r0 = 0, r1 = 0;
        const ulonglong2* const kk2 = reinterpret_cast<const ulonglong2*>(s_key);
        if (n <= 256) {
            for (int i = 0; i < (n2 >> 1); i++) { const ulonglong2 kk = kk2[i]; r0 += (int)(kk.x < k0) + (int)(kk.y < k0); }
        } else {
            for (int i = 0; i < (n2 >> 1); i++) {
                const ulonglong2 kk = kk2[i];
                r0 += (int)(kk.x < k0) + (int)(kk.y < k0); r1 += (int)(kk.x < k1) + (int)(kk.y < k1);
            }
        }
        if (tid < n) face_list[s0 + r0] = (uint32_t)k0;
        if (tid + 256 < n) face_list[s0 + r1] = (uint32_t)k1;
    } else if (n <= TILE_SORT_LDS) {
        for (int i = tid; i < N; i += 256) s_key[i] = i < n ? seg[i] : ~0ull;
        __syncthreads();
        for (int lk = 1; lk <= LN; lk++) {
            for (int lj = lk - 1; lj >= 0; lj--) {                           // lj == lk - 1: the mirror step of this merge
                const bool mirror = lj == lk - 1;
                for (int p = tid; p < half; p += 256) {
                    int i, j;
                    sort_pair_indices(p, mirror ? lk : 0, lj, i, j);
                    const uint64_t a = s_key[i], c = s_key[j];
                    if (c < a) { s_key[i] = c; s_key[j] = a; }
                }
                // span of this step's pairs: 2^(lj+1) keys (mirror: 2^lk = the same); the next step's is at most that,
                // except for the mirror of the next merge (2^(lk+1))
                const bool this_local = lj + 1 <= 7;
                const bool next_local = lj > 0 ? true : (lk + 1 <= 7);
                sort_step_sync(this_local && next_local, true);
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) face_list[s0 + i] = (uint32_t)s_key[i];
    } else {
        for (int lk = 1; lk <= LN; lk++) {
            for (int lj = lk - 1; lj >= 0; lj--) {
                const bool mirror = lj == lk - 1;
                for (int p = tid; p < half; p += 256) {
                    int i, j;
                    sort_pair_indices(p, mirror ? lk : 0, lj, i, j);
                    if (j < n) {
                        const uint64_t a = seg[i], c = seg[j];
                        if (c < a) { seg[i] = c; seg[j] = a; }
                    }
                }
                __syncthreads();                                              // (block scope: the block's own global writes are visible)
            }
        }
        for (int i = tid; i < n; i += 256) face_list[s0 + i] = (uint32_t)seg[i];
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
    if (idx == 0) { hit_valid[0] = 0u; hit_valid[1] = 0u; hit_valid[2] = 0u; hit_valid[3] = 0u; }   // new lists: earlier masks are stale; pool and tie queue empty
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

// (rocPRIM's size queries resolve a device configuration every time they are asked: a few microseconds each, and a forward
// asks half a dozen times between the plan's read-back and the run step's first launch -- the last answers are kept)
size_t scan_temp_bytes(int64_t BF) {
    thread_local int64_t last_bf = -1; thread_local size_t last_bytes = 0;
    if (BF == last_bf) return last_bytes;
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, p, p, (size_t)(BF > 0 ? BF : 1), rocprim::plus<uint32_t>());
    last_bf = BF; last_bytes = bytes;
    return bytes;
}

unsigned sort_end_bit(int64_t Tn) {
    const unsigned e = 32u + higher_msb((uint32_t)Tn);
    return e > 64u ? 64u : e;
}

size_t sort_temp_bytes(int64_t R, int64_t Tn) {
    thread_local int64_t last_r = -1, last_tn = -1; thread_local size_t last_bytes = 0;
    if (R == last_r && Tn == last_tn) return last_bytes;
    size_t bytes = 0;
    uint64_t* k = nullptr; uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(R > 0 ? R : 1), 0u, sort_end_bit(Tn));
    last_r = R; last_tn = Tn; last_bytes = bytes;
    return bytes;
}

void launch_tile_order_identity(int64_t Tn, uint32_t* tile_order, hipStream_t st) {
    if (Tn > 0) hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, st, Tn, nullptr, nullptr, nullptr, tile_order, (int)TILE_ORDER_INDEX);
}

// Order in which the workgroups of k_preprocess and k_bin_scatter take the chunks of 256 (view, face) items: block b takes
// chunk b * stride mod #chunks, a stride of ~1/13 of the chunks, coprime to their number (see k_preprocess).
static uint32_t chunk_stride(uint32_t chunks) {
    if (chunks <= 64u) return 1u;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
    uint32_t stride = chunks / 13u + 1u;
    while (gcd(stride, chunks) != 1u) stride++;
    return stride;
}

hipError_t launch_preprocess_scan(int B, int P, int F, int W, int H, const int32_t* patch_min, const int32_t* faces,
                                  const float* verts_ndc, const float* verts_image, FaceState fs, const dm2_render_desc* pack,
                                  uint32_t* host_meta, uint32_t host_seq, uint2* ranges_to_clear, uint32_t* tile_order, hipStream_t st) {
    const int64_t BF = (int64_t)B * F;
    if (BF == 0) return hipSuccess;
    const uint32_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int64_t Tn = (int64_t)B * gx * gy;
    StageTimer tm(ST_PREP, st);
    if (Tn == 0) return hipMemsetAsync(fs.plan_meta, 0, 4 * sizeof(uint32_t), st);
    launch_zero_words(fs.tile_cnt, 2 * Tn + 2 * PAIR_PARTS, st);           // (the tile counts and the pair-bound partial sums; k_tile_scan writes plan_meta)
    const int mode = pack && fs.recs ? PRE_FUSED : PRE_LAYERS;
    const int blocks = (int)((BF + 256 * pre_items(mode) - 1) / (256 * pre_items(mode)));
    const uint32_t stride = chunk_stride((uint32_t)blocks);
    if (mode == PRE_FUSED)
        hipLaunchKernelGGL(k_preprocess<PRE_FUSED>, dim3(blocks), dim3(256), 0, st, B, P, F, gx, gy, stride, patch_min, faces, verts_ndc, verts_image, fs, *pack);
    else
        hipLaunchKernelGGL(k_preprocess<PRE_LAYERS>, dim3(blocks), dim3(256), 0, st, B, P, F, gx, gy, stride, patch_min, faces, verts_ndc, verts_image, fs,
                           dm2_render_desc{});
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, Tn, fs.tile_cnt, fs.tile_cnt_big, fs.tile_start, fs.pair_part, fs.plan_meta,
                       host_meta, host_seq, ranges_to_clear);
    // (behind the kernel whose last store the host is polling for: it runs while the host sizes and enqueues the run step)
    if (tile_order) hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, st, Tn, fs.tile_cnt, fs.tile_cnt_big, fs.plan_meta, tile_order, tile_order_mode(Tn));
    return hipSuccess;
}

hipError_t launch_bin_sort(int B, int F, int W, int H, int64_t R, int64_t max_tile_entries, bool legacy, const float* key_depth,
                           FaceState fs, BinningState bs, uint2* ranges, bool ranges_cleared, uint32_t* tile_order, hipStream_t st) {
    const uint32_t gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int64_t Tn = (int64_t)B * gx * gy;
    if (!ranges_cleared) launch_zero_words(ranges, 2 * Tn, st);            // renderer.cu:211 (or done by the plan's last kernel)
    // tile_order: the composite kernels' block -> tile table, wanted and not yet written by the plan (which did both or neither)
    if (tile_order && !ranges_cleared)
        hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, st, Tn, R > 0 ? fs.tile_cnt : nullptr, fs.tile_cnt_big, fs.plan_meta, tile_order, tile_order_mode(Tn));
    hipError_t e = hipSuccess;
    if (R <= 0) return e;
    const int64_t BF = (int64_t)B * F;
    if (!legacy && max_tile_entries <= TILE_SORT_MAX) {
        {
            StageTimer tm(ST_EMIT, st);
            hipLaunchKernelGGL(k_bin_scatter, dim3((int)((BF + 255) / 256)), dim3(256), 0, st, B, F, gx, gy, chunk_stride((uint32_t)((BF + 255) / 256)), key_depth, fs, ranges, bs.keys);
        }
        StageTimer tm(ST_SORT, st);
        hipLaunchKernelGGL(k_tile_sort, dim3((unsigned)Tn), dim3(256), 0, st, Tn, (uint32_t)R, fs.tile_start, bs.keys, bs.face_list,
                           ranges, bs.hit_valid);
        return hipSuccess;
    }
    {
        StageTimer tm(ST_EMIT, st);
        size_t bytes = fs.scan_temp_bytes;
        e = rocprim::inclusive_scan(fs.scan_temp, bytes, fs.tiles_touched, fs.face_offsets, (size_t)BF, rocprim::plus<uint32_t>(), st);
        if (e != hipSuccess) return e;
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
