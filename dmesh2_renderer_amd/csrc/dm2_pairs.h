// dm2_pairs.h -- (pixel,face) pair enumeration for the dense composite kernels.
//
// Why: in the reference every pixel of a tile walks the tile's whole face list
// (forward.cu:307-416); with small triangles a wave spends most of its time with
// a handful of lanes inside the clipper.  Here the expensive per-(pixel,face)
// evaluation is decoupled from the per-pixel ordered blend:
//
//   A   per staged face: the EXACT set of tile pixels whose unit square passes the
//       clipper's bounding-box test (aa.h:96-101) is a rectangle; count it, scan the
//       counts -> every (pixel,face) pair of the chunk gets a dense index k.
//   B1  lane k classifies pair k (corner / half-plane tests only); the pairs that survive
//       are compacted in order into an LDS queue.
//   B2  one survivor per lane: clip, intersection, shading -> record in LDS.
//   C   each pixel blends ITS records in list order (cheap, sequential).
//   (dm2_forward_queue.hip, dm2_backward_queue.hip; the helpers below are the pair index arithmetic.)
//
// A pixel outside a face's rectangle contributes exactly nothing in the reference
// (bbox reject -> oarea 0 -> `continue`), so skipping it is not an approximation.
// With aa_temperature == 0 the reference performs no bbox test at all, so the
// rectangle is the whole tile.
#pragma once
#include "dm2_device_math.h"

namespace dm2 {

// ---- XCD-aware block -> tile order -----------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one), each XCD has its own L2, and
// neighbouring tiles share faces (1.6 tiles per face at the BASELINE workload) and pixels' rays.  A 1-D grid whose
// block L renders tile (L % 8) * ceil(Tn / 8) + L / 8 gives every XCD one contiguous band of tiles, so a face record
// is fetched into one L2 instead of up to eight.  (Only an ordering: correctness does not depend on where a block runs.)
__host__ __device__ __forceinline__ uint32_t tile_grid_blocks(uint32_t Tn) { return 8u * ((Tn + 7u) / 8u); }
// Inside its band an XCD takes the tiles in the order of `order` (k_tile_order, dm2_binning.hip: longest list first).
__device__ __forceinline__ bool tile_of_block(uint32_t Tn, const uint32_t* __restrict__ order, uint32_t& tile) {
    const uint32_t per = (Tn + 7u) / 8u;
    tile = order[(blockIdx.x & 7u) * per + (blockIdx.x >> 3)];
    return tile < Tn;
}

// rect packing: x0 | y0<<4 | (w-1)<<8 | (h-1)<<12 ; count==0 faces keep rect 0 and are never looked up
__device__ __forceinline__ uint32_t pack_rect(int x0, int y0, int w, int h) {
    return (uint32_t)x0 | ((uint32_t)y0 << 4) | ((uint32_t)(w - 1) << 8) | ((uint32_t)(h - 1) << 12);
}

// Exact tile-local pixel rectangle of a face (see header).  X0a/Y0a: absolute (full image)
// coordinate of the tile's first pixel; xlim/ylim: last valid local coordinate (image clip).
__device__ __forceinline__ int face_pixel_rect(const float bb[4], bool use_bbox, int X0a, int Y0a, int xlim, int ylim,
                                               uint32_t& rect) {
    int x0 = 0, x1 = xlim, y0 = 0, y1 = ylim;
    if (use_bbox) {
        // pixel a (integer, exactly representable) passes iff a + 1 >= txmin && a <= txmax
        //   <=> ceil(txmin) - 1 <= a <= floor(txmax)          (NaN bounds: the reference's half-plane
        // tests then reject every pixel, so an empty rectangle gives the same result)
        if (!(bb[0] == bb[0]) || !(bb[1] == bb[1]) || !(bb[2] == bb[2]) || !(bb[3] == bb[3])) { rect = 0; return 0; }
        const float lo_x = fminf(fmaxf(ceilf(bb[0]) - 1.0f - (float)X0a, -1.0f), 17.0f);
        const float hi_x = fminf(fmaxf(floorf(bb[1]) - (float)X0a, -1.0f), 17.0f);
        const float lo_y = fminf(fmaxf(ceilf(bb[2]) - 1.0f - (float)Y0a, -1.0f), 17.0f);
        const float hi_y = fminf(fmaxf(floorf(bb[3]) - (float)Y0a, -1.0f), 17.0f);
        x0 = max(x0, (int)lo_x); x1 = min(x1, (int)hi_x);
        y0 = max(y0, (int)lo_y); y1 = min(y1, (int)hi_y);
    }
    if (x1 < x0 || y1 < y0) { rect = 0; return 0; }
    rect = pack_rect(x0, y0, x1 - x0 + 1, y1 - y0 + 1);
    return (x1 - x0 + 1) * (y1 - y0 + 1);
}

// Inclusive scan of one int per lane inside a wave: DPP row shifts inside the 16-lane rows, then row_bcast:15 / row_bcast:31
// carry the row totals across (gfx9 DPP; ten VALU instructions, no LDS crossbar round trips as __shfl_up would cost).
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1 (lanes shifted in from outside the row read 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1 and 3 take lane 15 of the row before
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2 and 3 take lane 31
    return v;
}
// Inclusive running maximum of one unsigned per lane inside a wave (same network; 0 is the identity).
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t u) {
    int v = (int)u;                                                       // values stay below 2^31: signed max is the same
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return (uint32_t)v;
}

// Block-wide exclusive scan of one int per thread (256 threads = 4 waves).
// s_wave: 4 ints of LDS.  Returns the exclusive prefix; total in `total`.
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    const int w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    const int base = (wid > 0 ? w0 : 0) + (wid > 1 ? w1 : 0) + (wid > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    return base + inc - v;
}

// Face index of pair k: largest j with off[j] <= k (off is the exclusive scan, off[n] = total).
__device__ __forceinline__ int find_face_in(const int* off, int lo, int hi, int k) {
    while (hi - lo > 1) {        // invariant: off[lo] <= k < off[hi]
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int find_face(const int* off, int n, int k) { return find_face_in(off, 0, n, k); }

// local pair index -> (dx, dy) inside a rectangle of width w <= 16 (local < 256)
// inv_w[w] = ceil(65536 / w), w = 1..16 (LDS table filled by fill_inv_table)
__device__ __forceinline__ void fill_inv_table(int* inv_w) {
    if (threadIdx.x >= 1 && threadIdx.x <= 16) inv_w[threadIdx.x] = (65536 + (int)threadIdx.x - 1) / (int)threadIdx.x;
}
__device__ __forceinline__ void pair_xy(uint32_t rect, int local, const int* inv_w, int& lx, int& ly) {
    const int w = (int)((rect >> 8) & 15u) + 1;
    const int inv = inv_w[w];                      // (local * inv) >> 16 == floor(local / w) for local < 256, w <= 16
    const int dy = (local * inv) >> 16;
    lx = (int)(rect & 15u) + (local - dy * w);
    ly = (int)((rect >> 4) & 15u) + dy;
}

// pair index of pixel (lx,ly) in face rect, or -1
__device__ __forceinline__ int pixel_pair(uint32_t rect, int off, int lx, int ly) {
    const int dx = lx - (int)(rect & 15u), dy = ly - (int)((rect >> 4) & 15u);
    const int w = (int)((rect >> 8) & 15u) + 1, h = (int)((rect >> 12) & 15u) + 1;
    if ((unsigned)dx >= (unsigned)w || (unsigned)dy >= (unsigned)h) return -1;
    return off + dy * w + dx;
}

}  // namespace dm2
