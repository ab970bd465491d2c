// dm2_state.h -- scratch layouts shared by the host API and the kernels.
//
// Same role as the reference's FaceState / ImageState / BinningState /
// ImageRenderLayerState bump allocators (cuda_impl/state.h:10-69,
// renderer.cu:40-76,499-507): every buffer is carved out of one caller-owned
// byte array at 256-byte alignment, and backward re-derives the same pointers
// from the saved arrays.  The layouts themselves are this library's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dm2_hip.h"

namespace dm2 {

constexpr size_t ALIGN = 256;

struct Carver {
    uintptr_t p;
    explicit Carver(const void* base) : p(reinterpret_cast<uintptr_t>(base)) {}
    template <class T> T* take(size_t count) {
        p = (p + ALIGN - 1) & ~(uintptr_t)(ALIGN - 1);
        T* r = reinterpret_cast<T*>(p);
        p += count * sizeof(T);
        return r;
    }
    size_t used(const void* base) const { return (size_t)(p - reinterpret_cast<uintptr_t>(base)); }
};

// One packed record per (view, face): everything a (pixel,face) evaluation reads (the LDS FaceRec of dm2_stage.h,
// 240 bytes, padded to 256).  Written once per forward by the preprocess kernel, read by every list entry of the
// composite kernels as two full 128-byte lines (instead of ~16 separate 64-byte lines gathered through three levels
// of indices), and again by the backward: the gradients belong to the inputs the forward saw.
constexpr int FACE_REC_U4 = 16;            // uint4 per record (256 B)

constexpr int PAIR_PARTS = 1024;           // the plan's pair-bound sum is kept in this many partial sums (one atomic per wave, spread: atomics on one
                                            // address are serialised by the L2; 16 k of them on 64 addresses cost the plan 0.02 ms)

// per (batch,face): produced by the preprocess kernel
struct FaceState {
    float* depths;            // mean NDC z mapped to [0,1]  (sort key of Renderer)
    float* min_depths;        // min   "                     (sort key of LayeredRenderer)
    float* max_depths;
    uint32_t* tiles_touched;
    uint32_t* face_offsets;   // inclusive scan of tiles_touched
    uint32_t* rect_lo;        // x0 | y0 << 16   (half-open tile rect, reused by the key emitter)
    uint32_t* rect_hi;        // x1 | y1 << 16
    void* scan_temp; size_t scan_temp_bytes;
    uint4* recs;              // (BF, FACE_REC_U4) packed face records; nullptr when the caller needs none (layers)
    // tile-bucketed binning (dm2_binning.hip): filled by the plan step, consumed by the run step
    uint32_t* tile_cnt;       // (Tn)  list entries per tile from faces that touch at most four tiles (they keep their place, below)
    uint32_t* tile_cnt_big;   // (Tn)  list entries per tile from faces that touch more
    uint32_t* tile_start;     // (Tn)  exclusive scan of tile_cnt + tile_cnt_big
    uint32_t* plan_meta;      // [0] num_rendered  [1] entries of the longest tile list  [2,3] pair bound (lo, hi)
    unsigned long long* pair_part;   // (PAIR_PARTS) partial sums of the faces' pixel rectangles (their sum bounds the forward's (pixel,face) pairs)
    uint4* tile_rank;         // (BF)  faces with 1..4 tiles: the entry's place among the tile's small-face entries, rect order
    static FaceState carve(void* base, int64_t BF, int64_t Tn, size_t scan_temp_bytes, bool with_recs, size_t* total = nullptr) {
        Carver c(base); FaceState s;
        s.depths = c.take<float>(BF); s.min_depths = c.take<float>(BF); s.max_depths = c.take<float>(BF);
        s.tiles_touched = c.take<uint32_t>(BF); s.face_offsets = c.take<uint32_t>(BF);
        s.rect_lo = c.take<uint32_t>(BF); s.rect_hi = c.take<uint32_t>(BF);
        s.scan_temp = c.take<char>(scan_temp_bytes); s.scan_temp_bytes = scan_temp_bytes;
        s.recs = with_recs ? c.take<uint4>(BF * FACE_REC_U4) : nullptr;
        s.tile_cnt = c.take<uint32_t>(2 * Tn + 2 * PAIR_PARTS); s.tile_cnt_big = s.tile_cnt + Tn;      // (one fill clears all three)
        s.pair_part = reinterpret_cast<unsigned long long*>(s.tile_cnt + 2 * Tn);    // (8-byte aligned: the carve is 256-B aligned, 2 Tn words in front)
        s.tile_start = c.take<uint32_t>(Tn); s.plan_meta = c.take<uint32_t>(4);
        s.tile_rank = c.take<uint4>(Tn > 0 ? BF : 0);
        if (total) *total = c.used(base) + ALIGN;
        return s;
    }
};

// per pixel + per tile: what backward needs from forward
struct ImageState {
    float* final_T;           // (N)
    float* final_prev_T;      // (N)
    uint32_t* n_contrib;      // (N)
    uint2* ranges;            // (Tn) [start,end) into face_list
    uint32_t* tile_max_lc;    // (Tn) the largest n_contrib of the tile's pixels (written by the dense and the point-sampled forward)
    uint32_t* tile_order;     // (8 ceil(Tn / 8)) block -> tile of the composite kernels (k_tile_order, dm2_pairs.h tile_of_block)
    const uint4* face_recs;   // not part of the image scratch: FaceState::recs of the same forward, set by the host API
    static ImageState carve(void* base, int64_t N, int64_t Tn, size_t* total = nullptr) {
        Carver c(base); ImageState s; s.face_recs = nullptr;
        s.final_T = c.take<float>(N); s.final_prev_T = c.take<float>(N); s.n_contrib = c.take<uint32_t>(N);
        s.ranges = c.take<uint2>(Tn); s.tile_max_lc = c.take<uint32_t>(Tn);
        s.tile_order = c.take<uint32_t>(8 * ((Tn + 7) / 8));
        if (total) *total = c.used(base) + ALIGN;
        return s;
    }
};

struct LayerImageState {
    uint2* ranges;            // (Tn)
    int32_t* first_face;      // (N)
    int32_t* first_tet;       // (N)
    static LayerImageState carve(void* base, int64_t N, int64_t Tn, size_t* total = nullptr) {
        Carver c(base); LayerImageState s;
        s.ranges = c.take<uint2>(Tn); s.first_face = c.take<int32_t>(N); s.first_tet = c.take<int32_t>(N);
        if (total) *total = c.used(base) + ALIGN;
        return s;
    }
};

struct BinningState {
    uint32_t* face_list;          // sorted values (kept for backward)
    uint64_t* keys;               // sorted keys
    uint64_t* keys_unsorted;
    uint32_t* face_list_unsorted;
    void* sort_temp; size_t sort_temp_bytes;
    // per list entry and wave of the tile's block, the 64 pixels the entry blended into in the forward (dm2_forward_point.hip /
    // dm2_forward_queue.hip write, the mask-driven backward kernels read).  hit_valid[0]: what the forward left --
    // 0 nothing (stale), 1 point-sampled masks, 2 AA blend masks, 3 AA blend masks + the pair pool
    uint64_t* hit_masks;          // (4 R)
    uint32_t* hit_base;           // (R)  pair pool: slot of the entry's first blended pair (its pairs follow in (wave, pixel) order)
    uint32_t* hit_valid;          // (4)  [0] mode  [1] pool slots handed out  [2] entries in the backward's tie queue
    // pair pool (the tail of the buffer, whatever the caller appended to the fixed part): the coverage ratio
    // (forward.cu:375-378) of every blended (pixel, face) pair, so that the backward neither clips for an area nor depends
    // on reproducing it (dm2_backward_fast.hip)
    float* pool; int64_t pool_cap;
    static BinningState carve(void* base, int64_t R, size_t sort_temp_bytes, size_t* total = nullptr, size_t buffer_bytes = 0) {
        Carver c(base); BinningState s;
        s.face_list = c.take<uint32_t>(R); s.keys = c.take<uint64_t>(R);
        s.keys_unsorted = c.take<uint64_t>(R); s.face_list_unsorted = c.take<uint32_t>(R);
        s.sort_temp = c.take<char>(sort_temp_bytes); s.sort_temp_bytes = sort_temp_bytes;
        s.hit_masks = c.take<uint64_t>(4 * R); s.hit_base = c.take<uint32_t>(R); s.hit_valid = c.take<uint32_t>(4);
        const size_t fixed = c.used(base) + ALIGN;
        if (total) *total = fixed;
        s.pool = c.take<float>(0);
        const size_t off = c.used(base);
        s.pool_cap = buffer_bytes > off ? (int64_t)((buffer_bytes - off) / sizeof(float)) : 0;
        if (s.pool_cap > 0xFFFFFFF0ll) s.pool_cap = 0xFFFFFFF0ll;      // slots are 32-bit
        return s;
    }
    static size_t pool_bytes(int64_t pairs) { return pairs > 0 ? (((size_t)pairs * sizeof(float) + ALIGN - 1) & ~(ALIGN - 1)) + ALIGN : 0; }
};

// one entry of the backward's tie queue: a blended pair whose AA Jacobian the exact clipper has to supply
// (patch pixel coordinates have up to 20 bits: the low 16 of each in `pxy`, the rest beside the view)
struct __attribute__((aligned(16))) TieEntry { uint32_t face, view_hi, pxy; float dL_doarea; };
__host__ __device__ inline TieEntry tie_pack(uint32_t face, uint32_t view, uint32_t x, uint32_t y, float g) {
    TieEntry e; e.face = face; e.view_hi = view | ((x >> 16) << 16) | ((y >> 16) << 24); e.pxy = (x & 0xFFFFu) | (y << 16); e.dL_doarea = g;
    return e;
}
__host__ __device__ inline void tie_unpack(const TieEntry& e, uint32_t& view, uint32_t& x, uint32_t& y) {
    view = e.view_hi & 0xFFFFu; x = (e.pxy & 0xFFFFu) | (((e.view_hi >> 16) & 0xFFu) << 16); y = (e.pxy >> 16) | ((e.view_hi >> 24) << 16);
}

// ---- launchers implemented in the .hip files ------------------------------------
size_t scan_temp_bytes(int64_t BF);
size_t sort_temp_bytes(int64_t R, int64_t Tn);
unsigned sort_end_bit(int64_t Tn);

// plan: preprocess (forward.cu:16-108) + entries per tile + their scan -> fs.plan_meta = (num_rendered, longest list)
// pack != nullptr: also write the packed face records fs.recs from the op's inputs
// host_meta != nullptr: DEVICE pointer to three words of mapped host memory: (num_rendered, longest list, host_seq) are
// stored there by the last kernel, the sequence word last (system-scope release)
// ranges_to_clear != nullptr: the last kernel also zeroes these Tn tile ranges (what the run step would do first)
hipError_t launch_preprocess_scan(int B, int P, int F, int W, int H, const int32_t* patch_min, const int32_t* faces,
                                  const float* verts_ndc, const float* verts_image, FaceState fs, const dm2_render_desc* pack,
                                  uint32_t* host_meta, uint32_t host_seq, uint2* ranges_to_clear, uint32_t* tile_order, hipStream_t st);
// run: the sorted per-tile lists (renderer.cu:185-219): face_list ordered by (tile, depth key, emission order) + ranges.
// key depth = depths or min_depths.  max_tile_entries (from the plan) picks the method: per-tile sorts in LDS, or -- lists
// beyond TILE_SORT_MAX entries, or legacy = true -- the reference's way, one global stable radix sort.
constexpr int64_t TILE_SORT_MAX = 32768;
// ranges_cleared: the plan's last kernel has already zeroed `ranges` (launch_preprocess_scan's ranges_to_clear)
hipError_t launch_bin_sort(int B, int F, int W, int H, int64_t R, int64_t max_tile_entries, bool legacy, const float* key_depth,
                           FaceState fs, BinningState bs, uint2* ranges, bool ranges_cleared, uint32_t* tile_order, hipStream_t st);

// the composite kernels' block -> tile table in index order (a frame without faces: no plan ran)
void launch_tile_order_identity(int64_t Tn, uint32_t* tile_order, hipStream_t st);

void launch_render_forward_point(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, uint64_t* hit_masks,
                                 uint32_t* hit_valid, hipStream_t st);
// use_pool: also fill the pair pool of `bs` (the caller has checked its capacity against the plan's pair bound).
// Returns what it left for the backward (DM2_FWD_*).
// pairs_per_entry: the plan's pair bound / num_rendered (picks the forward's work distribution)
int launch_render_forward(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                          float* out_color, float* out_depth, int32_t* out_tri_cnt, const BinningState& bs, bool use_pool,
                          float pairs_per_entry, hipStream_t st);
void launch_prepare_faces(const dm2_prep_desc& d, hipStream_t st);
void launch_prepare_faces_backward(const dm2_prep_desc& d, const float* g_ndc, const float* g_image, const float* g_aa,
                                   float* image_grad_scratch, float* g_verts, hipStream_t st);
void launch_render_forward_queue(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, uint64_t* hit_masks,
                                 uint32_t* hit_valid, float* pool, int64_t pool_cap, uint32_t* hit_base, bool classes, hipStream_t st);
void launch_render_backward_point(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                  const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                  float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                  const uint64_t* hit_masks, const uint32_t* hit_valid, hipStream_t st);
void launch_render_backward_mask(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                 float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                 float* dL_daa_face_verts, const uint64_t* hit_masks, const uint32_t* hit_valid, hipStream_t st);
void launch_render_backward_fast(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                 float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                 float* dL_daa_face_verts, const BinningState& bs, TieEntry* tie_queue, int64_t tie_cap,
                                 bool check_mode, hipStream_t st);
// fwd_mode: DM2_FWD_* of the forward (DM2_FWD_UNKNOWN: every candidate kernel is launched and looks at hit_valid itself)
void launch_render_backward(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                            const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                            float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                            float* dL_daa_face_verts, const BinningState& bs, int fwd_mode, TieEntry* tie_queue, int64_t tie_cap,
                            hipStream_t st);
void launch_debug_aa_overlap(int variant, int64_t n, const float* tv, const float* te, const uint8_t* tz, const float* tr,
                             const float* tn, const float* tc, const float* pixmin, float* area, float* grad, int32_t* code,
                             hipStream_t st);
hipError_t launch_exchange_mark(int B, int P, int F, int N, const int32_t* faces, const uint32_t* tiles_touched, uint8_t* flags,
                                uint32_t* counts, hipStream_t st);
hipError_t launch_exchange_pack(int B, int P, int F, int N, const uint8_t* flags, const uint32_t* counts, uint32_t* cursors,
                                const float* dverts, const float* dcolor, const float* dopacity, const float* dintense, float* send,
                                hipStream_t st);
hipError_t launch_exchange_unpack(int B, int P, int F, int N, int rank, const float* recv, const uint32_t* recv_counts_host, int64_t rows,
                                  float* slice_v, float* slice_f, hipStream_t st);
int exchange_max_ranks();
size_t tet_scratch_bytes(int64_t T);
// tet_scratch (tet_scratch_bytes(T)) holds the packed per-tet records of the walk; nullptr = the reference-shaped walk
void launch_layers(const dm2_layers_desc& d, const FaceState& fs, const uint2* ranges, const uint32_t* face_list,
                   LayerImageState ls, void* tet_scratch, int32_t* render_layers, int32_t* render_layers_cnt, hipStream_t st);

}  // namespace dm2

// ---- optional per-stage hipEvent timing (dm2_profile_enable; defined in dm2_api.hip) ----
namespace dm2 {
enum Stage { ST_PREP = 0, ST_EMIT = 1, ST_SORT = 2, ST_RANGES = 3, ST_FWD = 4, ST_BWD = 5, ST_TIES = 6 };
void prof_begin(int stage, hipStream_t st);
void prof_end(int stage, hipStream_t st);
struct StageTimer {
    int s; hipStream_t st;
    StageTimer(int stage, hipStream_t stream) : s(stage), st(stream) { if (s >= 0) prof_begin(s, st); }     // stage < 0: times nothing
    ~StageTimer() { if (s >= 0) prof_end(s, st); }
};
}  // namespace dm2
