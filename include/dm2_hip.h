/* dm2_hip.h -- C ABI of libdm2_hip.so, the MI355X (gfx950) rasterizer hot path.
 *
 * Drop-in boundary: these entry points are what the reference's pybind module
 * `dmesh2_renderer._C` (ext.cpp:5-9) binds, with torch types replaced by raw
 * device pointers + sizes.  A binding (pybind/ctypes/cffi) allocates outputs
 * and scratch with its own allocator, then calls:
 *
 *   render_forward_cuda        (render.h:12-45,  render.cu:28-195)
 *       -> dm2_forward_plan() + dm2_forward_run()
 *   render_backward_cuda       (render.h:47-94,  render.cu:198-373)
 *       -> dm2_backward()
 *   generate_render_layers_cuda(render.h:101-119, render.cu:378-476)
 *       -> dm2_layers_plan() + dm2_layers_run()
 *
 * The plan/run split replaces the reference's resize-callback lambdas
 * (render.cu:20-26, renderer.cu:174-183): the number of (tile,face) pairs is
 * data dependent, so `plan` bins the faces, returns the pair count (and the
 * length of the longest tile list, which picks `run`'s sorting method), and
 * the caller sizes the binning scratch before `run`.
 *
 * All pointers are DEVICE pointers to contiguous row-major arrays unless noted.
 * All functions enqueue on `stream` (a hipStream_t passed as void*; NULL = the
 * null stream); only the plan functions wait for the GPU: for two words their last kernel stores
 * into mapped host memory (polled; a copy + event takes over where such stores are not seen in time).
 * Return value: 0 on success, non-zero on error with a message available from
 * dm2_last_error() (thread local).  The library keeps no global state besides
 * a per-thread pinned staging word.
 */
#ifndef DM2_HIP_H
#define DM2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM2_ABI_VERSION 6
#define DM2_TILE 16 /* config.h:4-5 BLOCK_X = BLOCK_Y = 16 */

/* Inputs of Renderer's op, same meaning and order as render.h:13-45. */
typedef struct dm2_render_desc {
    int32_t B, P, F;              /* views, vertices, faces */
    int32_t W, H;                 /* patch_width, patch_height */
    int32_t K;                    /* len_oarea_buffer (forced to 0 when aa_temperature == 0, render.cu:141-142) */
    float aa_temperature;         /* in [0,1] */
    int32_t flags;                /* DM2_FLAG_* */
    int32_t full_W, full_H;       /* DM2_FLAG_ANALYTIC_RAYS: size of the image the cameras' rays belong to (Renderer.width/height) */
    const float* background;      /* (3) */
    const int32_t* patch_min;     /* (B,2) */
    const float* verts;           /* (P,3) */
    const int32_t* faces;         /* (F,3) */
    const float* verts_color;     /* (P,3) */
    const float* faces_opacity;   /* (F) */
    const float* verts_ndc;       /* (B,P,3) */
    const float* verts_image;     /* (B,P,2) */
    const float* faces_intense;   /* (B,F) */
    const float* aa_face_verts;             /* (B,F,3,2) */
    const float* aa_face_edges;             /* (B,F,3,2) */
    const uint8_t* aa_face_edges_iszero;    /* (B,F,3,2) bool */
    const float* aa_face_edges_recip;       /* (B,F,3,2) */
    const float* aa_face_edges_normal;      /* (B,F,3,2) */
    const float* aa_face_edges_normal_c;    /* (B,F,3) */
    const float* image_ray_o;     /* (B,H,W,3); may be NULL with DM2_FLAG_ANALYTIC_RAYS */
    const float* image_ray_d;     /* (B,H,W,3); may be NULL with DM2_FLAG_ANALYTIC_RAYS */
    const float* ray_cam;         /* DM2_FLAG_ANALYTIC_RAYS: (B,32) = inv(mv) then inv(proj) of each view, row-major 4x4 each */
} dm2_render_desc;

/* flags */
#define DM2_FLAG_CORRECTED_DV 1  /* backward: use the true d(bary v)/d(verts) instead of the
                                    reference's as-written d(t)/d(verts) (auxiliary.h:272-280) */

#define DM2_FLAG_LEGACY_KERNELS 2 /* composite kernels: per-pixel list walk (reference-shaped work distribution)
                                    instead of the dense (pixel,face)-pair kernels; tile lists: one global radix sort
                                    instead of per-tile sorts; same results, kept for A/B */

#define DM2_FLAG_NO_BACKWARD 4    /* forward only: no backward will follow (inference, torch.no_grad()): the forward skips the per-entry
                                    blend masks it otherwise leaves for dm2_backward (32 B per list entry).  A backward called
                                    anyway still works: it takes the mask-free per-pixel walk. */

#define DM2_FLAG_ANALYTIC_RAYS 8  /* the primary rays are not read from image_ray_o / image_ray_d but computed per pixel from ray_cam
                                    in the operation order of the reference's Renderer._init_rays (__init__.py:198-237): pixel
                                    centre -> NDC (x, y, -1, 1) @ inv(proj)^T @ inv(mv)^T, NO perspective divide, direction
                                    normalised with + 1e-6 on the length.  Saves the two (B,H,W,3) tensors (49.8 MB per camera at
                                    1080p) and 24 B per pixel of reads in either pass. */

#define DM2_FLAG_AA_GRAD_TO_VERTS 16 /* dm2_backward: the sixth output is not dL/d(aa_face_verts) (B,F,3,2) but that gradient already
                                    scattered to the vertices it belongs to: a (B,P,2) array, zero-filled by the caller, that
                                    receives dL/d(aa corner) at [b, vertex of that corner] (the CCW reorder of pyrenderer.py:8,521-529
                                    undone: the forward's packed record remembers it).  What torch autograd does with
                                    dL/d(aa_face_verts) in the reference's host prep, without the (B,F,3,2) round trip: for a caller
                                    that owns the host prep as well (dmesh2_renderer_amd/prep.py).  The flag must be set in the
                                    forward call of the same frame too: that is when the record notes the reorder. */

#define DM2_FLAG_TABLES_FROM_IMAGE 32 /* dm2_forward*: the six aa_* tables are not read (their pointers may be NULL): the plan builds them
                                    per (view, face) from verts_image[faces] in registers -- CCW reorder, edges, |e| < 1e-3 flags,
                                    reciprocals, inward normals and their offsets, exactly as pyrenderer.py:6-30 / dm2_prepare_faces
                                    compute them -- straight into its packed face records.  For a caller that owns the host prep
                                    (SURVEY.md 8(f) rank 1): 114 B per face less to write and 114 to read.  The backward's dL/d(aa
                                    corners) then only exists per vertex: combine with DM2_FLAG_AA_GRAD_TO_VERTS. */

#define DM2_FLAG_NO_PAIR_POOL 64   /* dm2_forward*: leave blend masks only (DM2_FWD_MASKS), whatever room the binning scratch has behind
                                    its fixed part: for a caller that finds the plan's pair_bound too large for its memory budget */

/* Scratch kinds for dm2_scratch_bytes (state.h:18-61). */
enum {
    DM2_SCRATCH_FACE = 0,     /* count = B*F, aux = 2 * (B*tiles) + 1 for Renderer (holds the packed face records,
                                 256 B per (view, face), that forward AND backward read), 2 * (B*tiles) for
                                 LayeredRenderer; tiles = ceil(W/16) * ceil(H/16) */
    DM2_SCRATCH_IMAGE = 1,    /* count = B*H*W, aux = B*tiles      */
    DM2_SCRATCH_BINNING = 2,  /* count = num_rendered, aux = B*tiles */
    DM2_SCRATCH_LAYER_IMAGE = 3, /* count = B*H*W, aux = B*tiles    */
    DM2_SCRATCH_LAYER_TETS = 4,  /* count = T (tets): packed per-tet records of the layer walk, 256 B each */
    DM2_SCRATCH_PAIR_POOL = 5,   /* count = pair_bound of the plan: bytes to APPEND to the binning scratch (behind its
                                    DM2_SCRATCH_BINNING bytes for the same num_rendered) for the forward's pair pool: 4 B per
                                    (pixel, face) pair, the coverage the backward would otherwise clip for again */
    DM2_SCRATCH_TIE_QUEUE = 6    /* count = pairs the binning scratch's pool part holds (appended bytes / 4): scratch of ONE
                                    dm2_backward call, 16 B per pair, written only for the pairs the exact clipper has to redo */
};

/* What a forward left for its backward: returned by dm2_forward / dm2_forward_run, to be handed to dm2_backward. */
#define DM2_FWD_UNKNOWN 0  /* dm2_backward decides on the device: every candidate kernel is launched, all but one return at once */
#define DM2_FWD_NONE 1     /* nothing (DM2_FLAG_NO_BACKWARD, DM2_FLAG_LEGACY_KERNELS): the per-pixel walk recomputes everything */
#define DM2_FWD_MASKS 2    /* per list entry the pixels it blended into: the backward re-clips those pairs exactly */
#define DM2_FWD_POOL 3     /* masks + pair pool: the default -- no clip for an area, Jacobians without a polygon */
#define DM2_FWD_POINT 4    /* aa_temperature == 0: per list entry the pixels whose rays hit it (point-sampled coverage) */

int dm2_abi_version(void);
const char* dm2_last_error(void);

/* Bytes of scratch of `kind` for `count` items (replaces required<T>(), state.h:63-69). */
size_t dm2_scratch_bytes(int kind, int64_t count, int64_t aux);

/* Bin faces into 16x16 tiles (preprocessFaceCUDA forward.cu:16-108) and count the
 * entries of every tile list; returns the number of (tile,face) pairs (`num_rendered`,
 * the reference's InclusiveSum of tiles_touched, renderer.cu:165-179) and the length of
 * the longest list (`max_tile_entries`, to be handed to dm2_forward_run), and `pair_bound`: an upper bound of
 * the (pixel, face) pairs the composite will look at (the faces' pixel rectangles inside the patch, summed) -- the
 * size of the pair pool a caller may append to the binning scratch (DM2_SCRATCH_PAIR_POOL).
 * Waits for the 16-byte read-back of those numbers. */
int dm2_forward_plan(const dm2_render_desc* d, void* face_scratch, size_t face_bytes,
                     void* stream, int64_t* num_rendered, int64_t* max_tile_entries, int64_t* pair_bound);

/* Per-tile lists ordered by (depth key, emission order) + tile ranges + per-pixel composite
 * (renderer.cu:185-266, FORWARD::renderCUDA forward.cu:139-432).  The lists are the ones the
 * reference's global stable radix sort of (tile | depth) keys produces; they are built by
 * bucketing the entries per tile and sorting every tile's segment on chip, unless
 * max_tile_entries > 32768 or DM2_FLAG_LEGACY_KERNELS asks for the radix route.
 * out_color (B,H,W,3), out_depth (B,H,W): written for every pixel.
 * out_tri_cnt (B,H,W) int32: number of AA records the reference would hold
 * (min(#overlapping faces visited, K)); may be NULL.  The face / binning / image
 * scratch must be kept (unmodified) for dm2_backward -- they are the three byte
 * buffers the reference returns and takes back (render.cu:194, render.h:79-81).
 * The face scratch holds a packed copy of the per-face inputs as the forward saw
 * them; the backward differentiates with respect to those.
 * Pair pool: when binning_bytes >= DM2_SCRATCH_BINNING bytes + DM2_SCRATCH_PAIR_POOL bytes for pair_bound, the
 * composite also leaves the coverage ratio (forward.cu:375-378) of every blended pair in the appended part and
 * *forward_mode is DM2_FWD_POOL; otherwise DM2_FWD_MASKS (a caller that finds pair_bound too large for its memory
 * simply appends nothing), DM2_FWD_POINT at aa_temperature 0, or DM2_FWD_NONE under DM2_FLAG_NO_BACKWARD /
 * DM2_FLAG_LEGACY_KERNELS.  forward_mode may be NULL. */
int dm2_forward_run(const dm2_render_desc* d, int64_t num_rendered, int64_t max_tile_entries, int64_t pair_bound,
                    void* face_scratch, size_t face_bytes,
                    void* binning_scratch, size_t binning_bytes,
                    void* image_scratch, size_t image_bytes,
                    float* out_color, float* out_depth, int32_t* out_tri_cnt, void* stream, int32_t* forward_mode);

/* dm2_forward_plan + dm2_forward_run in ONE call for a caller that already holds a binning scratch of plausible size (a
 * training loop: last frame's size plus headroom): the run step is enqueued straight from the plan's read-back, without
 * the round trip through the caller that otherwise leaves the GPU idle (~15 us of a 25-us gap through Python).
 * Returns 0 (rendered; *num_rendered / *max_tile_entries / *pair_bound / *forward_mode set), 2 when binning_bytes is
 * smaller than dm2_scratch_bytes(DM2_SCRATCH_BINNING, *num_rendered, B*tiles) + dm2_scratch_bytes(DM2_SCRATCH_PAIR_POOL,
 * *pair_bound, 0) -- the plan is done and nothing else: allocate and call dm2_forward_run -- or 1 on error.  A
 * larger-than-needed binning scratch is fine, also for dm2_backward (which must be given the same size). */
int dm2_forward(const dm2_render_desc* d, void* face_scratch, size_t face_bytes,
                void* binning_scratch, size_t binning_bytes, void* image_scratch, size_t image_bytes,
                float* out_color, float* out_depth, int32_t* out_tri_cnt, void* stream,
                int64_t* num_rendered, int64_t* max_tile_entries, int64_t* pair_bound, int32_t* forward_mode);

/* Gradients (BACKWARD::renderCUDA backward.cu:17-532).  The six outputs must be
 * zero-filled by the caller (the reference's zeros_like, render.cu:313-318):
 * dL_dverts (P,3), dL_dverts_color (P,3), dL_dfaces_opacity (F),
 * dL_dverts_ndc (B,P,3) [only z written], dL_dfaces_intense (B,F),
 * dL_daa_face_verts (B,F,3,2).
 * forward_mode: what the forward of this frame returned (DM2_FWD_*; DM2_FWD_UNKNOWN costs a few idle launches).
 * tie_scratch: needed with DM2_FWD_POOL (and with DM2_FWD_UNKNOWN when the binning scratch has a pool part):
 * dm2_scratch_bytes(DM2_SCRATCH_TIE_QUEUE, pairs of the pool part, 0) bytes, scratch of this call only.  The binning
 * scratch is not const: the pool backward keeps its queue counters there (left as it found them). */
int dm2_backward(const dm2_render_desc* d, int64_t num_rendered, int32_t forward_mode,
                 const float* dL_dout_color, const float* dL_dout_depth,
                 const void* face_scratch, size_t face_bytes,
                 void* binning_scratch, size_t binning_bytes,
                 const void* image_scratch, size_t image_bytes,
                 void* tie_scratch, size_t tie_bytes,
                 float* dL_dverts, float* dL_dverts_color, float* dL_dfaces_opacity,
                 float* dL_dverts_ndc, float* dL_dfaces_intense, float* dL_daa_face_verts,
                 void* stream);

/* LayeredRenderer (render.h:101-119). */
typedef struct dm2_layers_desc {
    int32_t B, P, F, T;
    int32_t W, H, L;              /* full frame width/height, num_layers */
    int32_t flags;                /* DM2_FLAG_ANALYTIC_RAYS, DM2_FLAG_LEGACY_KERNELS */
    const float* verts;           /* (P,3) */
    const int32_t* faces;         /* (F,3) */
    const int32_t* tets;          /* (T,4) */
    const int32_t* face_tets;     /* (F,2), -1 = none */
    const int32_t* tet_faces;     /* (T,4) */
    const int32_t* face_existence;/* (F) */
    const float* verts_ndc;       /* (B,P,3) */
    const float* verts_image;     /* (B,P,2) */
    const float* image_ray_o;     /* (B,H,W,3); may be NULL with DM2_FLAG_ANALYTIC_RAYS */
    const float* image_ray_d;     /* (B,H,W,3); may be NULL with DM2_FLAG_ANALYTIC_RAYS */
    const float* ray_cam;         /* DM2_FLAG_ANALYTIC_RAYS: (B,32), see dm2_render_desc */
} dm2_layers_desc;

int dm2_layers_plan(const dm2_layers_desc* d, void* face_scratch, size_t face_bytes,
                    void* stream, int64_t* num_rendered, int64_t* max_tile_entries);
/* render_layers (B,H,W,L) must be pre-filled with -1 and render_layers_cnt (B,H,W)
 * with 0 by the caller (render.cu:437-438).  tet_scratch (DM2_SCRATCH_LAYER_TETS bytes for T tets; scratch of this call
 * only) receives one packed 256-B record per tet -- face vertices, outward normals, neighbours, existence flags -- so
 * that a step of the tet walk (forward.cu:853-996) is one contiguous fetch; NULL (or DM2_FLAG_LEGACY_KERNELS) selects the
 * reference's access pattern.  Same layers either way. */
int dm2_layers_run(const dm2_layers_desc* d, int64_t num_rendered, int64_t max_tile_entries,
                   void* face_scratch, size_t face_bytes,
                   void* binning_scratch, size_t binning_bytes,
                   void* image_scratch, size_t image_bytes,
                   void* tet_scratch, size_t tet_bytes,
                   int32_t* render_layers, int32_t* render_layers_cnt, void* stream);

/* Host prep of Renderer.forward / LayeredRenderer.generate, fused.
 * Replaces the ~20 torch kernels of the reference's Python host layer:
 *   compute_verts_ndc_image  dmesh2_renderer/__init__.py:239-262  (projection, |w| clamp, NDC -> image)
 *   Triangles                dmesh2_renderer/pyrenderer.py:6-30   (CCW reorder + the six AA tables)
 * mv / proj are the (B,4,4) matrices of the selected cameras, row-major, applied as
 * `hom @ mv^T @ proj^T`.  Any of the eight outputs may be NULL (LayeredRenderer needs the
 * first two only); with F == 0 or all aa_* NULL only the projection runs. */
typedef struct dm2_prep_desc {
    int32_t B, P, F;
    int32_t W, H;                 /* FULL image width / height (Renderer.width/height), not the patch */
    const float* verts;           /* (P,3) */
    const int32_t* faces;         /* (F,3) */
    const float* mv;              /* (B,4,4) */
    const float* proj;            /* (B,4,4) */
    float* verts_ndc;             /* (B,P,3) out */
    float* verts_image;           /* (B,P,2) out (input of the backward when non-NULL there) */
    float* aa_face_verts;         /* (B,F,3,2) out */
    float* aa_face_edges;         /* (B,F,3,2) out */
    uint8_t* aa_face_edges_iszero;/* (B,F,3,2) out, bool */
    float* aa_face_edges_recip;   /* (B,F,3,2) out */
    float* aa_face_edges_normal;  /* (B,F,3,2) out */
    float* aa_face_edges_normal_c;/* (B,F,3) out */
} dm2_prep_desc;

int dm2_prepare_faces(const dm2_prep_desc* d, void* stream);
/* The gradient torch autograd sends back through the host prep: upstream gradients of verts_ndc
 * (B,P,3), verts_image (B,P,2) and aa_face_verts (B,F,3,2) (each may be NULL) -> g_verts (P,3),
 * overwritten.  image_grad_scratch: B*P*2 floats of caller scratch.  Only the input fields of `d`
 * are read. */
int dm2_prepare_faces_backward(const dm2_prep_desc* d, const float* g_verts_ndc, const float* g_verts_image,
                               const float* g_aa_face_verts, float* image_grad_scratch, float* g_verts,
                               void* stream);

/* Device side of the sparse leaf-gradient exchange of a frame sharded by tile rows over N ranks (the
 * collectives themselves belong to the caller: dmesh2_renderer_amd/sharding.py drives torch.distributed / RCCL).  Rows are
 * owned by contiguous id ranges: face f by rank f / ceil(F/N), vertex v by rank v / ceil(P/N).
 *   dm2_exchange_mark    after dm2_forward*: flags (P + F bytes of caller scratch: F face flags, then P vertex flags) mark the
 *                        faces this rank's tile lists hold (any view) and their vertices; counts (2 N uint32, device):
 *                        [2 o] faces / [2 o + 1] vertices flagged in owner o's range = the rows this rank will send to o.
 *   dm2_exchange_pack    after the backward: the send buffer, per owner o [counts[2 o] rows of (2 + B) floats: id bits, dopacity,
 *                        dintense(b = 0..B-1) | counts[2 o + 1] rows of 7 floats: id bits, dverts(3), dverts_color(3)];
 *                        cursors: 2 N uint32 of scratch.  Rows of one segment in no particular order.
 *   dm2_exchange_unpack  owner `rank`: recv holds, per source s, [recv_counts[2 s] face rows | recv_counts[2 s + 1] vertex rows]
 *                        (`rows` rows in all; recv_counts is a HOST array -- the caller needed these numbers on the host for the
 *                        all-to-all anyway); they are summed, source by source, into slice_v (ceil(P/N), 6) and slice_f
 *                        (ceil(F/N), 1 + B), which the call zero-fills first. */
int dm2_exchange_mark(int32_t B, int32_t P, int32_t F, int32_t N, const int32_t* faces, const void* face_scratch, size_t face_bytes,
                      uint8_t* flags, uint32_t* counts, void* stream);
int dm2_exchange_pack(int32_t B, int32_t P, int32_t F, int32_t N, const uint8_t* flags, const uint32_t* counts, uint32_t* cursors,
                      const float* dverts, const float* dverts_color, const float* dfaces_opacity, const float* dfaces_intense,
                      float* send, void* stream);
int dm2_exchange_unpack(int32_t B, int32_t P, int32_t F, int32_t N, int32_t rank, const float* recv, const uint32_t* recv_counts,
                        int64_t rows, float* slice_v, float* slice_f, void* stream);

/* Introspection for tests/bench: copy pieces of the scratch state to caller
 * (device) buffers.  what: 0 ranges (B*tiles*2 u32, from image scratch),
 * 1 face_list (num_rendered u32, from binning scratch), 2 final_T, 3 final_prev_T
 * (N f32), 4 n_contrib (N u32), 5 first_face, 6 first_tet (N i32, layer image scratch),
 * 8 tiles_touched (count = B*F u32, from face scratch; aux = the aux of dm2_scratch_bytes). */
int dm2_debug_fetch(int what, int64_t count, int64_t aux, int64_t num_rendered,
                    const void* scratch, size_t scratch_bytes, void* dst, void* stream);

/* Test hook: run one of the device clippers on n independent (triangle, pixel) pairs.  Tables as the reference's
 * Triangles builds them (pyrenderer.py:6-30), n x the per-face layout of dm2_render_desc; pixmin (n,2) float: the
 * pixel's (x, y) origin, unit size.  variant 0: the generic clipper of the per-pixel-walk kernels, area + Jacobian in
 * the reference's order (aa.h:151-504); 1: the forward's area-only clipper; 2: the forward's accept / reject decision,
 * then the backward's segment formulation of area + Jacobian; 3: the same with the reference's fan sum over its corners
 * (the area bit-identical to the forward's, what the exact-clipper backward uses for faces with opacity > 0.9); 4: the
 * default backward's Jacobian without a polygon (code -1 and a zero Jacobian: the pair is a tie and takes variant 2's
 * route; area: the forward's).  Outputs: area (n), grad (n,3,2), code (n) int32 --
 * 0 = no error, non-zero = the reference reports one of its errors E00..E05 (dmesh2_renderer/README.md; the
 * composite kernels only ever test != 0); area and grad are zero where code != 0. */
int dm2_debug_aa_overlap(int variant, int64_t n, const float* aa_face_verts, const float* aa_face_edges,
                         const uint8_t* aa_face_edges_iszero, const float* aa_face_edges_recip,
                         const float* aa_face_edges_normal, const float* aa_face_edges_normal_c,
                         const float* pixmin, float* area, float* grad, int32_t* code, void* stream);

/* Optional per-stage timing (bench/profiling only; off by default).  When enabled, the
 * forward/backward/layers entry points record hipEvents on `stream` around every stage of
 * the calling thread's next calls; dm2_profile_read waits for the last one and returns the
 * milliseconds of each stage of the most recent forward_plan/forward_run/backward call:
 * [0] preprocess + tile scan  [1] scatter into the tile segments (radix route: scan + key emit)
 * [2] per-tile sorts (radix route: the radix sort)  [3] tile ranges (radix route only)
 * [4] forward composite [5] backward composite [6] the backward's tie pass (k_aa_ties; 0 when another backward kernel
 * ran).  Returns the number of values written. */
#define DM2_PROFILE_STAGES 7
void dm2_profile_enable(int on);
int dm2_profile_read(float* ms, int capacity);

/* Diagnostic builds only (-DDM2_STAMPS): copy the in-kernel cycle-stamp table (2 kernels x 16
 * segments, shader cycles summed over waves) to the HOST array `out`; returns the number of
 * values, or -1 in a product build (which contains no stamps). */
int dm2_debug_stamps(uint64_t* out, int capacity, int reset);

#ifdef __cplusplus
}
#endif
#endif /* DM2_HIP_H */
