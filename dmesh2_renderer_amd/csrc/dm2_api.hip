// dm2_api.hip -- extern "C" entry points of libdm2_hip.so (include/dm2_hip.h).
// Host orchestration equivalent to CudaRenderer::Renderer::forward/backward and
// RenderLayerGenerator::forward (renderer.cu:78-269, 271-392, 509-674), without
// the reference's per-stage cudaDeviceSynchronize (auxiliary.h:433-440): the
// only host wait is the read-back of the pair count in the plan step.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "dm2_device_math.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

#ifdef DM2_STAMPS
namespace dm2 {
unsigned long long* stamps_table() {
    static unsigned long long* p = nullptr;
    if (!p) { (void)hipMalloc((void**)&p, 2 * DM2_NSTAMP * sizeof(unsigned long long)); (void)hipMemset(p, 0, 2 * DM2_NSTAMP * sizeof(unsigned long long)); }
    return p;
}
}  // namespace dm2
#endif

namespace {

thread_local std::string g_err;

int fail(const std::string& m) { g_err = m; return 1; }

#define DM2_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct PinnedWord {     // per-thread mapped host words for the plan's read-back
    uint32_t* p = nullptr;      // host view: [0] num_rendered [1] longest tile list [2,3] pair bound [4] sequence word
    uint32_t* dp = nullptr;     // device view of the same memory, for device `dev`
    int dev = -1;
    uint32_t seq = 0;
    hipEvent_t ev = nullptr;
    ~PinnedWord() { if (p) (void)hipHostFree(p); if (ev) (void)hipEventDestroy(ev); }
};
thread_local PinnedWord g_pin;

// (num_rendered, entries of the longest tile list) of a plan: the one host read-back of a forward.  The plan's last kernel
// stores them into mapped host memory and then a sequence word; the host polls that word (microseconds instead of the
// ~30 us of a copy command + event).  Should the word not show up in time (a platform that delays such stores until the
// kernel has retired), the classical route takes over: an event behind a device-to-host copy of the device-side words.
int plan_meta_prepare(uint32_t** host_meta_dev, uint32_t* seq) {
    if (!g_pin.p) {
        DM2_HIP(hipHostMalloc((void**)&g_pin.p, 64, hipHostMallocMapped | hipHostMallocPortable));
        for (int k = 0; k < 8; k++) g_pin.p[k] = 0u;
    }
    int cur = 0;
    DM2_HIP(hipGetDevice(&cur));
    if (cur != g_pin.dev) {                                               // (a thread that moves to another GPU: that GPU's view)
        void* dp = nullptr;
        DM2_HIP(hipHostGetDevicePointer(&dp, g_pin.p, 0));
        g_pin.dp = (uint32_t*)dp; g_pin.dev = cur;
        if (g_pin.ev) { (void)hipEventDestroy(g_pin.ev); g_pin.ev = nullptr; }     // (events belong to a device as well)
    }
    if (!g_pin.ev) DM2_HIP(hipEventCreateWithFlags(&g_pin.ev, hipEventDisableTiming));
    g_pin.seq = g_pin.seq + 1u ? g_pin.seq + 1u : 1u;                      // never 0: the words start out as 0
    *host_meta_dev = g_pin.dp; *seq = g_pin.seq;
    return 0;
}
int plan_meta_wait(const uint32_t* plan_meta, hipStream_t st, int64_t* num_rendered, int64_t* max_tile_entries, int64_t* pair_bound) {
    DM2_HIP(hipEventRecord(g_pin.ev, st));                                 // (also tells whether the plan has retired)
    volatile uint32_t* hp = g_pin.p;
    uint32_t r = 0, longest = 0, plo = 0, phi = 0;
    bool seen = false;
    auto take = [&]() { r = hp[0]; longest = hp[1]; plo = hp[2]; phi = hp[3]; seen = true; };
    for (long spin = 0; ; spin++) {
        if (__atomic_load_n(&hp[4], __ATOMIC_ACQUIRE) == g_pin.seq) { take(); break; }
        if ((spin & 1023) == 1023) {
            if (hipEventQuery(g_pin.ev) != hipErrorNotReady) {
                // the plan's kernels are done (or the stream is in error): the stores are visible now, or they never will be
                if (__atomic_load_n(&hp[4], __ATOMIC_ACQUIRE) == g_pin.seq) take();
                break;
            }
            // earlier work still ahead of the plan on this stream (a caller that runs ahead): stop burning the core
            if (spin > (1l << 17)) { (void)hipEventSynchronize(g_pin.ev); if (__atomic_load_n(&hp[4], __ATOMIC_ACQUIRE) == g_pin.seq) take(); break; }
        }
        __builtin_ia32_pause();
    }
    if (!seen) {
        uint32_t tmp[4] = {0u, 0u, 0u, 0u};
        DM2_HIP(hipMemcpyAsync(tmp, plan_meta, sizeof(tmp), hipMemcpyDeviceToHost, st));
        DM2_HIP(hipStreamSynchronize(st));
        r = tmp[0]; longest = tmp[1]; plo = tmp[2]; phi = tmp[3];
    }
    *num_rendered = (int64_t)r;
    *max_tile_entries = (int64_t)longest;
    if (pair_bound) *pair_bound = (int64_t)(((uint64_t)phi << 32) | plo);
    if (longest == 0xFFFFFFFFu) { *num_rendered = 0; *max_tile_entries = 0; return fail("more than 2^31 - 1 (tile, face) pairs: render smaller patches"); }
    return 0;
}

int check_render_desc(const dm2_render_desc* d) {
    if (!d) return fail("null descriptor");
    if (d->B < 0 || d->P < 0 || d->F < 0 || d->W < 0 || d->H < 0) return fail("negative size in descriptor");
    if (d->aa_temperature < 0 || d->aa_temperature > 1) return fail("aa_temperature must be in the range [0, 1]");
    if (d->K < 0) return fail("len_oarea_buffer must be non-negative");
    const int64_t gx = (d->W + dm2::TILE - 1) / dm2::TILE, gy = (d->H + dm2::TILE - 1) / dm2::TILE;
    if (gx > 0xFFFF || gy > 0xFFFF) return fail("patch too large: more than 65535 tiles per axis");
    if (d->B > 65535) return fail("more than 65535 views per call");
    if ((int64_t)d->B * gx * gy >= (1ll << 31)) return fail("too many tiles");
    if (!(d->flags & DM2_FLAG_TABLES_FROM_IMAGE) && d->B > 0 && d->F > 0 && d->P > 0 &&
        (!d->aa_face_verts || !d->aa_face_edges || !d->aa_face_edges_iszero || !d->aa_face_edges_recip || !d->aa_face_edges_normal || !d->aa_face_edges_normal_c))
        return fail("the aa_face_* tables must not be null (or set DM2_FLAG_TABLES_FROM_IMAGE)");
    if (d->flags & DM2_FLAG_ANALYTIC_RAYS) {
        if (!d->ray_cam && d->B > 0) return fail("DM2_FLAG_ANALYTIC_RAYS needs ray_cam");
        if (d->full_W <= 0 || d->full_H <= 0) return fail("DM2_FLAG_ANALYTIC_RAYS needs the full image size");
    }
    return 0;
}

struct Profiler {
    bool on = false;
    hipEvent_t ev[2 * DM2_PROFILE_STAGES] = {};
    bool have[DM2_PROFILE_STAGES] = {};
    bool created = false;
};
thread_local Profiler g_prof;

inline int64_t tiles_of(int B, int W, int H) {
    return (int64_t)B * ((W + dm2::TILE - 1) / dm2::TILE) * ((H + dm2::TILE - 1) / dm2::TILE);
}

}  // namespace

namespace dm2 {
void prof_begin(int stage, hipStream_t st) {
    Profiler& p = g_prof;
    if (!p.on) return;
    if (!p.created) { for (auto& e : p.ev) (void)hipEventCreate(&e); p.created = true; }
    (void)hipEventRecord(p.ev[2 * stage], st);
}
void prof_end(int stage, hipStream_t st) {
    Profiler& p = g_prof;
    if (!p.on) return;
    (void)hipEventRecord(p.ev[2 * stage + 1], st);
    p.have[stage] = true;
}
}  // namespace dm2

extern "C" {

void dm2_profile_enable(int on) {
    g_prof.on = on != 0;
    for (auto& h : g_prof.have) h = false;
}

int dm2_profile_read(float* ms, int capacity) {
    Profiler& p = g_prof;
    int n = 0;
    for (int s = 0; s < DM2_PROFILE_STAGES && s < capacity; s++, n++) {
        ms[s] = 0.f;
        if (!p.on || !p.have[s]) continue;
        if (hipEventSynchronize(p.ev[2 * s + 1]) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.ev[2 * s], p.ev[2 * s + 1]) == hipSuccess) ms[s] = t;
    }
    return n;
}

int dm2_debug_stamps(uint64_t* out, int capacity, int reset) {
#ifdef DM2_STAMPS
    unsigned long long h[2][DM2_NSTAMP];
    if (hipMemcpy(h, dm2::stamps_table(), sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    int n = 0;
    for (int k = 0; k < 2; k++) for (int i = 0; i < DM2_NSTAMP && n < capacity; i++) out[n++] = h[k][i];
    if (reset) (void)hipMemset(dm2::stamps_table(), 0, sizeof(h));
    return n;
#else
    (void)out; (void)capacity; (void)reset;
    return -1;
#endif
}

int dm2_abi_version(void) { return DM2_ABI_VERSION; }
const char* dm2_last_error(void) { return g_err.c_str(); }

size_t dm2_scratch_bytes(int kind, int64_t count, int64_t aux) {
    size_t total = 0;
    if (count < 0) count = 0;
    if (aux < 0) aux = 0;
    switch (kind) {
        case DM2_SCRATCH_FACE: dm2::FaceState::carve(nullptr, count, aux >> 1, dm2::scan_temp_bytes(count), (aux & 1) != 0, &total); break;
        case DM2_SCRATCH_IMAGE: dm2::ImageState::carve(nullptr, count, aux, &total); break;
        case DM2_SCRATCH_BINNING: dm2::BinningState::carve(nullptr, count, dm2::sort_temp_bytes(count, aux), &total); break;
        case DM2_SCRATCH_LAYER_IMAGE: dm2::LayerImageState::carve(nullptr, count, aux, &total); break;
        case DM2_SCRATCH_LAYER_TETS: return dm2::tet_scratch_bytes(count);
        case DM2_SCRATCH_PAIR_POOL: return dm2::BinningState::pool_bytes(count);
        case DM2_SCRATCH_TIE_QUEUE: return count > 0 ? (size_t)count * sizeof(dm2::TieEntry) + dm2::ALIGN : 0;
        default: return 0;
    }
    return total;
}

static int forward_plan(const dm2_render_desc* d, void* face_scratch, size_t face_bytes, void* stream, int64_t* num_rendered,
                        int64_t* max_tile_entries, int64_t* pair_bound, uint2* ranges_to_clear, uint32_t* tile_order) {
    if (check_render_desc(d)) return 1;
    if (!num_rendered || !max_tile_entries) return fail("num_rendered / max_tile_entries is null");
    hipStream_t st = (hipStream_t)stream;
    const int64_t BF = (int64_t)d->B * d->F, Tn = tiles_of(d->B, d->W, d->H);
    *num_rendered = 0; *max_tile_entries = 0;
    if (pair_bound) *pair_bound = 0;
    if (d->P == 0 || BF == 0 || Tn == 0) return 0;                       // render.cu:149
    if (dm2_scratch_bytes(DM2_SCRATCH_FACE, BF, 2 * Tn + 1) > face_bytes) return fail("face scratch too small");
    dm2::FaceState fs = dm2::FaceState::carve(face_scratch, BF, Tn, dm2::scan_temp_bytes(BF), true);
    uint32_t* host_meta = nullptr; uint32_t seq = 0;
    if (plan_meta_prepare(&host_meta, &seq)) return 1;
    DM2_HIP(dm2::launch_preprocess_scan(d->B, d->P, d->F, d->W, d->H, d->patch_min, d->faces, d->verts_ndc, d->verts_image, fs, d,
                                        host_meta, seq, ranges_to_clear, tile_order, st));
    DM2_HIP(hipGetLastError());
    return plan_meta_wait(fs.plan_meta, st, num_rendered, max_tile_entries, pair_bound);
}

int dm2_forward_plan(const dm2_render_desc* d, void* face_scratch, size_t face_bytes, void* stream, int64_t* num_rendered,
                     int64_t* max_tile_entries, int64_t* pair_bound) {
    return forward_plan(d, face_scratch, face_bytes, stream, num_rendered, max_tile_entries, pair_bound, nullptr, nullptr);
}

static int forward_run(const dm2_render_desc* d, int64_t num_rendered, int64_t max_tile_entries, int64_t pair_bound, void* face_scratch, size_t face_bytes,
                       void* binning_scratch, size_t binning_bytes, void* image_scratch, size_t image_bytes,
                       float* out_color, float* out_depth, int32_t* out_tri_cnt, void* stream, bool ranges_cleared, int32_t* forward_mode) {
    if (forward_mode) *forward_mode = DM2_FWD_NONE;
    if (check_render_desc(d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t BF = (int64_t)d->B * d->F, N = (int64_t)d->B * d->H * d->W, Tn = tiles_of(d->B, d->W, d->H);
    if (N == 0) return 0;
    if (dm2_scratch_bytes(DM2_SCRATCH_IMAGE, N, Tn) > image_bytes) return fail("image scratch too small");
    dm2::ImageState is = dm2::ImageState::carve(image_scratch, N, Tn);
    const bool have_faces = (d->P != 0 && BF != 0);
    dm2::BinningState bs{};
    if (have_faces) {
        if (dm2_scratch_bytes(DM2_SCRATCH_FACE, BF, 2 * Tn + 1) > face_bytes) return fail("face scratch too small");
        if (dm2_scratch_bytes(DM2_SCRATCH_BINNING, num_rendered, Tn) > binning_bytes) return fail("binning scratch too small");
        dm2::FaceState fs = dm2::FaceState::carve(face_scratch, BF, Tn, dm2::scan_temp_bytes(BF), true);
        bs = dm2::BinningState::carve(binning_scratch, num_rendered, dm2::sort_temp_bytes(num_rendered, Tn), nullptr, binning_bytes);
        is.face_recs = fs.recs;
        DM2_HIP(dm2::launch_bin_sort(d->B, d->F, d->W, d->H, num_rendered, max_tile_entries, (d->flags & DM2_FLAG_LEGACY_KERNELS) != 0,
                                     fs.depths, fs, bs, is.ranges, ranges_cleared, is.tile_order, st));   // renderer.cu:192
    } else {
        DM2_HIP(hipMemsetAsync(is.ranges, 0, (size_t)Tn * sizeof(uint2), st));
        dm2::launch_tile_order_identity(Tn, is.tile_order, st);
    }
    // the pair pool is used when the caller appended room for every pair the plan counted (a smaller appendix is ignored)
    const bool use_pool = have_faces && pair_bound > 0 && bs.pool_cap >= pair_bound && !(d->flags & DM2_FLAG_NO_PAIR_POOL);
    const float pairs_per_entry = num_rendered > 0 ? (float)((double)pair_bound / (double)num_rendered) : 0.0f;
    const int mode = dm2::launch_render_forward(*d, is.ranges, bs.face_list, is, out_color, out_depth, out_tri_cnt, bs, use_pool, pairs_per_entry, st);
    if (forward_mode) *forward_mode = mode;
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_forward_run(const dm2_render_desc* d, int64_t num_rendered, int64_t max_tile_entries, int64_t pair_bound, void* face_scratch, size_t face_bytes,
                    void* binning_scratch, size_t binning_bytes, void* image_scratch, size_t image_bytes,
                    float* out_color, float* out_depth, int32_t* out_tri_cnt, void* stream, int32_t* forward_mode) {
    return forward_run(d, num_rendered, max_tile_entries, pair_bound, face_scratch, face_bytes, binning_scratch, binning_bytes, image_scratch,
                       image_bytes, out_color, out_depth, out_tri_cnt, stream, false, forward_mode);
}

int dm2_forward(const dm2_render_desc* d, void* face_scratch, size_t face_bytes, void* binning_scratch, size_t binning_bytes,
                void* image_scratch, size_t image_bytes, float* out_color, float* out_depth, int32_t* out_tri_cnt, void* stream,
                int64_t* num_rendered, int64_t* max_tile_entries, int64_t* pair_bound, int32_t* forward_mode) {
    if (check_render_desc(d)) return 1;
    if (!pair_bound) return fail("pair_bound is null");
    if (forward_mode) *forward_mode = DM2_FWD_NONE;
    const int64_t N = (int64_t)d->B * d->H * d->W, Tn = tiles_of(d->B, d->W, d->H);
    // the image scratch is at hand already: the plan's last kernel clears the tile ranges, one launch less in the run step
    uint2* ranges = nullptr; uint32_t* tile_order = nullptr;
    if (N > 0 && image_scratch && dm2_scratch_bytes(DM2_SCRATCH_IMAGE, N, Tn) <= image_bytes) {
        const dm2::ImageState is0 = dm2::ImageState::carve(image_scratch, N, Tn);
        ranges = is0.ranges; tile_order = is0.tile_order;
    }
    if (forward_plan(d, face_scratch, face_bytes, stream, num_rendered, max_tile_entries, pair_bound, ranges, tile_order)) return 1;
    const bool planned = d->P != 0 && (int64_t)d->B * d->F != 0 && Tn != 0;     // (otherwise no plan kernel ran)
    const bool wants_pool = d->aa_temperature > 0.0f && !(d->flags & (DM2_FLAG_NO_BACKWARD | DM2_FLAG_LEGACY_KERNELS | DM2_FLAG_NO_PAIR_POOL));
    if (dm2_scratch_bytes(DM2_SCRATCH_BINNING, *num_rendered, Tn) + (wants_pool ? dm2_scratch_bytes(DM2_SCRATCH_PAIR_POOL, *pair_bound, 0) : 0) > binning_bytes)
        return 2;                                                                // plan done; allocate, then dm2_forward_run
    return forward_run(d, *num_rendered, *max_tile_entries, *pair_bound, face_scratch, face_bytes, binning_scratch, binning_bytes,
                       image_scratch, image_bytes, out_color, out_depth, out_tri_cnt, stream, planned && ranges != nullptr, forward_mode);
}

int dm2_backward(const dm2_render_desc* d, int64_t num_rendered, int32_t forward_mode, const float* dL_dout_color, const float* dL_dout_depth,
                 const void* face_scratch, size_t face_bytes, void* binning_scratch, size_t binning_bytes,
                 const void* image_scratch, size_t image_bytes, void* tie_scratch, size_t tie_bytes,
                 float* dL_dverts, float* dL_dverts_color, float* dL_dfaces_opacity, float* dL_dverts_ndc,
                 float* dL_dfaces_intense, float* dL_daa_face_verts, void* stream) {
    if (check_render_desc(d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)d->B * d->H * d->W, Tn = tiles_of(d->B, d->W, d->H);
    if (d->F == 0 || d->P == 0 || N == 0 || num_rendered <= 0) return 0;       // render.cu:320
    const int64_t BF = (int64_t)d->B * d->F;
    if (dm2_scratch_bytes(DM2_SCRATCH_FACE, BF, 2 * Tn + 1) > face_bytes) return fail("face scratch too small");
    if (dm2_scratch_bytes(DM2_SCRATCH_IMAGE, N, Tn) > image_bytes) return fail("image scratch too small");
    if (dm2_scratch_bytes(DM2_SCRATCH_BINNING, num_rendered, Tn) > binning_bytes) return fail("binning scratch too small");
    dm2::ImageState is = dm2::ImageState::carve(const_cast<void*>(image_scratch), N, Tn);
    if (forward_mode < DM2_FWD_UNKNOWN || forward_mode > DM2_FWD_POINT) return fail("forward_mode must be one of DM2_FWD_*");
    dm2::BinningState bs = dm2::BinningState::carve(binning_scratch, num_rendered, dm2::sort_temp_bytes(num_rendered, Tn), nullptr, binning_bytes);
    is.face_recs = dm2::FaceState::carve(const_cast<void*>(face_scratch), BF, Tn, dm2::scan_temp_bytes(BF), true).recs;
    // the tie queue holds at most one entry per pair of the pool
    dm2::TieEntry* tie_queue = nullptr; int64_t tie_cap = 0;
    if (tie_scratch && tie_bytes >= sizeof(dm2::TieEntry) + dm2::ALIGN) {
        dm2::Carver c(tie_scratch);
        tie_queue = c.take<dm2::TieEntry>(0);
        tie_cap = (int64_t)((tie_bytes - c.used(tie_scratch)) / sizeof(dm2::TieEntry));
    }
    if (forward_mode == DM2_FWD_POOL && d->aa_temperature > 0.0f && !(d->flags & DM2_FLAG_LEGACY_KERNELS)) {
        if (bs.pool_cap <= 0) return fail("forward_mode is DM2_FWD_POOL but the binning scratch has no pool part");
        if (tie_cap < bs.pool_cap) return fail("tie scratch too small for the pool part of the binning scratch");
    }
    dm2::launch_render_backward(*d, is.ranges, bs.face_list, is, dL_dout_color, dL_dout_depth, dL_dverts, dL_dverts_color,
                                dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts, bs, forward_mode, tie_queue, tie_cap, st);
    DM2_HIP(hipGetLastError());
    return 0;
}

static int check_layers_desc(const dm2_layers_desc* d) {
    if (!d) return fail("null descriptor");
    if (d->B < 0 || d->P < 0 || d->F < 0 || d->T < 0 || d->W < 0 || d->H < 0) return fail("negative size in descriptor");
    if (d->L < 0) return fail("num_layers must be non-negative");
    const int64_t gx = (d->W + dm2::TILE - 1) / dm2::TILE, gy = (d->H + dm2::TILE - 1) / dm2::TILE;
    if (gx > 0xFFFF || gy > 0xFFFF || d->B > 65535) return fail("image too large");
    if ((int64_t)d->B * gx * gy >= (1ll << 31)) return fail("too many tiles");
    if ((d->flags & DM2_FLAG_ANALYTIC_RAYS) && !d->ray_cam && d->B > 0) return fail("DM2_FLAG_ANALYTIC_RAYS needs ray_cam");
    return 0;
}

int dm2_layers_plan(const dm2_layers_desc* d, void* face_scratch, size_t face_bytes, void* stream, int64_t* num_rendered,
                    int64_t* max_tile_entries) {
    if (check_layers_desc(d)) return 1;
    if (!num_rendered || !max_tile_entries) return fail("num_rendered / max_tile_entries is null");
    hipStream_t st = (hipStream_t)stream;
    const int64_t BF = (int64_t)d->B * d->F, Tn = tiles_of(d->B, d->W, d->H);
    *num_rendered = 0; *max_tile_entries = 0;
    if (BF == 0 || Tn == 0) return 0;
    if (dm2_scratch_bytes(DM2_SCRATCH_FACE, BF, 2 * Tn) > face_bytes) return fail("face scratch too small");
    dm2::FaceState fs = dm2::FaceState::carve(face_scratch, BF, Tn, dm2::scan_temp_bytes(BF), false);
    // patch_min = 0 (renderer.cu:557-558): a null patch_min means "all zeros"
    uint32_t* host_meta = nullptr; uint32_t seq = 0;
    if (plan_meta_prepare(&host_meta, &seq)) return 1;
    DM2_HIP(dm2::launch_preprocess_scan(d->B, d->P, d->F, d->W, d->H, nullptr, d->faces, d->verts_ndc, d->verts_image, fs, nullptr,
                                        host_meta, seq, nullptr, nullptr, st));
    DM2_HIP(hipGetLastError());
    return plan_meta_wait(fs.plan_meta, st, num_rendered, max_tile_entries, nullptr);
}

int dm2_layers_run(const dm2_layers_desc* d, int64_t num_rendered, int64_t max_tile_entries, void* face_scratch, size_t face_bytes,
                   void* binning_scratch, size_t binning_bytes, void* image_scratch, size_t image_bytes,
                   void* tet_scratch, size_t tet_bytes,
                   int32_t* render_layers, int32_t* render_layers_cnt, void* stream) {
    if (check_layers_desc(d)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t BF = (int64_t)d->B * d->F, N = (int64_t)d->B * d->H * d->W, Tn = tiles_of(d->B, d->W, d->H);
    if (N == 0) return 0;
    if (dm2_scratch_bytes(DM2_SCRATCH_LAYER_IMAGE, N, Tn) > image_bytes) return fail("image scratch too small");
    dm2::LayerImageState ls = dm2::LayerImageState::carve(image_scratch, N, Tn);
    dm2::FaceState fs{};
    dm2::BinningState bs{};
    if (BF != 0) {
        if (dm2_scratch_bytes(DM2_SCRATCH_FACE, BF, 2 * Tn) > face_bytes) return fail("face scratch too small");
        if (dm2_scratch_bytes(DM2_SCRATCH_BINNING, num_rendered, Tn) > binning_bytes) return fail("binning scratch too small");
        fs = dm2::FaceState::carve(face_scratch, BF, Tn, dm2::scan_temp_bytes(BF), false);
        bs = dm2::BinningState::carve(binning_scratch, num_rendered, dm2::sort_temp_bytes(num_rendered, Tn));
        DM2_HIP(dm2::launch_bin_sort(d->B, d->F, d->W, d->H, num_rendered, max_tile_entries, (d->flags & DM2_FLAG_LEGACY_KERNELS) != 0,
                                     fs.min_depths, fs, bs, ls.ranges, false, nullptr, st));   // renderer.cu:603
    } else {
        DM2_HIP(hipMemsetAsync(ls.ranges, 0, (size_t)Tn * sizeof(uint2), st));
    }
    if (tet_scratch && dm2_scratch_bytes(DM2_SCRATCH_LAYER_TETS, d->T, 0) > tet_bytes) return fail("tet scratch too small");
    dm2::launch_layers(*d, fs, ls.ranges, bs.face_list, ls, tet_scratch, render_layers, render_layers_cnt, st);
    DM2_HIP(hipGetLastError());
    return 0;
}

static int check_prep_desc(const dm2_prep_desc* d) {
    if (!d) return fail("null descriptor");
    if (d->B < 0 || d->P < 0 || d->F < 0 || d->W < 0 || d->H < 0) return fail("negative size in descriptor");
    if (d->B > 65535) return fail("more than 65535 views per call");
    if (d->P > 0 && d->B > 0 && (!d->verts || !d->mv || !d->proj)) return fail("verts, mv and proj must not be null");
    if (d->F > 0 && !d->faces) return fail("faces must not be null");
    return 0;
}

int dm2_prepare_faces(const dm2_prep_desc* d, void* stream) {
    if (check_prep_desc(d)) return 1;
    const bool tables = d->aa_face_verts || d->aa_face_edges || d->aa_face_edges_iszero || d->aa_face_edges_recip ||
                        d->aa_face_edges_normal || d->aa_face_edges_normal_c;
    if (tables && d->F > 0 && d->B > 0 && !d->verts_image) return fail("the AA tables are built from verts_image: it must not be null");
    dm2::launch_prepare_faces(*d, (hipStream_t)stream);
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_prepare_faces_backward(const dm2_prep_desc* d, const float* g_verts_ndc, const float* g_verts_image,
                               const float* g_aa_face_verts, float* image_grad_scratch, float* g_verts, void* stream) {
    if (check_prep_desc(d)) return 1;
    if (d->P > 0 && !g_verts) return fail("g_verts must not be null");
    if (g_aa_face_verts && d->F > 0 && d->B > 0 && d->P > 0 && !image_grad_scratch) return fail("image_grad_scratch must not be null");
    dm2::launch_prepare_faces_backward(*d, g_verts_ndc, g_verts_image, g_aa_face_verts, image_grad_scratch, g_verts, (hipStream_t)stream);
    DM2_HIP(hipGetLastError());
    return 0;
}

static int check_exchange(int32_t B, int32_t P, int32_t F, int32_t N) {
    if (B < 0 || P < 0 || F < 0) return fail("negative size");
    if (N < 1 || N > dm2::exchange_max_ranks()) return fail("dm2_exchange_*: 1 <= N <= 64 ranks");
    return 0;
}

int dm2_exchange_mark(int32_t B, int32_t P, int32_t F, int32_t N, const int32_t* faces, const void* face_scratch, size_t face_bytes,
                      uint8_t* flags, uint32_t* counts, void* stream) {
    if (check_exchange(B, P, F, N)) return 1;
    if (!flags || !counts) return fail("dm2_exchange_mark: null output");
    const int64_t BF = (int64_t)B * F;
    const uint32_t* touched = nullptr;
    if (BF > 0 && P > 0) {
        if (!faces || !face_scratch) return fail("dm2_exchange_mark: null input");
        dm2::FaceState fs = dm2::FaceState::carve(const_cast<void*>(face_scratch), BF, 0, dm2::scan_temp_bytes(BF), false);
        touched = fs.tiles_touched;
        if ((size_t)((const char*)(touched + BF) - (const char*)face_scratch) > face_bytes) return fail("dm2_exchange_mark: face scratch too small");
    }
    DM2_HIP(dm2::launch_exchange_mark(B, P, F, N, faces, touched, flags, counts, (hipStream_t)stream));
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_exchange_pack(int32_t B, int32_t P, int32_t F, int32_t N, const uint8_t* flags, const uint32_t* counts, uint32_t* cursors,
                      const float* dverts, const float* dverts_color, const float* dfaces_opacity, const float* dfaces_intense,
                      float* send, void* stream) {
    if (check_exchange(B, P, F, N)) return 1;
    if (!flags || !counts || !cursors) return fail("dm2_exchange_pack: null scratch");
    if (P > 0 && F > 0 && (!dverts || !dverts_color || !dfaces_opacity || !dfaces_intense)) return fail("dm2_exchange_pack: null gradient");
    DM2_HIP(dm2::launch_exchange_pack(B, P, F, N, flags, counts, cursors, dverts, dverts_color, dfaces_opacity, dfaces_intense, send, (hipStream_t)stream));
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_exchange_unpack(int32_t B, int32_t P, int32_t F, int32_t N, int32_t rank, const float* recv, const uint32_t* recv_counts,
                        int64_t rows, float* slice_v, float* slice_f, void* stream) {
    if (check_exchange(B, P, F, N)) return 1;
    if (rank < 0 || rank >= N) return fail("dm2_exchange_unpack: rank out of range");
    if (!slice_v || !slice_f || !recv_counts || (rows > 0 && !recv)) return fail("dm2_exchange_unpack: null pointer");
    DM2_HIP(dm2::launch_exchange_unpack(B, P, F, N, rank, recv, recv_counts, rows, slice_v, slice_f, (hipStream_t)stream));
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_debug_aa_overlap(int variant, int64_t n, const float* aa_face_verts, const float* aa_face_edges,
                         const uint8_t* aa_face_edges_iszero, const float* aa_face_edges_recip, const float* aa_face_edges_normal,
                         const float* aa_face_edges_normal_c, const float* pixmin, float* area, float* grad, int32_t* code,
                         void* stream) {
    if (variant < 0 || variant > 4) return fail("dm2_debug_aa_overlap: unknown variant");
    if (n < 0) return fail("dm2_debug_aa_overlap: negative count");
    if (n > 0 && (!aa_face_verts || !aa_face_edges || !aa_face_edges_iszero || !aa_face_edges_recip || !aa_face_edges_normal ||
                  !aa_face_edges_normal_c || !pixmin || !area || !grad || !code))
        return fail("dm2_debug_aa_overlap: null pointer");
    dm2::launch_debug_aa_overlap(variant, n, aa_face_verts, aa_face_edges, aa_face_edges_iszero, aa_face_edges_recip,
                                 aa_face_edges_normal, aa_face_edges_normal_c, pixmin, area, grad, code, (hipStream_t)stream);
    DM2_HIP(hipGetLastError());
    return 0;
}

int dm2_debug_fetch(int what, int64_t count, int64_t aux, int64_t num_rendered, const void* scratch, size_t scratch_bytes,
                    void* dst, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    void* base = const_cast<void*>(scratch);
    const void* src = nullptr; size_t bytes = 0;
    switch (what) {
        case 0: { auto s = dm2::ImageState::carve(base, count, aux); src = s.ranges; bytes = (size_t)aux * 8; break; }
        case 1: { auto s = dm2::BinningState::carve(base, num_rendered, dm2::sort_temp_bytes(num_rendered, aux)); src = s.face_list; bytes = (size_t)num_rendered * 4; break; }
        case 2: { auto s = dm2::ImageState::carve(base, count, aux); src = s.final_T; bytes = (size_t)count * 4; break; }
        case 3: { auto s = dm2::ImageState::carve(base, count, aux); src = s.final_prev_T; bytes = (size_t)count * 4; break; }
        case 4: { auto s = dm2::ImageState::carve(base, count, aux); src = s.n_contrib; bytes = (size_t)count * 4; break; }
        case 5: { auto s = dm2::LayerImageState::carve(base, count, aux); src = s.first_face; bytes = (size_t)count * 4; break; }
        case 6: { auto s = dm2::LayerImageState::carve(base, count, aux); src = s.first_tet; bytes = (size_t)count * 4; break; }
        case 7: { auto s = dm2::LayerImageState::carve(base, count, aux); src = s.ranges; bytes = (size_t)aux * 8; break; }
        case 8: { auto s = dm2::FaceState::carve(base, count, 0, dm2::scan_temp_bytes(count), false); src = s.tiles_touched; bytes = (size_t)count * 4; break; }
        default: return fail("dm2_debug_fetch: unknown item");
    }
    if (!scratch && bytes) return fail("dm2_debug_fetch: null scratch");
    if (bytes && (size_t)((const char*)src - (const char*)scratch) + bytes > scratch_bytes) return fail("dm2_debug_fetch: item lies beyond scratch_bytes");
    if (bytes) DM2_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // extern "C"
