// dm2_forward_queue.hip -- forward composite, dense pairs with survivor compaction.
//
// Same results as k_render_forward (dm2_forward.hip; FORWARD::renderCUDA<3>,
// forward.cu:139-432).  Work distribution (see dm2_pairs.h for the pair enumeration):
//
//   A   stage a chunk of faces, exact pixel rectangle per face, block scan -> pair index k
//   B1  one pair per lane: the corner / half-plane classification only (aa.h:103-149).
//       About a third of the rectangle pairs are rejected here; the survivors are
//       compacted, in pair order, into an LDS queue (wave ballot + prefix count, every
//       wave owns a contiguous quarter of the chunk's pairs so the order is global).
//   B2  one SURVIVOR per lane: polygon clip, Moeller-Trumbore, barycentric clamp, coverage
//       mix, interpolated colour / depth -> pair record in LDS.  All lanes carry work.
//   C   every pixel blends its own records in list order (64-bit face mask per pixel).
//
// The chunk is cut twice at a face boundary: to PAIRCAP pairs before B1 and to SURVCAP
// survivors after it; faces beyond a cut are staged again by the next chunk.
#include <hip/hip_runtime.h>

#include "dm2_clip_area.h"
#include "dm2_device_math.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_FQ_CHUNK
#define DM2_FQ_CHUNK 52       // staged faces per chunk (LDS-direct: 4 records per wave instruction)
#endif
#ifndef DM2_FQ_BLOCKS
#define DM2_FQ_BLOCKS 4       // resident blocks per CU the register / LDS budget is set for.  A/B at cfg4 on MI355X with a double-
                              // buffered prefetch of 32-face chunks: 3 blocks (512 records, 42.7 KB) 1.01 ms; 4 blocks (416 records,
                              // 40.4 KB) 0.83 ms; 4 blocks / 384 records / 640 pairs 0.86; 4 blocks / 28 faces 0.87 -- the
                              // chunk's fixed costs (five barriers, partly filled rounds) outweigh the longer prefetch distance
#endif
#ifndef DM2_FQ_PAIRCAP
#define DM2_FQ_PAIRCAP 768
#endif
#ifndef DM2_FQ_SURVCAP
#define DM2_FQ_SURVCAP 512
#endif
constexpr int FQ_CHUNK = DM2_FQ_CHUNK;
constexpr int FQ_PAIRCAP = DM2_FQ_PAIRCAP;
constexpr int FQ_SURVCAP = DM2_FQ_SURVCAP;
#ifndef DM2_FQ_TAILMIN
#define DM2_FQ_TAILMIN 0      // a last round with fewer survivors than this is cut off and its faces staged again
#endif
constexpr int FQ_TAILMIN = DM2_FQ_TAILMIN;
constexpr int FQ_QCAP = ((FQ_PAIRCAP + 3) / 4 + 63) & ~63;     // queue region of one wave
static_assert(FQ_CHUNK <= 64, "one mask bit per staged face; the id window is one wave wide");
constexpr int FQ_REC_CHUNKS = (int)(sizeof(FaceRec) / 16);
static_assert(FQ_PAIRCAP >= TILE_PIX && FQ_SURVCAP >= TILE_PIX, "a single face may own 256 pairs");
static_assert(FQ_PAIRCAP < 65536, "16-bit slots");
static_assert(FQ_SURVCAP == 2 * TILE_PIX && FQ_SURVCAP % 64 == 0 && FQ_QCAP <= 256, "one blend bit per survivor record; 8 bits of record index per queue entry");

constexpr uint32_t QF_REC = 1u;      // AA overlap found (the reference takes an AA record here)
constexpr uint32_t QF_BLEND = 2u;    // the face blends into the pixel

struct __attribute__((aligned(8))) FqPair { float alpha, c0, c1, c2, depth; uint32_t flags; };

// CLASSES: phase B2 takes the survivors class by class -- the ones that need the polygon clip first, the fully covered ones
// (aa.h:493-496: area = pixel area) behind them -- so that whole waves skip the clipper.  Pays with triangles of many pixels
// (256 x 256 / 2 k faces: forward -14 %, with depth complexity 60: -17 %), costs 4 % with the 2.5-pixel triangles of the
// 1080p / 1 M workload (a fifth of whose survivors are fully covered: one wave in eight): the launcher picks by the plan's
// candidate pairs per list entry.
template <bool CLASSES>
__global__ void __launch_bounds__(TILE_PIX, DM2_FQ_BLOCKS)
k_render_forward_queue(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, float* __restrict__ out_color, float* __restrict__ out_depth,
                       int32_t* __restrict__ out_tri_cnt, uint64_t* __restrict__ hit_masks,
                       uint32_t* __restrict__ hit_valid, float* __restrict__ pool, uint32_t pool_cap,
                       uint32_t* __restrict__ hit_base STAMP_PARAM) {
    __shared__ FaceRec recs[FQ_CHUNK];                   // this chunk's faces; refilled (LDS-direct) behind phase B2, its last reader
    __shared__ uint32_t s_ids[64];                       // face ids of the NEXT chunk's list entries
    __shared__ FqPair s_pair[FQ_SURVCAP];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ int s_off[FQ_CHUNK + 1];
    __shared__ uint32_t s_rect[FQ_CHUNK];
    __shared__ int s_kb[FQ_CHUNK];                       // pair index of the face's (virtual) tile pixel (0,0): off - y0*w - x0
    __shared__ int s_wtot[4];                            // survivors per wave
    __shared__ int s_inv[17];
    __shared__ uint16_t s_slot[FQ_PAIRCAP];              // per pair: survivors before it within its wave's range
    __shared__ uint32_t s_queue[4 * FQ_QCAP];            // survivors: q | face << 8 | corner mask << 14
    __shared__ unsigned long long s_mask[TILE_PIX];      // per pixel: faces of the chunk that left a record for it
    __shared__ __attribute__((aligned(16))) unsigned long long s_bmask[FQ_CHUNK * 4]; // per (face, wave of the tile): the pixels the face blends into
    __shared__ int s_wtotc[4];                           // survivors per wave that need the polygon clip (the others are fully covered)
    __shared__ __attribute__((aligned(16))) uint32_t s_rcnt[8];  // (without classes) [round][wave]: blending survivors of that wave's lanes in that B2 round
    __shared__ unsigned long long s_blend[FQ_SURVCAP / 64];  // one bit per survivor record: it blends
    __shared__ int s_bpre[4][FQ_SURVCAP / 64];           // [wave]: set bits of s_blend in front of each word (every wave scans for itself)
    __shared__ uint32_t s_cbase;                         // pair pool: first slot of this chunk
    __shared__ uint32_t s_maxlc;

    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    uint32_t tile;
    if (!tile_of_block(gx * gy * (uint32_t)d.B, is.tile_order, tile)) return;    // XCD-contiguous tile order (dm2_pairs.h)
    const int b = (int)(tile / (gx * gy));
    const uint32_t tyx = tile - (uint32_t)b * gx * gy;
    const int tile_y = (int)(tyx / gx), tile_x = (int)(tyx - (uint32_t)tile_y * gx);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    STAMP_DECL
    fill_inv_table(s_inv);
    s_mask[tid] = 0;
    if (hit_valid && blockIdx.x == 0 && tid == 0) hit_valid[0] = pool ? 3u : 2u;   // AA blend masks (+ the pair pool) are current
    const int lx = tid & 15, ly = tid >> 4;
    const int X0 = tile_x * TILE, Y0 = tile_y * TILE;
    const uint32_t px = X0 + lx, py = Y0 + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const int X0a = X0 + (int)pmx, Y0a = Y0 + (int)pmy;
    const int xlim = min(TILE - 1, d.W - 1 - X0), ylim = min(TILE - 1, d.H - 1 - Y0);

    if (inside) {
        f3 ro, rd;
        pixel_ray(d, b, pix, px + pmx, py + pmy, d.full_W, d.full_H, ro, rd);
        s_ray[tid * 6] = ro.x; s_ray[tid * 6 + 1] = ro.y; s_ray[tid * 6 + 2] = ro.z;
        s_ray[tid * 6 + 3] = rd.x; s_ray[tid * 6 + 4] = rd.y; s_ray[tid * 6 + 5] = rd.z;
    }
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const uint4* const grecs = is.face_recs + (int64_t)b * d.F * FACE_REC_U4;
    // Memory pipeline (see dm2_backward_mask.hip, dm2_stage.h): packed face records go straight from global memory into
    // LDS.  Phase B2 is the last reader of a chunk's records, so the NEXT chunk's are requested into the same array right
    // behind it (its cut is known by then) and land while the pixels blend (phase C) -- no second buffer: 4 blocks per CU
    // with 52-face chunks.  The ids a record address needs are requested one phase earlier and land during B2.
    const int rl = lane / FQ_REC_CHUNKS, rp = lane - rl * FQ_REC_CHUNKS;
    auto request_ids = [&](int nb) {
        if (wid == 2 && nb + lane < total) glds4(face_list + range.x + nb + lane, &s_ids[0]);
    };
    auto request_recs = [&](int nb) {
        const int nc2 = min(FQ_CHUNK, total - nb);
        for (int r0 = 4 * wid; r0 < nc2; r0 += 16) {                // this wave instruction's first record
            const int r = r0 + rl;
            if (rl < 4 && r < nc2) glds16(grecs + (int64_t)s_ids[r] * FACE_REC_U4 + rp, &recs[r0]);
        }
    };
    if (total > 0) {
        request_ids(0);
        lds_prefetch_wait();
        __syncthreads();
        request_recs(0);
    }
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
        lds_prefetch_wait();                                        // this wave's part of the chunk's records has landed ...
        if (__syncthreads_count(done) == TILE_PIX) break;          // ... everyone's; forward.cu:258-260; fences LDS reuse
        STAMP(1)
        // ---- phase A ----------------------------------------------------------------------
        n = min(FQ_CHUNK, total - base);
        const bool last_chunk = base + n >= total;
        int cnt = 0;
        if (tid < n) {
            uint32_t rect;
            cnt = face_pixel_rect(recs[tid].aa.bb, use_aa, X0a, Y0a, xlim, ylim, rect);
            s_rect[tid] = rect;
        }
        if (tid < FQ_CHUNK * 4) s_bmask[tid] = 0;
        if (tid < FQ_SURVCAP / 64) s_blend[tid] = 0;
        STAMP(2)
        if (wid == 0) {                                             // a chunk is at most 64 faces: all staging lanes are in wave 0
            const int inc = wave_inclusive_scan(cnt);
            if (tid < n) {
                const int ex = inc - cnt;
                s_off[tid] = ex;
                const uint32_t r = s_rect[tid];
                s_kb[tid] = ex - (int)((r >> 4) & 15u) * ((int)((r >> 8) & 15u) + 1) - (int)(r & 15u);
            }
            if (lane == 63) s_off[n] = inc;                         // lanes >= n count 0: lane 63 holds the total
        }
        __syncthreads();
        int tot = s_off[n];
        if (tot > FQ_PAIRCAP) {                                    // cut 1: faces [0, n) hold at most PAIRCAP pairs
            n = find_face(s_off, n, FQ_PAIRCAP);                    // >= 1: a face owns at most 256 pairs
            tot = s_off[n];
        }
        STAMP(3)

        // ---- phase B1: classify, compact survivors in pair order ---------------------------
        const int Q = (((tot + 3) >> 2) + 63) & ~63;               // pairs per wave, whole rounds of 64
        // Survivors are compacted in pair order (their record index) but QUEUED by class: the ones that need the polygon clip
        // from the front of the wave's queue region, the fully covered ones (all four pixel corners inside the three half
        // planes, aa.h:493-496: area = pixel area, no clip) from its back -- phase B2 then runs whole waves of one class and the
        // waves of the second class skip the clipper altogether.  An entry carries its record index within the wave.
        int wcount = 0, wfull = 0;
        for (int r = 0; r < Q; r += 64) {
            const int k = wid * Q + r + lane;
            bool surv = false, full = false;
            uint32_t entry = 0;
            if (k < tot) {
                const int j = find_face(s_off, n, k);
                int qx, qy;
                pair_xy(s_rect[j], k - s_off[j], s_inv, qx, qy);
                uint32_t cmask = 0xF;
                surv = true;
                if (use_aa) {
                    const float pxmin = (float)(uint32_t)(X0a + qx), pymin = (float)(uint32_t)(Y0a + qy);
                    // (the rectangle already is the exact set of pixels that pass the bbox test, aa.h:96-101)
                    surv = classify_pixel(recs[j].aa, pxmin, pxmin + 1, pymin, pymin + 1, cmask);
                }
                full = surv && (cmask == 0xFu);
                entry = (uint32_t)(qy * TILE + qx) | ((uint32_t)j << 8) | (cmask << 14);
            }
            const unsigned long long bal = __ballot(surv);
            const int before = wcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (k < tot) s_slot[k] = (uint16_t)before;
            if (CLASSES) {
                const unsigned long long balf = __ballot(full), balc = bal & ~balf;
                const int rf = wfull + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(balf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)balf, 0u));
                const int rc = (wcount - wfull) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(balc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)balc, 0u));
                if (surv) s_queue[wid * FQ_QCAP + (full ? FQ_QCAP - 1 - rf : rc)] = entry | ((uint32_t)before << 18);
                wfull += __popcll(balf);
            } else if (surv) s_queue[wid * FQ_QCAP + before] = entry;
            wcount += __popcll(bal);
        }
        if (lane == 0) { s_wtot[wid] = wcount; if (CLASSES) s_wtotc[wid] = wcount - wfull; }
        __syncthreads();
        STAMP(4)
        const int wb1 = s_wtot[0], wb2 = wb1 + s_wtot[1], wb3 = wb2 + s_wtot[2];
        int S = wb3 + s_wtot[3];
        const int ST0 = S;                                          // queue entries of the chunk (cut 2 below may retire the last faces' records)
        int wc1 = 0, wc2 = 0, wc3 = 0, SC = 0;                      // clip class: prefix over the waves
        if (CLASSES) { wc1 = s_wtotc[0]; wc2 = wc1 + s_wtotc[1]; wc3 = wc2 + s_wtotc[2]; SC = wc3 + s_wtotc[3]; }
        const int wf1 = wb1 - wc1, wf2 = wb2 - wc2, wf3 = wb3 - wc3;                                          // fully covered class
        // global survivor prefix at pair k (k <= tot)
        auto surv_before = [&](int k) -> int {
            if (k >= tot) return S;
            const int w = (k >= Q) + (k >= 2 * Q) + (k >= 3 * Q);
            return (w == 0 ? 0 : (w == 1 ? wb1 : (w == 2 ? wb2 : wb3))) + (int)s_slot[k];
        };
        // cut 2: at most SURVCAP records fit; and a nearly empty last round of 256 lanes is not worth running --
        // its faces go to the next chunk (unless this is the tile's last one)
        int target = S;
        if (S > FQ_SURVCAP) target = FQ_SURVCAP;
        else if (FQ_TAILMIN > 0 && !last_chunk && S > TILE_PIX && (S & (TILE_PIX - 1)) != 0 && (S & (TILE_PIX - 1)) < FQ_TAILMIN) target = S & ~(TILE_PIX - 1);
        if (target < S) {                                           // largest face prefix with <= target survivors
            int lo = 1, hi = n;                                     // surv_before(off[1]) <= 256 <= target holds, surv_before(off[n]) = S does not
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (surv_before(s_off[mid]) <= target) lo = mid; else hi = mid;
            }
            const int S2 = surv_before(s_off[lo]);
            n = lo; tot = s_off[lo]; S = S2;
        }

        if (base + n < total) request_ids(base + n);               // the next chunk's ids land during B2
        // pair pool: the chunk takes one slot per survivor (>= the pairs that blend) -- one atomic per chunk, issued here and
        // looked at behind B2
        uint32_t cbase = 0;
#if defined(DM2_FQ_POOL_EXP) && DM2_FQ_POOL_EXP == 1     // timing experiment only: no allocation
        if (pool && tid == 0) cbase = (tile * 4096u + (uint32_t)base * 16u) % (pool_cap - 600u);
#else
        if (pool && tid == 0) cbase = atomicAdd(hit_valid + 1, (uint32_t)S);
#endif
        // ---- phase B2: one survivor per lane ------------------------------------------------
        // Pool: a blending survivor's slot is the chunk's first slot + its rank among the chunk's blending survivors -- record
        // order is (entry, pixel) order, the order of the masks.  Without classes the lanes run in record order: a ballot per
        // wave and round, completed behind the barrier; with classes: one bit per record, summed behind the barrier.
        constexpr int FQ_ROUNDS = CLASSES ? (FQ_PAIRCAP + TILE_PIX - 1) / TILE_PIX : FQ_SURVCAP / TILE_PIX;   // (queue entries <= pairs of the chunk)
        const int ST = CLASSES ? ST0 : S;
        float pool_ratio[FQ_ROUNDS];
        int pool_s[FQ_ROUNDS];                                      // CLASSES: the record; else its rank among the wave's blending lanes of the round
#pragma unroll
        for (int rnd = 0; rnd < FQ_ROUNDS; rnd++) { pool_ratio[rnd] = 0.f; pool_s[rnd] = -1; }
        if (!CLASSES && pool && lane == 0) s_rcnt[4 + wid] = 0;    // (a second round that does not run; the barrier behind B1 separates this from the last chunk's readers)
#pragma unroll
        for (int rnd = 0; rnd < FQ_ROUNDS; rnd++) {
            if (rnd * TILE_PIX >= ST) break;                        // (block-uniform)
            const int t = tid + rnd * TILE_PIX;
            uint32_t entry = 0u;
            int s = t;
            if (CLASSES) {
                // queue position t: first the clip class of all four waves, then the fully covered class
                const bool isc = t < SC;
                const int tc = isc ? t : t - SC;
                const int b1 = isc ? wc1 : wf1, b2 = isc ? wc2 : wf2, b3 = isc ? wc3 : wf3;
                const int w = (tc >= b1) + (tc >= b2) + (tc >= b3);
                const int within = tc - (w == 0 ? 0 : (w == 1 ? b1 : (w == 2 ? b2 : b3)));
                if (t < ST) entry = s_queue[w * FQ_QCAP + (isc ? within : FQ_QCAP - 1 - within)];
                s = (w == 0 ? 0 : (w == 1 ? wb1 : (w == 2 ? wb2 : wb3))) + (int)((entry >> 18) & 255u);      // the survivor's record
            } else if (t < ST) {
                const int w = (s >= wb1) + (s >= wb2) + (s >= wb3);
                entry = s_queue[w * FQ_QCAP + (s - (w == 0 ? 0 : (w == 1 ? wb1 : (w == 2 ? wb2 : wb3))))];
            }
            bool blend_s = false;
            if (t < ST && s < S) {
            const int q = (int)(entry & 255u), j = (int)((entry >> 8) & 63u);
            const uint32_t cmask = (entry >> 14) & 15u;
            const FaceRec& fc = recs[j];
            const float pxmin = (float)(uint32_t)(X0a + (q & 15)), pxmax = pxmin + 1;
            const float pymin = (float)(uint32_t)(Y0a + (q >> 4)), pymax = pymin + 1;
            FqPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0;
            float oarea = 0.f;
            bool live = true;
            if (use_aa) {
                const int err = clip_area_classified(fc.aa, pxmin, pxmax, pymin, pymax, cmask, pix_area, oarea);
                live = !((err != 0) || (oarea == 0.0f));
                if (live) out.flags |= QF_REC;
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
                        out.flags |= QF_BLEND;
                        pool_ratio[rnd] = ratio; blend_s = true;
                        if (CLASSES) { pool_s[rnd] = s; if (pool) atomicOr(&s_blend[s >> 6], 1ull << (s & 63)); }
                    }
                }
            }
            if (out.flags) {
                s_pair[s] = out;
                atomicOr(&s_mask[q], 1ull << j);
                if (hit_masks && (out.flags & QF_BLEND)) atomicOr(&s_bmask[j * 4 + (q >> 6)], 1ull << (q & 63));
            }
            }
            if (!CLASSES && pool) {                                     // (block-uniform; every lane of the wave is here)
                const unsigned long long bal = __ballot(blend_s);
                pool_s[rnd] = blend_s ? (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) : -1;
                if (lane == 0) s_rcnt[rnd * 4 + wid] = __popcll(bal);
            }
        }
        STAMP(5)
        if (pool && tid == 0) s_cbase = cbase;
        lds_prefetch_wait();
        __syncthreads();
        if (base + n < total) request_recs(base + n);              // B2 was the records' last reader: refill behind it
        // what the backward needs to find its work without re-classifying (dm2_backward_mask.hip): per list entry and
        // wave of the tile's block, the pixels the entry blends into
        if (hit_masks && tid < n * 4) hit_masks[((int64_t)range.x + base + (tid >> 2)) * 4 + (tid & 3)] = s_bmask[tid];
        // ... and (dm2_backward_fast.hip) the coverage of every such pair, so that the backward neither clips for an area nor
        // depends on reproducing it: pool slots in mask order -- entry by entry, wave by wave, pixel by pixel -- behind the
        // chunk's first slot; per entry the slot of its first pair.
        if (pool) {
            const uint32_t cb = s_cbase;
            if (wid == 0) {
                int cj = 0;
                if (lane < n) {
                    const ulonglong2 m01 = reinterpret_cast<const ulonglong2*>(s_bmask)[2 * lane], m23 = reinterpret_cast<const ulonglong2*>(s_bmask)[2 * lane + 1];
                    cj = __popcll(m01.x) + __popcll(m01.y) + __popcll(m23.x) + __popcll(m23.y);
                }
                const int ex = wave_inclusive_scan(cj) - cj;
                if (lane < n) hit_base[(int64_t)range.x + base + lane] = cb + (uint32_t)ex;
            }
            if (!CLASSES) {
                const uint4 c03 = reinterpret_cast<const uint4*>(s_rcnt)[0], c47 = reinterpret_cast<const uint4*>(s_rcnt)[1];
                const uint32_t r0 = (wid > 0 ? c03.x : 0u) + (wid > 1 ? c03.y : 0u) + (wid > 2 ? c03.z : 0u);
                const uint32_t r1 = c03.x + c03.y + c03.z + c03.w + (wid > 0 ? c47.x : 0u) + (wid > 1 ? c47.y : 0u) + (wid > 2 ? c47.z : 0u);
                if (pool_s[0] >= 0) { const uint32_t slot = cb + r0 + (uint32_t)pool_s[0]; if (slot < pool_cap) pool[slot] = pool_ratio[0]; }
                if (FQ_ROUNDS > 1 && pool_s[FQ_ROUNDS > 1 ? 1 : 0] >= 0) { const uint32_t slot = cb + r1 + (uint32_t)pool_s[1]; if (slot < pool_cap) pool[slot] = pool_ratio[1]; }
            } else {
            // a blending survivor's slot: the chunk's first slot + the blending records in front of its own (record order is
            // (entry, pixel) order, the order of the masks): the bits of s_blend below it.  Every wave sums the words for itself.
            {
                const int c = lane < FQ_SURVCAP / 64 ? __popcll(s_blend[lane < FQ_SURVCAP / 64 ? lane : 0]) : 0;
                const int ex = wave_inclusive_scan(c) - c;
                if (lane < FQ_SURVCAP / 64) s_bpre[wid][lane] = ex;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (the lanes of the wave talk through s_bpre with no barrier in between)
            }
#pragma unroll
            for (int rnd = 0; rnd < FQ_ROUNDS; rnd++) {
                const int s = pool_s[rnd];
                if (s >= 0) {
                    const uint32_t slot = cb + (uint32_t)(s_bpre[wid][s >> 6] + __popcll(s_blend[s >> 6] & ((1ull << (s & 63)) - 1ull)));
                    if (slot < pool_cap) pool[slot] = pool_ratio[rnd];
                }
            }
            }
        }

        // ---- phase C: ordered blend of this pixel's records ---------------------------------
        {
            unsigned long long m = s_mask[tid];
            s_mask[tid] = 0;
            while (m && !done) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                // a mask bit is only set by a record for this pixel, so its pair index follows from the row pitch
                const int k = s_kb[j] + ly * ((int)((s_rect[j] >> 8) & 15u) + 1) + lx;
                const FqPair pr = s_pair[surv_before(k)];
                if ((pr.flags & QF_REC) && rec_cnt < K) rec_cnt++;           // forward.cu:344-352
                if (!(pr.flags & QF_BLEND)) continue;
                const float alpha = pr.alpha;
                const float test_T = T * (1 - alpha);
                C0 += pr.c0 * alpha * T; C1 += pr.c1 * alpha * T; C2 += pr.c2 * alpha * T;
                D += pr.depth * alpha * T;
                pT = T; T = test_T;
                last_contributor = (uint32_t)(base + j + 1);
                if (T < T_EPS) done = true;
            }
        }
        STAMP(6)
    }

    // the tile's largest n_contrib: where the backward's walk starts (dm2_backward_fast.hip)
    if (tid == 0) s_maxlc = 0;
    __syncthreads();
    { const uint32_t m = wave_inclusive_max(last_contributor); if (lane == 63 && m) atomicMax(&s_maxlc, m); }
    __syncthreads();
    if (tid == 0) is.tile_max_lc[tile] = s_maxlc;
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

// classes: take the survivors class by class (triangles of many pixels; the caller decides from the plan's numbers)
void launch_render_forward_queue(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, uint64_t* hit_masks,
                                 uint32_t* hit_valid, float* pool, int64_t pool_cap, uint32_t* hit_base, bool classes, hipStream_t st) {
    const uint32_t Tn = (uint32_t)(((d.W + TILE - 1) / TILE) * ((d.H + TILE - 1) / TILE) * d.B);
    StageTimer tm(ST_FWD, st);
    if (classes)
        hipLaunchKernelGGL(k_render_forward_queue<true>, dim3(tile_grid_blocks(Tn)), dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth,
                           out_tri_cnt, hit_masks, hit_valid, pool, (uint32_t)pool_cap, hit_base STAMP_ARG(0));
    else
        hipLaunchKernelGGL(k_render_forward_queue<false>, dim3(tile_grid_blocks(Tn)), dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth,
                           out_tri_cnt, hit_masks, hit_valid, pool, (uint32_t)pool_cap, hit_base STAMP_ARG(0));
}

}  // namespace dm2
