// dm2_backward_fast.hip -- backward composite driven by the forward's blend masks AND its pair pool (the default).
//
// Same results as k_render_backward (dm2_backward.hip; BACKWARD::renderCUDA<3>, backward.cu:17-532) up to fp32
// summation order of the scattered gradients.  dm2_forward_queue.hip leaves, per list entry and wave of the tile's
// block, the 64-bit mask of the pixels the entry blends into, and per blended pair its coverage ratio
// (forward.cu:375-378) in a pool ordered like the masks.  With those the backward
//   * neither enumerates nor classifies (pixel,face) pairs and never meets a clipper error path (the masks),
//   * replays the forward's alpha TO THE BIT without clipping for an area (the pool): nothing depends on reproducing
//     an area, nearly opaque faces included,
//   * owes the clipper only d(area)/d(corners), which needs no polygon (dm2_clip_fast.h: per triangle edge, the piece
//     inside the pixel; ~350 instead of ~950 VALU instructions per pair, no 160-register working set).  Pairs whose
//     reference polygon is not the geometric intersection (ties, 1-5 % at 1080p) are queued with their dL/d(area) and
//     get the exact segment clipper in k_aa_ties behind this kernel.
//
//   per chunk (walked back to front): the masks of the next <= 32 entries, scan of their hit counts (every wave for
//   itself, DPP), keep the leading entries whose hits fit 256 lanes (a ballot), pair lane -> (entry, pixel) through
//   start marks and a running maximum, then
//   B2  lane s: its (face, pixel) from the masks, its coverage from the pool; Moeller-Trumbore, clamp, alpha,
//       interpolated colour / depth -> record in LDS
//   C   pixel p: replay its records back to front (backward.cu:340-405)
//   D   lane s: chain rule (backward.cu:408-488) incl. the AA Jacobian, DPP pre-reduction over the lanes of one face,
//       ds_add_f32
//   flush with (entry,component) global atomics.
//
// Memory pipeline as in dm2_backward_mask.hip (the exact-clipper variant kept for forwards without a pool): a chunk's
// inputs -- face ids, blend masks, pool offsets, packed face records -- are requested ONE CHUNK AHEAD with LDS-direct
// loads into the other half of double-buffered LDS arrays.
#include <hip/hip_runtime.h>

#include "dm2_bwd_shared.h"
#include "dm2_clip_fast.h"
#include "dm2_clip_seg.h"
#include "dm2_device_math.h"
#include "dm2_dpp.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_BF_CAND
#define DM2_BF_CAND 30        // 30 candidates + 127 VGPRs: four blocks per CU (40.5 KB of LDS each).  A/B at cfg4: 32 candidates / 3 blocks 1.17 ms, 30 / 4 blocks 1.05 ms, 28 / 4 blocks 1.05 ms
#endif
constexpr int BM_CAND = DM2_BF_CAND;      // candidate entries per chunk
static_assert(BM_CAND * 4 <= TILE_PIX && 2 * BM_CAND <= 64, "one scan thread per (face, wave); one id window per wave");
constexpr int BM_SLOTS = BM_CAND * 4;
constexpr int REC_CHUNKS = FACE_RECB_PARTS;               // 12 of the 16 sixteen-byte parts of the global record (dm2_stage.h: FaceRecB)
constexpr int REC_PER_INSTR = 64 / REC_CHUNKS;            // records one wave instruction copies
static_assert(8 * REC_PER_INSTR >= BM_CAND, "two LDS-direct instructions per wave fetch a chunk");
#ifndef DM2_BF_BLOCKS
#define DM2_BF_BLOCKS 4       // resident blocks per CU the register budget is set for (no spills at 127 VGPRs)
#endif
// POINT: aa_temperature == 0 (dm2_forward_point.hip left the masks): coverage 1 for every pair of the masks, no pool, no AA terms
template <bool POINT>
__global__ void __launch_bounds__(TILE_PIX, DM2_BF_BLOCKS)
k_render_backward_fast(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                       float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                       float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                       float* __restrict__ dL_dfaces_intense, float* __restrict__ dL_daa_face_verts,
                       const uint64_t* __restrict__ hit_masks, uint32_t* __restrict__ hit_valid,
                       const uint32_t* __restrict__ hit_base, const float* __restrict__ pool,
                       TieEntry* __restrict__ tie_queue, uint32_t tie_cap, bool check_mode STAMP_PARAM) {
    if (check_mode && hit_valid[0] != (POINT ? 1u : 3u)) return;                  // (caller did not know what the forward left: not masks + pool -> another kernel runs)

    __shared__ FaceRecB recs2[2][BM_CAND];                     // [buffer]: this chunk's candidates / the next chunk's
    __shared__ float acc[BM_CAND * BM_ACC];
    __shared__ BfPair s_pair[TILE_PIX];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ __attribute__((aligned(16))) unsigned long long s_hit2[2][BM_SLOTS];   // [buffer][face][wave]: pixels of the wave the face blends into
    __shared__ uint16_t s_wbase[4][BM_CAND];                        // [wave][face]: pairs in front of slot (face, wave) -- written and read by that wave
    __shared__ uint16_t s_efirst[4][BM_CAND];                       // [wave][face]: the chunk's pairs in front of the face -- written and read by that wave
    __shared__ uint32_t s_hb2[2][BM_CAND];                     // [buffer][face]: pool slot of the entry's first blended pair
    __shared__ uint32_t s_mark[4][64];                         // [wave][pair lane]: (slot + 1) << 9 | first pair of the slot, where a slot starts
    __shared__ unsigned long long s_mask[TILE_PIX];            // per pixel: faces of the chunk with a record for it
    __shared__ uint32_t s_ids2[2][64];                         // [buffer]: face ids of the walk positions [base, base + 64): one wave-wide request
    __shared__ float s_pixc[6][TILE_PIX];                      // per pixel, read by phase C only: dL/dcolour, dL/ddepth, final T, T in front of the last contributor
    __shared__ float* s_fl_base[32];                           // flush, per component: destination of id 0 ...
    __shared__ int s_fl_sel[32];                               // ... which id of the record (face_id, vid[0..2]) | dwords per id << 2

    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    uint32_t tile;
    if (!tile_of_block(gx * gy * (uint32_t)d.B, is.tile_order, tile)) return;    // XCD-contiguous tile order (dm2_pairs.h)
    const int b = (int)(tile / (gx * gy));
    const uint32_t tyx = tile - (uint32_t)b * gx * gy;
    const int tile_y = (int)(tyx / gx), tile_x = (int)(tyx - (uint32_t)tile_y * gx);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    STAMP_DECL
    s_mask[tid] = 0;
    const int lx = tid & 15, ly = tid >> 4;
    const int X0 = tile_x * TILE, Y0 = tile_y * TILE;
    const uint32_t px = X0 + lx, py = Y0 + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const int X0a = X0 + (int)pmx, Y0a = Y0 + (int)pmy;
    const bool corrected = (d.flags & DM2_FLAG_CORRECTED_DV) != 0;

    uint2 range = ranges[tile];                                    // block-uniform: keep it in scalar registers
    range.x = __builtin_amdgcn_readfirstlane(range.x); range.y = __builtin_amdgcn_readfirstlane(range.y);
    // entries behind every pixel's last contributor are dead: the forward left the tile's largest n_contrib (no block-wide
    // maximum here, and the walk's first fetch leaves at once)
    const int total = (int)min((uint32_t)__builtin_amdgcn_readfirstlane(is.tile_max_lc[tile]), range.y - range.x);
    const uint4* const grecs = is.face_recs + (int64_t)b * d.F * FACE_REC_U4;
    // ---- the walk's position k (0 = the tile's deepest live entry) maps to list entry range.x + total - 1 - k
    // (backward.cu:171).  The next chunk's inputs go straight from global memory into the other LDS buffer
    // (global_load_lds: per-lane source address, destination = wave-uniform base + lane * size; no registers held).
    auto walk_entry = [&](int k) -> int64_t { return (int64_t)range.x + (uint32_t)(total - 1 - k); };
    const int rl = lane / REC_CHUNKS, rp = lane - rl * REC_CHUNKS;   // record copy: 5 records x 12 parts per wave instruction
    const int rsrc = recb_src_part(rp);
    // request the id window [nb, nb + 64) of the walk into s_ids2[buf]
    auto request_ids = [&](int buf, int nb) {
        if (wid == 2 && nb + lane < total)
            glds4(face_list + walk_entry(nb + lane), &s_ids2[buf][0]);
    };
    if (total > 0) request_ids(0, 0);                              // (lands while the pixels' own data is fetched)

    uint32_t last_contributor = 0;
    float T = 0.f;                                                 // starts as the T in front of the pixel's last contributor
    {
    float T_final = 0.f, prev_T_final = 0.f;
    float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f;
    if (inside) {
        f3 ro, rd;
        pixel_ray(d, b, pix, px + pmx, py + pmy, d.full_W, d.full_H, ro, rd);
        s_ray[tid * 6] = ro.x; s_ray[tid * 6 + 1] = ro.y; s_ray[tid * 6 + 2] = ro.z;
        s_ray[tid * 6 + 3] = rd.x; s_ray[tid * 6 + 4] = rd.y; s_ray[tid * 6 + 5] = rd.z;
        T_final = is.final_T[pix]; prev_T_final = is.final_prev_T[pix];
        last_contributor = is.n_contrib[pix];
        dLc0 = dL_dcolor[3 * pix]; dLc1 = dL_dcolor[3 * pix + 1]; dLc2 = dL_dcolor[3 * pix + 2];
        dLd = dL_ddepth[pix];
    }
    // phase C is their only reader: parked in LDS, not in six registers that would be live across B2 and D
    s_pixc[0][tid] = dLc0; s_pixc[1][tid] = dLc1; s_pixc[2][tid] = dLc2; s_pixc[3][tid] = dLd;
    s_pixc[4][tid] = T_final; s_pixc[5][tid] = prev_T_final;
    T = prev_T_final;
    }
    if (tid < M_N) fill_flush_table(tid, b, d.P, d.F, dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense,
                                    dL_daa_face_verts, s_fl_base, s_fl_sel, (d.flags & DM2_FLAG_AA_GRAD_TO_VERTS) != 0);

    const float temp = d.aa_temperature;                           // > 0 (the launcher dispatches on it)
    const float pix_area = 1.0f;
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];

    bool T_first_pass = true;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    // request masks + records of the chunk starting at walk position nb into buffer buf; ids[i]: face id of position nb + i
    auto request_chunk = [&](int buf, int nb, const uint32_t* ids) {
        const int nc2 = min(BM_CAND, total - nb);
        // (the masks first: should the compiler ever reload an address from scratch here, that reload waits for every
        // load issued before it -- vmcnt is in order -- and must not find this chunk's record loads in front of it)
        if (wid == 1 && (lane >> 1) < nc2)                         // lane: masks of (face lane / 2, waves 2 (lane & 1), + 1)
            glds16(hit_masks + walk_entry(nb + (lane >> 1)) * 4 + (lane & 1) * 2, &s_hit2[buf][0]);
        if (!POINT && wid == 3 && lane < nc2) glds4(hit_base + walk_entry(nb + lane), &s_hb2[buf][0]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = (i * 4 + wid) * REC_PER_INSTR;          // this wave instruction's first record
            const int r = r0 + rl;
            if (rl < REC_PER_INSTR && r < nc2) glds16(grecs + (int64_t)ids[r] * FACE_REC_U4 + rsrc, &recs2[buf][0] + r0);
        }
    };
    if (total > 0) {                                               // first chunk: synchronously
        lds_prefetch_wait();
        __syncthreads();
        request_chunk(0, 0, s_ids2[0]);
        lds_prefetch_wait();
    }
    for (int k = tid; k < BM_CAND * BM_ACC; k += TILE_PIX) acc[k] = 0.f;     // the flush re-zeroes what it consumes
    __syncthreads();                                                          // chunk 0's inputs and the zeroes are everyone's

    STAMP(0)
    int n = 0, cur = 0;
    // Three workgroup barriers per chunk: behind B2, behind C, behind D.  None between a chunk's flush and the next chunk's
    // scan + B2: the flush reads recs2[cur] and acc (and zeroes the entries it read); scan and B2 of the next chunk read
    // s_hit2 / recs2 of the OTHER buffer (landed and published before the flush), per-wave tables (s_wbase, s_mark) and
    // write s_pair / s_mask, whose last readers sit in front of the barriers behind C and D.
    for (int base = 0; base < total; base += n, cur ^= 1) {
        STAMP(1)
        const int nc = min(BM_CAND, total - base);
        // recs[j] / s_hit[j][.] = walk position base + j (requested by the previous chunk, or by the prologue)
        FaceRecB* const recs = recs2[cur];
        const unsigned long long* const s_hit = s_hit2[cur];
        const uint32_t* const s_ids = s_ids2[cur];
        const uint32_t* const s_hb = s_hb2[cur];
        STAMP(2)
        // ---- scan, cut and decode, by every wave for itself (no barrier, no LDS round trips through another wave): lane l
        // holds the hit words of slots 2l, 2l + 1 (slot = 4 face + pixel wave, i.e. face-major) and their exclusive scan
        const ulonglong2 hh = reinterpret_cast<const ulonglong2*>(s_hit)[lane];
        const unsigned long long h0 = (2 * lane < nc * 4) ? hh.x : 0ull, h1 = (2 * lane + 1 < nc * 4) ? hh.y : 0ull;
        const int c0 = __popcll(h0), c1 = __popcll(h1);
        const int inc = wave_inclusive_scan(c0 + c1);
        const int b0 = inc - c0 - c1, b1 = inc - c1;
        // keep the leading faces whose hits fit one round of 256 lanes (a face has at most 256): lane 2j + 1 holds the count
        // up to the end of face j
        n = max(1, __popcll(__ballot((lane & 1) && inc <= TILE_PIX && (lane >> 1) < nc)));
        const int S = __builtin_amdgcn_readlane(inc, 2 * n - 1);
        if ((lane & 1) == (wid >> 1) && (lane >> 1) < BM_CAND) s_wbase[wid][lane >> 1] = (uint16_t)((wid & 1) ? b1 : b0);    // phase C: slot (face, this wave)
        if (!POINT && (lane & 1) == 0 && (lane >> 1) < BM_CAND) s_efirst[wid][lane >> 1] = (uint16_t)b0;                              // B2: the face's first pair
        // pair lane -> slot: every non-empty slot that starts inside this wave's 64 pair lanes leaves a mark at its first
        // pair; a running maximum spreads it (slots and their first pairs grow together); the slot that covers the wave's
        // first lane comes from a ballot
        const int lo_pair = wid * 64;
        // (the lanes of the wave talk to each other through this row with no barrier in between: the fence below keeps the
        // compiler from forwarding this thread's own stores to its load; the hardware runs a wave's LDS operations in order)
        uint32_t* const mark_row = s_mark[wid];
        mark_row[lane] = 0u;
        { const int r0 = b0 - lo_pair, r1 = b1 - lo_pair;
          if (c0 > 0 && (uint32_t)r0 < 64u) mark_row[r0] = ((uint32_t)(2 * lane + 1) << 9) | (uint32_t)b0;
          if (c1 > 0 && (uint32_t)r1 < 64u) mark_row[r1] = ((uint32_t)(2 * lane + 2) << 9) | (uint32_t)b1; }
        uint32_t seed = 0u;
        { const unsigned long long e0 = __ballot(c0 > 0 && b0 <= lo_pair), e1 = __ballot(c1 > 0 && b1 <= lo_pair);
          if (e0) { const int l = 63 - __clzll((long long)e0); seed = ((uint32_t)(2 * l + 1) << 9) | (uint32_t)__builtin_amdgcn_readlane(b0, l); }
          if (e1) { const int l = 63 - __clzll((long long)e1); seed = max(seed, ((uint32_t)(2 * l + 2) << 9) | (uint32_t)__builtin_amdgcn_readlane(b1, l)); } }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const uint32_t mk = max(wave_inclusive_max(mark_row[lane]), seed);
        STAMP(3)

        // ---- phase B2: one blending (pixel,face) pair per lane -------------------------------
        const bool have = tid < S;
        int j = 0, q = 0;
        float i0 = 0.f, i1 = 0.f, i2 = 0.f, ratio = 0.f, alpha = 0.f;
        int code = 0;
        bool blend = false;
#if DM2_BM_CARRY
        // carried from B2 into phase D in registers (phase C in between needs few): ray, world corners, colours, NDC z
        f3 k_ro = {0, 0, 0}, k_rd = {0, 0, 0}, k_p0 = {0, 0, 0}, k_p1 = {0, 0, 0}, k_p2 = {0, 0, 0};
#if DM2_BM_CARRY > 1
        float k_col[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, k_dep[3] = {0, 0, 0}, k_int = 0.f, k_opa = 0.f;
#endif
#endif
        if (have) {
            const int lo = (int)(mk >> 9) - 1;                                    // the slot of pair tid, first pair mk & 511
            j = lo >> 2;
            q = ((lo & 3) << 6) + nth_set_bit64(s_hit[lo], tid - (int)(mk & 511u));
            // The forward blended this pair: its clip (aa.h:446-504) returned no error and a positive area, the ray met the
            // face's plane and the coverage was not 0.  None of those decisions is taken again, and the coverage itself --
            // hence alpha, to the bit -- comes from the forward's pool: the replay divides the running T by (1 - alpha)
            // (backward.cu:340-348) and the background term by it once more (backward.cu:396-401), so that one ulp of alpha is
            // ulp / (1 - alpha) of both; with the forward's own number a nearly opaque, nearly covering face is no special case.
            // Pool slot: the entry's first pair + this pair's place among the entry's pairs (the pool is in mask order).
            ratio = POINT ? 1.0f : pool[s_hb[j] + (uint32_t)(tid - (int)s_efirst[wid][j])];
            const FaceRecB& fc = recs[j];
            BfPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0;
            const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
            const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
            const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
            f3 tuv = {0, 0, 0};
#if DM2_BM_CARRY
            k_ro = ro; k_rd = rd; k_p0 = p0; k_p1 = p1; k_p2 = p2;
#if DM2_BM_CARRY > 1
#pragma unroll
            for (int c = 0; c < 9; c++) k_col[c] = fc.col[c];
            k_dep[0] = fc.dep[0]; k_dep[1] = fc.dep[1]; k_dep[2] = fc.dep[2]; k_int = fc.intense; k_opa = fc.opacity;
#endif
#endif
            if (ray_tri_intersection(ro, rd, p0, p1, p2, tuv)) {
                float iuc, ivc;
                clamp_bary_uv(tuv.y, tuv.z, iuc, ivc, code);
                i0 = 1 - iuc - ivc; i1 = iuc; i2 = ivc;
                float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                out.c0 = c0 * fc.intense; out.c1 = c1 * fc.intense; out.c2 = c2 * fc.intense;
                out.depth = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                alpha = fc.opacity * ratio;
                out.alpha = alpha;
                out.flags = MB_BLEND;
                blend = true;
            }
            s_pair[tid] = out;
            if (blend) atomicOr(&s_mask[q], 1ull << j);
        }
        STAMP(5)
        __syncthreads();
        // ---- the next chunk starts at base + n: request it now (every wave is past the previous chunk's flush, the last
        // reader of the other buffers); it has phases C and D to arrive (waited for before the flush, so that the flush's
        // atomics are never waited for)
        if (base + n < total) {
            request_chunk(cur ^ 1, base + n, s_ids + n);
            request_ids(cur ^ 1, base + n);
        }
        STAMP(6)

        // ---- phase C: per-pixel back-to-front replay ------------------------------------------
        {
            unsigned long long m = s_mask[tid];
            s_mask[tid] = 0;
            float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f, T_final = 0.f, prev_T_final = 0.f;
            if (m) {
                dLc0 = s_pixc[0][tid]; dLc1 = s_pixc[1][tid]; dLc2 = s_pixc[2][tid]; dLd = s_pixc[3][tid];
                T_final = s_pixc[4][tid]; prev_T_final = s_pixc[5][tid];
            }
            while (m) {                                                           // ascending face = back to front
                const int jj = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t e = (uint32_t)(total - 1 - base - jj);            // 0-based position in the list
                if (e >= last_contributor) continue;                              // backward.cu:219-221
                // slot of (face jj, this pixel): hits before (jj, this wave) + hits of lower pixels of this wave
                const int t = jj * 4 + wid;
                BfPair& pr = s_pair[(int)s_wbase[wid][jj] + __popcll(s_hit[t] & ((1ull << lane) - 1ull))];
                const float a = pr.alpha, iC0 = pr.c0, iC1 = pr.c1, iC2 = pr.c2, iD = pr.depth;
                // alpha == 1 exactly (backward.cu:396) is the forward's decision too: only a pixel's LAST contributor can
                // have it (T drops to 0 and the pixel is done), and then final_T is exactly 0
                const bool alpha_is_one = (a == 1.0f) || (T_first_pass && T_final == 0.0f);
                // (1 / (1 - alpha) once, to <= 1 ulp, for the running T and for the background term: the gradients owe the
                // reference 1e-5, not the bits of its two IEEE divisions)
                const float inv_1ma = rcp_refined(1.f - a);
                if (!T_first_pass) T = T * inv_1ma;                               // backward.cu:340-348
                T_first_pass = false;
                float dL_dalpha = 0.0f;
                accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = iC0;
                dL_dalpha += (iC0 - accum_rec0) * dLc0;
                accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = iC1;
                dL_dalpha += (iC1 - accum_rec1) * dLc1;
                accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = iC2;
                dL_dalpha += (iC2 - accum_rec2) * dLc2;
                accum_recd = last_alpha * last_depth + (1.f - last_alpha) * accum_recd; last_depth = iD;
                dL_dalpha += (iD - accum_recd) * dLd;
                dL_dalpha *= T;
                last_alpha = a;
                float bg_dot = 0.f;
                bg_dot += bg0 * dLc0; bg_dot += bg1 * dLc1; bg_dot += bg2 * dLc2;
                const float bd_dot = (float)(0.0 + 1.0 * (double)dLd);            // backward.cu:394
                if (alpha_is_one) {
                    dL_dalpha += (-prev_T_final) * bg_dot;
                    dL_dalpha += (-prev_T_final) * bd_dot;
                } else {
                    dL_dalpha += (-T_final * inv_1ma) * bg_dot;
                    dL_dalpha += (-T_final * inv_1ma) * bd_dot;
                }
                pr.depth = T; pr.alpha = dL_dalpha; pr.flags = MB_BLEND | MB_ACTIVE;   // (phase D: T, dL/dalpha in place of depth, alpha)
            }
        }
        STAMP(7)
        __syncthreads();
        STAMP(8)

        // ---- phase D: chain rule + per-entry accumulation ----------------------------------------
        {
            const int jkey = have ? j : -1;
            const int l16 = tid & 15;
            // NB: every DPP read must execute with all lanes enabled, hence the unconditional reads and `&`, `|`.
            const int k1 = dpp_shr_i<1>(jkey), k2 = dpp_shr_i<2>(jkey), k4 = dpp_shr_i<4>(jkey), k8 = dpp_shr_i<8>(jkey);
            const int kn = dpp_shl_i<1>(jkey);
            const bool s1 = (l16 >= 1) & (k1 == jkey);
            const bool s2 = (l16 >= 2) & (k2 == jkey);
            const bool s4 = (l16 >= 4) & (k4 == jkey);
            const bool s8 = (l16 >= 8) & (k8 == jkey);
            const float m1 = s1 ? 1.f : 0.f, m2 = s2 ? 1.f : 0.f, m4 = s4 ? 1.f : 0.f, m8 = s8 ? 1.f : 0.f;
            uint32_t pflags = 0; float pr_T = 0.f, pr_dL_dalpha = 0.f;
            if (have && blend) { const BfPair& pr = s_pair[tid]; pflags = pr.flags; pr_T = pr.depth; pr_dL_dalpha = pr.alpha; }
            const bool active = (pflags & MB_ACTIVE) != 0;
            float nact = active ? 1.f : 0.f;
            seg_scan16(nact, s1, s2, s4, s8);
            const bool emit = ((l16 == 15) | (kn != jkey)) & (jkey >= 0) & (nact > 0.f);
            float* const arow = acc + j * BM_ACC;
#if DM2_BM_CARRY < 2
            const FaceRecB& fcD = recs[j];
#endif
            float dL_diu = 0.f, dL_div = 0.f, dL_doarea = 0.f;
            ISA_MARK(D_group1)
            {   // group 1: vertex colours, NDC depth, intensity, opacity
                float g1[14];
#pragma unroll
                for (int c = 0; c < 14; c++) g1[c] = 0.f;
                if (active) {
                    const float Tq = pr_T, dL_dalpha = pr_dL_dalpha;
                    const float qc0 = s_pixc[0][q], qc1 = s_pixc[1][q], qc2 = s_pixc[2][q], qd = s_pixc[3][q];   // dL/dcolour, dL/ddepth of the pixel
#if DM2_BM_CARRY > 1
                    const float intense = k_int, opacity = k_opa;
                    const float* const colD = k_col; const float* const depD = k_dep;
#else
                    const float intense = fcD.intense, opacity = fcD.opacity;
                    const float* const colD = fcD.col; const float* const depD = fcD.dep;
#endif
                    const float dics[3] = {qc0 * alpha * Tq, qc1 * alpha * Tq, qc2 * alpha * Tq};
                    const float did = qd * alpha * Tq;
                    g1[12] = dL_dalpha * ratio;
                    const float dL_dratio = (dL_dalpha * opacity) * temp;
                    dL_doarea = dL_dratio / pix_area;
                    float dL_di0 = 0.f, dL_di1 = 0.f, dL_di2 = 0.f, dL_dfint = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dL_di0 += colD[ch] * dics[ch] * intense;
                        dL_di1 += colD[3 + ch] * dics[ch] * intense;
                        dL_di2 += colD[6 + ch] * dics[ch] * intense;
                        g1[ch] = 0.f + i0 * dics[ch] * intense;
                        g1[3 + ch] = 0.f + i1 * dics[ch] * intense;
                        g1[6 + ch] = 0.f + i2 * dics[ch] * intense;
                        dL_dfint += (i0 * colD[ch] + i1 * colD[3 + ch] + i2 * colD[6 + ch]) * dics[ch];
                    }
                    g1[13] = dL_dfint;
                    dL_di0 += depD[0] * did; dL_di1 += depD[1] * did; dL_di2 += depD[2] * did;
                    g1[9] = 0.f + i0 * did; g1[10] = 0.f + i1 * did; g1[11] = 0.f + i2 * did;
                    float diuc_diu, diuc_div, divc_diu, divc_div;
                    clamp_bary_uv_grad(code, diuc_diu, diuc_div, divc_diu, divc_div);
                    const float di0_diu = -1.f * diuc_diu + -1.f * divc_diu, di0_div = -1.f * diuc_div + -1.f * divc_div;
                    const float di1_diu = 1.f * diuc_diu + 0.f * divc_diu, di1_div = 1.f * diuc_div + 0.f * divc_div;
                    const float di2_diu = 0.f * diuc_diu + 1.f * divc_diu, di2_div = 0.f * diuc_div + 1.f * divc_div;
                    dL_diu = dL_di0 * di0_diu + dL_di1 * di1_diu + dL_di2 * di2_diu;
                    dL_div = dL_di0 * di0_div + dL_di1 * di1_div + dL_di2 * di2_div;
                }
                seg_scan16_safe(g1, m1, m2, m4, m8, s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 12; c++) atomicAdd(arow + M_DC + c, g1[c]);      // M_DC..+8 and M_DZ..+2 are contiguous
                    atomicAdd(arow + M_OP, g1[12]);
                    atomicAdd(arow + M_IN, g1[13]);
                    arow[M_FLAG] = 1.0f;
                }
            }
            ISA_MARK(D_group2)
            // group 2: AA corners.  d(area)/d(corners) without a polygon (dm2_clip_fast.h); a pair it flags as a tie adds nothing
            // here: it is queued with its dL/d(area) for the exact clipper (k_aa_ties below).  One atomic per wave with such a
            // pair, its return value looked at behind group 3.
            bool tie_push = false;
            uint32_t tie_base = 0;
            unsigned long long tie_bal = 0;
            if (!POINT) {
                float g2[6];
                bool tie = false;
                {
                    const FaceRecB& fb = recs[j];
                    const AAFaceB fa = {fb.v2, fb.e, fb.r, fb.zmask};
                    const float pxmin = (float)(uint32_t)(X0a + (q & 15)), pymin = (float)(uint32_t)(Y0a + (q >> 4));
                    fast_area_grad(fa, pxmin, pxmin + 1, pymin, pymin + 1, g2, tie);
                }
                tie_push = tie && active && (dL_doarea != 0.0f);
                const bool use = active && !tie;
#pragma unroll
                for (int c = 0; c < 6; c++) g2[c] = use ? dL_doarea * g2[c] : 0.0f;     // (a select: a tie's or an idle lane's entries need not be finite)
                tie_bal = __ballot(tie_push);
                if (tie_bal && lane == (int)(__ffsll((long long)tie_bal) - 1)) tie_base = atomicAdd(hit_valid + 2, (uint32_t)__popcll(tie_bal));
                seg_scan16_safe(g2, m1, m2, m4, m8, s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 6; c++) atomicAdd(arow + M_AA + c, g2[c]);
                }
            }
            ISA_MARK(D_group3)
            {   // group 3: world-space corners through the ray/triangle intersection
                float g3[9];
#pragma unroll
                for (int c = 0; c < 9; c++) g3[c] = 0.f;
                if (active) {
#if DM2_BM_CARRY
                    const f3 ro = k_ro, rd = k_rd, p0 = k_p0, p1 = k_p1, p2 = k_p2;
#else
                    const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                    const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                    const f3 p0 = {fcD.v[0], fcD.v[1], fcD.v[2]}, p1 = {fcD.v[3], fcD.v[4], fcD.v[5]}, p2 = {fcD.v[6], fcD.v[7], fcD.v[8]};
#endif
                    f3 du0, du1, du2, dv0, dv1, dv2;
                    ray_tri_intersection_grad<true>(ro, rd, p0, p1, p2, corrected, du0, du1, du2, dv0, dv1, dv2);
                    const f3 dp0 = dL_diu * du0 + dL_div * dv0;
                    const f3 dp1 = dL_diu * du1 + dL_div * dv1;
                    const f3 dp2 = dL_diu * du2 + dL_div * dv2;
                    g3[0] = dp0.x; g3[1] = dp0.y; g3[2] = dp0.z;
                    g3[3] = dp1.x; g3[4] = dp1.y; g3[5] = dp1.z;
                    g3[6] = dp2.x; g3[7] = dp2.y; g3[8] = dp2.z;
                }
                seg_scan16_safe(g3, m1, m2, m4, m8, s1, s2, s4, s8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 9; c++) atomicAdd(arow + M_DV + c, g3[c]);
                }
            }
            if (tie_bal) {                                                       // (wave-uniform)
                const uint32_t tb = (uint32_t)__shfl((int)tie_base, __ffsll((long long)tie_bal) - 1);
                const uint32_t at = tb + __builtin_amdgcn_mbcnt_hi((uint32_t)(tie_bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tie_bal, 0u));
                if (tie_push && at < tie_cap)
                    tie_queue[at] = tie_pack((uint32_t)recs[j].face_id, (uint32_t)b, (uint32_t)(X0 + (q & 15)), (uint32_t)(Y0 + (q >> 4)), dL_doarea);
            }
        }
        STAMP(9)
        lds_prefetch_wait();                                        // the next chunk's records, masks and ids are in LDS ...
        __syncthreads();                                            // ... for every wave once all of them are here
        STAMP(10)

        // ---- flush: lane = (entry, component); 8 entries per pass ------------------------------
        // Branch-free: every component's destination is  base + 4 (id * mult),  id one of the record's (face_id, vid[0..2]).
        const int comp = tid & 31;
        if (comp < (POINT ? M_AA : M_N)) {                                    // (no AA components at temperature 0)
            float* const basep = s_fl_base[comp];                                 // (per-component table, filled in the prologue)
            const int entry = s_fl_sel[comp];
            const bool corner = (entry & 0x80) != 0;                              // DM2_FLAG_AA_GRAD_TO_VERTS: an AA corner on its way to its vertex
            int sel0, mult;
            flush_id_and_mult(entry, 0u, sel0, mult);
            for (int e = tid >> 5; e < n; e += TILE_PIX / 32) {
                float* a = acc + e * BM_ACC;
                const float flag = a[M_FLAG];                                     // (the 32 lanes of an entry sit in one wave)
                const float val = a[comp];
                int sel = sel0;
                if (corner) { int m_; flush_id_and_mult(entry, recs[e].zmask, sel, m_); }      // (the record knows the reorder)
                const int id = (&recs[e].face_id)[sel];
                if (flag != 0.f) {
                    a[comp] = 0.f;                                                // ready for the next chunk
                    if (comp == 0) a[M_FLAG] = 0.f;
#if !defined(DM2_BM_FLUSH_MODE) || DM2_BM_FLUSH_MODE == 0
                    atomicAdd(basep + (int64_t)id * mult, val);
#else
                    if (val == 12345.678f) basep[(int64_t)id * mult] = val;     // timing experiment only: no global traffic
#endif
                }
            }
        }
        STAMP(11)
    }
    STAMP_FLUSH
}

// The ties of k_render_backward_fast: one queued (pixel, face) pair per lane, the exact segment clipper's Jacobian
// (dm2_clip_seg.h: the reference's polygon also where it is not the geometric intersection) times the pair's dL/d(area),
// added to dL/d(aa_face_verts) -- or, under DM2_FLAG_AA_GRAD_TO_VERTS, to the image-space gradient of the vertex the corner
// came from.  Grid-stride over the queue; the block that finishes last empties the queue for a second backward of the
// same forward.
__global__ void __launch_bounds__(256)
k_aa_ties(dm2_render_desc d, const uint4* __restrict__ face_recs, const TieEntry* __restrict__ queue, uint32_t cap,
          uint32_t* __restrict__ counters, float* __restrict__ dL_daa_face_verts, bool check_mode) {
    if (check_mode && counters[0] != 3u) return;
    const uint32_t n = min(counters[2], cap);
    // the grid is sized for a full queue; the blocks behind the queue's end leave without a ticket (1024 tickets on one
    // address are serialised by the L2 in front of the working blocks' atomics: 8 us of a 0.2-ms step at cfg 1)
    const uint32_t working = min(gridDim.x, (n + blockDim.x - 1u) / blockDim.x);
    if (blockIdx.x >= working) return;
    const bool to_verts = (d.flags & DM2_FLAG_AA_GRAD_TO_VERTS) != 0;
    // (wave-uniform trip count: the lanes of a wave reduce over runs of equal faces with DPP before the atomics -- a face's ties
    // come from neighbouring lanes of one wave of the main kernel and sit next to each other in the queue)
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        const bool valid = i < n;
        TieEntry e = tie_pack(0u, 0u, 0u, 0u, 0.0f);
        if (valid) e = queue[i];
        uint32_t b, x, y;
        tie_unpack(e, b, x, y);
        const int64_t bf = (int64_t)b * d.F + e.face;
        const uint4* src = face_recs + bf * FACE_REC_U4;
        AAFace f;
        float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid) {
            uint4* dst = reinterpret_cast<uint4*>(&f);
#pragma unroll
            for (int k = 0; k < 8; k++) dst[k] = src[k];                          // the AA tables: the record's first 128 bytes
            const float pxmin = (float)(x + (uint32_t)d.patch_min[2 * b]), pymin = (float)(y + (uint32_t)d.patch_min[2 * b + 1]);
            float area;
            seg_area_grad(f, pxmin, pxmin + 1, pymin, pymin + 1, 1.0f, area, g, false);
#pragma unroll
            for (int c = 0; c < 6; c++) g[c] *= e.dL_doarea;
        }
        // runs of equal (view, face) inside 16-lane rows
        // (runs by CONTIGUITY: the same face may come back further down the queue -- another tile's batch -- with other faces in
        // between; lane l - K is in lane l's run only if every lane between them is)
        const int key = valid ? (int)e.face : -1 - (int)lane, kv = (int)b, l16 = (int)(lane & 15u);
        const int cont = (int)(l16 >= 1) & (int)(dpp_shr_i<1>(key) == key) & (int)(dpp_shr_i<1>(kv) == kv);   // continues the lane before (every DPP read executes)
        const int c2 = cont & dpp_shr_i<1>(cont), c4 = c2 & dpp_shr_i<2>(c2), c8 = c4 & dpp_shr_i<4>(c4);
        const bool s1 = cont != 0, s2 = (c2 != 0) & (l16 >= 2), s4 = (c4 != 0) & (l16 >= 4), s8 = (c8 != 0) & (l16 >= 8);
        const bool last = (l16 == 15) | (dpp_shl_i<1>(cont) == 0);
#pragma unroll
        for (int c = 0; c < 6; c++) seg_scan16(g[c], s1, s2, s4, s8);
        const bool emit = valid & last;
        if (!emit) continue;
        if (to_verts) {
            const uint4 ids = src[14];                                            // vid[0..2] are dwords 56..58 of the record
            const int vid[3] = {(int)ids.x, (int)ids.y, (int)ids.z};
            const uint32_t zm = src[4].z;                                         // zmask: dword 18
            const bool flip = (zm >> 8) & 1u;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int v = vid[c == 0 ? 0 : (flip ? 3 - c : c)];
                float* dstp = dL_daa_face_verts + ((int64_t)b * d.P + v) * 2;
                atomicAdd(dstp, g[2 * c]); atomicAdd(dstp + 1, g[2 * c + 1]);
            }
        } else {
            float* dstp = dL_daa_face_verts + bf * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) atomicAdd(dstp + c, g[c]);
        }
    }
    __shared__ uint32_t s_ticket;
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(counters + 3, 1u);
    __syncthreads();
    if (s_ticket == working - 1 && threadIdx.x == 0) { counters[2] = 0u; counters[3] = 0u; }
}

// check_mode: the caller does not know what the forward left (DM2_FWD_UNKNOWN): both kernels look at hit_valid themselves
void launch_render_backward_fast(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                 float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                 float* dL_daa_face_verts, const BinningState& bs, TieEntry* tie_queue, int64_t tie_cap,
                                 bool check_mode, hipStream_t st) {
    const uint32_t Tn = (uint32_t)(((d.W + TILE - 1) / TILE) * ((d.H + TILE - 1) / TILE) * d.B);
    const uint32_t cap = (uint32_t)(tie_cap > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : tie_cap);
    if (!(d.aa_temperature > 0.0f)) {                      // point-sampled coverage: the masks of dm2_forward_point.hip, nothing else
        StageTimer tm(check_mode ? -1 : ST_BWD, st);
        hipLaunchKernelGGL(k_render_backward_fast<true>, dim3(tile_grid_blocks(Tn)), dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                           dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                           bs.hit_masks, bs.hit_valid, bs.hit_base, bs.pool, tie_queue, 0u, check_mode STAMP_ARG(1));
        return;
    }
    {
        StageTimer tm(check_mode ? -1 : ST_BWD, st);       // (under DM2_FWD_UNKNOWN the caller times the whole cascade)
        hipLaunchKernelGGL(k_render_backward_fast<false>, dim3(tile_grid_blocks(Tn)), dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                           dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                           bs.hit_masks, bs.hit_valid, bs.hit_base, bs.pool, tie_queue, cap, check_mode STAMP_ARG(1));
    }
    const unsigned blocks = (unsigned)((cap + 255u) / 256u < 1024u ? (cap + 255u) / 256u : 1024u);
    StageTimer tm(check_mode ? -1 : ST_TIES, st);
    if (blocks) hipLaunchKernelGGL(k_aa_ties, dim3(blocks), dim3(256), 0, st, d, is.face_recs, tie_queue, cap, bs.hit_valid, dL_daa_face_verts, check_mode);
}

}  // namespace dm2
