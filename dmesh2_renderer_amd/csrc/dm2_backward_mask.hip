// dm2_backward_mask.hip -- backward composite driven by the forward's blend masks.
//
// Same results as k_render_backward (dm2_backward.hip; BACKWARD::renderCUDA<3>, backward.cu:17-532) up to fp32
// summation order of the scattered gradients.  dm2_forward_queue.hip leaves, per list entry and wave of the
// tile's block, the 64-bit mask of the pixels the entry blends into.  With those the backward neither enumerates
// nor classifies (pixel,face) pairs, it never meets a clipper error path, and it can size every chunk to exactly
// one full round:
//
//   per chunk (walked back to front): the masks of the next <= 32 entries, scan of their hit counts (every wave for
//   itself, DPP), keep the leading entries whose hits fit 256 lanes (a ballot), pair lane -> (entry, pixel) through
//   start marks and a running maximum, then
//   B2  lane s: its (face, pixel) from the masks; AA area + Jacobian (dm2_clip_seg.h), Moeller-Trumbore, clamp,
//       coverage, alpha, interpolated colour / depth -> record in LDS
//   C   pixel p: replay its records back to front (backward.cu:340-405)
//   D   lane s: chain rule (backward.cu:408-488), DPP pre-reduction over the lanes of one face, ds_add_f32
//   flush with (entry,component) global atomics.
//
// Memory pipeline.  A chunk's inputs -- face ids, blend masks, packed face records (dm2_stage.h: 256 B per
// (view,face), two full lines) -- are requested ONE CHUNK AHEAD, right after the current chunk's cut is known, with
// LDS-direct loads (global_load_lds_dwordx4: 4 records per wave instruction, no staging registers) into the OTHER
// half of double-buffered LDS arrays: the global latency is covered by phases B2, C and D of the current chunk instead
// of stalling the block (in-kernel stamps of the round-1 kernel: 48 % of a block's time went into that stall).  The
// ids a record address needs are read one chunk earlier still, as a window of 64 entries.
//
// The masks are only valid when this frame's forward was dm2_forward_queue.hip (hit_valid[0] == 2); otherwise the
// kernel returns at once and k_render_backward, launched behind it, does the work.
#include <hip/hip_runtime.h>

#include "dm2_bwd_shared.h"
#include "dm2_clip_seg.h"
#include "dm2_device_math.h"
#include "dm2_dpp.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_stamps.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int BM_CAND = 32;      // candidate entries per chunk: two per 16-lane group of the cooperative record copy
static_assert(BM_CAND * 4 <= TILE_PIX && 2 * BM_CAND <= 64, "one scan thread per (face, wave); one id window per wave");
constexpr int BM_SLOTS = BM_CAND * 4;
constexpr int REC_CHUNKS = (int)(sizeof(FaceRec) / 16);   // 15 x 16 B of the 256-B global record are live
#ifndef DM2_BM_BLOCKS
#define DM2_BM_BLOCKS 3       // resident blocks per CU the register budget is set for.  A/B at cfg4 on MI355X: 4 blocks (128 VGPRs) spill
                              // 39 registers to scratch, and a scratch reload waits for every LDS-direct load issued before it: 2.08 ms;
                              // 3 blocks (no spill) 1.63 ms
#endif
__global__ void __launch_bounds__(TILE_PIX, DM2_BM_BLOCKS)
k_render_backward_mask(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                       float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                       float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                       float* __restrict__ dL_dfaces_intense, float* __restrict__ dL_daa_face_verts,
                       const uint64_t* __restrict__ hit_masks, const uint32_t* __restrict__ hit_valid STAMP_PARAM) {
    if (hit_valid[0] != 2u) return;                                // the masks are not this frame's: k_render_backward runs

    __shared__ FaceRec recs2[2][BM_CAND];                      // [buffer]: this chunk's candidates / the next chunk's
    __shared__ float acc[BM_CAND * BM_ACC];
    __shared__ BmPair s_pair[TILE_PIX];
    __shared__ float s_ray[TILE_PIX * 6];
    __shared__ __attribute__((aligned(16))) unsigned long long s_hit2[2][BM_SLOTS];   // [buffer][face][wave]: pixels of the wave the face blends into
    __shared__ int s_wbase[4][BM_CAND];                        // [wave][face]: pairs in front of slot (face, wave) -- written and read by that wave
    __shared__ uint32_t s_mark[4][64];                         // [wave][pair lane]: (slot + 1) << 9 | first pair of the slot, where a slot starts
    __shared__ unsigned long long s_mask[TILE_PIX];            // per pixel: faces of the chunk with a record for it
    __shared__ uint32_t s_ids2[2][2 * BM_CAND];                // [buffer]: face ids of the walk positions [base, base + 64)
    __shared__ float s_pixc[6][TILE_PIX];                      // per pixel, read by phase C only: dL/dcolour, dL/ddepth, final T, T in front of the last contributor
    __shared__ float* s_fl_base[32];                           // flush, per component: destination of id 0 ...
    __shared__ int s_fl_sel[32];                               // ... which id of the record (face_id, vid[0..2]) | dwords per id << 2
    __shared__ uint32_t s_max_lc;

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
    uint2 range = ranges[tile];                                    // block-uniform: keep it in scalar registers
    range.x = __builtin_amdgcn_readfirstlane(range.x); range.y = __builtin_amdgcn_readfirstlane(range.y);

    if (tid == 0) s_max_lc = 0;
    if (tid < M_N) fill_flush_table(tid, b, d.P, d.F, dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense,
                                    dL_daa_face_verts, s_fl_base, s_fl_sel, (d.flags & DM2_FLAG_AA_GRAD_TO_VERTS) != 0);
    __syncthreads();
    if (last_contributor) atomicMax(&s_max_lc, last_contributor);
    __syncthreads();
    const int total = (int)min((uint32_t)__builtin_amdgcn_readfirstlane(s_max_lc), range.y - range.x);   // entries behind every pixel's last contributor are dead

    const float temp = d.aa_temperature;                           // > 0 (the launcher dispatches on it)
    const float pix_area = 1.0f;
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];
    const uint4* const grecs = is.face_recs + (int64_t)b * d.F * FACE_REC_U4;

    bool T_first_pass = true;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    // ---- the walk's position k (0 = the tile's deepest live entry) maps to list entry range.x + total - 1 - k
    // (backward.cu:171).  The next chunk's inputs go straight from global memory into the other LDS buffer
    // (global_load_lds: per-lane source address, destination = wave-uniform base + lane * size; no registers held).
    auto walk_entry = [&](int k) -> int64_t { return (int64_t)range.x + (uint32_t)(total - 1 - k); };
    const int rl = lane / REC_CHUNKS, rp = lane - rl * REC_CHUNKS;   // record copy: 4 records x 15 parts per wave instruction
    // request the id window [nb, nb + 64) of the walk into s_ids2[buf]
    auto request_ids = [&](int buf, int nb) {
        if (wid == 2 && nb + lane < total)
            glds4(face_list + walk_entry(nb + lane), &s_ids2[buf][0]);
    };
    // request masks + records of the chunk starting at walk position nb into buffer buf; ids[i]: face id of position nb + i
    auto request_chunk = [&](int buf, int nb, const uint32_t* ids) {
        const int nc2 = min(BM_CAND, total - nb);
        // (the masks first: should the compiler ever reload an address from scratch here, that reload waits for every
        // load issued before it -- vmcnt is in order -- and must not find this chunk's record loads in front of it)
        if (wid == 1 && (lane >> 1) < nc2)                         // lane: masks of (face lane / 2, waves 2 (lane & 1), + 1)
            glds16(hit_masks + walk_entry(nb + (lane >> 1)) * 4 + (lane & 1) * 2, &s_hit2[buf][0]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = (i * 4 + wid) * 4;                      // this wave instruction's first record
            const int r = r0 + rl;
            if (rl < 4 && r < nc2) glds16(grecs + (int64_t)ids[r] * FACE_REC_U4 + rp, &recs2[buf][r0]);
        }
    };
    if (total > 0) {                                               // first chunk: synchronously
        request_ids(0, 0);
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
        FaceRec* const recs = recs2[cur];
        const unsigned long long* const s_hit = s_hit2[cur];
        const uint32_t* const s_ids = s_ids2[cur];
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
        if ((lane & 1) == (wid >> 1)) s_wbase[wid][lane >> 1] = (wid & 1) ? b1 : b0;    // phase C: slot (face, this wave)
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
        float dg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
            const FaceRec& fc = recs[j];
            const float pxmin = (float)(uint32_t)(X0a + (q & 15)), pxmax = pxmin + 1;
            const float pymin = (float)(uint32_t)(Y0a + (q >> 4)), pymax = pymin + 1;
            // The forward blended this pair: its clip (aa.h:446-504) returned no error and a positive area, the ray met the
            // face's plane and the coverage was not 0.  Those decisions are NOT taken again here (the segment formulation
            // may round an area of 1e-9 to 0, or 1 to 1 - 1 ulp): a pair of the masks is replayed whatever the values say,
            // otherwise the per-pixel replay below would lose its place in the list.
            // The replay divides the running T by (1 - alpha) (backward.cu:340-348) and the background term by it once more
            // (backward.cu:396-401): an error of one ulp in alpha is an error of ulp / (1 - alpha) in both.  The segment
            // formulation's shoelace area is good to 2 ulp of the pixel area (2.4e-7): at most 2.4e-6 of T per entry up to
            // alpha = 0.9, but a nearly opaque, nearly covering face needs the forward's alpha to the bit (found by the
            // randomised sweep: opacity 1, coverage 0.999998, 1.3 % error in one opacity gradient).  alpha <= opacity, so
            // faces with opacity > 0.9 get the reference's fan sum over the same corners -- the forward's area exactly
            // (dm2_clip_seg.h); a wave without such a face skips it (the benchmark scene, opacities up to 0.9, always does).
            float oarea;
            seg_area_grad(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea, dg, fc.opacity > 0.9f);
            oarea = fmaxf(oarea, 0.0f);
            BmPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0; out.T = 0.f; out.dL_dalpha = 0.f;
            ratio = oarea / pix_area;
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
                ratio = mix_coverage(code, ratio, temp);
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
                BmPair& pr = s_pair[s_wbase[wid][jj] + __popcll(s_hit[t] & ((1ull << lane) - 1ull))];
                const float a = pr.alpha, iC0 = pr.c0, iC1 = pr.c1, iC2 = pr.c2, iD = pr.depth;
                // alpha == 1 exactly (backward.cu:396) is the forward's decision too: only a pixel's LAST contributor can
                // have it (T drops to 0 and the pixel is done), and then final_T is exactly 0
                const bool alpha_is_one = (a == 1.0f) || (T_first_pass && T_final == 0.0f);
                if (!T_first_pass) T = T / (1.f - a);                             // backward.cu:340-348
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
                    dL_dalpha += (-T_final / (1.f - a)) * bg_dot;
                    dL_dalpha += (-T_final / (1.f - a)) * bd_dot;
                }
                pr.T = T; pr.dL_dalpha = dL_dalpha; pr.flags = MB_BLEND | MB_ACTIVE;
                // phase D needs this pixel's loss gradients, not the colours any more: hand them over in place
                pr.c0 = dLc0; pr.c1 = dLc1; pr.c2 = dLc2; pr.depth = dLd;
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
            BmPair pr; pr.flags = 0; pr.T = 0.f; pr.dL_dalpha = 0.f; pr.c0 = pr.c1 = pr.c2 = pr.depth = 0.f;
            if (have && blend) pr = s_pair[tid];
            const bool active = (pr.flags & MB_ACTIVE) != 0;
            float nact = active ? 1.f : 0.f;
            seg_scan16(nact, s1, s2, s4, s8);
            const bool emit = ((l16 == 15) | (kn != jkey)) & (jkey >= 0) & (nact > 0.f);
            float* const arow = acc + j * BM_ACC;
#if DM2_BM_CARRY < 2
            const FaceRec& fcD = recs[j];
#endif
            float dL_diu = 0.f, dL_div = 0.f, dL_doarea = 0.f;
            {   // group 1: vertex colours, NDC depth, intensity, opacity
                float g1[14];
#pragma unroll
                for (int c = 0; c < 14; c++) g1[c] = 0.f;
                if (active) {
                    const float Tq = pr.T, dL_dalpha = pr.dL_dalpha;
                    const float qc0 = pr.c0, qc1 = pr.c1, qc2 = pr.c2, qd = pr.depth;   // dL/dcolour, dL/ddepth of the pixel
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
                seg_scan16_n(g1, m1, m2, m4, m8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 12; c++) atomicAdd(arow + M_DC + c, g1[c]);      // M_DC..+8 and M_DZ..+2 are contiguous
                    atomicAdd(arow + M_OP, g1[12]);
                    atomicAdd(arow + M_IN, g1[13]);
                    arow[M_FLAG] = 1.0f;
                }
            }
            {   // group 2: AA corners
                float g2[6];
#pragma unroll
                for (int c = 0; c < 6; c++) g2[c] = dL_doarea * dg[c];                   // dL_doarea is 0 on inactive lanes
                seg_scan16_n(g2, m1, m2, m4, m8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 6; c++) atomicAdd(arow + M_AA + c, g2[c]);
                }
            }
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
                    ray_tri_intersection_grad(ro, rd, p0, p1, p2, corrected, du0, du1, du2, dv0, dv1, dv2);
                    const f3 dp0 = dL_diu * du0 + dL_div * dv0;
                    const f3 dp1 = dL_diu * du1 + dL_div * dv1;
                    const f3 dp2 = dL_diu * du2 + dL_div * dv2;
                    g3[0] = dp0.x; g3[1] = dp0.y; g3[2] = dp0.z;
                    g3[3] = dp1.x; g3[4] = dp1.y; g3[5] = dp1.z;
                    g3[6] = dp2.x; g3[7] = dp2.y; g3[8] = dp2.z;
                }
                seg_scan16_n(g3, m1, m2, m4, m8);
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 9; c++) atomicAdd(arow + M_DV + c, g3[c]);
                }
            }
        }
        STAMP(9)
        lds_prefetch_wait();                                        // the next chunk's records, masks and ids are in LDS ...
        __syncthreads();                                            // ... for every wave once all of them are here
        STAMP(10)

        // ---- flush: lane = (entry, component); 8 entries per pass ------------------------------
        // Branch-free: every component's destination is  base + 4 (id * mult),  id one of the record's (face_id, vid[0..2]).
        const int comp = tid & 31;
        if (comp < M_N) {
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
                if (corner) { int m_; flush_id_and_mult(entry, recs[e].aa.zmask, sel, m_); }   // (the record knows the reorder)
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

void launch_render_backward_mask(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                 float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                 float* dL_daa_face_verts, const uint64_t* hit_masks, const uint32_t* hit_valid, hipStream_t st) {
    const uint32_t Tn = (uint32_t)(((d.W + TILE - 1) / TILE) * ((d.H + TILE - 1) / TILE) * d.B);
    hipLaunchKernelGGL(k_render_backward_mask, dim3(tile_grid_blocks(Tn)), dim3(TILE_PIX), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                       dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                       hit_masks, hit_valid STAMP_ARG(1));
}

}  // namespace dm2
