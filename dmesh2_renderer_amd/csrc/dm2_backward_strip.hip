// dm2_backward_strip.hip -- backward composite driven by the forward's blend masks, ONE WAVE PER STRIP.
//
// Same arithmetic per (pixel, face) pair and per pixel as dm2_backward_mask.hip (BACKWARD::renderCUDA<3>,
// backward.cu:17-532, up to fp32 summation order of the scattered gradients), different decomposition: a workgroup is one
// wave and owns one 4-row strip (64 pixels) of a 16x16 tile.  The forward's masks are already per (list entry, strip), so
// a strip's blending pairs, its pixels' replays and its share of every face's gradient involve no other wave:
//
//   * no workgroup barrier anywhere (the tile-wide kernel has three per chunk, 14 % of its wave-cycles, and its four waves
//     move in lock step); 12 independent waves per CU interleave instead of 3 blocks of 4;
//   * entries that blend into no pixel of the strip cost this wave nothing: no record is fetched for them;
//   * the walk stops at the strip's own deepest contributor, not the tile's.
//
// Per round: the next <= 64 list entries' mask words (walked back to front) -> scan of their hit counts, keep the leading
// entries whose hits fit the wave's 64 lanes (at most 16 entries with hits: the LDS budget of 12 waves per CU), compact
// them into a slot table; then B2 (lane = blending pair), C (lane = pixel: replay), D (lane = pair: chain rule, DPP
// pre-reduction over runs of equal slots, ds_add_f32) and the flush with (slot, component)-shaped global atomics.
//
// Memory pipeline: window k+2 (mask words + face ids, LDS-direct) is requested when window k+1 has been scanned, at the top
// of round k; the records of round k+1 are requested into the SAME array right behind round k's phase B2, their last reader
// (phase D takes its operands from B2 in registers, the flush takes the four ids it needs from a copy); both are waited
// for before the flush, so that the flush's atomics are never waited for.
#include <hip/hip_runtime.h>

#include "dm2_bwd_shared.h"
#include "dm2_clip_seg.h"
#include "dm2_device_math.h"
#include "dm2_dpp.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

static_assert(DM2_BM_CARRY == 2, "the records are refilled behind phase B2: phase D must not read them");
constexpr int BS_SLOTS = 16;                          // entries with hits per round
constexpr int BS_WAVES_PER_SIMD = 3;                  // the register budget (<= 168 VGPRs); 12 one-wave workgroups per CU, 13 KB of LDS each
constexpr int BS_REC_CHUNKS = (int)(sizeof(FaceRec) / 16);

__global__ void __launch_bounds__(64, BS_WAVES_PER_SIMD)
k_render_backward_strip(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                        ImageState is, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                        float* __restrict__ dL_dverts, float* __restrict__ dL_dverts_color,
                        float* __restrict__ dL_dfaces_opacity, float* __restrict__ dL_dverts_ndc,
                        float* __restrict__ dL_dfaces_intense, float* __restrict__ dL_daa_face_verts,
                        const uint64_t* __restrict__ hit_masks, const uint32_t* __restrict__ hit_valid) {
    if (hit_valid[0] != 2u) return;                                // the masks are not this frame's: k_render_backward runs

    __shared__ FaceRec recs[BS_SLOTS];                             // slot r's face (refilled behind B2)
    __shared__ BmPair s_pair[64];                                  // (its first 256 B double as the decode's start marks)
    __shared__ float acc[BS_SLOTS * BM_ACC];
    __shared__ int4 s_ids4[BS_SLOTS];                              // (face_id, vid[0..2]) of slot r for the flush
    __shared__ unsigned long long s_tab_hit[2][BS_SLOTS];          // slot tables of this round / the next one:
    __shared__ int s_tab_base[2][BS_SLOTS];                        //   pixels of the strip the entry blends into, pairs in front of it,
    __shared__ uint32_t s_tab_id[2][BS_SLOTS];                     //   face id,
    __shared__ int s_tab_pos[2][BS_SLOTS];                         //   walk position
    __shared__ uint32_t s_win_lo[64], s_win_hi[64], s_win_id[64];  // the next window: mask words and face ids of 64 walk positions
    __shared__ float s_ray[64 * 6];
    __shared__ float s_pixc[6][64];                                // per pixel, read by phase C only: dL/dcolour, dL/ddepth, final T, T in front of the last contributor
    __shared__ float* s_fl_base[32];
    __shared__ int s_fl_sel[32];

    // ---- block -> (tile, strip): the four strips of a tile and neighbouring tiles share an XCD (one L2 for their records)
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const uint32_t Tn = gx * gy * (uint32_t)d.B, per = (Tn + 7) / 8;
    const uint32_t in_xcd = blockIdx.x >> 3, strip = in_xcd & 3u, tile = (blockIdx.x & 7u) * per + (in_xcd >> 2);
    if (tile >= Tn) return;
    const int b = (int)(tile / (gx * gy));
    const uint32_t tyx = tile - (uint32_t)b * gx * gy;
    const int tile_y = (int)(tyx / gx), tile_x = (int)(tyx - (uint32_t)tile_y * gx);
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = (int)strip * 4 + (lane >> 4);
    const int X0 = tile_x * TILE, Y0 = tile_y * TILE;
    const uint32_t px = X0 + lx, py = Y0 + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    const uint32_t pmx = (uint32_t)d.patch_min[2 * b], pmy = (uint32_t)d.patch_min[2 * b + 1];
    const int X0a = X0 + (int)pmx, Y0a = Y0 + (int)pmy + (int)strip * 4;      // absolute coordinates of the strip's pixel (0, 0)
    const bool corrected = (d.flags & DM2_FLAG_CORRECTED_DV) != 0;

    uint32_t last_contributor = 0;
    float T = 0.f;                                                 // starts as the T in front of the pixel's last contributor
    {
        float T_final = 0.f, prev_T_final = 0.f, dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f;
        if (inside) {
            f3 ro, rd;
            pixel_ray(d, b, pix, px + pmx, py + pmy, d.full_W, d.full_H, ro, rd);
            s_ray[lane * 6] = ro.x; s_ray[lane * 6 + 1] = ro.y; s_ray[lane * 6 + 2] = ro.z;
            s_ray[lane * 6 + 3] = rd.x; s_ray[lane * 6 + 4] = rd.y; s_ray[lane * 6 + 5] = rd.z;
            T_final = is.final_T[pix]; prev_T_final = is.final_prev_T[pix];
            last_contributor = is.n_contrib[pix];
            dLc0 = dL_dcolor[3 * pix]; dLc1 = dL_dcolor[3 * pix + 1]; dLc2 = dL_dcolor[3 * pix + 2];
            dLd = dL_ddepth[pix];
        }
        s_pixc[0][lane] = dLc0; s_pixc[1][lane] = dLc1; s_pixc[2][lane] = dLc2; s_pixc[3][lane] = dLd;
        s_pixc[4][lane] = T_final; s_pixc[5][lane] = prev_T_final;
        T = prev_T_final;
    }
    uint2 range = ranges[tile];
    range.x = __builtin_amdgcn_readfirstlane(range.x); range.y = __builtin_amdgcn_readfirstlane(range.y);
    // entries behind every pixel's last contributor are dead -- of THIS strip's pixels
    const uint32_t max_lc = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(last_contributor), 63);
    const int total = (int)min(max_lc, range.y - range.x);
    if (total <= 0) return;
    if (lane < M_N) fill_flush_table(lane, b, d.P, d.F, dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense,
                                     dL_daa_face_verts, s_fl_base, s_fl_sel);
    for (int k = lane; k < BS_SLOTS * BM_ACC; k += 64) acc[k] = 0.f;          // the flush re-zeroes what it consumes

    const float temp = d.aa_temperature;                           // > 0 (the launcher dispatches on it)
    const float pix_area = 1.0f;
    const float bg0 = d.background[0], bg1 = d.background[1], bg2 = d.background[2];
    const uint4* const grecs = is.face_recs + (int64_t)b * d.F * FACE_REC_U4;
    const uint32_t* const masks32 = reinterpret_cast<const uint32_t*>(hit_masks);

    bool T_first_pass = true;
    float accum_rec0 = 0.f, accum_rec1 = 0.f, accum_rec2 = 0.f, accum_recd = 0.f;
    float last_alpha = 0.f, last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, last_depth = 0.f;

    // walk position k (0 = the strip's deepest live entry) <-> list entry range.x + total - 1 - k (backward.cu:171)
    auto walk_entry = [&](int k) -> int64_t { return (int64_t)range.x + (uint32_t)(total - 1 - k); };
    // request mask words and face ids of the walk positions [pos, pos + 64) into the window arrays
    auto request_window = [&](int pos) {
        if (pos + lane < total) {
            const int64_t e = walk_entry(pos + lane);
            const uint32_t* m = masks32 + (e * 4 + strip) * 2;
            glds4(m, s_win_lo); glds4(m + 1, s_win_hi); glds4(face_list + e, s_win_id);
        }
    };
    // window -> slot table `buf`: the leading positions whose hits fit 64 lanes and BS_SLOTS slots.
    // n_win = positions consumed (with or without hits), n_slots = entries with hits among them, S = their hits
    auto scan_window = [&](int buf, int pos, int& n_win, int& n_slots, int& S) {
        const bool valid = pos + lane < total;
        const unsigned long long h = valid ? (((unsigned long long)s_win_hi[lane] << 32) | s_win_lo[lane]) : 0ull;
        const int c = __popcll(h);
        const int inc = wave_inclusive_scan(c);
        const unsigned long long nzb = __ballot(c > 0);
        const int nzinc = __popcll(nzb & ((2ull << lane) - 1ull));             // entries with hits up to and including this lane
        const bool ok = inc <= 64 && nzinc <= BS_SLOTS;                        // monotone: the ok lanes are a prefix (lane 0 always is)
        n_win = min(__popcll(__ballot(ok)), total - pos);
        S = __builtin_amdgcn_readlane(inc, n_win - 1);
        const bool sel = ok && c > 0;
        n_slots = __popcll(__ballot(sel));
        if (sel) {
            const int r = nzinc - 1;
            s_tab_hit[buf][r] = h; s_tab_base[buf][r] = inc - c; s_tab_id[buf][r] = s_win_id[lane]; s_tab_pos[buf][r] = pos + lane;
        }
    };
    // request the records of slot table `buf` (4 records x 15 parts of 16 B per wave instruction)
    const int rl = lane / BS_REC_CHUNKS, rp = lane - rl * BS_REC_CHUNKS;
    auto request_records = [&](int buf, int ns) {
#pragma unroll
        for (int i = 0; i < BS_SLOTS / 4; i++) {
            const int r = i * 4 + rl;
            if (rl < 4 && r < ns) glds16(grecs + (int64_t)s_tab_id[buf][r] * FACE_REC_U4 + rp, &recs[i * 4]);
        }
    };

    // ---- prologue: window 0 -> table 0, window 1 requested, records of round 0
    int cur = 0, pos = 0, nw = 0, ns = 0, S = 0;
    request_window(0);
    lds_prefetch_wait();
    __syncthreads();
    scan_window(0, 0, nw, ns, S);
    __syncthreads();                                               // (one wave: orders this wave's LDS traffic for the compiler)
    if (nw < total) request_window(nw);
    request_records(0, ns);
    lds_prefetch_wait();
    __syncthreads();

    volatile uint32_t* const s_mark = reinterpret_cast<volatile uint32_t*>(s_pair);
    while (true) {
        // recs / table[cur] = this round (ns slots, S pairs, nw walk positions from pos); window arrays = positions from pos + nw
        const int pos1 = pos + nw;
        int nw1 = 0, ns1 = 0, S1 = 0;
        if (pos1 < total) {
            scan_window(cur ^ 1, pos1, nw1, ns1, S1);
            __syncthreads();
            if (pos1 + nw1 < total) request_window(pos1 + nw1);
        }
        const unsigned long long* const t_hit = s_tab_hit[cur];
        const int* const t_base = s_tab_base[cur];
#ifdef DM2_STRIP_DEBUG
        if (strip == 0 && tile == 0 && lane < 16) {
            static __device__ int dbg_round;
            const int rr = (lane == 0) ? atomicAdd(&dbg_round, 1) : 0;
            const int r0 = __builtin_amdgcn_readfirstlane(rr);
            float* o = dL_daa_face_verts + 36 + r0 * 40;
            if (lane == 0) { o[0] = (float)nw; o[1] = (float)ns; o[2] = (float)S; o[3] = (float)pos; o[4] = (float)total; o[5] = (float)max_lc; }
            if (lane < ns) { o[8 + lane] = (float)t_base[lane]; o[24 + lane] = (float)s_tab_id[cur][lane]; }
        }
#endif

        int j = 0, q = 0;
        float dg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float i0 = 0.f, i1 = 0.f, i2 = 0.f, ratio = 0.f, alpha = 0.f;
        int code = 0;
        bool blend = false;
        f3 k_ro = {0, 0, 0}, k_rd = {0, 0, 0}, k_p0 = {0, 0, 0}, k_p1 = {0, 0, 0}, k_p2 = {0, 0, 0};
        float k_col[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, k_dep[3] = {0, 0, 0}, k_int = 0.f, k_opa = 0.f;
        const bool have = lane < S;
        if (S > 0) {
            // ---- pair lane -> slot: every slot leaves a mark at its first pair, a running maximum spreads it
            // (volatile: lanes talk to each other through these words with no barrier in between -- without it the compiler
            // may forward this thread's own stores to its load)
            s_mark[lane] = 0u;
            if (lane < ns) { const int bs = t_base[lane]; s_mark[bs] = ((uint32_t)(lane + 1) << 9) | (uint32_t)bs; }
            const uint32_t mk = wave_inclusive_max(s_mark[lane]);
            __syncthreads();                                       // the marks are read before s_pair is written
            // ---- phase B2: one blending (pixel, face) pair per lane ---------------------------------
            if (have) {
                j = (int)(mk >> 9) - 1;
                q = nth_set_bit64(t_hit[j], lane - (int)(mk & 511u));            // pixel of the strip
                const FaceRec& fc = recs[j];
                const float pxmin = (float)(uint32_t)(X0a + (q & 15)), pxmax = pxmin + 1;
                const float pymin = (float)(uint32_t)(Y0a + (q >> 4)), pymax = pymin + 1;
                // (the forward's decisions are not taken again: see dm2_backward_mask.hip)
                float oarea;
                seg_area_grad(fc.aa, pxmin, pxmax, pymin, pymax, pix_area, oarea, dg);
                oarea = fmaxf(oarea, 0.0f);
                BmPair out; out.alpha = 0.f; out.c0 = out.c1 = out.c2 = out.depth = 0.f; out.flags = 0; out.T = 0.f; out.dL_dalpha = 0.f;
                ratio = oarea / pix_area;
                const f3 ro = {s_ray[q * 6], s_ray[q * 6 + 1], s_ray[q * 6 + 2]};
                const f3 rd = {s_ray[q * 6 + 3], s_ray[q * 6 + 4], s_ray[q * 6 + 5]};
                const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                f3 tuv = {0, 0, 0};
                k_ro = ro; k_rd = rd; k_p0 = p0; k_p1 = p1; k_p2 = p2;
#pragma unroll
                for (int c = 0; c < 9; c++) k_col[c] = fc.col[c];
                k_dep[0] = fc.dep[0]; k_dep[1] = fc.dep[1]; k_dep[2] = fc.dep[2]; k_int = fc.intense; k_opa = fc.opacity;
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
                s_pair[lane] = out;
            }
            if (lane < ns) {                                       // what the flush needs of the records, before they are refilled
                const FaceRec& fr = recs[lane];
                s_ids4[lane] = make_int4(fr.face_id, fr.vid[0], fr.vid[1], fr.vid[2]);
            }
        }
        __syncthreads();                                           // every read of recs has returned
        if (pos1 < total) request_records(cur ^ 1, ns1);           // B2 was the records' last reader: refill behind it

        if (S > 0) {
            // ---- phase C: per-pixel back-to-front replay -------------------------------------------
            {
                uint32_t m = 0;                                    // slots with a record for this pixel (ascending slot = back to front)
                for (int r = 0; r < ns; r++) m |= (uint32_t)((t_hit[r] >> lane) & 1ull) << r;
                float dLc0 = 0.f, dLc1 = 0.f, dLc2 = 0.f, dLd = 0.f, T_final = 0.f, prev_T_final = 0.f;
                if (m) {
                    dLc0 = s_pixc[0][lane]; dLc1 = s_pixc[1][lane]; dLc2 = s_pixc[2][lane]; dLd = s_pixc[3][lane];
                    T_final = s_pixc[4][lane]; prev_T_final = s_pixc[5][lane];
                }
                while (m) {
                    const int jj = __ffs((int)m) - 1;
                    m &= m - 1;
                    const uint32_t e = (uint32_t)(total - 1 - s_tab_pos[cur][jj]);        // 0-based position in the list
                    if (e >= last_contributor) continue;                                  // backward.cu:219-221
                    BmPair& pr = s_pair[t_base[jj] + __popcll(t_hit[jj] & ((1ull << lane) - 1ull))];
                    if (!(pr.flags & MB_BLEND)) continue;                                 // (the ray missed the face's plane)
                    const float a = pr.alpha, iC0 = pr.c0, iC1 = pr.c1, iC2 = pr.c2, iD = pr.depth;
                    // alpha == 1 exactly (backward.cu:396) is the forward's decision too: only a pixel's LAST contributor can
                    // have it (T drops to 0 and the pixel is done), and then final_T is exactly 0
                    const bool alpha_is_one = (a == 1.0f) || (T_first_pass && T_final == 0.0f);
                    if (!T_first_pass) T = T / (1.f - a);                                 // backward.cu:340-348
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
                    const float bd_dot = (float)(0.0 + 1.0 * (double)dLd);                // backward.cu:394
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
            __syncthreads();

            // ---- phase D: chain rule + per-slot accumulation ------------------------------------------
            {
                const int jkey = have ? j : -1;
                const int l16 = lane & 15;
                // NB: every DPP read must execute with all lanes enabled, hence the unconditional reads and `&`, `|`.
                const int k1 = dpp_shr_i<1>(jkey), k2 = dpp_shr_i<2>(jkey), k4 = dpp_shr_i<4>(jkey), k8 = dpp_shr_i<8>(jkey);
                const int kn = dpp_shl_i<1>(jkey);
                const bool s1 = (l16 >= 1) & (k1 == jkey);
                const bool s2 = (l16 >= 2) & (k2 == jkey);
                const bool s4 = (l16 >= 4) & (k4 == jkey);
                const bool s8 = (l16 >= 8) & (k8 == jkey);
                const float m1 = s1 ? 1.f : 0.f, m2 = s2 ? 1.f : 0.f, m4 = s4 ? 1.f : 0.f, m8 = s8 ? 1.f : 0.f;
                BmPair pr; pr.flags = 0; pr.T = 0.f; pr.dL_dalpha = 0.f; pr.c0 = pr.c1 = pr.c2 = pr.depth = 0.f;
                if (have && blend) pr = s_pair[lane];
                const bool active = (pr.flags & MB_ACTIVE) != 0;
                float nact = active ? 1.f : 0.f;
                seg_scan16(nact, s1, s2, s4, s8);
                const bool emit = ((l16 == 15) | (kn != jkey)) & (jkey >= 0) & (nact > 0.f);
#ifdef DM2_STRIP_DEBUG
                if (strip == 0 && tile == 0) {
                    const int na = __popcll(__ballot(active)), ne = __popcll(__ballot(emit)), nb = __popcll(__ballot(have && blend)), nh = __popcll(__ballot(have));
                    const int nf = __popcll(__ballot((pr.flags & MB_BLEND) != 0));
                    if (lane == 0) { float* o = dL_daa_face_verts + 36 + 200 + pos; o[0] = (float)(na + 100 * ne + 10000 * nb); o[1] = (float)(nh + 100 * nf); }
                }
#endif
                float* const arow = acc + j * BM_ACC;
                float dL_diu = 0.f, dL_div = 0.f, dL_doarea = 0.f;
                {   // group 1: vertex colours, NDC depth, intensity, opacity
                    float g1[14];
#pragma unroll
                    for (int c = 0; c < 14; c++) g1[c] = 0.f;
                    if (active) {
                        const float Tq = pr.T, dL_dalpha = pr.dL_dalpha;
                        const float qc0 = pr.c0, qc1 = pr.c1, qc2 = pr.c2, qd = pr.depth;   // dL/dcolour, dL/ddepth of the pixel
                        const float intense = k_int, opacity = k_opa;
                        const float dics[3] = {qc0 * alpha * Tq, qc1 * alpha * Tq, qc2 * alpha * Tq};
                        const float did = qd * alpha * Tq;
                        g1[12] = dL_dalpha * ratio;
                        const float dL_dratio = (dL_dalpha * opacity) * temp;
                        dL_doarea = dL_dratio / pix_area;
                        float dL_di0 = 0.f, dL_di1 = 0.f, dL_di2 = 0.f, dL_dfint = 0.f;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            dL_di0 += k_col[ch] * dics[ch] * intense;
                            dL_di1 += k_col[3 + ch] * dics[ch] * intense;
                            dL_di2 += k_col[6 + ch] * dics[ch] * intense;
                            g1[ch] = 0.f + i0 * dics[ch] * intense;
                            g1[3 + ch] = 0.f + i1 * dics[ch] * intense;
                            g1[6 + ch] = 0.f + i2 * dics[ch] * intense;
                            dL_dfint += (i0 * k_col[ch] + i1 * k_col[3 + ch] + i2 * k_col[6 + ch]) * dics[ch];
                        }
                        g1[13] = dL_dfint;
                        dL_di0 += k_dep[0] * did; dL_di1 += k_dep[1] * did; dL_di2 += k_dep[2] * did;
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
                        f3 du0, du1, du2, dv0, dv1, dv2;
                        ray_tri_intersection_grad(k_ro, k_rd, k_p0, k_p1, k_p2, corrected, du0, du1, du2, dv0, dv1, dv2);
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
        }
        lds_prefetch_wait();                                       // the next round's records and the window behind it are in LDS
        __syncthreads();

        // ---- flush: lane = (slot, component); 2 slots per pass ------------------------------------
        if (S > 0) {
            const int comp = lane & 31;
            if (comp < M_N) {
                float* const basep = s_fl_base[comp];
                const int sel = s_fl_sel[comp] & 3, mult = s_fl_sel[comp] >> 2;
                for (int e = lane >> 5; e < ns; e += 2) {
                    float* a = acc + e * BM_ACC;
                    const float flag = a[M_FLAG];
                    const float val = a[comp];
                    const int id = reinterpret_cast<const int*>(&s_ids4[e])[sel];
                    if (flag != 0.f) {
                        a[comp] = 0.f;                                                // ready for the next round
                        if (comp == 0) a[M_FLAG] = 0.f;
                        atomicAdd(basep + (int64_t)id * mult, val);
                    }
                }
            }
            __syncthreads();
        }
        if (pos1 >= total) break;
        pos = pos1; nw = nw1; ns = ns1; S = S1; cur ^= 1;
    }
}

void launch_render_backward_strip(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                  const float* dL_dcolor, const float* dL_ddepth, float* dL_dverts, float* dL_dverts_color,
                                  float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                  float* dL_daa_face_verts, const uint64_t* hit_masks, const uint32_t* hit_valid, hipStream_t st) {
    const uint32_t Tn = (uint32_t)(((d.W + TILE - 1) / TILE) * ((d.H + TILE - 1) / TILE) * d.B);
    const uint32_t per = (Tn + 7) / 8;
    hipLaunchKernelGGL(k_render_backward_strip, dim3(8 * per * 4), dim3(64), 0, st, d, ranges, face_list, is, dL_dcolor, dL_ddepth,
                       dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts,
                       hit_masks, hit_valid);
}

}  // namespace dm2
