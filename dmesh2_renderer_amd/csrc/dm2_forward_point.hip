// dm2_forward_point.hip -- forward composite for aa_temperature == 0 (point-sampled coverage).
//
// Same results as k_render_forward (dm2_forward.hip; FORWARD::renderCUDA<3>, forward.cu:139-432): with temperature
// 0 the reference applies no bounding-box test (forward.cu:314), every face of a tile's list is intersected with
// all 256 pixel rays and blends where the clamped barycentrics say "inside".  This is that per-pixel walk, written
// so that all lanes of a wave stay in lock-step through the intersection test of a face (no per-lane `continue` /
// `break` in front of it), which makes one more product free: a ballot per (list entry, wave) of the pixels the
// entry blended into.  The backward (dm2_backward_fast.hip, POINT) reads these 64-bit masks instead of repeating the
// dense intersection test.
//
// Two-stage test.  A tile's list holds ~200 faces, a strip of 16 x 4 pixels is hit by a few of them; the exact test
// (Moeller-Trumbore as the reference writes it, an IEEE division included: ~85 instructions) is only needed to DECIDE
// for rays near a face.  When all rays of the tile start at one point (a camera: always, with the reference's ray
// tensors) the barycentrics are ratios of three dot products of the ray direction with per-face vectors,
//     u = rd . (E2 x T) / rd . (E2 x E1),   v = rd . (T x E1) / rd . (E2 x E1),        T = ro - p0,
// staged once per chunk (one lane per face).  A wave first evaluates those (~20 instructions) against an error bound from
// the un-cancelled magnitudes of the same sums (per face: ray directions are unit vectors); only if some ray is inside or
// within the bound of the triangle does the wave run the reference's arithmetic (~60 instructions), which alone decides
// and alone produces values.  Bit-identical output.  (Measured at 1080p / 1 M faces: a first version that carried the
// magnitudes per component -- 39 instructions + 8 LDS reads per test -- gained nothing over the exact test alone.)
// In front of both, once per chunk and wave: the same linear forms over the BOX of the strip's ray directions, one lane per
// face (see "Strip prefilter" in the kernel) -- the walk visits only the faces some pixel of the strip may be near.
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_pairs.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int FP_CHUNK = 128;
constexpr float FP_EPS = 1.0e-5f;     // relative bound of |quick dot - exact numerator| in units of the un-cancelled magnitude (a few ulp would do)

// per staged face, for rays from `ro`: A = E2 x T, Bv = T x E1, Nn = E2 x E1 and the same sums with every term's magnitude
struct __attribute__((aligned(16))) FpQuick { float A[3], ma, Bv[3], mb, Nn[3], md; };    // m*: FP_EPS x the sums' un-cancelled magnitudes for |rd_i| <= 1

__global__ void __launch_bounds__(TILE_PIX)
k_render_forward_point(dm2_render_desc d, const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                       ImageState is, float* __restrict__ out_color, float* __restrict__ out_depth,
                       int32_t* __restrict__ out_tri_cnt, uint64_t* __restrict__ hit_masks, uint32_t* __restrict__ hit_valid) {
    __shared__ FaceRec recs[FP_CHUNK];
    __shared__ FpQuick s_quick[FP_CHUNK];
    __shared__ float s_ro[3];

    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    uint32_t tile;
    if (!tile_of_block(gx * gy * (uint32_t)d.B, is.tile_order, tile)) return;    // XCD-contiguous bands, longest list first (dm2_pairs.h)
    const int b = (int)(tile / (gx * gy));
    const uint32_t tyx = tile - (uint32_t)b * gx * gy;
    const uint32_t tile_y = tyx / gx, tile_x = tyx - tile_y * gx;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t lx = tid & 15, ly = tid >> 4;
    const uint32_t px = tile_x * TILE + lx, py = tile_y * TILE + ly;
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    if (blockIdx.x == 0 && tid == 0) hit_valid[0] = 1u;   // the masks of this launch are current

    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) {
        pixel_ray(d, b, pix, px + (uint32_t)d.patch_min[2 * b], py + (uint32_t)d.patch_min[2 * b + 1], d.full_W, d.full_H, ro, rd);
    }
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    // do all rays of the tile share their origin (pixel (0,0) of the tile is always inside the image)?
    if (tid == 0) { s_ro[0] = ro.x; s_ro[1] = ro.y; s_ro[2] = ro.z; }
    __syncthreads();
    const f3 ro0 = {s_ro[0], s_ro[1], s_ro[2]};
    const bool quick = __syncthreads_and(!inside || (ro.x == ro0.x && ro.y == ro0.y && ro.z == ro0.z &&
                                                     fabsf(rd.x) <= 1.0001f && fabsf(rd.y) <= 1.0001f && fabsf(rd.z) <= 1.0001f)) != 0;
    // Strip prefilter (quick rays only).  A wave's 64 pixels are a strip of 16 x 4; its ray directions fill a small box
    // [dc - dh, dc + dh].  The per-pixel quick test rejects on linear forms of the direction (A . rd, Bv . rd, Nn . rd and their
    // sum), so a form's range over the box -- centre value +- sum |coefficient| x half-width -- decides for the whole strip at
    // once: one lane per face, 64 faces per pass, and the per-pixel walk below only visits the faces some pixel of the strip may
    // still be near (a third of a tile's list).  Conservative by construction: it only skips a face if EVERY pixel's quick test
    // would reject it, with 10 % of the quick test's own margins to spare for the rounding of both evaluations.
    float dcx = 0.f, dcy = 0.f, dcz = 0.f, dhx = 0.f, dhy = 0.f, dhz = 0.f;
    if (quick) {
        const float BIG = 3.0e38f;
        float lx0 = inside ? rd.x : BIG, hx0 = inside ? rd.x : -BIG, ly0 = inside ? rd.y : BIG, hy0 = inside ? rd.y : -BIG;
        float lz0 = inside ? rd.z : BIG, hz0 = inside ? rd.z : -BIG;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lx0 = fminf(lx0, __shfl_xor(lx0, o)); hx0 = fmaxf(hx0, __shfl_xor(hx0, o));
            ly0 = fminf(ly0, __shfl_xor(ly0, o)); hy0 = fmaxf(hy0, __shfl_xor(hy0, o));
            lz0 = fminf(lz0, __shfl_xor(lz0, o)); hz0 = fmaxf(hz0, __shfl_xor(hz0, o));
        }
        if (hx0 >= lx0) {                                               // (a strip with a pixel inside the image)
            dcx = 0.5f * (lx0 + hx0); dcy = 0.5f * (ly0 + hy0); dcz = 0.5f * (lz0 + hz0);
            dhx = 0.5001f * (hx0 - lx0) + 1.0e-7f; dhy = 0.5001f * (hy0 - ly0) + 1.0e-7f; dhz = 0.5001f * (hz0 - lz0) + 1.0e-7f;
        }
    }
    bool done = !inside;
    float pT = 1.0f, T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;

    for (int base = 0; base < total; base += FP_CHUNK) {
        if (__syncthreads_count(done) == TILE_PIX) break;          // forward.cu:258-260 (also guards LDS reuse)
        const int n = min(FP_CHUNK, total - base);
        if (tid < n) {
            stage_face(is.face_recs, (int64_t)b * d.F + face_list[range.x + base + tid], recs[tid]);
            if (quick) {
                const FaceRec& fc = recs[tid];
                const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
                const f3 T = ro0 - p0, E1 = p1 - p0, E2 = p2 - p0;
                const f3 aT = {fabsf(T.x), fabsf(T.y), fabsf(T.z)}, aE1 = {fabsf(E1.x), fabsf(E1.y), fabsf(E1.z)}, aE2 = {fabsf(E2.x), fabsf(E2.y), fabsf(E2.z)};
                auto crs = [](f3 a, f3 c) -> f3 { return {a.y * c.z - a.z * c.y, a.z * c.x - a.x * c.z, a.x * c.y - a.y * c.x}; };
                auto crs_abs = [](f3 a, f3 c) -> f3 { return {a.y * c.z + a.z * c.y, a.z * c.x + a.x * c.z, a.x * c.y + a.y * c.x}; };
                const f3 A = crs(E2, T), Bv = crs(T, E1), Nn = crs(E2, E1);
                const f3 Aa = crs_abs(aE2, aT), Ba = crs_abs(aT, aE1), Na = crs_abs(aE2, aE1);
                FpQuick& q = s_quick[tid];
                q.A[0] = A.x; q.A[1] = A.y; q.A[2] = A.z; q.Bv[0] = Bv.x; q.Bv[1] = Bv.y; q.Bv[2] = Bv.z; q.Nn[0] = Nn.x; q.Nn[1] = Nn.y; q.Nn[2] = Nn.z;
                q.ma = FP_EPS * 1.0001f * (Aa.x + Aa.y + Aa.z); q.mb = FP_EPS * 1.0001f * (Ba.x + Ba.y + Ba.z); q.md = FP_EPS * 1.0001f * (Na.x + Na.y + Na.z);
            }
        }
        __syncthreads();

        for (int sub = 0; sub < FP_CHUNK / 64; sub++) {
        if (64 * sub >= n) break;
        unsigned long long todo;
        {
            const int jf = 64 * sub + lane;
            bool keep = jf < n;
            if (quick && jf < n) {
                const FpQuick& q = s_quick[jf];
                const float ac = __builtin_fmaf(dcx, q.A[0], __builtin_fmaf(dcy, q.A[1], dcz * q.A[2]));
                const float bc = __builtin_fmaf(dcx, q.Bv[0], __builtin_fmaf(dcy, q.Bv[1], dcz * q.Bv[2]));
                const float nc = __builtin_fmaf(dcx, q.Nn[0], __builtin_fmaf(dcy, q.Nn[1], dcz * q.Nn[2]));
                const float ar = dhx * fabsf(q.A[0]) + dhy * fabsf(q.A[1]) + dhz * fabsf(q.A[2]);
                const float br = dhx * fabsf(q.Bv[0]) + dhy * fabsf(q.Bv[1]) + dhz * fabsf(q.Bv[2]);
                const float nr = dhx * fabsf(q.Nn[0]) + dhy * fabsf(q.Nn[1]) + dhz * fabsf(q.Nn[2]);
                const float g0 = q.A[0] + q.Bv[0] - q.Nn[0], g1 = q.A[1] + q.Bv[1] - q.Nn[1], g2 = q.A[2] + q.Bv[2] - q.Nn[2];
                const float gc = __builtin_fmaf(dcx, g0, __builtin_fmaf(dcy, g1, dcz * g2));
                const float gr = dhx * fabsf(g0) + dhy * fabsf(g1) + dhz * fabsf(g2) + 1.0e-6f * (fabsf(q.A[0]) + fabsf(q.A[1]) + fabsf(q.A[2]) + fabsf(q.Bv[0]) + fabsf(q.Bv[1]) + fabsf(q.Bv[2]) + fabsf(q.Nn[0]) + fabsf(q.Nn[1]) + fabsf(q.Nn[2]));
                const float ma = 1.1f * q.ma, mb = 1.1f * q.mb, md = 1.1f * q.md, M = 1.1f * (q.ma + q.mb + q.md);
                bool rej = false;
                if (nc - nr > md) rej = (ac + ar < -ma) || (bc + br < -mb) || (gc - gr > M);          // the denominator positive all over the strip
                else if (nc + nr < -md) rej = (ac - ar > ma) || (bc - br > mb) || (gc + gr < -M);     // negative all over the strip
                keep = !rej;                                                                           // (a NaN anywhere keeps the face)
            }
            todo = __ballot(keep);
            // the faces the strip skips blend into none of its pixels
            if (jf < n && !keep) hit_masks[((int64_t)range.x + base + jf) * 4 + wid] = 0ull;
        }
        while (todo) {
            const int j = 64 * sub + (__ffsll((long long)todo) - 1);
            todo &= todo - 1ull;
            contributor = (uint32_t)(base + j) + 1u;                    // the entry's 1-based position in the tile's list (forward.cu:310)
            if (quick) {
                // numerators and denominator of (u, v) up to rounding, and how far the reference's own evaluation can be from them
                const FpQuick& q = s_quick[j];
                const float na = __builtin_fmaf(rd.x, q.A[0], __builtin_fmaf(rd.y, q.A[1], rd.z * q.A[2]));
                const float nb = __builtin_fmaf(rd.x, q.Bv[0], __builtin_fmaf(rd.y, q.Bv[1], rd.z * q.Bv[2]));
                const float dn = __builtin_fmaf(rd.x, q.Nn[0], __builtin_fmaf(rd.y, q.Nn[1], rd.z * q.Nn[2]));
                const float ma = q.ma, mb = q.mb, md = q.md;
                // certainly outside: u < 0 or v < 0 or u + v > 1 by more than the bound, for the sign of the denominator at hand
                const float sg = dn < 0.0f ? -1.0f : 1.0f;
                const float a = sg * na, c = sg * nb, dd = fabsf(dn);
                const bool out = (dd > md) && ((a < -ma) || (c < -mb) || (a + c - dd > ma + mb + md));
                if (__ballot(!done && !out) == 0ull) {                  // (wave-uniform) nobody near this face
                    if (lane == 0) hit_masks[((int64_t)range.x + base + j) * 4 + wid] = 0ull;
                    continue;
                }
            }
            const FaceRec& fc = recs[j];
            const f3 p0 = {fc.v[0], fc.v[1], fc.v[2]}, p1 = {fc.v[3], fc.v[4], fc.v[5]}, p2 = {fc.v[6], fc.v[7], fc.v[8]};
            f3 tuv = {0, 0, 0};
            const bool ok = ray_tri_intersection(ro, rd, p0, p1, p2, tuv);
            // At temperature 0 the coverage is 1 where the clamped barycentrics say "inside" (clamp_bary_uv's first region,
            // auxiliary.h:294: code 0, u_c = u, v_c = v) and 0 everywhere else (forward.cu:375-381: mix_coverage(code, 0, 0)):
            // the other six regions and the double-precision mix need not be evaluated to know that.
            const float iuc = tuv.y, ivc = tuv.z;
            const float ratio = 1.0f;
            const bool hit = !done && ok && (iuc >= 0.0f) && (ivc >= 0.0f) && (iuc + ivc <= 1.0f);
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) hit_masks[((int64_t)range.x + base + j) * 4 + wid] = bal;
            if (hit) {
                const float i0 = 1 - iuc - ivc, i1 = iuc, i2 = ivc;
                float c0 = i0 * fc.col[0] + i1 * fc.col[3] + i2 * fc.col[6];
                float c1 = i0 * fc.col[1] + i1 * fc.col[4] + i2 * fc.col[7];
                float c2 = i0 * fc.col[2] + i1 * fc.col[5] + i2 * fc.col[8];
                c0 = c0 * fc.intense; c1 = c1 * fc.intense; c2 = c2 * fc.intense;
                const float iD = i0 * fc.dep[0] + i1 * fc.dep[1] + i2 * fc.dep[2];
                const float alpha = fc.opacity * ratio;
                const float test_T = T * (1 - alpha);
                C0 += c0 * alpha * T; C1 += c1 * alpha * T; C2 += c2 * alpha * T;
                D += iD * alpha * T;
                pT = T; T = test_T;
                last_contributor = contributor;
                if (T < T_EPS) done = true;
            }
        }
        }   // sub-chunks of 64 faces
    }

    // the tile's largest n_contrib: where the backward's walk starts (dm2_backward_fast.hip)
    __shared__ uint32_t s_maxlc;
    __syncthreads();
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
        if (out_tri_cnt) out_tri_cnt[pix] = 0;                     // no AA records at temperature 0 (K is forced to 0)
    }
}

void launch_render_forward_point(const dm2_render_desc& d, const uint2* ranges, const uint32_t* face_list, ImageState is,
                                 float* out_color, float* out_depth, int32_t* out_tri_cnt, uint64_t* hit_masks,
                                 uint32_t* hit_valid, hipStream_t st) {
    const dim3 grid(tile_grid_blocks((uint32_t)(((d.W + TILE - 1) / TILE) * ((d.H + TILE - 1) / TILE) * d.B)));
    StageTimer tm(ST_FWD, st);
    hipLaunchKernelGGL(k_render_forward_point, grid, dim3(TILE_PIX), 0, st, d, ranges, face_list, is, out_color, out_depth,
                       out_tri_cnt, hit_masks, hit_valid);
}

}  // namespace dm2
