// dm2_bwd_shared.h -- constants, pair record and small helpers of the mask-driven backward kernel (dm2_backward_mask.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dm2 {

#ifndef DM2_BM_CARRY
#define DM2_BM_CARRY 2         // phase D takes ray, corners, colours, NDC z of its pair from phase B2 in registers (0: re-reads LDS; A/B at cfg4: -3.7 %)
#endif
#ifndef DM2_BM_ACC
#define DM2_BM_ACC 33       // odd pitch: the emit lanes of different faces add to different LDS banks (A/B at cfg4: 32 -> 33, -0.4 %)
#endif
constexpr int BM_ACC = DM2_BM_ACC;       // pitch of an accumulator row (dwords)
// accumulator row of one list entry: d/dverts (9), d/dverts_color (9), d/dndc.z (3), d/dopacity, d/dintense, d/daa (6); flag
constexpr int M_DV = 0, M_DC = 9, M_DZ = 18, M_OP = 21, M_IN = 22, M_AA = 23, M_N = 29, M_FLAG = 31;
constexpr uint32_t MB_BLEND = 1u, MB_ACTIVE = 2u;

struct __attribute__((aligned(16))) BmPair { float alpha, c0, c1, c2, depth; uint32_t flags; float T, dL_dalpha; };
static_assert(sizeof(BmPair) == 32, "BmPair");
// dm2_backward_fast.hip: phase B2 leaves (alpha, colour, depth), phase C replaces alpha by dL/dalpha and depth by the T in
// front of the pair and sets the flags; phase D reads the pixel's loss gradients from the per-pixel LDS rows itself
struct __attribute__((aligned(8))) BfPair { float alpha, c0, c1, c2, depth; uint32_t flags; };
static_assert(sizeof(BfPair) == 24, "BfPair");

// index of the n-th (0-based) set bit of m; n < popcount(m)
__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int n) {
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const int cl = __popc(lo);
    const bool up = n >= cl;
    const uint32_t w = up ? hi : lo;
    n = up ? n - cl : n;
    int pos = 0;
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const int c = __popc((w >> pos) & ((1u << s) - 1u));
        if (n >= c) { n -= c; pos += s; }
    }
    return pos + (up ? 32 : 0);
}

// Flush table: component comp of an accumulator row goes to  base + 4 * (id * mult),  id one of the record's
// (face_id, vid[0..2]) -- filled once per block by the lanes comp < M_N.
// aa_to_verts (DM2_FLAG_AA_GRAD_TO_VERTS): dL_daa_face_verts is a (B,P,2) array and corner c of the record goes to the
// vertex the CCW reorder took it from: entry = corner | 0x80, resolved per record by flush_id_and_mult.
template <class SelT>
__device__ __forceinline__ void fill_flush_table(int comp, int b, int P, int F, float* dL_dverts, float* dL_dverts_color,
                                                 float* dL_dfaces_opacity, float* dL_dverts_ndc, float* dL_dfaces_intense,
                                                 float* dL_daa_face_verts, float** fl_base, SelT* fl_sel, bool aa_to_verts = false) {
    if (aa_to_verts && comp >= M_AA) {
        const int within = comp - M_AA;
        fl_base[comp] = dL_daa_face_verts + (int64_t)b * P * 2 + (within & 1);
        fl_sel[comp] = (SelT)((within >> 1) | 0x80);
        return;
    }
    const int g = (comp >= M_DC) + (comp >= M_DZ) + (comp >= M_OP) + (comp >= M_IN) + (comp >= M_AA);   // 0..5: dverts, dcolor, dndc.z, dopacity, dintense, daa
    const int within = comp - (g == 0 ? M_DV : g == 1 ? M_DC : g == 2 ? M_DZ : g == 3 ? M_OP : g == 4 ? M_IN : M_AA);
    const int sel = g < 2 ? 1 + within / 3 : (g == 2 ? 1 + within : 0);
    const int mult = g < 3 ? 3 : (g == 5 ? 6 : 1);
    const int64_t add = g < 2 ? (int64_t)(within % 3) : g == 2 ? (int64_t)b * P * 3 + 2 : g == 3 ? (int64_t)0
                      : g == 4 ? (int64_t)b * F : (int64_t)b * F * 6 + within;
    fl_base[comp] = (g == 0 ? dL_dverts : g == 1 ? dL_dverts_color : g == 2 ? dL_dverts_ndc : g == 3 ? dL_dfaces_opacity
                    : g == 4 ? dL_dfaces_intense : dL_daa_face_verts) + add;
    fl_sel[comp] = (SelT)(sel | (mult << 2));
}

// id selector (0: face_id, 1..3: vid[0..2]) and dwords per id of a flush-table entry, for a record with edge-flag word zmask
__device__ __forceinline__ void flush_id_and_mult(int entry, uint32_t zmask, int& sel, int& mult) {
    if (entry & 0x80) {                                   // an AA corner on its way to its vertex
        const int c = entry & 3;
        const bool flip = (zmask >> 8) & 1u;
        sel = 1 + (c == 0 ? 0 : (flip ? 3 - c : c));
        mult = 2;
    } else { sel = entry & 3; mult = entry >> 2; }
}

}  // namespace dm2
