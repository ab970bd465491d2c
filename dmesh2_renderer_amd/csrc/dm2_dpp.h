// dm2_dpp.h -- DPP row-shift helpers and the segmented row scan used by the backward kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dm2 {

// DPP row shifts (16-lane rows; lanes shifted in from outside the row read 0)
template <int N> __device__ __forceinline__ int dpp_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true); }
template <int N> __device__ __forceinline__ int dpp_shl_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
template <int N> __device__ __forceinline__ float dpp_shr_f(float v) { return __int_as_float(dpp_shr_i<N>(__float_as_int(v))); }
// inclusive segmented sum inside a row of 16 lanes; sK = "lane l-K belongs to the same run"
// (written as "add the shifted value, then keep the sum where the lane continues a run" so that the compiler can fold
// the DPP move into the add: v_add_f32_dpp + v_cndmask per step instead of v_mov_dpp + v_cndmask + v_add)
__device__ __forceinline__ void seg_scan16(float& v, bool s1, bool s2, bool s4, bool s8) {
    float t = v + dpp_shr_f<1>(v); v = s1 ? t : v;
    t = v + dpp_shr_f<2>(v); v = s2 ? t : v;
    t = v + dpp_shr_f<4>(v); v = s4 ? t : v;
    t = v + dpp_shr_f<8>(v); v = s8 ? t : v;
}

// The same segmented scan for N values at once with ONE instruction per value and step: v_fmac_f32 with a DPP source,
//     v += shifted(v) * mK,      mK = 1.0 where lane l - K continues the lane's run, else 0.0
// (t * 1.0 + v rounds once, exactly like t + v; lanes shifted in from outside the row read 0).  Inline assembly: the
// compiler has no pattern that folds a select into a DPP multiply-add.  A DPP read needs two wait states behind the VALU
// write of its source; the steps of one value are N >= 3 instructions apart, and an s_nop covers the producers.
// Requires finite values (0 * inf would poison a lane that is not part of the run).
template <int N>
__device__ __forceinline__ void seg_scan16_n(float (&g)[N], float m1, float m2, float m4, float m8) {
    static_assert(N >= 3, "the steps of one value must be at least three instructions apart");
    asm volatile("s_nop 1");
#pragma unroll
    for (int c = 0; c < N; c++) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(g[c]) : "v"(m1));
#pragma unroll
    for (int c = 0; c < N; c++) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(g[c]) : "v"(m2));
#pragma unroll
    for (int c = 0; c < N; c++) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(g[c]) : "v"(m4));
#pragma unroll
    for (int c = 0; c < N; c++) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(g[c]) : "v"(m8));
}

// seg_scan16_n for values that may not be finite (a sliver face whose determinant underflows, a NaN vertex): the
// multiply-by-0 continuation mask of the fast form would turn a neighbouring run's Inf into NaN in THIS run, so a wave in
// which any lane holds a non-finite value takes the select form (two instructions per value and step), which keeps a
// non-finite gradient inside the rows of the face that produced it, as the reference's per-pair atomics do.
template <int N>
__device__ __forceinline__ void seg_scan16_safe(float (&g)[N], float m1, float m2, float m4, float m8, bool s1, bool s2, bool s4, bool s8) {
    float mag = 0.f;
#pragma unroll
    for (int c = 0; c < N; c++) mag += fabsf(g[c]);
    if (__builtin_expect(__ballot(!(mag < 3.0e38f)) == 0ull, 1)) { seg_scan16_n(g, m1, m2, m4, m8); return; }
#pragma unroll
    for (int c = 0; c < N; c++) seg_scan16(g[c], s1, s2, s4, s8);
}

}  // namespace dm2
