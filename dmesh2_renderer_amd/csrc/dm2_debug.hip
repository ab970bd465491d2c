// dm2_debug.hip -- test hooks: run the device clippers on caller-supplied (triangle tables, pixel) pairs so that
// the reference's own AA vectors (tests/golden/aa_pairs.npz, aa_error_pairs.npz <- pyrenderer.py:207-425) reach
// every clipper variant directly, not only through whole-frame comparisons.
#include <hip/hip_runtime.h>

#include "dm2_clip_area.h"
#include "dm2_clip_fast.h"
#include "dm2_clip_seg.h"
#include "dm2_device_math.h"
#include "dm2_state.h"

namespace dm2 {

__global__ void __launch_bounds__(64)
k_debug_aa_overlap(int variant, int64_t n, const float* __restrict__ tv, const float* __restrict__ te,
                   const uint8_t* __restrict__ tz, const float* __restrict__ tr, const float* __restrict__ tn,
                   const float* __restrict__ tc, const float* __restrict__ pixmin, float* __restrict__ area,
                   float* __restrict__ grad, int32_t* __restrict__ code) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AAFace f;
    uint32_t zm = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        f.v[k] = tv[6 * i + k]; f.e[k] = te[6 * i + k]; f.r[k] = tr[6 * i + k]; f.n[k] = tn[6 * i + k];
        zm |= (tz[6 * i + k] ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) f.c[k] = tc[3 * i + k];
    f.zmask = zm;
    f.bb[0] = fminf(fminf(f.v[0], f.v[2]), f.v[4]); f.bb[1] = fmaxf(fmaxf(f.v[0], f.v[2]), f.v[4]);
    f.bb[2] = fminf(fminf(f.v[1], f.v[3]), f.v[5]); f.bb[3] = fmaxf(fmaxf(f.v[1], f.v[3]), f.v[5]);
    const float pxmin = pixmin[2 * i], pymin = pixmin[2 * i + 1], pxmax = pxmin + 1, pymax = pymin + 1;
    float a = 0.f, g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int err = 0;
    if (variant == 0) {                // legacy kernels: generic clipper with the reference's per-fan-triangle Jacobians
        err = tri_pix_overlap_area<true>(f, pxmin, pxmax, pymin, pymax, 1.0f, a, g);
    } else if (variant == 1) {         // forward (dm2_forward_queue.hip): bbox reject, classification, straight-line area clipper
        err = tri_pix_overlap_area_only(f, pxmin, pxmax, pymin, pymax, 1.0f, a);
    } else if (variant == 4) {         // backward (dm2_backward_fast.hip): the forward's decision and area, the Jacobian without a polygon;
                                       // code -1: the pair is a tie and goes to the segment formulation instead (grad zeroed here)
        err = tri_pix_overlap_area_only(f, pxmin, pxmax, pymin, pymax, 1.0f, a);
        if (err == 0 && a != 0.0f) {
            bool tie;
            fast_area_grad(f, pxmin, pxmax, pymin, pymax, g, tie);
            if (tie) { err = -1; for (int k = 0; k < 6; k++) g[k] = 0.f; }
        }
    } else {                           // backward (dm2_backward_mask.hip): the forward's decision, then the segment formulation
        float af = 0.f;
        err = tri_pix_overlap_area_only(f, pxmin, pxmax, pymin, pymax, 1.0f, af);
        if (err == 0 && af != 0.0f) seg_area_grad(f, pxmin, pxmax, pymin, pymax, 1.0f, a, g, variant == 3);   // 3: with the exact fan-sum area
    }
    if (err > 0) { a = 0.f; for (int k = 0; k < 6; k++) g[k] = 0.f; }
    area[i] = a; code[i] = err;
    for (int k = 0; k < 6; k++) grad[6 * i + k] = g[k];
}

void launch_debug_aa_overlap(int variant, int64_t n, const float* tv, const float* te, const uint8_t* tz, const float* tr,
                             const float* tn, const float* tc, const float* pixmin, float* area, float* grad, int32_t* code,
                             hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_debug_aa_overlap, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, variant, n, tv, te, tz, tr, tn, tc,
                       pixmin, area, grad, code);
}

}  // namespace dm2
