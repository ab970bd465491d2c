// dm2_prep.hip -- the host prep of Renderer.forward, fused.
//
// The reference prepares the op's inputs with ~20 torch kernels (and as many again in autograd's
// backward): projection `hom @ mv^T @ proj^T`, |w| clamp, NDC -> image units
// (dmesh2_renderer/__init__.py:239-262), then per face the CCW reorder and the six AA tables
// (pyrenderer.py:6-30, 521-535).  Every table is a pure function of three image-space corners,
// so one pass per vertex and one per face produce all eight tensors; all of it is HBM streaming:
//
//   k_project        12 B read, 20 B written per (view, vertex)
//   k_aa_tables      12 B + 3 gathers of 8 B read, 114 B written per (view, face)
//   k_aa_scatter     24 B read, 6 atomics per (view, face)               (backward)
//   k_project_bwd    12 B + B * 20 B read, 12 B written per vertex      (backward)
//
// Arithmetic: the operation order of the oracle (oracle/dm2_oracle_prep.cpp), no FMA contraction:
// tables are bit-identical to torch's element-wise results for the same verts_image; the two 4x4
// products are summed k = 0..3 (the BLAS order of the reference run is not defined).
#include <hip/hip_runtime.h>

#include "dm2_state.h"

namespace dm2 {

namespace {

constexpr float W_EPS = 1e-4f;      // __init__.py:254-255

struct ClipPt { float c0, c1, c2, w; bool clamped; };

// mv, proj: this view's matrices (block-uniform -> scalar loads)
__device__ __forceinline__ ClipPt project_vertex(float x, float y, float z, const float* __restrict__ mv, const float* __restrict__ proj) {
    float t[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = ((x * mv[4 * j] + y * mv[4 * j + 1]) + z * mv[4 * j + 2]) + 1.0f * mv[4 * j + 3];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = ((t[0] * proj[4 * j] + t[1] * proj[4 * j + 1]) + t[2] * proj[4 * j + 2]) + t[3] * proj[4 * j + 3];
    ClipPt o;
    float w = c[3];
    o.clamped = false;
    if (w >= 0.0f && w < W_EPS) { w = W_EPS; o.clamped = true; }
    if (w < 0.0f && w > -W_EPS) { w = -W_EPS; o.clamped = true; }
    o.c0 = c[0]; o.c1 = c[1]; o.c2 = c[2]; o.w = w;
    return o;
}

__device__ __forceinline__ float2 image_of(const ClipPt& c, float Wf, float Hf) {
    const float nx = c.c0 / c.w, ny = c.c1 / c.w;
    return make_float2(((nx + 1.0f) * 0.5f) * Wf, ((ny + 1.0f) * 0.5f) * Hf);      // __init__.py:258-260
}

__global__ void __launch_bounds__(256)
k_project(int P, float Wf, float Hf, const float* __restrict__ verts, const float* __restrict__ mv,
          const float* __restrict__ proj, float* __restrict__ ndc, float* __restrict__ image) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const ClipPt c = project_vertex(verts[3 * (int64_t)p], verts[3 * (int64_t)p + 1], verts[3 * (int64_t)p + 2], mv + 16 * b, proj + 16 * b);
    const int64_t o = (int64_t)b * P + p;
    const float nx = c.c0 / c.w, ny = c.c1 / c.w, nz = c.c2 / c.w;
    if (ndc) { ndc[3 * o] = nx; ndc[3 * o + 1] = ny; ndc[3 * o + 2] = nz; }
    if (image) reinterpret_cast<float2*>(image)[o] = make_float2(((nx + 1.0f) * 0.5f) * Wf, ((ny + 1.0f) * 0.5f) * Hf);
}

// CCW decision of pyrenderer.py:521-535: swap corners 1 and 2 when the signed area is negative
__device__ __forceinline__ bool is_clockwise(float2 p0, float2 p1, float2 p2) {
    const float area = 0.5f * ((p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y));
    return area < 0.0f;
}

__global__ void __launch_bounds__(256)
k_aa_tables(int P, int F, const int32_t* __restrict__ faces, const float* __restrict__ image,
            float* __restrict__ aa_verts, float* __restrict__ aa_edges, uint8_t* __restrict__ aa_iszero,
            float* __restrict__ aa_recip, float* __restrict__ aa_normal, float* __restrict__ aa_normal_c) {
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float2* im = reinterpret_cast<const float2*>(image) + (int64_t)b * P;
    const float2 p0 = im[faces[3 * (int64_t)f]], p1 = im[faces[3 * (int64_t)f + 1]], p2 = im[faces[3 * (int64_t)f + 2]];
    const bool flip = is_clockwise(p0, p1, p2);
    const float2 q[3] = {p0, flip ? p2 : p1, flip ? p1 : p2};
    const int64_t bf = (int64_t)b * F + f;
    float2* ov = reinterpret_cast<float2*>(aa_verts) + bf * 3;
    float2* oe = reinterpret_cast<float2*>(aa_edges) + bf * 3;
    float2* orc = reinterpret_cast<float2*>(aa_recip) + bf * 3;
    float2* on = reinterpret_cast<float2*>(aa_normal) + bf * 3;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float2 s = q[i], e = q[(i + 1) % 3];
        const float ex = e.x - s.x, ey = e.y - s.y;
        const float nx = -ey, ny = ex;
        if (aa_verts) ov[i] = s;
        if (aa_edges) oe[i] = make_float2(ex, ey);
        if (aa_iszero) { aa_iszero[bf * 6 + 2 * i] = fabsf(ex) < 1e-3f; aa_iszero[bf * 6 + 2 * i + 1] = fabsf(ey) < 1e-3f; }
        if (aa_recip) orc[i] = make_float2(1.0f / ex, 1.0f / ey);
        if (aa_normal) on[i] = make_float2(nx, ny);
        if (aa_normal_c) aa_normal_c[bf * 3 + i] = nx * s.x + ny * s.y;
    }
}

// backward, step 1: d(aa_face_verts) -> d(verts_image), undoing the CCW reorder
__global__ void __launch_bounds__(256)
k_aa_scatter(int P, int F, float Wf, float Hf, const float* __restrict__ verts, const int32_t* __restrict__ faces,
             const float* __restrict__ mv, const float* __restrict__ proj, const float* __restrict__ g_aa,
             float* __restrict__ g_image) {
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float2* g = reinterpret_cast<const float2*>(g_aa) + ((int64_t)b * F + f) * 3;
    const float2 g0 = g[0], g1 = g[1], g2 = g[2];
    // a face the rasteriser never listed (culled, outside the patch or -- multi-GPU -- outside this rank's band) has an
    // all-zero row: nothing to scatter, and no need to project its corners to find their order
    if (g0.x == 0.f && g0.y == 0.f && g1.x == 0.f && g1.y == 0.f && g2.x == 0.f && g2.y == 0.f) return;
    const int v0 = faces[3 * (int64_t)f], v1 = faces[3 * (int64_t)f + 1], v2 = faces[3 * (int64_t)f + 2];
    const int vs[3] = {v0, v1, v2};
    float2 p[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float* v = verts + 3 * (int64_t)vs[i];
        p[i] = image_of(project_vertex(v[0], v[1], v[2], mv + 16 * b, proj + 16 * b), Wf, Hf);
    }
    const bool flip = is_clockwise(p[0], p[1], p[2]);
    const int dst[3] = {v0, flip ? v2 : v1, flip ? v1 : v2};
    float* gi = g_image + (int64_t)b * P * 2;
    const float2 gs[3] = {g0, g1, g2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float2 gv = gs[i];
        atomicAdd(gi + 2 * (int64_t)dst[i], gv.x);
        atomicAdd(gi + 2 * (int64_t)dst[i] + 1, gv.y);
    }
}

// backward, step 2: d(verts_ndc) + d(verts_image) -> d(verts), summed over the views in-thread
__global__ void __launch_bounds__(256)
k_project_bwd(int B, int P, float Wf, float Hf, const float* __restrict__ verts, const float* __restrict__ mv,
              const float* __restrict__ proj, const float* __restrict__ g_ndc, const float* __restrict__ g_image_a,
              const float* __restrict__ g_image_b, float* __restrict__ g_verts) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float x = verts[3 * (int64_t)p], y = verts[3 * (int64_t)p + 1], z = verts[3 * (int64_t)p + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < B; b++) {
        const float* M = mv + 16 * b;
        const float* Pm = proj + 16 * b;
        const ClipPt c = project_vertex(x, y, z, M, Pm);
        const int64_t o = (int64_t)b * P + p;
        float gn0 = 0.f, gn1 = 0.f, gn2 = 0.f;
        if (g_ndc) { gn0 = g_ndc[3 * o]; gn1 = g_ndc[3 * o + 1]; gn2 = g_ndc[3 * o + 2]; }
        float gi0 = 0.f, gi1 = 0.f;
        if (g_image_a) { gi0 += g_image_a[2 * o]; gi1 += g_image_a[2 * o + 1]; }
        if (g_image_b) { gi0 += g_image_b[2 * o]; gi1 += g_image_b[2 * o + 1]; }
        gn0 += (gi0 * Wf) * 0.5f;
        gn1 += (gi1 * Hf) * 0.5f;
        const float w2 = c.w * c.w;
        float gc[4];
        gc[0] = gn0 / c.w; gc[1] = gn1 / c.w; gc[2] = gn2 / c.w;
        float gw = 0.f;
        gw += -gn0 * c.c0 / w2; gw += -gn1 * c.c1 / w2; gw += -gn2 * c.c2 / w2;
        gc[3] = c.clamped ? 0.f : gw;               // torch.where passes no gradient into the clamped branch
        float gt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) gt[k] = ((gc[0] * Pm[k] + gc[1] * Pm[4 + k]) + gc[2] * Pm[8 + k]) + gc[3] * Pm[12 + k];
        a0 += ((gt[0] * M[0] + gt[1] * M[4]) + gt[2] * M[8]) + gt[3] * M[12];
        a1 += ((gt[0] * M[1] + gt[1] * M[5]) + gt[2] * M[9]) + gt[3] * M[13];
        a2 += ((gt[0] * M[2] + gt[1] * M[6]) + gt[2] * M[10]) + gt[3] * M[14];
    }
    g_verts[3 * (int64_t)p] = a0; g_verts[3 * (int64_t)p + 1] = a1; g_verts[3 * (int64_t)p + 2] = a2;
}

}  // namespace

void launch_prepare_faces(const dm2_prep_desc& d, hipStream_t st) {
    const float Wf = (float)d.W, Hf = (float)d.H;
    if (d.P > 0 && d.B > 0 && (d.verts_ndc || d.verts_image)) {
        const dim3 grid((d.P + 255) / 256, d.B);
        hipLaunchKernelGGL(k_project, grid, dim3(256), 0, st, d.P, Wf, Hf, d.verts, d.mv, d.proj, d.verts_ndc, d.verts_image);
    }
    const bool tables = d.aa_face_verts || d.aa_face_edges || d.aa_face_edges_iszero || d.aa_face_edges_recip ||
                        d.aa_face_edges_normal || d.aa_face_edges_normal_c;
    if (d.F > 0 && d.B > 0 && tables) {
        const dim3 grid((d.F + 255) / 256, d.B);
        hipLaunchKernelGGL(k_aa_tables, grid, dim3(256), 0, st, d.P, d.F, d.faces, d.verts_image, d.aa_face_verts, d.aa_face_edges,
                           d.aa_face_edges_iszero, d.aa_face_edges_recip, d.aa_face_edges_normal, d.aa_face_edges_normal_c);
    }
}

void launch_prepare_faces_backward(const dm2_prep_desc& d, const float* g_ndc, const float* g_image, const float* g_aa,
                                   float* image_grad_scratch, float* g_verts, hipStream_t st) {
    const float Wf = (float)d.W, Hf = (float)d.H;
    const bool scatter = g_aa && d.F > 0 && d.B > 0 && d.P > 0;
    if (scatter) {
        (void)hipMemsetAsync(image_grad_scratch, 0, (size_t)d.B * d.P * 2 * sizeof(float), st);
        const dim3 grid((d.F + 255) / 256, d.B);
        hipLaunchKernelGGL(k_aa_scatter, grid, dim3(256), 0, st, d.P, d.F, Wf, Hf, d.verts, d.faces, d.mv, d.proj, g_aa, image_grad_scratch);
    }
    if (d.P > 0) {
        hipLaunchKernelGGL(k_project_bwd, dim3((d.P + 255) / 256), dim3(256), 0, st, d.B, d.P, Wf, Hf, d.verts, d.mv, d.proj, g_ndc,
                           g_image, scatter ? image_grad_scratch : nullptr, g_verts);
    }
}

}  // namespace dm2
