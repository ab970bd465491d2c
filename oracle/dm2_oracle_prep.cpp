// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// CPU restatement of the reference's Python host prep:
//   projection   compute_verts_ndc_image   dmesh2_renderer/__init__.py:239-262
//   AA tables    Triangles                 dmesh2_renderer/pyrenderer.py:6-30, order_ccw :521-529, tri_area :531-535
// and of the gradient torch autograd sends back through them to `verts`
// (only `verts_ndc` and `aa_face_verts` receive gradients from the op, render.cu:372;
// the other five tables are treated as constants by the reference).
//
// Parity status: PINNED by reference-produced vectors -- tests/golden/boundary_*.npz hold
// the reference Python's own verts_ndc / verts_image / six tables for the same inputs
// (tests/test_oracle_prep.py).  The tables are bit-exact given verts_image; the projection
// is a 4x4 matmul whose summation order inside the BLAS the reference calls is not defined,
// so verts_ndc / verts_image are compared at 1e-6 relative; here the order is k = 0..3 with
// separate multiply and add.
//
// Templated on R: float is the oracle, double validates the backward by finite differences.
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

template <class R> struct Clip { R c[4]; R w; bool clamped; };

template <class R>
inline Clip<R> project(const R* v, const R* mv, const R* proj) {
    const R hom[4] = {v[0], v[1], v[2], R(1)};
    R t[4];
    for (int j = 0; j < 4; j++) {                       // hom @ mv^T
        R s = hom[0] * mv[4 * j];
        for (int k = 1; k < 4; k++) s = s + hom[k] * mv[4 * j + k];
        t[j] = s;
    }
    Clip<R> o;
    for (int j = 0; j < 4; j++) {                       // (.) @ proj^T
        R s = t[0] * proj[4 * j];
        for (int k = 1; k < 4; k++) s = s + t[k] * proj[4 * j + k];
        o.c[j] = s;
    }
    // |w| clamp, sign kept (__init__.py:254-255)
    const R eps = R(1e-4f);
    R w = o.c[3];
    o.clamped = false;
    if (w >= R(0) && w < eps) { w = eps; o.clamped = true; }
    if (w < R(0) && w > -eps) { w = -eps; o.clamped = true; }
    o.w = w;
    return o;
}

template <class R>
void prepare(int B, int P, int F, int W, int H, const R* verts, const int* faces, const R* mv, const R* proj,
             R* ndc, R* image, R* aa_verts, R* aa_edges, uint8_t* aa_iszero, R* aa_recip, R* aa_normal, R* aa_normal_c) {
    for (int b = 0; b < B; b++) {
        for (int p = 0; p < P; p++) {
            const Clip<R> c = project<R>(verts + 3 * p, mv + 16 * b, proj + 16 * b);
            R* n = ndc + ((int64_t)b * P + p) * 3;
            n[0] = c.c[0] / c.w; n[1] = c.c[1] / c.w; n[2] = c.c[2] / c.w;
            R* im = image + ((int64_t)b * P + p) * 2;
            im[0] = ((n[0] + R(1)) * R(0.5)) * R(W);      // __init__.py:258-260
            im[1] = ((n[1] + R(1)) * R(0.5)) * R(H);
        }
        for (int f = 0; f < F; f++) {
            const R* im = image + (int64_t)b * P * 2;
            const R* p0 = im + 2 * (int64_t)faces[3 * f];
            const R* p1 = im + 2 * (int64_t)faces[3 * f + 1];
            const R* p2 = im + 2 * (int64_t)faces[3 * f + 2];
            // tri_area / order_ccw (pyrenderer.py:521-535): swap corners 1 and 2 of clockwise triangles
            const R area = R(0.5) * ((p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]));
            const bool flip = area < R(0);
            const R* q[3] = {p0, flip ? p2 : p1, flip ? p1 : p2};
            const int64_t o = ((int64_t)b * F + f) * 6;
            for (int i = 0; i < 3; i++) {
                const R* s = q[i];
                const R* e = q[(i + 1) % 3];
                const R ex = e[0] - s[0], ey = e[1] - s[1];
                aa_verts[o + 2 * i] = s[0]; aa_verts[o + 2 * i + 1] = s[1];
                aa_edges[o + 2 * i] = ex; aa_edges[o + 2 * i + 1] = ey;
                aa_iszero[o + 2 * i] = std::fabs(ex) < R(1e-3f); aa_iszero[o + 2 * i + 1] = std::fabs(ey) < R(1e-3f);
                aa_recip[o + 2 * i] = R(1) / ex; aa_recip[o + 2 * i + 1] = R(1) / ey;
                const R nx = -ey, ny = ex;
                aa_normal[o + 2 * i] = nx; aa_normal[o + 2 * i + 1] = ny;
                aa_normal_c[((int64_t)b * F + f) * 3 + i] = nx * s[0] + ny * s[1];
            }
        }
    }
}

// d(verts) from d(verts_ndc) (B,P,3), d(verts_image) (B,P,2) and d(aa_face_verts) (B,F,3,2); any may be null.
template <class R>
void prepare_backward(int B, int P, int F, int W, int H, const R* verts, const int* faces, const R* mv, const R* proj,
                      const R* g_ndc, const R* g_image, const R* g_aa, R* g_verts) {
    std::vector<R> gi((size_t)B * P * 2, R(0));
    std::vector<R> image((size_t)P * 2);
    for (int64_t i = 0; i < (int64_t)P * 3; i++) g_verts[i] = R(0);
    for (int b = 0; b < B; b++) {
        for (int p = 0; p < P; p++) {
            const Clip<R> c = project<R>(verts + 3 * p, mv + 16 * b, proj + 16 * b);
            image[2 * p] = ((c.c[0] / c.w + R(1)) * R(0.5)) * R(W);
            image[2 * p + 1] = ((c.c[1] / c.w + R(1)) * R(0.5)) * R(H);
        }
        R* gib = gi.data() + (size_t)b * P * 2;
        if (g_image) for (int64_t i = 0; i < (int64_t)P * 2; i++) gib[i] = g_image[(int64_t)b * P * 2 + i];
        if (g_aa) {
            for (int f = 0; f < F; f++) {
                const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
                const R* p0 = &image[2 * (size_t)v0]; const R* p1 = &image[2 * (size_t)v1]; const R* p2 = &image[2 * (size_t)v2];
                const R area = R(0.5) * ((p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]));
                const bool flip = area < R(0);
                const int dst[3] = {v0, flip ? v2 : v1, flip ? v1 : v2};      // un-permute the CCW reorder
                const R* g = g_aa + ((int64_t)b * F + f) * 6;
                for (int i = 0; i < 3; i++) { gib[2 * (size_t)dst[i]] += g[2 * i]; gib[2 * (size_t)dst[i] + 1] += g[2 * i + 1]; }
            }
        }
        for (int p = 0; p < P; p++) {
            const R* M = mv + 16 * b; const R* Pm = proj + 16 * b;
            const Clip<R> c = project<R>(verts + 3 * p, M, Pm);
            R gn[3] = {R(0), R(0), R(0)};
            if (g_ndc) for (int i = 0; i < 3; i++) gn[i] = g_ndc[((int64_t)b * P + p) * 3 + i];
            gn[0] += (gib[2 * p] * R(W)) * R(0.5);
            gn[1] += (gib[2 * p + 1] * R(H)) * R(0.5);
            R gc[4];
            R gw = R(0);
            for (int i = 0; i < 3; i++) {
                gc[i] = gn[i] / c.w;
                gw += -gn[i] * c.c[i] / (c.w * c.w);
            }
            gc[3] = c.clamped ? R(0) : gw;                 // torch.where passes no gradient into the clamped branch
            R gt[4], gh[4];
            for (int k = 0; k < 4; k++) { R s = R(0); for (int j = 0; j < 4; j++) s += gc[j] * Pm[4 * j + k]; gt[k] = s; }
            for (int k = 0; k < 4; k++) { R s = R(0); for (int j = 0; j < 4; j++) s += gt[j] * M[4 * j + k]; gh[k] = s; }
            for (int i = 0; i < 3; i++) g_verts[3 * p + i] += gh[i];
        }
    }
}

}  // namespace

extern "C" {

#define ORC_PREP_API(SUF, R)                                                                                           \
    void orc_prepare_##SUF(int B, int P, int F, int W, int H, const R* verts, const int* faces, const R* mv,            \
                           const R* proj, R* ndc, R* image, R* aa_verts, R* aa_edges, uint8_t* aa_iszero, R* aa_recip,  \
                           R* aa_normal, R* aa_normal_c) {                                                             \
        prepare<R>(B, P, F, W, H, verts, faces, mv, proj, ndc, image, aa_verts, aa_edges, aa_iszero, aa_recip,          \
                   aa_normal, aa_normal_c);                                                                            \
    }                                                                                                                  \
    void orc_prepare_backward_##SUF(int B, int P, int F, int W, int H, const R* verts, const int* faces, const R* mv,   \
                                    const R* proj, const R* g_ndc, const R* g_image, const R* g_aa, R* g_verts) {      \
        prepare_backward<R>(B, P, F, W, H, verts, faces, mv, proj, g_ndc, g_image, g_aa, g_verts);                      \
    }

ORC_PREP_API(f32, float)
ORC_PREP_API(f64, double)

}  // extern "C"
