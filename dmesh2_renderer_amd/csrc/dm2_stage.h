// dm2_stage.h -- per-list-entry face record staged in LDS by the composite kernels.
//
// The reference stages 96-108 B per entry in shared memory and re-reads the AA
// tables from global memory for every (pixel,face) (forward.cu:228-243,314-317,
// aa.h:111-120,184-203).  Here everything a (pixel,face) evaluation needs is
// packed ONCE per (view,face) and forward into a 256-byte global record (by the
// preprocess kernel, dm2_binning.hip) and copied per (tile,entry) into a 240-byte
// LDS record; the per-pixel code reads it with ds_reads only.
#pragma once
#include "dm2_device_math.h"
#include "dm2_state.h"

namespace dm2 {

#ifndef DM2_FACEREC_PAD
#define DM2_FACEREC_PAD 1
#endif
struct __attribute__((aligned(16))) FaceRec {
    AAFace aa;          // 32 dwords
    float v[9];         // world-space corners
    float col[9];       // vertex colours
    float dep[3];       // NDC z of the corners
    float opacity, intense;
    int face_id;
    int vid[3];         // vertex ids (backward scatter)
    float pad[DM2_FACEREC_PAD];   // 240-B stride (60 dwords = -4 mod 32 banks): records of consecutive entries start 4 LDS
                                  // banks apart, so lanes that read the same field of different records do not collide
                                  // (256 B would be a 64-way conflict; 272 B, +4 banks, behaves the same and costs 32 B more)
};
static_assert(sizeof(FaceRec) == 236 + 4 * DM2_FACEREC_PAD && sizeof(FaceRec) % 16 == 0, "FaceRec layout");

// The default backward's shorter LDS record: of the AA tables only corners, edges, reciprocals and the edge flags (parts 0-4
// of the global record; the normals, their offsets and the bbox serve the forward's classification only), then world
// corners ... vertex ids (parts 8-14): 12 of the 16 sixteen-byte parts, 192 bytes.
#ifndef DM2_FACERECB_PAD
#define DM2_FACERECB_PAD 0
#endif
struct __attribute__((aligned(16))) FaceRecB {
    float v2[6], e[6], r[6];   // aa_face_verts / edges / recip  (named v2: `v` are the world-space corners below)
    uint32_t zmask;
    float c0_unused;
    float v[9];         // world-space corners
    float col[9];
    float dep[3];
    float opacity, intense;
    int face_id;
    int vid[3];
    float pad;
#if DM2_FACERECB_PAD
    float pad2[4];      // 208-B stride (52 dwords = 20 mod 32 banks): the same field of eight consecutive records lies in eight different
                        // banks (192 B = 16 mod 32: only two)
#endif
};
static_assert(sizeof(FaceRecB) == 192 + 16 * DM2_FACERECB_PAD, "FaceRecB layout");
constexpr int FACE_RECB_PARTS = 12 + DM2_FACERECB_PAD;       // (the pad is copied as a part of its own: LDS-direct destinations are lane-contiguous)
// global part (16 B) of the packed record that holds part rp of a FaceRecB
__device__ __forceinline__ int recb_src_part(int rp) { return rp < 5 ? rp : (rp < 12 ? rp + 3 : 15); }
// view of a FaceRecB's AA members under the names dm2_clip_fast.h uses
struct AAFaceB { const float* v; const float* e; const float* r; uint32_t zmask; };

// Copy the packed record of (view, face) `bf` into LDS (one lane per record: 15 loads of 16 bytes from two 128-byte
// lines; the kernels with a prefetch pipeline copy cooperatively instead, 16 lanes per record).
__device__ __forceinline__ void stage_face(const uint4* __restrict__ recs, int64_t bf, FaceRec& r) {
    const uint4* src = recs + bf * FACE_REC_U4;
    uint4* dst = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(FaceRec) / 16); k++) dst[k] = src[k];
}

// LDS-direct loads (global_load_lds_*): lane l's 16 (4) bytes at `gsrc` land at lds_base + 16 (4) * l; lds_base must be
// wave-uniform.  Written as inline assembly ON PURPOSE: hipcc tracks an LDS-direct load issued through its builtin as a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next LDS read of ANY array and of every barrier, which
// turns the prefetch into a synchronous copy.  Issued this way the load is invisible to the compiler's wait-count pass;
// the kernels wait for it themselves (lds_prefetch_wait(), then a workgroup barrier) before they read the destination
// buffer.  Such a kernel must not spill: a scratch reload would have to wait for every load issued before it.
// (M0 holds the destination; it is compiler-reserved, hence saved and restored inside the statement.)
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    const uint32_t dst = lds_offset_of(lds_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_base) {
    const uint32_t dst = lds_offset_of(lds_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// wait for every LDS-direct load this wave issued (a workgroup barrier must follow before another wave's part is read)
__device__ __forceinline__ void lds_prefetch_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Primary ray of patch pixel (px, py) of view b: read from the op's ray tensors, or computed (DM2_FLAG_ANALYTIC_RAYS)
template <class Desc>
__device__ __forceinline__ void pixel_ray(const Desc& d, int b, int64_t pix, uint32_t x_abs, uint32_t y_abs, int full_W, int full_H,
                                          f3& ro, f3& rd) {
    if (d.flags & DM2_FLAG_ANALYTIC_RAYS) {
        analytic_ray(d.ray_cam + 32 * b, (float)x_abs, (float)y_abs, (float)full_W, (float)full_H, ro, rd);
    } else {
        ro = {d.image_ray_o[3 * pix], d.image_ray_o[3 * pix + 1], d.image_ray_o[3 * pix + 2]};
        rd = {d.image_ray_d[3 * pix], d.image_ray_d[3 * pix + 1], d.image_ray_d[3 * pix + 2]};
    }
}

// Gather face `face_id` of view `b` from the op's input tensors into `r` (the preprocess kernel packs with it).
// iv1: verts_image of the face's second vertex (the caller has it at hand)
__device__ __forceinline__ void pack_face(const dm2_render_desc& d, int b, int face_id, float2 iv1, FaceRec& r) {
    const int v0 = d.faces[3 * face_id], v1 = d.faces[3 * face_id + 1], v2 = d.faces[3 * face_id + 2];
    r.face_id = face_id; r.vid[0] = v0; r.vid[1] = v1; r.vid[2] = v2;
    const int vs[3] = {v0, v1, v2};
    const float* ndc = d.verts_ndc + (int64_t)b * d.P * 3;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float* pv = d.verts + 3 * (int64_t)vs[i];
        const float* pc = d.verts_color + 3 * (int64_t)vs[i];
        r.v[3 * i] = pv[0]; r.v[3 * i + 1] = pv[1]; r.v[3 * i + 2] = pv[2];
        r.col[3 * i] = pc[0]; r.col[3 * i + 1] = pc[1]; r.col[3 * i + 2] = pc[2];
        r.dep[i] = ndc[3 * (int64_t)vs[i] + 2];
    }
    r.opacity = d.faces_opacity[face_id];
    const int64_t bf = (int64_t)b * d.F + face_id;
    r.intense = d.faces_intense[bf];
    uint32_t zm = 0;
    if (d.flags & DM2_FLAG_TABLES_FROM_IMAGE) {
        // The six AA tables are pure functions of the face's three image-space corners (pyrenderer.py:6-30, 521-535): built
        // here in registers, in the operation order of the fused host prep (dm2_prep.hip k_aa_tables: bit-identical tables),
        // instead of being written by the host prep (114 B per face) and read back here.
        const float2* im = reinterpret_cast<const float2*>(d.verts_image) + (int64_t)b * d.P;
        const float2 p0 = im[v0], p1 = im[v1], p2 = im[v2];
        const float area2 = 0.5f * ((p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y));
        const bool flip = area2 < 0.0f;                               // clockwise: corners 1 and 2 swap (pyrenderer.py:521-535)
        const float2 q[3] = {p0, flip ? p2 : p1, flip ? p1 : p2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float2 sv = q[i], ev = q[(i + 1) % 3];
            const float ex = ev.x - sv.x, ey = ev.y - sv.y;
            const float nx = -ey, ny = ex;
            r.aa.v[2 * i] = sv.x; r.aa.v[2 * i + 1] = sv.y;
            r.aa.e[2 * i] = ex; r.aa.e[2 * i + 1] = ey;
            r.aa.r[2 * i] = 1.0f / ex; r.aa.r[2 * i + 1] = 1.0f / ey;
            r.aa.n[2 * i] = nx; r.aa.n[2 * i + 1] = ny;
            r.aa.c[i] = nx * sv.x + ny * sv.y;
            zm |= (fabsf(ex) < 1e-3f ? 1u : 0u) << (2 * i);
            zm |= (fabsf(ey) < 1e-3f ? 1u : 0u) << (2 * i + 1);
        }
        if (flip) zm |= 1u << 8;                                      // (DM2_FLAG_AA_GRAD_TO_VERTS routes the corner gradients back with it)
    } else {
    const float2* av = reinterpret_cast<const float2*>(d.aa_face_verts + bf * 6);
    const float2* ae = reinterpret_cast<const float2*>(d.aa_face_edges + bf * 6);
    const float2* ar = reinterpret_cast<const float2*>(d.aa_face_edges_recip + bf * 6);
    const float2* an = reinterpret_cast<const float2*>(d.aa_face_edges_normal + bf * 6);
    const float* ac = d.aa_face_edges_normal_c + bf * 3;
    const uint8_t* az = d.aa_face_edges_iszero + bf * 6;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float2 a = av[i], e = ae[i], rr = ar[i], n = an[i];
        r.aa.v[2 * i] = a.x; r.aa.v[2 * i + 1] = a.y;
        r.aa.e[2 * i] = e.x; r.aa.e[2 * i + 1] = e.y;
        r.aa.r[2 * i] = rr.x; r.aa.r[2 * i + 1] = rr.y;
        r.aa.n[2 * i] = n.x; r.aa.n[2 * i + 1] = n.y;
        r.aa.c[i] = ac[i];
        zm |= (az[2 * i] ? 1u : 0u) << (2 * i);
        zm |= (az[2 * i + 1] ? 1u : 0u) << (2 * i + 1);
    }
    // bit 8: the CCW reorder swapped corners 1 and 2 (pyrenderer.py:8,521-529): aa corner 1 is not this face's vertex 1
    // (DM2_FLAG_AA_GRAD_TO_VERTS routes the corner gradients back with it; bits 0..5 are the edge flags).  Noted by every
    // forward, whatever its flags: a backward that asks for the routing never depends on what the forward was told.
    if (!(av[1].x == iv1.x && av[1].y == iv1.y)) zm |= 1u << 8;
    }
    r.aa.zmask = zm;
    // aa_face_verts.min(2)/.max(2)  (forward.cu:480-481, backward.cu:589-590)
    r.aa.bb[0] = fminf(fminf(r.aa.v[0], r.aa.v[2]), r.aa.v[4]);
    r.aa.bb[1] = fmaxf(fmaxf(r.aa.v[0], r.aa.v[2]), r.aa.v[4]);
    r.aa.bb[2] = fminf(fminf(r.aa.v[1], r.aa.v[3]), r.aa.v[5]);
    r.aa.bb[3] = fmaxf(fmaxf(r.aa.v[1], r.aa.v[3]), r.aa.v[5]);
}

}  // namespace dm2
