// dm2_layers.hip -- LayeredRenderer kernels for gfx950 (non-differentiable):
//   k_first_intersect  firstIntersectCUDA        forward.cu:538-709
//   k_tet_walk         generateRenderLayersCUDA  forward.cu:744-1000   (DM2_FLAG_LEGACY_KERNELS: the reference's access pattern)
//   k_pack_tets + k_tet_walk_rec                 the same walk over packed per-tet records (one 256-B fetch per step)
//
// The reference's unguarded accesses are not reproduced: out-of-image lanes
// never write first_face/first_tet (forward.cu:584-585 aliases other pixels
// when W or H is not a multiple of 16), the "other faces" table cannot overflow
// (forward.cu:884-896), layer writes are bounded by L, and the tet walk is
// capped at T+1 steps so every wave terminates.
#include <hip/hip_runtime.h>

#include "dm2_device_math.h"
#include "dm2_stage.h"
#include "dm2_state.h"

namespace dm2 {

constexpr int LAY_CHUNK = 256;

struct __attribute__((aligned(16))) LayRec {
    float v[9];
    float min_d, max_d;
    int face_id;
};
static_assert(sizeof(LayRec) == 48, "LayRec");

__device__ __forceinline__ f3 load_vert(const float* verts, int i) {
    return {verts[3 * (int64_t)i], verts[3 * (int64_t)i + 1], verts[3 * (int64_t)i + 2]};
}

// auxiliary.h:382-431
__device__ __forceinline__ f3 tet_face_outward_normal(const float* verts, const int32_t* faces, const int32_t* tets,
                                                      int face_idx, int tet_idx) {
    const f3 p0 = load_vert(verts, faces[3 * face_idx]);
    const f3 p1 = load_vert(verts, faces[3 * face_idx + 1]);
    const f3 p2 = load_vert(verts, faces[3 * face_idx + 2]);
    f3 n = cross(p1 - p0, p2 - p0);
    float n_norm = sqrtf(dot(n, n));
    n_norm = fmaxf(n_norm, 0.0001f);
    n = n / n_norm;
    const f3 q0 = load_vert(verts, tets[4 * tet_idx]), q1 = load_vert(verts, tets[4 * tet_idx + 1]);
    const f3 q2 = load_vert(verts, tets[4 * tet_idx + 2]), q3 = load_vert(verts, tets[4 * tet_idx + 3]);
    const f3 c = (((q0 + q1) + q2) + q3) * 0.25f;
    const f3 dd = c - p0;
    if (dot(n, dd) > 0.0f) n = -n;
    return n;
}

__global__ void __launch_bounds__(TILE_PIX)
k_first_intersect(dm2_layers_desc d, const float* __restrict__ min_depths, const float* __restrict__ max_depths,
                  const uint2* __restrict__ ranges, const uint32_t* __restrict__ face_list,
                  int32_t* __restrict__ first_face, int32_t* __restrict__ first_tet) {
    __shared__ LayRec recs[LAY_CHUNK];
    const int b = blockIdx.z;
    const uint32_t gx = (d.W + TILE - 1) / TILE, gy = (d.H + TILE - 1) / TILE;
    const int tid = threadIdx.x;
    const uint32_t px = blockIdx.x * TILE + (tid & 15), py = blockIdx.y * TILE + (tid >> 4);
    const bool inside = (px < (uint32_t)d.W) && (py < (uint32_t)d.H);
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    f3 ro = {0, 0, 0}, rd = {0, 0, 0};
    if (inside) {
        pixel_ray(d, b, pix, px, py, d.W, d.H, ro, rd);
    }
    const uint32_t tile = ((uint32_t)b * gy + blockIdx.y) * gx + blockIdx.x;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    bool done = !inside;
    float min_T = -1.0f, min_T_max_depth = -1.0f;
    int ff = -1;
    for (int base = 0; base < total; base += LAY_CHUNK) {
        if (__syncthreads_count(done) == TILE_PIX) break;
        const int n = min(LAY_CHUNK, total - base);
        if (tid < n) {
            const int f = (int)face_list[range.x + base + tid];
            LayRec& r = recs[tid];
            r.face_id = f;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const f3 p = load_vert(d.verts, d.faces[3 * f + i]);
                r.v[3 * i] = p.x; r.v[3 * i + 1] = p.y; r.v[3 * i + 2] = p.z;
            }
            r.min_d = min_depths[(int64_t)b * d.F + f];
            r.max_d = max_depths[(int64_t)b * d.F + f];
        }
        __syncthreads();
        for (int j = 0; !done && j < n; j++) {
            const LayRec& r = recs[j];
            if (min_T >= 0.0f && r.min_d > min_T_max_depth) { done = true; continue; }   // forward.cu:648-651
            f3 tuv;
            if (!ray_tri_intersection(ro, rd, {r.v[0], r.v[1], r.v[2]}, {r.v[3], r.v[4], r.v[5]}, {r.v[6], r.v[7], r.v[8]}, tuv)) continue;
            const bool hit = (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
            if (!hit) continue;
            if (min_T < 0.0f || tuv.x < min_T) { min_T = tuv.x; min_T_max_depth = r.max_d; ff = r.face_id; }
        }
    }
    if (!inside) return;
    int ft = -1;
    if (ff >= 0) {
        for (int i = 0; i < 2; i++) {                                   // forward.cu:689-708
            const int tet = d.face_tets[2 * ff + i];
            if (tet < 0) continue;
            const f3 n = tet_face_outward_normal(d.verts, d.faces, d.tets, ff, tet);
            if (dot(n, rd) < 0.0f) ft = tet;
        }
    }
    first_face[pix] = ff;
    first_tet[pix] = ft;
}

__global__ void __launch_bounds__(TILE_PIX)
k_tet_walk(dm2_layers_desc d, const int32_t* __restrict__ first_face, const int32_t* __restrict__ first_tet,
           int32_t* __restrict__ layers, int32_t* __restrict__ layers_cnt) {
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const uint32_t px = blockIdx.x * TILE + (tid & 15), py = blockIdx.y * TILE + (tid >> 4);
    if (!((px < (uint32_t)d.W) && (py < (uint32_t)d.H))) return;
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    f3 ro, rd;
    pixel_ray(d, b, pix, px, py, d.W, d.H, ro, rd);
    int curr_face = first_face[pix], curr_tet = first_tet[pix];
    bool done = (curr_face == -1 || curr_tet == -1);
    int ndone = 0, steps = 0;
    const int L = d.L;
    while (!done) {
        if (++steps > d.T + 1) break;
        if (d.face_existence[curr_face]) {                              // forward.cu:853-860
            if (ndone < L) layers[pix * L + ndone] = curr_face;
            ndone++;
            if (ndone >= L) done = true;
        }
        if (curr_tet == -1) done = true;
        if (done) break;
        int others[3] = {-1, -1, -1};
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int tf = d.tet_faces[4 * curr_tet + i];
            if (tf == curr_face) continue;
            if (cnt == 0) others[0] = tf; else if (cnt == 1) others[1] = tf; else if (cnt == 2) others[2] = tf;
            cnt++;
        }
        if (cnt != 3) break;                                            // forward.cu:892-896
        const f3 ncur = tet_face_outward_normal(d.verts, d.faces, d.tets, curr_face, curr_tet);
        if (dot(ncur, rd) >= 0.0f) break;                               // forward.cu:919-922
        int next_face = -1, ncand = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int of = others[i];
            const f3 p0 = load_vert(d.verts, d.faces[3 * of]), p1 = load_vert(d.verts, d.faces[3 * of + 1]);
            const f3 p2 = load_vert(d.verts, d.faces[3 * of + 2]);
            f3 tuv;
            if (!ray_tri_intersection(ro, rd, p0, p1, p2, tuv)) continue;
            const bool hit = (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
            const f3 no = tet_face_outward_normal(d.verts, d.faces, d.tets, of, curr_tet);
            if (hit && dot(no, rd) > 0.0f) { next_face = of; ncand++; }
        }
        if (ncand != 1) break;                                          // forward.cu:977-981
        int next_tet = -1;
        for (int i = 0; i < 2; i++) {
            const int pt = d.face_tets[2 * next_face + i];
            if (pt == curr_tet) continue;
            next_tet = pt; break;
        }
        curr_face = next_face; curr_tet = next_tet;
    }
    layers_cnt[pix] = ndone;
}


// ---- the walk over packed per-tet records ------------------------------------------------------------------------
// A step of the reference's walk (forward.cu:853-996) chases, for the current tet: its 4 face ids, then per face 3
// vertex ids and 3 vertices, per outward normal the tet's 4 vertex ids and vertices again, the existence flag, and the
// neighbour through face_tets -- ~15 index loads and ~100 float loads in chains three deep, per pixel and step.  Everything
// a step reads is a function of the tet alone, so k_pack_tets writes it once per call into one 256-byte record per tet
// (the vertices move every iteration of an optimisation, hence per call: 24 MB at 93 750 tets, ~0.01 ms) and a step of
// k_tet_walk_rec is ONE contiguous fetch of two 128-byte lines.  The arithmetic is the reference's, operation for
// operation (same intersection test, normals computed by the same function), so the layers are identical.
struct TetFaceRec {
    float v[9];          // the face's vertices in faces[] order (ray_tri_intersection operands)
    float n[3];          // outward normal w.r.t. this tet (auxiliary.h:382-431)
    int face_id;
    int next_tet;        // the tet on the other side: first entry of face_tets[face] that is not this tet (forward.cu:984-992), -1 = none
};
struct __attribute__((aligned(16))) TetRec {
    TetFaceRec f[4];     // tet_faces[] order
    int exist_mask;      // bit i: face_existence[f[i].face_id] != 0
    int pad[7];
};
static_assert(sizeof(TetFaceRec) == 56 && sizeof(TetRec) == 256, "TetRec layout");

size_t tet_scratch_bytes(int64_t T) { return (size_t)(T > 0 ? T : 0) * sizeof(TetRec) + ALIGN; }

__global__ void __launch_bounds__(256)
k_pack_tets(dm2_layers_desc d, TetRec* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.T) return;
    TetRec r;
    r.exist_mask = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) r.pad[k] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int tf = d.tet_faces[4 * t + i];
        TetFaceRec& fr = r.f[i];
        if ((unsigned)tf >= (unsigned)d.F) {
            // a padded or invalid entry of a tet the walk may never reach (the reference only touches tets a ray enters): no face
            // here -- the walk's "current face among the tet's four" test fails as the reference's cnt != 3 would
            fr.face_id = -1; fr.next_tet = -1;
#pragma unroll
            for (int c = 0; c < 9; c++) fr.v[c] = 0.f;
            fr.n[0] = fr.n[1] = fr.n[2] = 0.f;
            continue;
        }
        fr.face_id = tf;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const f3 p = load_vert(d.verts, d.faces[3 * tf + c]);
            fr.v[3 * c] = p.x; fr.v[3 * c + 1] = p.y; fr.v[3 * c + 2] = p.z;
        }
        const f3 n = tet_face_outward_normal(d.verts, d.faces, d.tets, tf, t);
        fr.n[0] = n.x; fr.n[1] = n.y; fr.n[2] = n.z;
        int nt = -1;
        for (int k = 0; k < 2; k++) {
            const int pt = d.face_tets[2 * tf + k];
            if (pt == t) continue;
            nt = pt; break;
        }
        fr.next_tet = nt;
        if (d.face_existence[tf]) r.exist_mask |= 1 << i;
    }
    const uint4* src = reinterpret_cast<const uint4*>(&r);
    uint4* dst = reinterpret_cast<uint4*>(out + t);
#pragma unroll
    for (int k = 0; k < 16; k++) dst[k] = src[k];
}

__global__ void __launch_bounds__(TILE_PIX)
k_tet_walk_rec(dm2_layers_desc d, const TetRec* __restrict__ trecs, const int32_t* __restrict__ first_face,
               const int32_t* __restrict__ first_tet, int32_t* __restrict__ layers, int32_t* __restrict__ layers_cnt) {
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const uint32_t px = blockIdx.x * TILE + (tid & 15), py = blockIdx.y * TILE + (tid >> 4);
    if (!((px < (uint32_t)d.W) && (py < (uint32_t)d.H))) return;
    const int64_t pix = ((int64_t)b * d.H + py) * d.W + px;
    f3 ro, rd;
    pixel_ray(d, b, pix, px, py, d.W, d.H, ro, rd);
    int curr_face = first_face[pix], curr_tet = first_tet[pix];
    bool done = (curr_face == -1 || curr_tet == -1);
    int ndone = 0, steps = 0;
    const int L = d.L;
    while (!done) {
        if (++steps > d.T + 1) break;
        if (curr_tet == -1) {                                           // the walk left the mesh through curr_face: forward.cu:853-866
            if (d.face_existence[curr_face]) {
                if (ndone < L) layers[pix * L + ndone] = curr_face;
                ndone++;
            }
            break;
        }
        // the current tet's record: 16 x 16 B, one latency
        TetRec r;
        {
            const uint4* src = reinterpret_cast<const uint4*>(trecs + curr_tet);
            uint4* dst = reinterpret_cast<uint4*>(&r);
#pragma unroll
            for (int k = 0; k < 16; k++) dst[k] = src[k];
        }
        // (static indices only: a run-time index into the record would send it to scratch memory)
        int ci = -1, cnt = 0;
        f3 ncur = {0.f, 0.f, 0.f};
        bool cur_exists = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (r.f[i].face_id == curr_face) {
                if (ci < 0) { ci = i; ncur = {r.f[i].n[0], r.f[i].n[1], r.f[i].n[2]}; cur_exists = ((r.exist_mask >> i) & 1) != 0; }
            } else cnt++;
        }
        // forward.cu:853-860: the layer is the face the ray entered through
        const bool exists = ci >= 0 ? cur_exists : d.face_existence[curr_face] != 0;
        if (exists) {
            if (ndone < L) layers[pix * L + ndone] = curr_face;
            ndone++;
            if (ndone >= L) break;
        }
        if (cnt != 3) break;                                            // forward.cu:892-896
        if (dot(ncur, rd) >= 0.0f) break;                               // forward.cu:919-922
        int next_face = -1, next_tet = -1, ncand = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const TetFaceRec& fr = r.f[i];
            if (i == ci) continue;
            f3 tuv;
            if (!ray_tri_intersection(ro, rd, {fr.v[0], fr.v[1], fr.v[2]}, {fr.v[3], fr.v[4], fr.v[5]}, {fr.v[6], fr.v[7], fr.v[8]}, tuv)) continue;
            const bool hit = (tuv.x >= 0.0f && tuv.y >= 0.0f && tuv.z >= 0.0f && tuv.y + tuv.z <= 1.0f);
            const f3 no = {fr.n[0], fr.n[1], fr.n[2]};
            if (hit && dot(no, rd) > 0.0f) { next_face = fr.face_id; next_tet = fr.next_tet; ncand++; }
        }
        if (ncand != 1) break;                                          // forward.cu:977-981
        curr_face = next_face; curr_tet = next_tet;
    }
    layers_cnt[pix] = ndone;
}

void launch_layers(const dm2_layers_desc& d, const FaceState& fs, const uint2* ranges, const uint32_t* face_list,
                   LayerImageState ls, void* tet_scratch, int32_t* render_layers, int32_t* render_layers_cnt, hipStream_t st) {
    const dim3 grid((d.W + TILE - 1) / TILE, (d.H + TILE - 1) / TILE, d.B);
    const bool recs = tet_scratch && d.T > 0 && !(d.flags & DM2_FLAG_LEGACY_KERNELS);
    TetRec* trecs = nullptr;
    if (recs) {       // (independent of the first intersections: packed first, so that the walk's inputs are ready)
        trecs = reinterpret_cast<TetRec*>(((uintptr_t)tet_scratch + ALIGN - 1) & ~(uintptr_t)(ALIGN - 1));
        hipLaunchKernelGGL(k_pack_tets, dim3((d.T + 255) / 256), dim3(256), 0, st, d, trecs);
    }
    hipLaunchKernelGGL(k_first_intersect, grid, dim3(TILE_PIX), 0, st, d, fs.min_depths, fs.max_depths, ranges, face_list,
                       ls.first_face, ls.first_tet);
    if (recs)
        hipLaunchKernelGGL(k_tet_walk_rec, grid, dim3(TILE_PIX), 0, st, d, trecs, ls.first_face, ls.first_tet, render_layers, render_layers_cnt);
    else
        hipLaunchKernelGGL(k_tet_walk, grid, dim3(TILE_PIX), 0, st, d, ls.first_face, ls.first_tet, render_layers, render_layers_cnt);
}

}  // namespace dm2
