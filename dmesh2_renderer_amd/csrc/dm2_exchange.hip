// dm2_exchange.hip -- device side of the sharded step's sparse leaf-gradient exchange (dmesh2_renderer_amd/sharding.py).
//
// A rank that renders one tile-row band of the frame holds partial gradients that are non-zero only in the rows of the
// faces its band binned and of their vertices (geometry replicated, pixels sharded).  Rows are owned by
// contiguous id ranges (face f by rank f / ceil(F/N), vertex v by rank v / ceil(P/N)); every rank sends each owner its touched
// rows of that owner's range ([id | row] records, one all-to-all), the owner sums what arrives into its dense slice, and an
// all-gather of the slices leaves every rank with the full gradients.  The three kernels below are the local work of that:
//
//   k_xchg_mark / k_xchg_count   after the forward: flag the faces of the band's tile lists (tiles_touched of the plan) and
//                                their vertices, count the flagged rows per owner -- the all-to-all's split sizes, which
//                                the host can read back while the backward still runs
//   k_xchg_pack                  after the backward: [id | dopacity | dintense(B)] and [id | dverts(3) | dcolor(3)] rows into the
//                                send buffer, per owner [face rows | vertex rows]; places inside a segment by wave-aggregated
//                                atomics (the lanes of a wave mostly share one owner: one atomic per wave and owner)
//   k_xchg_unpack                owner: every received row is added to the dense slice (fp32 atomics; a row has one contributor
//                                but for the faces that straddle a band edge)
//
// All HBM streaming: 4 B of flags + ~40 B per touched row; tools/exchange_time.py times them against the torch formulation
// they replace (15 torch kernels + a sort: 0.6-0.8 ms per step at 1080p / 1 M faces).
#include <hip/hip_runtime.h>

#include "dm2_state.h"

namespace dm2 {

namespace {

constexpr int XCHG_MAX_RANKS = 64;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One count per active lane into cnt[2 * owner + kind], the lanes of a wave that share an owner through ONE atomic;
// returns the lane's place (as its own atomicAdd(.., 1) would).  Call with the whole wave (act = the lane has a row).
__device__ __forceinline__ uint32_t wave_owner_place(uint32_t* cnt, uint32_t slot, bool act) {
    const int lane = lane_id();
    unsigned long long todo = __ballot(act);
    uint32_t place = 0;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)slot, l);
        const unsigned long long m = __ballot(act && slot == sl);
        uint32_t base = 0;
        if (lane == l) base = atomicAdd(cnt + sl, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
        if (act && slot == sl) place = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return place;
}

__global__ void __launch_bounds__(256)
k_xchg_mark(int B, int F, const int32_t* __restrict__ faces, const uint32_t* __restrict__ tiles_touched,
            uint8_t* __restrict__ flag_f, uint8_t* __restrict__ flag_v) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    uint32_t t = 0;
    for (int b = 0; b < B; b++) t |= tiles_touched[(int64_t)b * F + f];
    if (!t) return;
    flag_f[f] = 1;
    flag_v[faces[3 * (int64_t)f]] = 1; flag_v[faces[3 * (int64_t)f + 1]] = 1; flag_v[faces[3 * (int64_t)f + 2]] = 1;
}

__global__ void __launch_bounds__(256)
k_xchg_count(int P, int F, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
             uint32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (whole waves: the lanes past the end stay for the ballots)
    const bool af = i < F && flag_f[i], av = i < P && flag_v[i];
    wave_owner_place(counts, 2u * (uint32_t)(af ? i / Fs : 0), af);
    wave_owner_place(counts, 2u * (uint32_t)(av ? i / Ps : 0) + 1u, av);
}

// element offset of owner o's segment in the send buffer: per owner [nf rows of (2 + B) | nv rows of 7]
__device__ __forceinline__ uint32_t seg_start(const uint32_t* __restrict__ counts, int o, int B) {
    uint32_t off = 0;
    for (int k = 0; k < o; k++) off += counts[2 * k] * (uint32_t)(2 + B) + counts[2 * k + 1] * 7u;
    return off;
}

__global__ void __launch_bounds__(256)
k_xchg_pack(int B, int P, int F, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
            const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, const float* __restrict__ dverts,
            const float* __restrict__ dcolor, const float* __restrict__ dopacity, const float* __restrict__ dintense,
            float* __restrict__ send) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool af = i < F && flag_f[i], av = i < P && flag_v[i];
    const int of = af ? (int)(i / Fs) : 0, ov = av ? (int)(i / Ps) : 0;
    const uint32_t pf = wave_owner_place(cursors, 2u * (uint32_t)of, af);
    const uint32_t pv = wave_owner_place(cursors, 2u * (uint32_t)ov + 1u, av);
    if (af) {
        float* dst = send + seg_start(counts, of, B) + (size_t)pf * (size_t)(2 + B);
        dst[0] = __int_as_float((int)i);                               // (the id's bits travel in a float slot)
        dst[1] = dopacity[i];
        for (int b = 0; b < B; b++) dst[2 + b] = dintense[(int64_t)b * F + i];
    }
    if (av) {
        float* dst = send + seg_start(counts, ov, B) + (size_t)counts[2 * ov] * (size_t)(2 + B) + (size_t)pv * 7u;
        dst[0] = __int_as_float((int)i);
        dst[1] = dverts[3 * i]; dst[2] = dverts[3 * i + 1]; dst[3] = dverts[3 * i + 2];
        dst[4] = dcolor[3 * i]; dst[5] = dcolor[3 * i + 1]; dst[6] = dcolor[3 * i + 2];
    }
}

// recv: per source s [nf_s rows of (2 + B) | nv_s rows of 7], recv_counts[2 s], [2 s + 1]; one thread per row
__global__ void __launch_bounds__(256)
k_xchg_unpack(int B, int N, int rank, int Ps, int Fs, const float* __restrict__ recv, const uint32_t* __restrict__ recv_counts,
              float* __restrict__ slice_v, float* __restrict__ slice_f) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    size_t off = 0;
    for (int s = 0; s < N; s++) {
        const uint32_t nf = recv_counts[2 * s], nv = recv_counts[2 * s + 1];
        if (r < (int64_t)nf) {
            const float* row = recv + off + (size_t)r * (size_t)(2 + B);
            const int64_t id = (int64_t)__float_as_int(row[0]) - (int64_t)rank * Fs;
            if (id < 0 || id >= Fs) return;                            // (a row that is not this owner's: never sent by a correct peer)
            for (int k = 0; k < 1 + B; k++) atomicAdd(slice_f + id * (1 + B) + k, row[1 + k]);
            return;
        }
        r -= nf; off += (size_t)nf * (size_t)(2 + B);
        if (r < (int64_t)nv) {
            const float* row = recv + off + (size_t)r * 7u;
            const int64_t id = (int64_t)__float_as_int(row[0]) - (int64_t)rank * Ps;
            if (id < 0 || id >= Ps) return;
#pragma unroll
            for (int k = 0; k < 6; k++) atomicAdd(slice_v + id * 6 + k, row[1 + k]);
            return;
        }
        r -= nv; off += (size_t)nv * 7u;
    }
}

}  // namespace

hipError_t launch_exchange_mark(int B, int P, int F, int N, const int32_t* faces, const uint32_t* tiles_touched, uint8_t* flags,
                                uint32_t* counts, hipStream_t st) {
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)P + (size_t)F, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counts, 0, (size_t)2 * N * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (F == 0 || P == 0) return hipSuccess;
    const int Fs = (F + N - 1) / N, Ps = (P + N - 1) / N;
    uint8_t* flag_f = flags; uint8_t* flag_v = flags + F;
    hipLaunchKernelGGL(k_xchg_mark, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, B, F, faces, tiles_touched, flag_f, flag_v);
    const int64_t m = P > F ? P : F;
    hipLaunchKernelGGL(k_xchg_count, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, P, F, Ps, Fs, flag_f, flag_v, counts);
    return hipSuccess;
}

hipError_t launch_exchange_pack(int B, int P, int F, int N, const uint8_t* flags, const uint32_t* counts, uint32_t* cursors,
                                const float* dverts, const float* dcolor, const float* dopacity, const float* dintense, float* send,
                                hipStream_t st) {
    hipError_t e = hipMemsetAsync(cursors, 0, (size_t)2 * N * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (F == 0 || P == 0) return hipSuccess;
    const int Fs = (F + N - 1) / N, Ps = (P + N - 1) / N;
    const int64_t m = P > F ? P : F;
    hipLaunchKernelGGL(k_xchg_pack, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, B, P, F, Ps, Fs, flags, flags + F, counts, cursors,
                       dverts, dcolor, dopacity, dintense, send);
    return hipSuccess;
}

hipError_t launch_exchange_unpack(int B, int P, int F, int N, int rank, const float* recv, const uint32_t* recv_counts, int64_t rows,
                                  float* slice_v, float* slice_f, hipStream_t st) {
    const int Fs = (F + N - 1) / N, Ps = (P + N - 1) / N;
    hipError_t e = hipMemsetAsync(slice_v, 0, (size_t)Ps * 6 * sizeof(float), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(slice_f, 0, (size_t)Fs * (1 + B) * sizeof(float), st);
    if (e != hipSuccess) return e;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_xchg_unpack, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, B, N, rank, Ps, Fs, recv, recv_counts, slice_v, slice_f);
    return hipSuccess;
}

int exchange_max_ranks() { return XCHG_MAX_RANKS; }

}  // namespace dm2
