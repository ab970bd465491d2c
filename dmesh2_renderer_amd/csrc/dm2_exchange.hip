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
//                                send buffer, per owner [face rows | vertex rows]; places inside a segment: one global atomic per
//                                block of 8192 ids and owner, LDS cursors inside the block
//   k_xchg_unpack                owner: every received row is added to the dense slice, one launch per source (rows of one source
//                                have distinct ids: plain adds, and the same summation order on every rank)
//
// All HBM streaming: 4 B of flags + ~40 B per touched row; tools/exchange_time.py times them against the torch formulation
// they replace (15 torch kernels + a sort: 0.6-0.8 ms per step at 1080p / 1 M faces).
#include <hip/hip_runtime.h>

#include "dm2_state.h"

namespace dm2 {

constexpr int XCHG_MAX_RANKS = 64;

// One count per active lane into cnt[slot] (LDS), the lanes of a wave that share a slot through ONE atomic (neighbouring ids share
// their owner: one or two distinct slots per wave); returns the lane's place, as its own atomicAdd(.., 1) would.  Whole waves only.
__device__ __forceinline__ uint32_t wave_slot_place(uint32_t* cnt, uint32_t slot, bool act) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
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

constexpr int XCHG_IDS_PER_BLOCK = 8192;      // a block takes a contiguous chunk of ids: a few hundred global atomics per call in all
                                              // (one atomic per wave on the 2 N counters -- the same few addresses -- was 0.7 ms: atomics on
                                              // one address are serialised by the L2)

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

// flagged ids of [i0, i0 + XCHG_IDS_PER_BLOCK) per owner, counted in LDS first
__global__ void __launch_bounds__(256)
k_xchg_count(int P, int F, int N, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
             uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_cnt[2 * XCHG_MAX_RANKS];
    for (int k = threadIdx.x; k < 2 * N; k += 256) s_cnt[k] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * XCHG_IDS_PER_BLOCK;
    // (all of the block's flags first: with the loads inside the loop every iteration waited for its own round trip to memory,
    // 0.1 ms per launch; whole waves, uniform trip count)
    uint8_t ff[XCHG_IDS_PER_BLOCK / 256], fv[XCHG_IDS_PER_BLOCK / 256];
#pragma unroll
    for (int r = 0; r < XCHG_IDS_PER_BLOCK / 256; r++) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        ff[r] = i < F ? flag_f[i] : 0; fv[r] = i < P ? flag_v[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < XCHG_IDS_PER_BLOCK / 256; r++) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        const bool af = ff[r] != 0, av = fv[r] != 0;
        wave_slot_place(s_cnt, 2u * (uint32_t)(af ? i / Fs : 0), af);
        wave_slot_place(s_cnt, 2u * (uint32_t)(av ? i / Ps : 0) + 1u, av);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * N; k += 256) if (s_cnt[k]) atomicAdd(counts + k, s_cnt[k]);
}

// the rows of the block's chunk of ids: counted per owner in LDS, ONE global atomic per (block, owner, kind) reserves their
// places in the owner's segment, the rows take them in whatever order the LDS cursor hands out
__global__ void __launch_bounds__(256)
k_xchg_pack(int B, int P, int F, int N, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
            const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, const float* __restrict__ dverts,
            const float* __restrict__ dcolor, const float* __restrict__ dopacity, const float* __restrict__ dintense,
            float* __restrict__ send) {
    __shared__ uint32_t s_cnt[2 * XCHG_MAX_RANKS], s_base[2 * XCHG_MAX_RANKS], s_seg[XCHG_MAX_RANKS + 1];
    for (int k = threadIdx.x; k < 2 * N; k += 256) s_cnt[k] = 0;
    if (threadIdx.x == 0) {
        uint32_t off = 0;
        for (int o = 0; o < N; o++) { s_seg[o] = off; off += counts[2 * o] * (uint32_t)(2 + B) + counts[2 * o + 1] * 7u; }
    }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * XCHG_IDS_PER_BLOCK;
    uint8_t ff[XCHG_IDS_PER_BLOCK / 256], fv[XCHG_IDS_PER_BLOCK / 256];      // (all of the block's flags first, see k_xchg_count)
#pragma unroll
    for (int r = 0; r < XCHG_IDS_PER_BLOCK / 256; r++) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        ff[r] = i < F ? flag_f[i] : 0; fv[r] = i < P ? flag_v[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < XCHG_IDS_PER_BLOCK / 256; r++) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        const bool af = ff[r] != 0, av = fv[r] != 0;
        wave_slot_place(s_cnt, 2u * (uint32_t)(af ? i / Fs : 0), af);
        wave_slot_place(s_cnt, 2u * (uint32_t)(av ? i / Ps : 0) + 1u, av);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * N; k += 256) { s_base[k] = s_cnt[k] ? atomicAdd(cursors + k, s_cnt[k]) : 0u; s_cnt[k] = 0; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < XCHG_IDS_PER_BLOCK / 256; r++) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        const bool af = ff[r] != 0, av = fv[r] != 0;
        const uint32_t pfl = wave_slot_place(s_cnt, 2u * (uint32_t)(af ? i / Fs : 0), af);
        const uint32_t pvl = wave_slot_place(s_cnt, 2u * (uint32_t)(av ? i / Ps : 0) + 1u, av);
        if (af) {
            const int o = (int)(i / Fs);
            const uint32_t pf = s_base[2 * o] + pfl;
            float* dst = send + s_seg[o] + (size_t)pf * (size_t)(2 + B);
            dst[0] = __int_as_float((int)i);                           // (the id's bits travel in a float slot)
            dst[1] = dopacity[i];
            for (int b = 0; b < B; b++) dst[2 + b] = dintense[(int64_t)b * F + i];
        }
        if (av) {
            const int o = (int)(i / Ps);
            const uint32_t pv = s_base[2 * o + 1] + pvl;
            float* dst = send + s_seg[o] + (size_t)counts[2 * o] * (size_t)(2 + B) + (size_t)pv * 7u;
            dst[0] = __int_as_float((int)i);
            dst[1] = dverts[3 * i]; dst[2] = dverts[3 * i + 1]; dst[3] = dverts[3 * i + 2];
            dst[4] = dcolor[3 * i]; dst[5] = dcolor[3 * i + 1]; dst[6] = dcolor[3 * i + 2];
        }
    }
}

// One source at a time (launch order = source order: the same summation order on every rank): the nf face rows and nv
// vertex rows of one source carry distinct ids, so plain read-add-write is race free within a launch.
__global__ void __launch_bounds__(256)
k_xchg_unpack(int B, int rank, int Ps, int Fs, const float* __restrict__ rows_f, uint32_t nf, const float* __restrict__ rows_v, uint32_t nv,
              float* __restrict__ slice_v, float* __restrict__ slice_f) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < (int64_t)nf) {
        const float* row = rows_f + (size_t)r * (size_t)(2 + B);
        const int64_t id = (int64_t)__float_as_int(row[0]) - (int64_t)rank * Fs;
        if (id >= 0 && id < Fs)                                        // (a row that is not this owner's: never sent by a correct peer)
            for (int k = 0; k < 1 + B; k++) slice_f[id * (1 + B) + k] += row[1 + k];
    } else if (r - nf < (int64_t)nv) {
        const float* row = rows_v + (size_t)(r - nf) * 7u;
        const int64_t id = (int64_t)__float_as_int(row[0]) - (int64_t)rank * Ps;
        if (id >= 0 && id < Ps) {
#pragma unroll
            for (int k = 0; k < 6; k++) slice_v[id * 6 + k] += row[1 + k];
        }
    }
}

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
    hipLaunchKernelGGL(k_xchg_count, dim3((unsigned)((m + XCHG_IDS_PER_BLOCK - 1) / XCHG_IDS_PER_BLOCK)), dim3(256), 0, st, P, F, N, Ps, Fs, flag_f, flag_v, counts);
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
    hipLaunchKernelGGL(k_xchg_pack, dim3((unsigned)((m + XCHG_IDS_PER_BLOCK - 1) / XCHG_IDS_PER_BLOCK)), dim3(256), 0, st, B, P, F, N, Ps, Fs, flags, flags + F,
                       counts, cursors, dverts, dcolor, dopacity, dintense, send);
    return hipSuccess;
}

// recv_counts_host: the owner's HOST copy of the (N, 2) row counts (the caller read them back for the all-to-all's split sizes)
hipError_t launch_exchange_unpack(int B, int P, int F, int N, int rank, const float* recv, const uint32_t* recv_counts_host, int64_t rows,
                                  float* slice_v, float* slice_f, hipStream_t st) {
    const int Fs = (F + N - 1) / N, Ps = (P + N - 1) / N;
    hipError_t e = hipMemsetAsync(slice_v, 0, (size_t)Ps * 6 * sizeof(float), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(slice_f, 0, (size_t)Fs * (1 + B) * sizeof(float), st);
    if (e != hipSuccess) return e;
    if (rows <= 0) return hipSuccess;
    size_t off = 0;
    for (int s = 0; s < N; s++) {
        const uint32_t nf = recv_counts_host[2 * s], nv = recv_counts_host[2 * s + 1];
        const float* rows_f = recv + off; off += (size_t)nf * (size_t)(2 + B);
        const float* rows_v = recv + off; off += (size_t)nv * 7u;
        const int64_t n = (int64_t)nf + nv;
        if (n > 0) hipLaunchKernelGGL(k_xchg_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B, rank, Ps, Fs, rows_f, nf, rows_v, nv, slice_v, slice_f);
    }
    return hipSuccess;
}

int exchange_max_ranks() { return XCHG_MAX_RANKS; }

}  // namespace dm2
