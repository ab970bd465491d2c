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
//                                block of 32768 ids and owner, LDS cursors inside the block
//   k_xchg_unpack                owner: every received row is added to the dense slice, one launch per source (rows of one source
//                                have distinct ids: plain adds, and the same summation order on every rank)
//
// All HBM streaming: 4 B of flags + ~40 B per touched row; tools/exchange_time.py times them against the torch formulation
// they replace (15 torch kernels + a sort: 0.6-0.8 ms per step at 1080p / 1 M faces).
#include <hip/hip_runtime.h>

#include "dm2_state.h"

namespace dm2 {

constexpr int XCHG_MAX_RANKS = 64;

constexpr int XCHG_THREADS = 1024;
constexpr int XCHG_IDS_PER_WAVE = 2048;                   // a wave takes 2048 contiguous ids, 32 rounds of 64: one or two owners per wave
constexpr int XCHG_ROUNDS = XCHG_IDS_PER_WAVE / 64;
constexpr int XCHG_BATCH = 8;                            // rounds whose gathers are in flight together (k_xchg_pack)
constexpr int XCHG_IDS_PER_BLOCK = XCHG_IDS_PER_WAVE * (XCHG_THREADS / 64);    // 32768 ids per block: ~100 blocks, a couple of hundred
                                              // global atomics per call in all.  (Atomics on one address are serialised by the L2: one per
                                              // wave and round on the 2 N counters was 0.7 ms per launch.)

// bit r of the result: the flag of id w0 + 64 r + lane (ids >= n read as unflagged).  A wave whose whole range is below n (all
// but the last) loads unguarded, 32 loads in flight off one address register: a load behind a branch waits for itself, and the
// rounds ran one memory round trip after the other (0.1 ms per launch).
__device__ __forceinline__ uint32_t xchg_load_flags(const uint8_t* __restrict__ flag, int64_t w0, int lane, int64_t n) {
    uint32_t bits = 0;
    if (w0 + XCHG_IDS_PER_WAVE <= n) {
        const uint8_t* p = flag + w0 + lane;
#pragma unroll
        for (int r = 0; r < XCHG_ROUNDS; r++) bits |= (uint32_t)(p[64 * r] != 0) << r;
    } else {
        for (int r = 0; r < XCHG_ROUNDS; r++) {
            const int64_t i = w0 + 64 * r + lane;
            bits |= (uint32_t)(i < n && flag[i] != 0) << r;
        }
    }
    return bits;
}

// flagged ids of the wave's range that belong to the owner whose slice is [lo, hi) (relative to the wave's first id); wave-uniform
__device__ __forceinline__ uint32_t xchg_wave_count(uint32_t bits, int lane, int lo, int hi) {
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < XCHG_ROUNDS; r++) {
        const int rel = 64 * r + lane;
        cnt += (uint32_t)__popcll(__ballot(((bits >> r) & 1u) && rel >= lo && rel < hi));
    }
    return cnt;
}

__device__ __forceinline__ int xchg_hi(int lo, int S) { const int64_t h = (int64_t)lo + S; return h < XCHG_IDS_PER_WAVE ? (int)h : XCHG_IDS_PER_WAVE; }

// the owners [o0, o1] a wave's ids [w0, w0 + XCHG_IDS_PER_WAVE) & [0, n) fall to, slices of S ids; false: no id of the wave is < n
__device__ __forceinline__ bool xchg_wave_owners(int64_t w0, int64_t n, int S, int& o0, int& o1) {
    if (w0 >= n) return false;
    const int64_t last = (w0 + XCHG_IDS_PER_WAVE - 1 < n ? w0 + XCHG_IDS_PER_WAVE - 1 : n - 1);
    o0 = (int)((uint32_t)w0 / (uint32_t)S); o1 = (int)((uint32_t)last / (uint32_t)S);       // (ids < 2^31; wave-uniform: scalar divisions)
    return true;
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

// flagged ids per owner: a ballot count per wave and owner, summed in LDS, one global atomic per block and owner
__global__ void __launch_bounds__(XCHG_THREADS)
k_xchg_count(int P, int F, int N, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
             uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_cnt[2 * XCHG_MAX_RANKS];
    for (int k = threadIdx.x; k < 2 * N; k += XCHG_THREADS) s_cnt[k] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (int64_t)blockIdx.x * XCHG_IDS_PER_BLOCK + (int64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * XCHG_IDS_PER_WAVE;
    const uint32_t ff = xchg_load_flags(flag_f, w0, lane, F);
    __builtin_amdgcn_sched_barrier(0);                  // (32 loads in flight at a time: with all 64 the kernel spills)
    const uint32_t fv = xchg_load_flags(flag_v, w0, lane, P);
    __builtin_amdgcn_sched_barrier(0);
    int o0, o1;
    if (xchg_wave_owners(w0, F, Fs, o0, o1))
        for (int o = o0; o <= o1; o++) {
            const int lo = (int)((int64_t)o * Fs - w0);
            const uint32_t c = xchg_wave_count(ff, lane, lo, xchg_hi(lo, Fs));
            if (lane == 0 && c) atomicAdd(&s_cnt[2 * o], c);
        }
    if (xchg_wave_owners(w0, P, Ps, o0, o1))
        for (int o = o0; o <= o1; o++) {
            const int lo = (int)((int64_t)o * Ps - w0);
            const uint32_t c = xchg_wave_count(fv, lane, lo, xchg_hi(lo, Ps));
            if (lane == 0 && c) atomicAdd(&s_cnt[2 * o + 1], c);
        }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * N; k += XCHG_THREADS) if (s_cnt[k]) atomicAdd(counts + k, s_cnt[k]);
}

// the rows of the block's chunk of ids: counted per owner in LDS as above, ONE global atomic per (block, owner, kind) reserves their
// places in the owner's segment, every wave takes its share of the block's places with one LDS atomic per owner and hands
// them out by ballot rank (the order of the rows inside a segment is free: the owner adds rows with distinct ids)
__global__ void __launch_bounds__(XCHG_THREADS)
k_xchg_pack(int B, int P, int F, int N, int Ps, int Fs, const uint8_t* __restrict__ flag_f, const uint8_t* __restrict__ flag_v,
            const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, const float* __restrict__ dverts,
            const float* __restrict__ dcolor, const float* __restrict__ dopacity, const float* __restrict__ dintense,
            float* __restrict__ send) {
    __shared__ uint32_t s_cnt[2 * XCHG_MAX_RANKS], s_base[2 * XCHG_MAX_RANKS], s_seg[XCHG_MAX_RANKS + 1];
    for (int k = threadIdx.x; k < 2 * N; k += XCHG_THREADS) s_cnt[k] = 0;
    if (threadIdx.x == 0) {
        uint32_t off = 0;
        for (int o = 0; o < N; o++) { s_seg[o] = off; off += counts[2 * o] * (uint32_t)(2 + B) + counts[2 * o + 1] * 7u; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (int64_t)blockIdx.x * XCHG_IDS_PER_BLOCK + (int64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * XCHG_IDS_PER_WAVE;
    const uint32_t ff = xchg_load_flags(flag_f, w0, lane, F);
    __builtin_amdgcn_sched_barrier(0);                  // (32 loads in flight at a time: with all 64 the kernel spills)
    const uint32_t fv = xchg_load_flags(flag_v, w0, lane, P);
    __builtin_amdgcn_sched_barrier(0);
    int of0 = 0, of1 = -1, ov0 = 0, ov1 = -1;
    xchg_wave_owners(w0, F, Fs, of0, of1);
    xchg_wave_owners(w0, P, Ps, ov0, ov1);
    for (int o = of0; o <= of1; o++) {
        const int lo = (int)((int64_t)o * Fs - w0);
        const uint32_t c = xchg_wave_count(ff, lane, lo, xchg_hi(lo, Fs));
        if (lane == 0 && c) atomicAdd(&s_cnt[2 * o], c);
    }
    for (int o = ov0; o <= ov1; o++) {
        const int lo = (int)((int64_t)o * Ps - w0);
        const uint32_t c = xchg_wave_count(fv, lane, lo, xchg_hi(lo, Ps));
        if (lane == 0 && c) atomicAdd(&s_cnt[2 * o + 1], c);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * N; k += XCHG_THREADS) { s_base[k] = s_cnt[k] ? atomicAdd(cursors + k, s_cnt[k]) : 0u; s_cnt[k] = 0; }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int o = of0; o <= of1; o++) {
        const int lo = (int)((int64_t)o * Fs - w0), hi = xchg_hi(lo, Fs);
        const uint32_t c = xchg_wave_count(ff, lane, lo, hi);
        if (!c) continue;
        uint32_t place = 0;
        if (lane == 0) place = atomicAdd(&s_cnt[2 * o], c);
        place = (uint32_t)__builtin_amdgcn_readfirstlane((int)place) + s_base[2 * o];
        float* seg = send + s_seg[o];
        // XCHG_BATCH rounds at a time: all of the batch's gathers first (unguarded, from row 0 for idle lanes), then its stores --
        // a load behind a branch, or behind the previous round's stores, waits for itself and the rounds ran one memory round
        // trip after the other
#pragma unroll 1
        for (int r0 = 0; r0 < XCHG_ROUNDS; r0 += XCHG_BATCH) {
            float op[XCHG_BATCH];
#pragma unroll
            for (int k = 0; k < XCHG_BATCH; k++) {
                const int rel = 64 * (r0 + k) + lane;
                const bool act = ((ff >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                op[k] = dopacity[act ? w0 + rel : 0];
            }
            uint32_t pl = place;
#pragma unroll
            for (int k = 0; k < XCHG_BATCH; k++) {
                const int rel = 64 * (r0 + k) + lane;
                const bool act = ((ff >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                const unsigned long long m = __ballot(act);
                if (act) {
                    float* dst = seg + (size_t)(pl + (uint32_t)__popcll(m & below)) * (size_t)(2 + B);
                    dst[0] = __int_as_float((int)(w0 + rel)); dst[1] = op[k];      // (the id's bits travel in a float slot)
                }
                pl += (uint32_t)__popcll(m);
            }
            for (int b = 0; b < B; b++) {
                const float* src = dintense + (int64_t)b * F;
#pragma unroll
                for (int k = 0; k < XCHG_BATCH; k++) {
                    const int rel = 64 * (r0 + k) + lane;
                    const bool act = ((ff >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                    op[k] = src[act ? w0 + rel : 0];
                }
                pl = place;
#pragma unroll
                for (int k = 0; k < XCHG_BATCH; k++) {
                    const int rel = 64 * (r0 + k) + lane;
                    const bool act = ((ff >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                    const unsigned long long m = __ballot(act);
                    if (act) seg[(size_t)(pl + (uint32_t)__popcll(m & below)) * (size_t)(2 + B) + 2 + b] = op[k];
                    pl += (uint32_t)__popcll(m);
                }
            }
            place = pl;
        }
    }
    for (int o = ov0; o <= ov1; o++) {
        const int lo = (int)((int64_t)o * Ps - w0), hi = xchg_hi(lo, Ps);
        const uint32_t c = xchg_wave_count(fv, lane, lo, hi);
        if (!c) continue;
        uint32_t place = 0;
        if (lane == 0) place = atomicAdd(&s_cnt[2 * o + 1], c);
        place = (uint32_t)__builtin_amdgcn_readfirstlane((int)place) + s_base[2 * o + 1];
        float* seg = send + s_seg[o] + (size_t)counts[2 * o] * (size_t)(2 + B);
#pragma unroll 1
        for (int r0 = 0; r0 < XCHG_ROUNDS; r0 += XCHG_BATCH) {
            float g[XCHG_BATCH][6];
#pragma unroll
            for (int k = 0; k < XCHG_BATCH; k++) {
                const int rel = 64 * (r0 + k) + lane;
                const bool act = ((fv >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                const int64_t i = act ? w0 + rel : 0;
                g[k][0] = dverts[3 * i]; g[k][1] = dverts[3 * i + 1]; g[k][2] = dverts[3 * i + 2];
                g[k][3] = dcolor[3 * i]; g[k][4] = dcolor[3 * i + 1]; g[k][5] = dcolor[3 * i + 2];
            }
#pragma unroll
            for (int k = 0; k < XCHG_BATCH; k++) {
                const int rel = 64 * (r0 + k) + lane;
                const bool act = ((fv >> (r0 + k)) & 1u) && rel >= lo && rel < hi;
                const unsigned long long m = __ballot(act);
                if (act) {
                    float* dst = seg + (size_t)(place + (uint32_t)__popcll(m & below)) * 7u;
                    dst[0] = __int_as_float((int)(w0 + rel));
#pragma unroll
                    for (int c = 0; c < 6; c++) dst[1 + c] = g[k][c];
                }
                place += (uint32_t)__popcll(m);
            }
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
    hipLaunchKernelGGL(k_xchg_count, dim3((unsigned)((m + XCHG_IDS_PER_BLOCK - 1) / XCHG_IDS_PER_BLOCK)), dim3(XCHG_THREADS), 0, st, P, F, N, Ps, Fs, flag_f, flag_v, counts);
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
    hipLaunchKernelGGL(k_xchg_pack, dim3((unsigned)((m + XCHG_IDS_PER_BLOCK - 1) / XCHG_IDS_PER_BLOCK)), dim3(XCHG_THREADS), 0, st, B, P, F, N, Ps, Fs, flags, flags + F,
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
