// valu_calib.hip -- measures what one SIMD of gfx950 sustains for wave64 vector instructions, so that
// bench.py's `valu_issue_util` uses a MEASURED cycles-per-instruction constant instead of an assumed one
// (MI355X_MICROARCH.md:52-54 says 2 cycles with >= 2 waves per SIMD, 4 for a lone wave).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_calib valu_calib.hip && ./valu_calib
//
// Every kernel runs ITER iterations of UNROLL instructions of one kind on NCHAIN independent register chains
// (NCHAIN = 1 is a fully dependent stream).  Grid = 256 CUs x (waves per SIMD x 4) waves, one block per wave so
// that the dispatcher spreads them over the SIMDs.  Cycles come from s_memtime (shader clock), wall time from
// hipEvents; cycles per wave-instruction per SIMD = elapsed_cycles / (instructions per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITER = 4096;
constexpr int UNROLL = 64;

enum Kind { K_FMA = 0, K_ADD = 1, K_CNDMASK = 2, K_ADD_DPP = 3, K_MUL_ADD = 4, K_LSHL64 = 5, K_CMP_CND = 6, K_RCP = 7, K_DIV = 8, K_MIN3 = 9, K_BCNT = 10, K_NKIND = 11 };
static const char* kind_name[K_NKIND] = {"v_fma_f32", "v_add_f32", "v_cndmask_b32", "v_add_f32_dpp(row_shr)", "v_mul_f32+v_add_f32 (pair)", "v_lshlrev_b64",
                                         "v_cmp_gt_f32+v_cndmask (pair)", "v_rcp_f32", "fp32 IEEE divide (x/y)", "v_min3_f32", "v_bcnt_u32_b32"};

template <int KIND, int NCHAIN>
__global__ void __launch_bounds__(64) k_calib(float* out, unsigned long long* cyc, float seed) {
    float a[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) a[c] = seed + (float)(threadIdx.x + c);
    const float m = seed * 0.999f, b = seed * 0.001f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / NCHAIN; u++) {
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) {
                float& x = a[c];
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(b));
                else if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
                else if (KIND == K_ADD_DPP) {                // a DPP read needs 2 wait states behind the VALU write of its source: only the dependent stream needs the nop
                    if (NCHAIN == 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1" : "+v"(x));
                    else asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
                }
                else if (KIND == K_MUL_ADD) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x) : "v"(m), "v"(b));
                else if (KIND == K_LSHL64) { unsigned long long y = (unsigned long long)__float_as_uint(x); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(y)); x = __uint_as_float((unsigned)y); }
                else if (KIND == K_CMP_CND) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(m), "v"(b) : "vcc");
                else if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
                else if (KIND == K_DIV) { x = m / x; asm volatile("" : "+v"(x)); }
                else if (KIND == K_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(b));
                else if (KIND == K_BCNT) { unsigned y = __float_as_uint(x); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(y) : "v"(1u)); x = __uint_as_float(y); }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += a[c];
    if (s == 123.456f) out[0] = s;                         // keeps the chains alive; never true
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NCHAIN>
static void run(int waves_per_simd, float* d_out, unsigned long long* d_cyc, int ncu) {
    const int blocks = ncu * 4 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_calib<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0f);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_calib<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks);
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : cyc) mean += (double)c; mean /= blocks;
    const int per_pair = (KIND == K_MUL_ADD || KIND == K_CMP_CND) ? 2 : 1;
    const double instr = (double)ITER * UNROLL * per_pair;                 // per wave (K_DIV: source-level operations)
    // in-wave cycles per instruction: what ONE wave sees; per-SIMD: divided by the waves that share the SIMD
    printf("{\"kind\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_instr_seen_by_wave\": %.3f, \"cycles_per_instr_per_simd\": %.3f, "
           "\"wall_ms\": %.4f, \"wall_ginstr_per_s_per_simd\": %.3f}\n",
           kind_name[KIND], NCHAIN, waves_per_simd, mean / instr, mean / instr / waves_per_simd, ms,
           instr * waves_per_simd / (ms * 1e-3) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int KIND> static void sweep(float* d_out, unsigned long long* d_cyc, int ncu) {
    for (int w : {1, 2, 4, 8}) run<KIND, 8>(w, d_out, d_cyc, ncu);
    for (int w : {1, 2, 4}) run<KIND, 1>(w, d_out, d_cyc, ncu);
}


// packed fp32 (two fp32 operations per lane and instruction: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on 64-bit register pairs)
typedef float v2f __attribute__((ext_vector_type(2)));
enum PkKind { PK_MUL = 0, PK_ADD = 1, PK_FMA = 2, PK_NKIND = 3 };
static const char* pk_name[PK_NKIND] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"};
template <int KIND, int NCHAIN>
__global__ void __launch_bounds__(64) k_calib_pk(float* out, unsigned long long* cyc, float seed) {
    v2f a[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) a[c] = v2f{seed + (float)(threadIdx.x + c), seed - (float)c};
    const v2f m = {seed * 0.999f, seed * 0.998f}, b = {seed * 0.001f, seed * 0.002f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / NCHAIN; u++) {
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) {
                v2f& x = a[c];
                if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
                else if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += a[c].x + a[c].y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int NCHAIN>
static void run_pk(int waves_per_simd, float* d_out, unsigned long long* d_cyc, int ncu) {
    const int blocks = ncu * 4 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_calib_pk<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_calib_pk<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks);
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : cyc) mean += (double)c; mean /= blocks;
    const double instr = (double)ITER * UNROLL;
    printf("{\"kind\": \"%s (2 fp32 ops per lane)\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_instr_seen_by_wave\": %.3f, \"cycles_per_instr_per_simd\": %.3f, "
           "\"wall_ms\": %.4f, \"wall_ginstr_per_s_per_simd\": %.3f}\n",
           pk_name[KIND], NCHAIN, waves_per_simd, mean / instr, mean / instr / waves_per_simd, ms, instr * waves_per_simd / (ms * 1e-3) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}
template <int KIND> static void sweep_pk(float* d_out, unsigned long long* d_cyc, int ncu) {
    for (int w : {1, 2, 4, 8}) run_pk<KIND, 8>(w, d_out, d_cyc, ncu);
    for (int w : {1, 2, 4}) run_pk<KIND, 1>(w, d_out, d_cyc, ncu);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", p.gcnArchName, ncu, p.clockRate);
    float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * ncu * 4 * 8));
    sweep_pk<PK_MUL>(d_out, d_cyc, ncu);
    sweep_pk<PK_ADD>(d_out, d_cyc, ncu);
    sweep_pk<PK_FMA>(d_out, d_cyc, ncu);
    if (argc > 1) return 0;                               // any argument: the packed kinds only
    sweep<K_FMA>(d_out, d_cyc, ncu);
    sweep<K_ADD>(d_out, d_cyc, ncu);
    sweep<K_CNDMASK>(d_out, d_cyc, ncu);
    sweep<K_ADD_DPP>(d_out, d_cyc, ncu);
    sweep<K_MUL_ADD>(d_out, d_cyc, ncu);
    sweep<K_LSHL64>(d_out, d_cyc, ncu);
    sweep<K_CMP_CND>(d_out, d_cyc, ncu);
    sweep<K_RCP>(d_out, d_cyc, ncu);
    sweep<K_DIV>(d_out, d_cyc, ncu);
    sweep<K_MIN3>(d_out, d_cyc, ncu);
    sweep<K_BCNT>(d_out, d_cyc, ncu);
    return 0;
}
