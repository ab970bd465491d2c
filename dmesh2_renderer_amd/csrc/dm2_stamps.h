// dm2_stamps.h -- in-kernel cycle stamps, DIAGNOSTIC BUILDS ONLY
// (make EXTRA=-DDM2_STAMPS BUILD=build_stamps OUT=libdm2_hip_stamps.so).
// Each wave accumulates shader cycles per code segment in registers and adds them once, at
// the end, to a device table that no kernel reads (guide: cdna_hip_programming.md §7,
// "In-kernel stamps").  In the product build every macro below expands to nothing.
#pragma once
#include <hip/hip_runtime.h>

#define DM2_NSTAMP 16
#ifdef DM2_STAMPS
namespace dm2 { unsigned long long* stamps_table(); }   // device pointer to [2][DM2_NSTAMP], defined in dm2_api.hip
#define STAMP_PARAM , unsigned long long* st_out
#define STAMP_ARG(which) , (dm2::stamps_table() + (which) * DM2_NSTAMP)
// The per-segment sums live in LDS (one row per wave, 12 segments) and the last stamp in scalar registers, so that the
// diagnostic build keeps the product build's VGPR budget: with the sums in VGPRs the backward kernel spilled to scratch,
// and a scratch reload waits for every outstanding global atomic / LDS-direct load (vmcnt is in order) -- the profile
// then showed the spills, not the kernel.
#define DM2_NSTAMP_LDS 12
#define STAMP_DECL __shared__ unsigned long long st_lds[4][DM2_NSTAMP_LDS]; \
    if ((threadIdx.x & 63) < DM2_NSTAMP_LDS) st_lds[threadIdx.x >> 6][threadIdx.x & 63] = 0ull; \
    unsigned long long st_last = __builtin_readcyclecounter();
#define STAMP(i) { const unsigned long long st_now = __builtin_readcyclecounter(); \
    if ((threadIdx.x & 63) == 0) atomicAdd(&st_lds[threadIdx.x >> 6][i], st_now - st_last); st_last = st_now; }
#define STAMP_FLUSH { if ((threadIdx.x & 63) < DM2_NSTAMP_LDS) { const unsigned long long v_ = st_lds[threadIdx.x >> 6][threadIdx.x & 63]; \
    if (v_) atomicAdd(&st_out[threadIdx.x & 63], v_); } }
#else
#define STAMP_PARAM
#define STAMP_ARG(which)
#define STAMP_DECL
#ifdef DM2_ISA_MARKS    // static budget builds (tools/isa_budget.py): a comment line in the -save-temps ISA at every stamp point
#define STAMP(i) asm volatile("; DM2_MARK after_stamp_" #i);
#else
#define STAMP(i)
#endif
#define STAMP_FLUSH
#endif

// a comment line in the -save-temps ISA of a -DDM2_ISA_MARKS build (tools/isa_budget.py), nothing otherwise
#ifdef DM2_ISA_MARKS
#define ISA_MARK(name) asm volatile("; DM2_MARK " #name);
#else
#define ISA_MARK(name)
#endif
