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
#define STAMP_DECL unsigned long long st_acc[DM2_NSTAMP] = {}; unsigned long long st_last = __builtin_readcyclecounter();
#define STAMP(i) { const unsigned long long st_now = __builtin_readcyclecounter(); st_acc[i] += st_now - st_last; st_last = st_now; }
#define STAMP_FLUSH { if ((threadIdx.x & 63) == 0) { for (int s_ = 0; s_ < DM2_NSTAMP; s_++) if (st_acc[s_]) atomicAdd(&st_out[s_], st_acc[s_]); } }
#else
#define STAMP_PARAM
#define STAMP_ARG(which)
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif
