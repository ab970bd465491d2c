#!/bin/bash
# Collect the rocprofv3 evidence behind the bench line on the GPU box (one gpurun call):
#   kernel-trace + stats of the bench command, then separate --pmc passes (never combined with tracing):
#   FETCH_SIZE, WRITE_SIZE, two sets of SQ_* counters; the same for LayeredRenderer.generate at cfg3.
# usage: bash tools/profile_run.sh <outdir under gpurun_out> <tag>
set -u
OUT=${1:-gpurun_out/prof}; TAG=${2:-r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu"
S="python3 bench.py --steps 3 --warmup 1 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/${TAG}_bench_under_rocprof.log 2>&1
python3 tools/trim_rocprof.py $OUT/kt > $OUT/${TAG}_cfg4_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $S > /dev/null 2>&1
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1 --no-cpu, cfg4; per launch means"; python3 tools/pmc_summary.py $OUT/fetch; python3 tools/pmc_summary.py $OUT/write; } > $OUT/${TAG}_cfg4_hbm_traffic.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq1 -- $S > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -- $S > /dev/null 2>&1
{ echo "# rocprofv3 --pmc SQ_* (two passes of eight counters), bench.py --steps 3 --warmup 1 --no-cpu, cfg4; per launch means"; python3 tools/pmc_summary.py $OUT/sq1; python3 tools/pmc_summary.py $OUT/sq2; } > $OUT/${TAG}_cfg4_sq_counters.txt
# LayeredRenderer.generate at cfg3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktl -- python3 tests/layers_time.py > $OUT/${TAG}_layers_under_rocprof.log 2>&1
python3 tools/trim_rocprof.py $OUT/ktl > $OUT/${TAG}_layers_cfg3_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/lfetch -- python3 tests/layers_time.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/lwrite -- python3 tests/layers_time.py > /dev/null 2>&1
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tests/layers_time.py (cfg3: 1024x1024, 93 750 tets, 191 250 faces, 4 layers); per launch means"; python3 tools/pmc_summary.py $OUT/lfetch; python3 tools/pmc_summary.py $OUT/lwrite; } > $OUT/${TAG}_layers_cfg3_hbm_traffic.txt
# point-sampled coverage (aa_temperature 0) and the sharded step's local kernels (exchange, band plan)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt0 -- python3 bench.py --steps 20 --warmup 5 --no-cpu --aa-temperature 0 > $OUT/${TAG}_temp0_under_rocprof.log 2>&1
python3 tools/trim_rocprof.py $OUT/kt0 > $OUT/${TAG}_cfg4_temp0_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktx -- python3 tools/exchange_time.py 8 > $OUT/${TAG}_exchange_time_8.txt 2>&1
python3 tools/trim_rocprof.py $OUT/ktx > $OUT/${TAG}_exchange_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2 $OUT/ktl $OUT/lfetch $OUT/lwrite $OUT/kt0 $OUT/ktx
ls -la $OUT
