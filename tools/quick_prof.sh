#!/bin/bash
# One gpurun call: bench line + rocprofv3 kernel stats of the bench command (cfg4 unless BENCH_ARGS says otherwise).
# usage: bash tools/quick_prof.sh <outdir under gpurun_out> [tag]
set -u
OUT=${1:-gpurun_out/qp}; TAG=${2:-q}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS=${BENCH_ARGS:-}
python3 bench.py --steps 20 --warmup 5 --no-cpu $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu $ARGS > $OUT/${TAG}_under_rocprof.log 2>&1
python3 tools/trim_rocprof.py $OUT/kt > $OUT/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt
cat $OUT/${TAG}_bench.json | cut -c1-900
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
