#!/bin/bash
# One gpurun call: bench lines of every configuration + rocprofv3 kernel stats at cfg4 (AA and point-sampled) + the sharded
# step's local costs.  usage: bash tools/round_bench.sh <outdir under gpurun_out> <tag>
set -u
OUT=${1:-gpurun_out/rb}; TAG=${2:-r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg4 cfg1 cfg2 cfg5 cfg4_b4 cfg1_dc60; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu --config $cfg > $OUT/${TAG}_bench_$cfg.json 2>> $OUT/${TAG}_bench.err
done
python3 bench.py --steps 20 --warmup 5 --no-cpu --aa-temperature 0 > $OUT/${TAG}_bench_cfg4_temp0.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/${TAG}_under_rocprof.log 2>&1
python3 tools/trim_rocprof.py $OUT/kt > $OUT/${TAG}_cfg4_kernel_stats.csv 2>/dev/null; rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt0 -- python3 bench.py --steps 20 --warmup 5 --no-cpu --aa-temperature 0 > $OUT/${TAG}_under_rocprof_temp0.log 2>&1
python3 tools/trim_rocprof.py $OUT/kt0 > $OUT/${TAG}_cfg4_temp0_kernel_stats.csv 2>/dev/null; rm -rf $OUT/kt0
for n in 8 4 2; do python3 tools/exchange_time.py $n > $OUT/${TAG}_exchange_time_$n.txt 2>&1; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print(f.split("bench_")[-1][:-5], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in c["stage_ms_rank0"].items()}, c.get("grad_Mtris_per_s"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
head -8 $OUT/${TAG}_cfg4_kernel_stats.csv | cut -c1-110; head -6 $OUT/${TAG}_cfg4_temp0_kernel_stats.csv | cut -c1-110; cat $OUT/${TAG}_exchange_time_8.txt
