#!/bin/bash
# A/B several builds of the library on the bench workload (one GPU call); prints value + stage times per build.
for lib in "$@"; do
  echo "== $lib"
  DM2_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_rank0'])"
done
