for i in 1 2; do for lib in dmesh2_renderer_amd/csrc/libdm2_hip.so dmesh2_renderer_amd/csrc/ab/lib_nocoal.so; do
  echo "== $lib"; DM2_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --config cfg4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_rank0'])"
done; done
