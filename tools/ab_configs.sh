for cfg in cfg4 cfg1 cfg1_dc60 cfg2; do for lib in dmesh2_renderer_amd/csrc/libdm2_hip.so dmesh2_renderer_amd/csrc/ab/lib_noclass.so; do
  echo "== $cfg $lib"; DM2_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_rank0'])"
done; done
