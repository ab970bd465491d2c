# A/B of builds: usage: bash tools/ab_libs.sh "<lib> <lib> ..." "<bench args>;<bench args>;..."
LIBS=${1:-dmesh2_renderer_amd/csrc/libdm2_hip.so}
IFS=';' read -ra RUNS <<< "${2:---config cfg4}"
for args in "${RUNS[@]}"; do for i in 1 2; do for lib in $LIBS; do
  echo -n "== $args | $lib | "; DM2_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_rank0']; print(d['ms_per_step'], 'plan', s['preprocess_scan'], 'fwd', s['forward_composite'], 'bwd', s['backward_composite'])"
done; done; done
