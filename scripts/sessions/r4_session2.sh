# SEQ (sequential value + gradient solves, double): parity, then A/B of occupancy / gathers in flight against round 3's stacked dynamic kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "gradient or fp64 or mixed_degree or gpr" > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s2_pytest.log
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc4_')[-1].replace('_C2','').replace('_tab',''), k['pairs'], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
S_ALL="16 20 21 24 25 26 28 29 31 35 36"
occ() { o=""; for s in $S_ALL; do o="$o,1:$s:$1"; done; echo ${o#,}; }
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --gradient --dtype f64 --no-api --no-cpu-baseline --steps 30 > gpurun_out/s2_$name.json 2> gpurun_out/s2_$name.err || tail -3 gpurun_out/s2_$name.err
  echo -n "$name: "; show gpurun_out/s2_$name.json
}
run r3_stacked GD_OC_SEQ=0
run seq_default X=1
run seq_occ2 GD_OCCUPANCY=$(occ 2)
run seq_occ3 GD_OCCUPANCY=$(occ 3)
run seq_occ3_gch8 GD_OCCUPANCY=$(occ 3) GD_HIPCC_EXTRA=-DGD_OC_GCH=8
run seq_occ2_gch8 GD_OCCUPANCY=$(occ 2) GD_HIPCC_EXTRA=-DGD_OC_GCH=8
run seq_occ4 GD_OCCUPANCY=$(occ 4)
run seq_f32too_default GD_HIPCC_EXTRA=-DGD_OC_SEQ=2
