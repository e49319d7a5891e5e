# SEQ: x in LDS (XLDS) x occupancy, per layout
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc4_')[-1].replace('_C2','').replace('_tab','').split('_L')[-1], round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
S_ALL="16 20 21 24 25 26 28 29 31 35 36"
occ() { o=""; for s in $S_ALL; do o="$o,1:$s:$1"; done; echo ${o#,}; }
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --gradient --dtype f64 --no-api --no-cpu-baseline --steps 30 > gpurun_out/s4_$name.json 2> gpurun_out/s4_$name.err || tail -3 gpurun_out/s4_$name.err
  echo -n "$name: "; show gpurun_out/s4_$name.json; }
run xlds0_occ2 GD_OCCUPANCY=$(occ 2) GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=0
run xlds1_occ2 GD_OCCUPANCY=$(occ 2) GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=1
run xlds0_occ3 GD_OCCUPANCY=$(occ 3) GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=0
run xlds1_occ3 GD_OCCUPANCY=$(occ 3) GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=1
run xlds1_occ4 GD_OCCUPANCY=$(occ 4) GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=1
run xlds1_occ3_gch8 GD_OCCUPANCY=$(occ 3) "GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=1 -DGD_OC_GCH=8"
run xlds1_occ2_f32 GD_OCCUPANCY=$(occ 2) "GD_HIPCC_EXTRA=-DGD_OC_SEQ_XLDS=1 -DGD_OC_SEQ=2"
