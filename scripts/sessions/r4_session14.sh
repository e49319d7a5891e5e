# MIXED (iterative refinement: float iteration, double residual) on the fp64 headline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('fp64_converged') or {}; print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms its', round(d.get('mean_cg_iterations') or 0,2), 'acc', (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'), '| conv', round((c.get('value') or 0)/1e6,2), 'its', c.get('mean_cg_iterations'), 'acc', c.get('max_rel_err_vs_converged_oracle'), [(k['kernel'].split('_L')[-1].replace('_tab',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
S_ALL="16 20 21 24 25 26 28 29 31 35 36"
occ() { o=""; for s in $S_ALL; do o="$o,1:$s:$1"; done; echo ${o#,}; }
run() { name=$1; shift; env "$@" timeout 900 python bench.py --no-api --no-f32 --steps 50 --cpu-seconds 2 > gpurun_out/s14_$name.json 2> gpurun_out/s14_$name.err || tail -3 gpurun_out/s14_$name.err; echo -n "$name: "; show gpurun_out/s14_$name.json; }
run base X=1
run mixed GD_HIPCC_EXTRA=-DGD_OC_MIXED=1
run mixed_occ2 GD_HIPCC_EXTRA=-DGD_OC_MIXED=1 GD_OCCUPANCY=$(occ 2)
run mixed_occ3 GD_HIPCC_EXTRA=-DGD_OC_MIXED=1 GD_OCCUPANCY=$(occ 3)
