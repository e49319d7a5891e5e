cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('fp64_converged') or {}; print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms its', round(d.get('mean_cg_iterations') or 0,2), 'acc', (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'), '| conv', round((c.get('value') or 0)/1e6,2), [(k['kernel'].split('_L')[-1].replace('_tab',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']][:9])"; }
run() { name=$1; shift; timeout 900 python bench.py --no-api --cpu-seconds 2 "$@" > gpurun_out/s16_$name.json 2> gpurun_out/s16_$name.err || tail -3 gpurun_out/s16_$name.err; echo -n "$(date +%T) $name: "; show gpurun_out/s16_$name.json; }
run grad64 --gradient --steps 50 --no-f32
run f64 --steps 100 --no-f32
GD_OCCUPANCY=1:21:4 run f64_21at4 --steps 100 --no-f32
run grad64b --gradient --steps 50 --no-f32
