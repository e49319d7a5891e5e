# two-wave static layouts for the pairs of five and more row batches (GD_OC_STATIC2=1)
# (prototype removed again, DESIGN.md "tried and dropped": OCStatic2 / static2_layouts in _backend_hip.py,
#  a two-instantiation entry point, W <= 2 static layouts in mgk_oc.h)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GD_OC_STATIC2=1
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -x -k "full_size_gram or self_similarity or cross_similarity or fp64_build" 2>&1 | tail -4
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1','').replace('_tab',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']], (d.get('accuracy') or {}).get('max_rel_err_vs_converged_oracle'))"; }
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-api --steps 50 --cpu-seconds 2 --no-f32 > gpurun_out/s31_$name.json 2> gpurun_out/s31_$name.err || tail -3 gpurun_out/s31_$name.err | cut -c1-300; echo -n "$name: "; show gpurun_out/s31_$name.json; }
run f64 --dtype f64
run f32 --dtype f32
for w in 3 5; do export GD_OC_STATIC2_WAVES=$w; run f64_w$w --dtype f64; done
