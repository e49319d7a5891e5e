# what a workgroup barrier per CG iteration costs in the multi-wave kernels of configuration 2
# (the hook was removed again: `#ifdef GD_EXPERIMENT_BARRIERS` / a loop of that many `job_sync<W>()`
#  behind the "p published" barrier of the CG loop in mgk_oc.h)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [(k['kernel'].split('_oc')[-1].replace('_C1',''), round(k['isolated_ms'] or 0,3)) for k in d['kernels']])"; }
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-api --steps 30 --cpu-seconds 2 --no-f32 --no-cpu-baseline > gpurun_out/s32_$name.json 2> gpurun_out/s32_$name.err || tail -3 gpurun_out/s32_$name.err | cut -c1-300; echo -n "$name: "; show gpurun_out/s32_$name.json; }
run c2_0 --config 2 --dtype f32
GD_HIPCC_EXTRA=-DGD_EXPERIMENT_BARRIERS=1 run c2_1 --config 2 --dtype f32
GD_HIPCC_EXTRA=-DGD_EXPERIMENT_BARRIERS=3 run c2_3 --config 2 --dtype f32
GD_HIPCC_EXTRA=-DGD_EXPERIMENT_BARRIERS=3 run c2f64_3 --config 2 --dtype f64
