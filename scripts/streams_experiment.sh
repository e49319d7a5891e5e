#!/bin/bash
# How many streams should the solver launches of a step be dealt onto?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms'))"; }
for q in ${STREAMS:-0 2 3 4 6}; do
  for tag in "f64:" "f32:--dtype f32" "g64:--gradient" "c2:--config 2"; do
    name=${tag%%:*}; args=${tag#*:}
    GD_MAX_STREAMS=$q timeout 600 python bench.py --no-cpu-baseline --no-api --isolated-steps 0 $args > gpurun_out/st_${q}_$name.json 2> gpurun_out/st_${q}_$name.err
    echo -n "streams=$q $name: "; show gpurun_out/st_${q}_$name.json
  done
  for tag in "small32p:--graphs 354 --steps 200 --dtype f32 --pipeline" "small32:--graphs 354 --steps 200 --dtype f32" "small64p:--graphs 354 --steps 200 --pipeline" "small64:--graphs 354 --steps 200"; do
    name=${tag%%:*}; args=${tag#*:}
    GD_MAX_STREAMS=$q MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --sharded --no-cpu-baseline --no-api --isolated-steps 0 $args > gpurun_out/st_${q}_$name.json 2> gpurun_out/st_${q}_$name.err
    echo -n "streams=$q $name: "; show gpurun_out/st_${q}_$name.json
  done
done
