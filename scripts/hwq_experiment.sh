#!/bin/bash
# Does the number of hardware queues the HIP runtime multiplexes the streams
# onto (GPU_MAX_HW_QUEUES, default 4) matter for the concurrent launches?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms'))"; }
for q in ${QUEUES:-4 8 16}; do
  for tag in "f64:" "f32:--dtype f32" "g64:--gradient" "c2:--config 2"; do
    name=${tag%%:*}; args=${tag#*:}
    GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --no-api --isolated-steps 0 $args > gpurun_out/hwq_${q}_$name.json 2> gpurun_out/hwq_${q}_$name.err
    echo -n "q=$q $name: "; show gpurun_out/hwq_${q}_$name.json
  done
  for tag in "small32:--graphs 354 --steps 200 --dtype f32 --pipeline" "small32np:--graphs 354 --steps 200 --dtype f32"; do
    name=${tag%%:*}; args=${tag#*:}
    GPU_MAX_HW_QUEUES=$q MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --sharded --no-cpu-baseline --no-api --isolated-steps 0 $args > gpurun_out/hwq_${q}_$name.json 2> gpurun_out/hwq_${q}_$name.err
    echo -n "q=$q $name: "; show gpurun_out/hwq_${q}_$name.json
  done
done
