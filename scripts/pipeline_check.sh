#!/bin/bash
# Sharded step on one rank (nccl group of size 1) with and without the
# pipelining of consecutive steps, plus the distributed tests.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x > gpurun_out/pytest_dist.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_dist.log
# (354 graphs = the per-rank share of the 1000-graph matrix on 8 ranks)
for tag in "pipe:--pipeline" "nopipe:" "pipe32:--dtype f32 --pipeline" "nopipe32:--dtype f32" "small:--graphs 354 --steps 200 --pipeline" "smallnp:--graphs 354 --steps 200" "small32:--graphs 354 --steps 200 --dtype f32 --pipeline" "small32np:--graphs 354 --steps 200 --dtype f32" "small32local:--graphs 354 --steps 200 --dtype f32 --local"; do
  name=${tag%%:*}; args=${tag#*:}
  mode=--sharded
  case "$args" in *--local*) mode=; args=${args/--local/};; esac
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py $mode --no-cpu-baseline --no-api $args > gpurun_out/sh_$name.json 2> gpurun_out/sh_$name.err
  echo "$name rc=$?"; tail -1 gpurun_out/sh_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms'), (d.get('sharded_check') or {}).get('max_rel_diff_vs_oracle'))"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "bench 2 ranks rc=$?"; tail -c 700 gpurun_out/bench_2ranks.json; echo
