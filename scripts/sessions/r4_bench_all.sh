# every bench line of the round -> gpurun_out/r4b_<name>.json (copied to profiles/r04_bench_<name>.json)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for tag in "f64:" "f32:--dtype f32" "grad64:--gradient" "grad32:--gradient --dtype f32" "c2:--config 2" "c2f64:--config 2 --dtype f64" "tang:--config tang2019" "tang64:--config tang2019 --dtype f64" "tanggrad:--config tang2019 --gradient" "gpr:--gpr" "gpr64:--gpr --dtype f64" "fit64:--gpr --fit" "fit32:--gpr --fit --dtype f32" "nws48:--config nws48"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/r4b_$name.json 2> gpurun_out/r4b_$name.err
  echo "bench $name rc=$? $(tail -1 gpurun_out/r4b_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M/s',round(d['ms_per_step'],3),'ms')" 2>/dev/null)"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 3 --share-devices > gpurun_out/r4b_2ranks_one_gpu.json 2> gpurun_out/r4b_2ranks_one_gpu.err; echo "2 ranks rc=$?"
