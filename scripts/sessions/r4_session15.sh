cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/s15_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s15_pytest.log
for tag in "fit64:--gpr --fit" "fit32:--gpr --fit --dtype f32" "f64:"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $args > gpurun_out/r4b_$name.json 2> gpurun_out/r4b_$name.err
  echo "bench $name rc=$? $(tail -1 gpurun_out/r4b_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M/s',round(d['ms_per_step'],3),'ms', d.get('fit'), d.get('api_inclusive'))" 2>/dev/null)"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
