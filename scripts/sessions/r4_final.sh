# final pass of the round: GPU suite, smoke, every bench line, profiles of the changed workloads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/sessions/r4_bench_all.sh 2>&1 | tail -18
PROFILE_TAGS="c2f64:--config\ 2\ --dtype\ f64" true
for tag in "c2f64:--config 2 --dtype f64" "f64:"; do
  name=${tag%%:*}; args=${tag#*:}
  BENCH_ARGS="$args" bash scripts/profile.sh > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name && mv gpurun_out/prof gpurun_out/prof_$name
  find gpurun_out/prof_$name -name "*kernel_trace.csv" -delete
  echo "profiled $name"
done
