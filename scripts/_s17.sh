cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python scripts/calibrate_cost.py --out=gpurun_out/cost_table.json > gpurun_out/s17_calibrate.log 2>&1; echo "calibrate rc=$?"; tail -3 gpurun_out/s17_calibrate.log
export GD_COST_TABLE=$PWD/gpurun_out/cost_table.json
for tag in "f64:" "f32:--f32" "grad64:--gradient" "grad32:--gradient --f32"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 1200 python scripts/shard_sim.py $args > gpurun_out/r5_shard_sim_$name.log 2>&1; echo "$name rc=$?"; grep -E "^full step|^world" gpurun_out/r5_shard_sim_$name.log
done
timeout 900 python scripts/gpr_step_sim.py > gpurun_out/r5_gpr_step_sim_f64.log 2>&1; tail -8 gpurun_out/r5_gpr_step_sim_f64.log
timeout 900 python scripts/gpr_step_sim.py --f32 > gpurun_out/r5_gpr_step_sim_f32.log 2>&1; tail -8 gpurun_out/r5_gpr_step_sim_f32.log
