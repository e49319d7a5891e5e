#!/bin/bash
# round 3, session 11: SQ counters of the fp32 step with and without the grid walk / scalar loads
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --dtype f32 --no-api --serial --isolated-steps 0"
for tag in on off; do
  if [ $tag = off ]; then export GD_HIPCC_EXTRA="-DGD_OC_GRID=0 -DGD_OC_SLOAD=0"; fi
  OUT=gpurun_out/ab_prof_$tag
  rm -rf $OUT && mkdir -p $OUT
  python3 bench.py $ARGS > $OUT/warm.json 2> $OUT/warm.err     # JIT outside the profiler
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > $OUT/pmc_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > $OUT/pmc_b.log 2>&1
  ls $OUT
done
