#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run through gpurun from the
# repo root).  Kernel trace + stats, then SQ counters, then the two TCC
# traffic counters in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC
# slots).  python3 is named directly: the profiler's preloaded library must
# not see an exec() hop.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
# (--serial: every launch alone on one stream, so that the per-kernel averages of
# the trace are the isolated durations bench.py quotes in roofline.avg_launch_ms)
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-api --serial --isolated-steps 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
python3 bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
