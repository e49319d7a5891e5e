#!/bin/bash
# Two SQ counter passes over bench.py (no kernel trace): instruction mix and
# LDS / wait cycles per kernel, printed as a table.  Run through gpurun from
# the repo root; BENCH_ARGS and GD_HIPCC_EXTRA are passed through.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-api --isolated-steps 0 ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
python3 scripts/pmc_table.py $OUT
