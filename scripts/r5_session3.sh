#!/bin/bash
# Round 5, session 3: every step under its own timeout, timestamps in the log.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
log() { echo "[$(date +%H:%M:%S)] $*"; }
log start
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed or large_pair or undefined_on_empty" > gpurun_out/s3_pytest_large.log 2>&1
log "pytest large rc=$?"; tail -5 gpurun_out/s3_pytest_large.log
for dt in f32 f64; do
  timeout 400 python bench.py --config large --graphs 16 --dtype $dt --steps 3 --warmup 1 --no-api --no-f32 --cpu-seconds 4 > gpurun_out/s3_large16_stream_$dt.json 2> gpurun_out/s3_large16_stream_$dt.err
  log "stream16 $dt rc=$?"; head -c 330 gpurun_out/s3_large16_stream_$dt.json; echo
done
timeout 300 python scripts/mfma_experiment.py > gpurun_out/s3_mfma.json 2> gpurun_out/s3_mfma.err
log "mfma rc=$?"; head -c 1500 gpurun_out/s3_mfma.json; echo; tail -3 gpurun_out/s3_mfma.err
timeout 1200 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x -k "ranks_through" > gpurun_out/s3_pytest_dist.log 2>&1
log "pytest dist rc=$?"; tail -5 gpurun_out/s3_pytest_dist.log
timeout 400 python bench.py --gpr --gpus 2 --share-devices --graphs 300 --steps 3 --warmup 1 > gpurun_out/s3_gpr2.json 2> gpurun_out/s3_gpr2.err
log "gpr 2 ranks rc=$?"; tail -c 1200 gpurun_out/s3_gpr2.json; echo; tail -3 gpurun_out/s3_gpr2.err
GD_GPR_OVERLAP=1 timeout 400 python bench.py --gpr --gpus 2 --share-devices --graphs 300 --steps 3 --warmup 1 > gpurun_out/s3_gpr2_overlap.json 2> gpurun_out/s3_gpr2_overlap.err
log "gpr 2 ranks overlapped rc=$?"; tail -c 600 gpurun_out/s3_gpr2_overlap.json; echo; tail -3 gpurun_out/s3_gpr2_overlap.err
timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -k "pairlist or features or spatial or sharded" > gpurun_out/s3_pytest_fuzz.log 2>&1
log "pytest fuzz rc=$?"; tail -12 gpurun_out/s3_pytest_fuzz.log
log done
