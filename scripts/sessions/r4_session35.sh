# do page faults of fresh large host arrays matter in a first call?  glibc malloc tunables from the environment
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
one() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); a=d['api_inclusive']; print(round(a['first_call_ms'],2), a['fresh_backend_call_ms'], round(a['repeat_call_ms'],2))"; }
for rep in 1 2; do
timeout 600 python bench.py --steps 50 --cpu-seconds 2 --no-f32 > gpurun_out/s35_a.json 2> gpurun_out/s35_a.err; echo -n "default malloc: "; one gpurun_out/s35_a.json
MALLOC_MMAP_THRESHOLD_=268435456 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864 timeout 600 python bench.py --steps 50 --cpu-seconds 2 --no-f32 > gpurun_out/s35_b.json 2> gpurun_out/s35_b.err; echo -n "no mmap, no trim: "; one gpurun_out/s35_b.json
done
