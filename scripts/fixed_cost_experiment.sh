cd $GRAFT_REPO_ROOT
for dt in f64 f32; do
for ms in 1 2 3; do
GD_MAX_STREAMS=$ms timeout 300 python bench.py --sharded --graphs 354 --dtype $dt --no-cpu-baseline --no-api --no-f32 --steps 200 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$dt streams $ms:', round(d['ms_per_step'],4), 'ms', d.get('phases_per_rank'), 'launches', len(d.get('kernels',[])))"
done; done
