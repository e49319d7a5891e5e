for occ in "" "1:16:4,1:24:3,1:32:2" "1:16:5,1:24:4,1:32:3" "1:16:6,1:24:5,1:32:4" "1:16:8,1:24:6,1:32:5"; do
GD_OCCUPANCY="$occ" python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(repr(os.environ.get('GD_OCCUPANCY')), round(d['value']/1e6,2), 'Mpairs/s', round(d['ms_per_step'],2), [(k['kernel'][4:], round(k['avg_ms'],3)) for k in d['kernels']])"
done
