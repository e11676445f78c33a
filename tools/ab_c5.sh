# A/B of one environment switch on C5 (5 M points, batch of 2, one context) and on single-frame latency of C3
mkdir -p gpurun_out
v=$1; vals=${2:-"0 1"}
for rep in 1 2; do for x in $vals; do
  env $v=$x python bench.py --config C5 --batch 2 --contexts 1 --steps 3 --warmup 1 --rounds 3 --no-cpu-baseline --no-end-to-end > gpurun_out/abc5_${x}_$rep.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/abc5_${x}_$rep.json').read().strip().splitlines()[-1]); print('$v=$x rep $rep: C5', round(d['value'],1), 'frames/s, latency', round(d['latency_ms_single_frame'],3), 'ms')"
done; done
