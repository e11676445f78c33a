# A/B of one environment switch on the bench line, both settings in one run on one box (twice each, interleaved).
# usage: tools/ab_env.sh VAR "0 1" [kernel name pattern for the per-kernel table]
set -e
mkdir -p gpurun_out
v=$1; vals=${2:-"0 1"}; pat=${3:-scatter}
for rep in 1 2; do
  for x in $vals; do
    env $v=$x python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary > gpurun_out/ab_${v}_${x}_$rep.json 2>gpurun_out/ab_${v}_${x}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${v}_${x}_$rep.json").read().strip().splitlines()[-1])
print("$v=$x rep $rep:", round(d["value"], 1), "frames/s; dominant", d["roofline"]["kernel"], round(1e3 * d["roofline"]["avg_launch_ms"], 1), "us")
PY
    python tools/kernel_table.py gpurun_out/ab_${v}_${x}_$rep.json | grep -E "$pat" || true
  done
done
