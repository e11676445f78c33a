# The bench line against frames per launch and contexts in flight.  usage: tools/sweep_batch.sh "16 24 32" "2 3"
set -e
mkdir -p gpurun_out
for b in ${1:-16 24 32}; do
  for c in ${2:-2 3}; do
    python bench.py --batch $b --contexts $c --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary > gpurun_out/sw_${b}_${c}.json 2>gpurun_out/sw_${b}_${c}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/sw_${b}_${c}.json").read().strip().splitlines()[-1])
print("batch $b ctx $c", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step")
PY
  done
done
