set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_box.py tests/test_gpu_tiled.py -x -q > gpurun_out/t_box.log 2>&1
tail -3 gpurun_out/t_box.log
for b in 16 24 32; do
  for c in 2 3; do
    python bench.py --batch $b --contexts $c --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/sw_${b}_${c}.json 2>gpurun_out/sw_${b}_${c}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/sw_${b}_${c}.json").read().strip().splitlines()[-1])
print("batch $b ctx $c", d["value"], d["ms_per_step"])
PY
  done
done
