# single-frame latency and C5 throughput against the digit width of the sort path (PCOP_RADIX_BITS)
set -e
mkdir -p gpurun_out
for bits in 8 11; do
  PCOP_RADIX_BITS=$bits python bench.py --steps 3 --warmup 1 --rounds 5 --no-cpu-baseline --no-end-to-end > gpurun_out/rb_c3_$bits.json 2>gpurun_out/rb_c3_$bits.err
  PCOP_RADIX_BITS=$bits python bench.py --config C5 --batch 2 --contexts 1 --steps 3 --warmup 1 --rounds 3 --no-cpu-baseline --no-end-to-end > gpurun_out/rb_c5_$bits.json 2>gpurun_out/rb_c5_$bits.err
  python - <<PY
import json
for c in ("c3","c5"):
    d=json.loads(open("gpurun_out/rb_%s_$bits.json"%c).read().strip().splitlines()[-1])
    print("bits $bits", c, "fps", round(d["value"],1), "latency_ms", d.get("latency_ms_single_frame"))
PY
done
