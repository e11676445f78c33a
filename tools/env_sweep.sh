#!/bin/bash
# usage: tools/env_sweep.sh VAR v1 v2 ... ; runs bench.py for each value and prints value + top kernels
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$var=$v FAILED"; tail -3 gpurun_out/sw.err; continue; }
  python - "$var=$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sw.json"))
k=d["kernel_ms"]
sel=["k_voxel_stream","k_vid_scatter","k_vox_mark","k_vid_hist","k_cgrid_union","k_cgrid_hook","k_cell_flatten","k_emit"]
print(sys.argv[1], round(d["value"],1), "lat", round(d["latency_ms_single_frame"],3), {s[2:]:k.get(s) for s in sel})
PY
done
