# A/B of a compile-time constant in one run on one box: the bench line with the library as built, then with the library rebuilt
# with EXTRA flags (twice each, the default build restored at the end).  usage: tools/ab_build.sh "-DPCOP_REDUCE_BPB=4" [kernel pattern]
mkdir -p gpurun_out
flags="$1"; pat=${2:-reduce}
run() {
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/abb_$1.json 2>gpurun_out/abb_$1.err || { echo "bench failed ($1)"; tail -3 gpurun_out/abb_$1.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abb_$1.json").read().strip().splitlines()[-1])
print("$1:", round(d["value"], 1), "frames/s; parity", d.get("parity_checked"))
PY
  python tools/kernel_table.py gpurun_out/abb_$1.json | grep -E "$pat" || true
}
run default_1 || exit 3
make -s -C pointcloud_obstacle_processing_amd/csrc clean && make -s -C pointcloud_obstacle_processing_amd/csrc EXTRA="$flags" || exit 3
run variant_1 || exit 3; run variant_2 || exit 3
make -s -C pointcloud_obstacle_processing_amd/csrc clean && make -s -C pointcloud_obstacle_processing_amd/csrc || exit 3
run default_2
