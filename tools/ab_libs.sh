# A/B of compile-time variants in ONE run on one box WITHOUT building there: the variants are built here (make EXTRA=...) into
# variants/libpcop_<tag>.so, travel with the snapshot, and bench.py picks each through PCOP_LIB.
#   build (here):  tools/ab_libs.sh build tag1 "-DFLAG1" tag2 "-DFLAG2" ...
#   run (box):     [PARITY=1] tools/ab_libs.sh run [kernel pattern] [bench args]     every library under variants, twice, default first and last
cmd=$1; shift
C=pointcloud_obstacle_processing_amd/csrc
if [ "$cmd" = build ]; then
  mkdir -p variants
  while [ $# -ge 2 ]; do
    tag=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -Wno-unused-result $flags -shared -o variants/libpcop_$tag.so $C/pcop_capi.hip 2>&1 | grep -E " error" ; ls -la variants/libpcop_$tag.so
  done
  exit 0
fi
pat=${1:-reduce}; shift
EXTRA_ARGS="$*"
run() {  # tag lib
  # PARITY=1: two frames of the timed batch are compared with the CPU restatement in every run (a variant nobody has run yet); default: no CPU leg
  local cpu="--no-cpu-baseline"; [ -n "$PARITY" ] && cpu="--cpu-frames 2"
  PCOP_LIB=$2 python bench.py --steps 8 --warmup 2 $cpu --no-end-to-end --no-secondary $EXTRA_ARGS > gpurun_out/abl_$1.json 2> gpurun_out/abl_$1.err || { echo "bench failed ($1)"; tail -3 gpurun_out/abl_$1.err; return 0; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$1.json").read().strip().splitlines()[-1])
print("$1: %.0f frames/s" % d["value"], "| parity_checked", d.get("parity_checked"), "|", d.get("parity", "no parity leg in this run"))
PY
  python tools/kernel_table.py gpurun_out/abl_$1.json | grep -E "$pat" || true
}
D=$PWD/$C/libpcop.so
run default_1 $D
for l in variants/libpcop_*.so; do t=$(basename $l .so); t=${t#libpcop_}; run ${t}_1 $PWD/$l; done
for l in variants/libpcop_*.so; do t=$(basename $l .so); t=${t#libpcop_}; run ${t}_2 $PWD/$l; done
run default_2 $D
