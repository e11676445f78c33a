# Clouds up to 1 M points on one long-lived context (FUZZ_BIG=1), two seeds, progress to gpurun_out/soak_big_*.log.
set -e
mkdir -p gpurun_out
s=${1:-21}
FUZZ_BIG=1 timeout -k 10 520 python tests/tools/fuzz_stateful.py 60 $s > gpurun_out/soak_big_$s.log 2>&1 || true
tail -2 gpurun_out/soak_big_$s.log
FUZZ_BIG=1 PCOP_VOXEL_PATH=auto timeout -k 10 520 python tests/tools/fuzz_stateful.py 60 $((s+1)) > gpurun_out/soak_big_$((s+1)).log 2>&1 || true
tail -2 gpurun_out/soak_big_$((s+1)).log
