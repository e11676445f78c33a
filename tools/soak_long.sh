# The long campaign: thousands of fuzz cases and many hundred stateful steps, summary lines to gpurun_out/soak_long.txt
# usage: tools/soak_long.sh [first fuzz seed] [first stateful seed]   (about 17 minutes; the batch fuzzer has its own script, tools/soak_batches.sh, about 15)
set -e
mkdir -p gpurun_out
export FUZZ_CHECK_CLEAN=1   # after every call: occupancy flags zero, cell table empty
F0=${1:-31}; S0=${2:-41}
: > gpurun_out/soak_long.txt
for s in $F0 $((F0+1)) $((F0+2)); do
  timeout -k 10 400 python tests/tools/fuzz.py 3000 $s > gpurun_out/soak_long_fuzz_$s.log 2>&1 || echo "fuzz seed $s: exit $?" >> gpurun_out/soak_long.txt
  echo "fuzz.py 3000 cases seed $s: $(tail -1 gpurun_out/soak_long_fuzz_$s.log)" >> gpurun_out/soak_long.txt
done
for s in $S0 $((S0+1)); do
  timeout -k 10 420 python tests/tools/fuzz_stateful.py 600 $s > gpurun_out/soak_long_stateful_$s.log 2>&1 || echo "stateful seed $s: exit $?" >> gpurun_out/soak_long.txt
  echo "fuzz_stateful.py 600 steps seed $s (PCOP_VOXEL_PATH=stream): $(tail -1 gpurun_out/soak_long_stateful_$s.log)" >> gpurun_out/soak_long.txt
done
PCOP_VOXEL_PATH=auto timeout -k 10 420 python tests/tools/fuzz_stateful.py 600 $((S0+2)) > gpurun_out/soak_long_stateful_$((S0+2)).log 2>&1 || echo "stateful seed $((S0+2)): exit $?" >> gpurun_out/soak_long.txt
echo "fuzz_stateful.py 600 steps seed $((S0+2)) (default path choice): $(tail -1 gpurun_out/soak_long_stateful_$((S0+2)).log)" >> gpurun_out/soak_long.txt
PCOP_CELL_PATH=hash timeout -k 10 420 python tests/tools/fuzz_stateful.py 300 $((S0+3)) > gpurun_out/soak_long_stateful_$((S0+3)).log 2>&1 || echo "stateful seed $((S0+3)): exit $?" >> gpurun_out/soak_long.txt
echo "fuzz_stateful.py 300 steps seed $((S0+3)) (PCOP_CELL_PATH=hash): $(tail -1 gpurun_out/soak_long_stateful_$((S0+3)).log)" >> gpurun_out/soak_long.txt
cat gpurun_out/soak_long.txt
