# the batch fuzzer alone, long: usage tools/soak_batches.sh [first seed]
mkdir -p gpurun_out
export FUZZ_CHECK_CLEAN=1
S0=${1:-71}
: > gpurun_out/soak_batches.txt
for s in $S0 $((S0+1)) $((S0+2)); do
  timeout -k 10 330 python tests/tools/fuzz_batches.py 1200 $s > gpurun_out/soak_batches_$s.log 2>&1; rc=$?
  echo "fuzz_batches.py 1200 steps seed $s: exit $rc: $(tail -1 gpurun_out/soak_batches_$s.log)" >> gpurun_out/soak_batches.txt
  if grep -q "Memory access fault" gpurun_out/soak_batches_$s.log; then cat gpurun_out/soak_batches.txt; exit 3; fi
done
PCOP_VOXEL_BOX=0 timeout -k 10 330 python tests/tools/fuzz_batches.py 1200 $((S0+3)) > gpurun_out/soak_batches_$((S0+3)).log 2>&1; rc=$?
echo "fuzz_batches.py 1200 steps seed $((S0+3)) (PCOP_VOXEL_BOX=0): exit $rc: $(tail -1 gpurun_out/soak_batches_$((S0+3)).log)" >> gpurun_out/soak_batches.txt
cat gpurun_out/soak_batches.txt
