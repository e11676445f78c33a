# Several processes at once (three by default) on the one GPU, each its own long-lived context and its own seed of the batch fuzzer: the look-back
# waits and the re-plans under contention from other processes' kernels.  usage: tools/soak_concurrent.sh [first seed] [steps] [processes, at most 5]
mkdir -p gpurun_out
export FUZZ_CHECK_CLEAN=1
S0=${1:-141}; N=${2:-500}; P=${3:-3}
seeds=$(seq $S0 $((S0+P-1)))
pids=""
for s in $seeds; do
  timeout -k 10 900 python tests/tools/fuzz_batches.py $N $s > gpurun_out/soak_conc_$s.log 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=1; done
: > gpurun_out/soak_concurrent.txt
for s in $seeds; do echo "fuzz_batches.py $N steps seed $s (one of $P at once): $(tail -1 gpurun_out/soak_conc_$s.log)" >> gpurun_out/soak_concurrent.txt; done
cat gpurun_out/soak_concurrent.txt
if grep -q "Memory access fault" gpurun_out/soak_conc_*.log; then exit 3; fi
exit $rc
