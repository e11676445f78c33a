# after the fix: the segment that faulted, with the cleanliness check, then the whole seed
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t_fuzz.log 2>&1; rc=$?; tail -3 gpurun_out/t_fuzz.log
if [ $rc -ne 0 ]; then exit 3; fi
timeout -k 10 300 python tests/tools/fuzz.py 3000 32 > gpurun_out/fuzz32_full.log 2>&1; rc=$?; echo "seed 32 full: exit $rc: $(tail -1 gpurun_out/fuzz32_full.log)"
if [ $rc -ne 0 ]; then exit 3; fi
