# A longer shake of the adaptive host state than the test suite has time for: the two fuzzers with fresh seeds, on the default
# voxel path choice, on the streaming path for single frames too, and with the bounds pass kept (PCOP_VOXEL_BOX=0).  Each line of progress
# goes to gpurun_out/soak_*.log as it is printed.  usage: tools/soak.sh [seed0]
set -e
mkdir -p gpurun_out
s=${1:-11}
python tests/tools/fuzz.py 400 $s > gpurun_out/soak_fuzz_$s.log 2>&1 && tail -1 gpurun_out/soak_fuzz_$s.log
python tests/tools/fuzz_stateful.py 240 $((s+1)) > gpurun_out/soak_stateful_stream_$((s+1)).log 2>&1 && tail -1 gpurun_out/soak_stateful_stream_$((s+1)).log
PCOP_VOXEL_PATH=auto python tests/tools/fuzz_stateful.py 240 $((s+2)) > gpurun_out/soak_stateful_auto_$((s+2)).log 2>&1 && tail -1 gpurun_out/soak_stateful_auto_$((s+2)).log
PCOP_VOXEL_BOX=0 python tests/tools/fuzz_stateful.py 160 $((s+3)) > gpurun_out/soak_stateful_nobox_$((s+3)).log 2>&1 && tail -1 gpurun_out/soak_stateful_nobox_$((s+3)).log
