# Who runs beside whom: rocprofv3 --kernel-trace of the default bench line, reduced by tools/timeline.py to per-queue busy
# fractions, how many heavy / light kernels run at once, mean durations under co-running, and the gaps between heavy kernels.
# usage (on the GPU box): bash tools/timeline.sh [tag]     -> gpurun_out/timeline_<tag>.txt (+ the trace CSV)
set -e
tag=${1:-default}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary > $R/gpurun_out/timeline_$tag.json 2>$R/gpurun_out/timeline_$tag.err
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
cp $f $R/gpurun_out/timeline_${tag}_kernel_trace.csv
python3 $R/tools/timeline.py $f | tee $R/gpurun_out/timeline_$tag.txt
