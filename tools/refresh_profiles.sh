#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel trace and the two PMC passes of the
# same command, reduced to the files kept under profiles/.  usage: tools/refresh_profiles.sh <round tag, e.g. r02> [batch]
set -e
tag=${1:-r02}
B=${2:-32}
PART=${3:-all}   # all | a (the bench line, the traces, the counters) | b (configs, the steps next to the path, the pool, the fuzzer)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh_$tag
if [ "$PART" != b ]; then
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-secondary > $O/trace.log 2>&1
echo "trace done"
# the same trace with ONE context (the kernels alone on the GPU): what bench.py's HIP-event durations can be reproduced from
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_solo -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --no-secondary > $O/trace_solo.log 2>&1
echo "solo trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --rounds 3 --no-cpu-baseline --no-end-to-end --no-secondary --contexts 1 > $O/pmc_$c.log 2>&1
done
# the "why" of the LDS-heavy kernels: bank conflicts against LDS-active cycles, LDS issue stalls, wave cycles (one pass: 8 SQ counters fit)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --rounds 3 --no-cpu-baseline --no-end-to-end --no-secondary --contexts 1 > $O/pmc_sq.log 2>&1
echo "pmc done"
# single-frame latency: the kernel trace of one-frame calls
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -o trace -- python3 $R/tools/single_frame.py > $O/trace1.log 2>&1
echo "single-frame trace done"
cd $R
T=$(dirname $(find $O/trace -name trace_kernel_trace.csv | head -1))
python3 tools/prof_summary.py $T $B > $O/kernel_summary_batch$B.txt
TS=$(dirname $(find $O/trace_solo -name trace_kernel_trace.csv | head -1))
python3 tools/prof_summary.py $TS $B > $O/kernel_summary_batch${B}_solo.txt
cp $(find $O/trace -name trace_kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats.csv
T1=$(dirname $(find $O/trace1 -name trace_kernel_trace.csv | head -1))
python3 tools/prof_summary.py $T1 1 > $O/kernel_summary_single_frame.txt
F=$(dirname $(find $O/pmc_FETCH_SIZE -name pmc_counter_collection.csv | head -1))
W=$(dirname $(find $O/pmc_WRITE_SIZE -name pmc_counter_collection.csv | head -1))
python3 tools/pmc_summary.py $F $W $B $O/pmc_traffic.json > $O/pmc_summary.txt
python3 tools/pmc_table.py $O/pmc_sq > $O/sq_lds.txt
head -14 $O/kernel_summary_batch$B.txt; head -10 $O/pmc_summary.txt; tail -3 $O/kernel_summary_single_frame.txt
python3 - <<PY
import json; d=json.load(open("$O/bench.json")); print(d["value"], d["latency_ms_single_frame"], d["roofline"], d.get("cpu_baseline"), d.get("end_to_end"), d.get("parity"))
PY
fi
if [ "$PART" != a ]; then
mkdir -p $O; cd $R
python3 tools/pool_bench.py 1 512 16 > $O/pool.json 2> $O/pool.err || true
cat $O/pool.json
python3 tests/tools/config_report.py > $O/configs.txt 2> $O/configs.err
python3 tests/tools/occ_bench.py > $O/occupancy.json
python3 tests/tools/sor_bench.py > $O/sor.json
python3 tests/tools/node_bench.py > $O/node_chain.json
python3 tests/tools/fuzz_stateful.py 90 5 2>&1 | grep -v "^  \.\.\." | tail -3 > $O/fuzz_stateful.txt
echo "configs / occupancy / sor / node chain done"
fi
