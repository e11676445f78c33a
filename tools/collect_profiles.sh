#!/bin/bash
# Run locally after `gpurun -- bash tools/refresh_profiles.sh <tag> <batch>`: copies the reduced files from gpurun_out/ into profiles/.
set -e
tag=${1:-r02}; B=${2:-32}
O=gpurun_out/refresh_$tag
cp $O/bench.json profiles/${tag}_bench_n1.json
cp $O/rocprofv3_kernel_stats.csv profiles/${tag}_rocprofv3_kernel_stats.csv
cp $O/kernel_summary_batch$B.txt profiles/${tag}_kernel_summary_batch$B.txt
cp $O/kernel_summary_batch${B}_solo.txt profiles/${tag}_kernel_summary_batch${B}_solo.txt
cp $O/kernel_summary_single_frame.txt profiles/${tag}_kernel_summary_single_frame.txt
cp $O/pmc_summary.txt profiles/${tag}_pmc_summary.txt
cp $O/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $O/configs.txt profiles/${tag}_configs.txt
cp $O/occupancy.json profiles/${tag}_occupancy.json
cp $O/sor.json profiles/${tag}_sor.json
cp $O/node_chain.json profiles/${tag}_node_chain.json
cp $O/sq_lds.txt profiles/${tag}_sq_lds.txt
cp $O/fuzz_stateful.txt profiles/${tag}_fuzz_stateful.txt
[ -s $O/pool.json ] && cp $O/pool.json profiles/${tag}_pool.json
ls -la profiles/${tag}_*
