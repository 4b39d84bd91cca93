#!/bin/bash
# Here (not on the GPU box): copy what tools/run_profiles.sh left under gpurun_out/<round>prof into profiles/<round> (TDV_ROUND, default r4) and fold the PMC
# passes.  gpurun merges new files into old directories, so the newest file of each pass is taken.
set -eu
cd "$(dirname "$0")/.."
ROUND="${TDV_ROUND:-r4}"
O="gpurun_out/${ROUND}prof"; P="profiles/${ROUND}"
[ -d "$O" ] || { echo "nothing under $O" >&2; exit 1; }
mkdir -p "$P"
nf() { ls -t $O/$1/runc/*$2 | head -1; }
cp $O/bench.json $O/bench_under_rocprof.json $O/bench_ops.jsonl $O/bench_batch_c4_256.jsonl $P/
cp $(nf kt_bench kernel_stats.csv) $P/kernel_stats_bench.csv
cp $(nf kt_ops kernel_stats.csv) $P/kernel_stats_ops.csv
[ -d $O/kt_c5 ] && cp $(nf kt_c5 kernel_stats.csv) $P/kernel_stats_c5.csv
[ -f $O/bench_c5_1gpu.jsonl ] && cp $O/bench_c5_1gpu.jsonl $P/
[ -d $O/kt_icpref ] && cp $(nf kt_icpref kernel_stats.csv) $P/kernel_stats_icp_reference_order.csv && grep "^{" $O/icp_reference_order.jsonl > $P/icp_reference_order.jsonl
cp $O/pmc_summary.json $O/pmc_ops_summary.json $P/
