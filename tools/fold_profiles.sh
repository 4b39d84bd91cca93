#!/bin/bash
# Here (not on the GPU box): copy what tools/run_profiles.sh left under gpurun_out/<round>prof into profiles/<round> (TDV_ROUND, default r3) and fold the PMC
# passes.  gpurun merges new files into old directories, so the newest file of each pass is taken.
set -eu
cd "$(dirname "$0")/.."
ROUND="${TDV_ROUND:-r3}"
O="gpurun_out/${ROUND}prof"; P="profiles/${ROUND}"
[ -d "$O" ] || { echo "nothing under $O" >&2; exit 1; }
mkdir -p "$P"
nf() { ls -t $O/$1/runc/*$2 | head -1; }
cp $O/bench.json $O/bench_under_rocprof.json $O/bench_ops.jsonl $O/bench_batch_c4_256.jsonl $P/
cp $(nf kt_bench kernel_stats.csv) $P/kernel_stats_bench.csv
cp $(nf kt_ops kernel_stats.csv) $P/kernel_stats_ops.csv
[ -d $O/kt_c5 ] && cp $(nf kt_c5 kernel_stats.csv) $P/kernel_stats_c5.csv
[ -f $O/bench_c5_1gpu.jsonl ] && cp $O/bench_c5_1gpu.jsonl $P/
N1="rocprofv3 --pmc passes, one counter group per pass, no tracing combined; means over every dispatch of the command. Command: python3 bench.py --no-cpu-baseline --no-operators (100 steps + 5 warm-up at 200k x 200k)."
N2="rocprofv3 --pmc passes, one counter group per pass, no tracing combined; means over every dispatch of the command. Command: python3 tools/bench_ops.py (every operator of tools/opbench.py, median-of-3 loops: each kernel is dispatched several times)."
python tools/summarize_pmc.py --out $P/pmc_summary.json --n 200000 --note "$N1" $(nf pmc_fetch_bench counter_collection.csv) $(nf pmc_write_bench counter_collection.csv)
python tools/summarize_pmc.py --out $P/pmc_ops_summary.json --n 200000 --note "$N2" $(nf pmc_fetch_ops counter_collection.csv) $(nf pmc_write_ops counter_collection.csv) $(nf pmc_sq_ops counter_collection.csv)
