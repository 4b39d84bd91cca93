#!/bin/bash
# One gpurun call: the bench line, the kernel trace of the same command, the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) for
# bench.py and for tools/bench_ops.py, and the full C4 batch.  Programs are started directly after `rocprofv3 ... --`.
set -eu
: "${GRAFT_REPO_ROOT:?must be set (gpurun exports it on the GPU box)}"
ROUND="${TDV_ROUND:-r4}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/${ROUND}prof"
[ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }
rm -rf "$O"; mkdir -p "$O"
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-operators"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- $B > $O/bench_under_rocprof.json 2> $O/kt_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_bench -- $B > /dev/null 2> $O/pmc_fetch_bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_bench -- $B > /dev/null 2> $O/pmc_write_bench.err
P="python3 $R/tools/bench_ops.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ops -- $P > $O/bench_ops_under_rocprof.jsonl 2> $O/kt_ops.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_ops -- $P > /dev/null 2> $O/pmc_fetch_ops.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_ops -- $P > /dev/null 2> $O/pmc_write_ops.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_ops -- $P > /dev/null 2> $O/pmc_sq_ops.err || echo "SQ pass failed" >> $O/notes.txt
python3 $R/tools/bench_ops.py > $O/bench_ops.jsonl 2> $O/bench_ops.err
python3 $R/tools/bench_batch.py --instances 256 > $O/bench_batch_c4_256.jsonl 2> $O/bench_batch_c4_256.err
python3 $R/tools/bench_batch.py --instances 256 --order first >> $O/bench_batch_c4_256.jsonl 2>> $O/bench_batch_c4_256.err
python3 $R/tools/bench_batch.py --instances 256 --voxel-px 2.0 >> $O/bench_batch_c4_256.jsonl 2>> $O/bench_batch_c4_256.err
TDV_BATCH_LANES=1 python3 $R/tools/bench_batch.py --instances 64 >> $O/bench_batch_c4_256.jsonl 2>> $O/bench_batch_c4_256.err
# C5, one rank's share (1,024 instances from one label image): the line, its kernel trace, and the unbatched shapes for comparison
python3 $R/tools/c5_tray.py --reps 3 > $O/bench_c5_1gpu.jsonl 2> $O/bench_c5_1gpu.err
TDV_RANSAC_BATCH=0 python3 $R/tools/c5_tray.py --reps 2 >> $O/bench_c5_1gpu.jsonl 2>> $O/bench_c5_1gpu.err
TDV_BATCH_STAGED=0 python3 $R/tools/c5_tray.py --reps 2 >> $O/bench_c5_1gpu.jsonl 2>> $O/bench_c5_1gpu.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -- python3 $R/tools/c5_tray.py --reps 1 > /dev/null 2> $O/kt_c5.err
# ICP with the reference's accumulation order beside the tree, 200k x 200k, both ICP modes: rates and the kernel trace (k_icp_flags / scan_counts / rows / fold_ref)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_icpref -- python3 $R/tools/studies/icp_ref_order_probe.py 50000 200000 > $O/icp_reference_order.jsonl 2> $O/kt_icpref.err
# fold the PMC passes here (the per-dispatch counter tables are hundreds of MB with the 256- and 1,024-instance batches in the run;
# gpurun copies back at most 64 MiB), then keep only statistics and summaries
nf() { ls -t $O/$1/*/*$2 2>/dev/null | head -1; }
N1="rocprofv3 --pmc passes, one counter group per pass, no tracing combined; means over every dispatch of the command. Command: python3 bench.py --no-cpu-baseline --no-operators (100 steps + 5 warm-up at 200k x 200k)."
N2="rocprofv3 --pmc passes, one counter group per pass, no tracing combined; means over every dispatch of the command. Command: python3 tools/bench_ops.py (every operator of tools/opbench.py: each kernel is dispatched several times)."
python3 $R/tools/summarize_pmc.py --out $O/pmc_summary.json --n 200000 --note "$N1" $(nf pmc_fetch_bench counter_collection.csv) $(nf pmc_write_bench counter_collection.csv) || echo "pmc fold (bench) failed" >> $O/notes.txt
python3 $R/tools/summarize_pmc.py --out $O/pmc_ops_summary.json --n 200000 --note "$N2" $(nf pmc_fetch_ops counter_collection.csv) $(nf pmc_write_ops counter_collection.csv) $(nf pmc_sq_ops counter_collection.csv) || echo "pmc fold (ops) failed" >> $O/notes.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O; cat $O/bench.json | head -c 600
