set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/bench_v7.json 2> $O/bench_v7.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v7 -- python3 $R/bench.py --no-cpu-baseline > $O/prof_v7_bench.json 2> $O/prof_v7.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_v7 -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2> $O/pmc_fetch_v7.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_v7 -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2> $O/pmc_write_v7.err
cat $O/bench_v7.json
