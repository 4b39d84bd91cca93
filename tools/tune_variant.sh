#!/bin/bash
# Tuning helper (GPU box): rebuild ONE source with extra -D flags and run a command.
#   tools/tune_variant.sh fmatch.hip "-DFMQ_WAVES_PER_SIMD=6" python tools/bench_fm.py --paths index
src=$1; flags=$2; shift 2
touch 3dvision_amd/csrc/$src
TDV_HIPCC_FLAGS="$flags" python 3dvision_amd/build.py > /dev/null || exit 1
echo "== $src $flags"
"$@"
