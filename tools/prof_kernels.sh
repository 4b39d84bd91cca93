#!/bin/bash
# GPU box: rocprofv3 kernel trace of a python tool, top kernels printed.  usage: tools/prof_kernels.sh <tag> <script.py> [args...]
set -u
: "${GRAFT_REPO_ROOT:?must be set (gpurun exports it on the GPU box)}"
tag="${1:?usage: tools/prof_kernels.sh <tag> <script.py> [args...]}"; shift
[ -n "$tag" ] && [ $# -ge 1 ] || { echo "usage: tools/prof_kernels.sh <tag> <script.py> [args...]" >&2; exit 2; }
out="$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
rm -rf "$out"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/"$@" > $out.stdout 2> $out.stderr)
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-58s calls %5s avg %10.3f us total %9.3f ms %6s%%" % (r["Name"].split("(")[0][-58:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
