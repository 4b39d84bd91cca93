#!/bin/bash
# GPU box: SQ counters of one python tool, per kernel (own pass, no tracing combined).  usage: tools/pmc_sq.sh <tag> <script.py> [args...]
set -u
: "${GRAFT_REPO_ROOT:?must be set (gpurun exports it on the GPU box)}"
tag="${1:?usage: tools/pmc_sq.sh <tag> <script.py> [args...]}"; shift
[ -n "$tag" ] && [ $# -ge 1 ] || { echo "usage: tools/pmc_sq.sh <tag> <script.py> [args...]" >&2; exit 2; }
out="$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag"
rm -rf "$out"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/"$@" > $out.stdout 2> $out.stderr)
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "tdv::" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    w = max(v["SQ_WAVES"], 1); wc = max(v["SQ_WAVE_CYCLES"], 1)
    print("  %-44s disp %4d waves/disp %7d valu/wave %5d salu/wave %5d active %3.0f%% wait_any %3.0f%% wait_inst %3.0f%%" % (
        k[-44:], n[k], w / max(n[k], 1), v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, 100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc))
PY
