#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files (one pass per counter) into profiles/<round>/pmc_summary.json.

    python tools/summarize_pmc.py --fetch <FETCH_SIZE csv> --write <WRITE_SIZE csv> --out profiles/r1/pmc_summary.json \
        --n 200000 --note "bench.py --steps 20 --warmup 3 --no-cpu-baseline"

Per kernel: mean FETCH_SIZE / WRITE_SIZE per dispatch (KB as rocprofv3 reports them) and HBM bytes per launch
(= (FETCH + WRITE) * 1024).  bench.py reads the entry of its dominant kernel as roofline.traffic.
"""
import argparse
import csv
import json
from collections import defaultdict


def fold(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if "tdv::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").strip()
            a = acc[short]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    fe, wr = fold(a.fetch, "FETCH_SIZE"), fold(a.write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe[k][0] / max(fe[k][1], 1); w = wr[k][0] / max(wr[k][1], 1)
        kernels[k] = {"FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "dispatches": max(fe[k][1], wr[k][1]),
                      "hbm_bytes_per_launch": (f + w) * 1024.0}
    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no tracing combined). " + a.note +
                " Values in KB per dispatch as reported by rocprofv3, mean over every dispatch of the command. FETCH_SIZE: the "
                "guide's gfx950 x2 correction applies to wide 16-B/lane streaming reads; these kernels read through the scalar "
                "path and 4-B lane loads, for which the counter is uncalibrated, so it is reported uncorrected.",
        "workload": {"n_src": a.n, "n_tgt": a.n},
        "kernels": kernels,
    }
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out, "with", len(kernels), "kernels")


if __name__ == "__main__":
    main()
