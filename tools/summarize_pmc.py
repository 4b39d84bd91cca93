#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files (one pass per counter group) into one per-kernel JSON summary.

    python tools/summarize_pmc.py --out profiles/r2/pmc_summary.json --n 200000 --note "..." pass1.csv pass2.csv ...

Per kernel of this library: the mean of every collected counter per dispatch, and for the HBM counters
    hbm_bytes_per_launch = (fetch_factor * FETCH_SIZE + WRITE_SIZE) * 1024        (rocprofv3 reports both in KB)
fetch_factor: /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read (16 B per lane), exact WRITE_SIZE for 16-B stores; other access widths are uncalibrated.  The
factor 2 is therefore applied only to the kernels in WIDE_READERS (their dominant reads are 16-B-per-lane streams); every
other kernel's FETCH_SIZE is reported as counted, and the entry says which rule was used.  bench.py reads its dominant
kernel's hbm_bytes_per_launch as roofline.traffic.
"""
import argparse
import csv
import json
from collections import defaultdict

WIDE_READERS = {
    "k_depth_bits": "16-B mask and depth loads per lane",
    "k_depth_preprocess": "8-B depth + 4-B mask per lane (narrower than the calibrated shape: factor applied as an upper estimate)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in a.csv:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "tdv::" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "").strip()
                c = acc[short][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"]); c[1] += 1
    kernels = {}
    for k in sorted(acc):
        e = {"dispatches": max(v[1] for v in acc[k].values())}
        for cname, (tot, cnt) in sorted(acc[k].items()):
            e[cname + "_mean"] = tot / max(cnt, 1)
        if "FETCH_SIZE_mean" in e or "WRITE_SIZE_mean" in e:
            base = k.split("::")[-1].split("<")[0]
            factor = 2.0 if base in WIDE_READERS else 1.0
            e["fetch_factor"] = factor
            e["fetch_rule"] = WIDE_READERS.get(base, "as counted (access shape not one of the guide's calibrated ones)")
            e["hbm_bytes_per_launch"] = (factor * e.get("FETCH_SIZE_mean", 0.0) + e.get("WRITE_SIZE_mean", 0.0)) * 1024.0
        kernels[k] = e
    out = {
        "note": "rocprofv3 --pmc passes, one counter group per pass, no tracing combined; means over every dispatch of the command. " + a.note,
        "workload": {"n_src": a.n, "n_tgt": a.n},
        "kernels": kernels,
    }
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out, "with", len(kernels), "kernels")


if __name__ == "__main__":
    main()
