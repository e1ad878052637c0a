#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>/ directory (written by profiles/run_profile.sh on the GPU box) into the
committed summaries: profiles/<round>/kernel_stats_<suffix>.csv, pmc_summary_<suffix>.csv and, with
--traffic, profiles/traffic.json (HBM bytes per launch per kernel, read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB from separate --pmc passes; on gfx950
FETCH_SIZE reports half the bytes of the reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor
was checked on the filter kernel, whose reads are known exactly (60 B per 150-bp read, each read once)."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("suffix")
    ap.add_argument("--round", default="r03")
    ap.add_argument("--traffic", action="store_true")
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    a = ap.parse_args()
    base = os.path.join(ROOT, "gpurun_out", "prof_" + a.tag)
    out = os.path.join(ROOT, "profiles", a.round)
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(out, "kernel_stats_%s.csv" % a.suffix))
    ks = glob.glob(os.path.join(base, "trace_serial", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(out, "kernel_stats_%s_serial.csv" % a.suffix))
    summary = {}
    with open(os.path.join(out, "pmc_summary_%s.csv" % a.suffix), "w") as f:
        f.write("kernel,counter,dispatches,avg_per_dispatch\n")
        for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
            fs = glob.glob(os.path.join(base, d, "*", "*_counter_collection.csv"))
            if not fs:
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(fs[0])):
                agg[(r["Kernel_Name"].split("(")[0].replace(",", ";"), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(agg.items()):
                if "trew" in k:
                    f.write("%s,%s,%d,%.3f\n" % (k, c, len(v), sum(v) / len(v)))
                    summary.setdefault(k, {})[c] = sum(v) / len(v)
    if a.traffic:
        traffic = {"config": {"reads_per_gpu": a.reads, "read_len": a.read_len}, "source": "profiles/%s/pmc_summary_%s.csv" % (a.round, a.suffix), "kernels": {}}
        for k, c in summary.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                name = "filter_kernel" if "filter_kernel" in k else "exact_kernel" if "exact_kernel" in k else None
                if name:
                    traffic["kernels"][name] = {
                        "fetch_kib": c["FETCH_SIZE"], "write_kib": c["WRITE_SIZE"],
                        "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
                    }
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
