#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

  tools/pmc_summary.py --round r01 --config cfg2 --kt <kernel-trace dir> --pmc <pmc dir> [<pmc dir> ...]

Writes profiles/<round>_kernel_stats.csv (the --stats table, our kernels only),
profiles/<round>_pmc_<config>.csv (mean counter value per kernel) and profiles/pmc_<config>.json
(what bench.py reports as roofline.traffic): HBM bytes per k_move launch from FETCH_SIZE and
WRITE_SIZE, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 - both are
in KiB; FETCH_SIZE counts 128-byte requests as 64 bytes for wide (16 B/lane) streaming reads, so
it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for k in ("k_move_dma", "k_move_t", "k_move", "k_advance", "k_reset", "k_refresh", "k_remi", "k_done", "k_cars_on_roads"):
        if k in name:
            return k
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--kt", default=None)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if a.kt:
        for f in glob.glob(os.path.join(a.kt, "**", "*_kernel_stats.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if short(r["Name"])]
            with open(os.path.join(out, "%s_kernel_stats_%s.csv" % (a.round, a.config)), "w") as g:
                w = csv.writer(g)
                w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
                for r in rows:
                    w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    agg = collections.defaultdict(list)
    for d in a.pmc:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    if agg:
        with open(os.path.join(out, "%s_pmc_%s.csv" % (a.round, a.config)), "w") as g:
            w = csv.writer(g)
            w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
            for (k, c), v in sorted(agg.items()):
                w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
        mv = next((k for k in ("k_move_t", "k_move_dma", "k_move") if (k, "FETCH_SIZE") in agg), "k_move")
        if (mv, "FETCH_SIZE") in agg and (mv, "WRITE_SIZE") in agg:
            fetch = sum(agg[(mv, "FETCH_SIZE")]) / len(agg[(mv, "FETCH_SIZE")])
            write = sum(agg[(mv, "WRITE_SIZE")]) / len(agg[(mv, "WRITE_SIZE")])
            js = {"kernel": mv, "config": a.config, "round": a.round,
                  "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
                  "read_bytes": fetch * 1024 * 2, "write_bytes": write * 1024,
                  "k_move_hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
                  "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests of wide streaming reads as 64 B); "
                                "WRITE_SIZE as reported", "note": a.note}
            json.dump(js, open(os.path.join(out, "pmc_%s.json" % a.config), "w"), indent=1)
            print(json.dumps(js))


if __name__ == "__main__":
    main()
