#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

  tools/pmc_summary.py --round r02 --config cfg2 --kt <kernel-trace dir> --pmc <pmc dir> [<pmc dir> ...]

Writes profiles/<round>_kernel_stats_<config>.csv (the --stats table, our kernels only),
profiles/<round>_pmc_<config>.csv (mean counter value per kernel) and profiles/pmc_<config>.json
(what bench.py reports as roofline.traffic): HBM bytes per launch of the kernel that moves the cars
(--kernel, default: the one of ours with the largest total time) from FETCH_SIZE and WRITE_SIZE,
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 - both are in KiB;
FETCH_SIZE counts 128-byte requests as 64 bytes for wide streaming reads, so it is doubled;
WRITE_SIZE is exact for streaming stores.  The json carries `kernel_hash`: the hash of the measured kernel's MACHINE
CODE in the library that ran under the profiler, recorded on the box next to the counters (kernel_hashes.json in the
profile directory, written by tools/profile_round.sh with tools/kernel_hash.py).  This script only copies it - it
refuses to write a summary for counters that came without one - so a summary cannot be re-stamped for other code;
bench.py reports the traffic only while the library it drives has the same hash for that kernel.
"""
import argparse
import collections
import csv
import glob
import json
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOVERS = ("k_res", "k_move_dma", "k_move_tts", "k_move_ts", "k_move_tt", "k_move_tt1", "k_move_t", "k_move")


def _code_only(text):
    """C++ source without comments and blank lines (string literals kept intact)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return "\n".join(l.rstrip() for l in "".join(out).split("\n") if l.strip())


def csrc_hash():
    """sha256 over the kernel sources and the ABI header (sorted by name) with comments and blank lines removed - an
    edited comment does not unpin the profiles, an edited statement does; first 16 hex digits."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "traffic-env_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hpp", ".hip", ".cpp")))
    for f in files + [os.path.join(ROOT, "include", "tfx.h")]:
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def short(name):
    if "k_move_tt<false" in name:
        return "k_move_tt1"          # the one-tick form that ends a call of two-tick passes
    for k in MOVERS + ("k_tail", "k_edge", "k_advance", "k_risk", "k_tick_add", "k_reset", "k_refresh", "k_remi", "k_done", "k_cars_on_roads",
                       "k_poisson", "k_greedy", "k_agent_obs"):
        if k + "<" in name or k + "(" in name or name.strip().endswith(k):
            return k
    for k in MOVERS:
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
    ap.add_argument("--kernel", default=None, help="the kernel whose traffic goes into pmc_<config>.json")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="ticks one launch of that kernel covers (k_res: the n of tfx_step(n); k_move_tt: 2)")
    ap.add_argument("--read-bytes-expected", type=float, default=0.0,
                    help="bytes the kernel is known to read per launch (for the record of the FETCH_SIZE factor)")
    ap.add_argument("--hash", action="store_true", help="print the current source hash and exit (informational)")
    ap.add_argument("--out", default=None, help="directory the summaries go to (default: profiles/)")
    ap.add_argument("--hashes", default=None,
                    help="kernel_hashes.json recorded on the box by the profile run (default: next to the --pmc directories)")
    a = ap.parse_args()
    if a.hash:
        print(csrc_hash())
        return
    hashes = None
    if a.pmc:  # counters travel with the hash of the code they measured, or not at all
        hpath = a.hashes or os.path.join(os.path.dirname(os.path.abspath(a.pmc[0].rstrip("/"))), "kernel_hashes.json")
        if not os.path.exists(hpath):
            raise SystemExit("pmc_summary: %s is missing - these counters were not taken together with the hash of the "
                             "code they measured (tools/profile_round.sh writes it on the box); no summary written" % hpath)
        hashes = json.load(open(hpath))
    out = a.out or os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    total_ns = collections.Counter()
    if a.kt:
        for f in glob.glob(os.path.join(a.kt, "**", "*_kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if short(r["Name"]) in MOVERS:
                    total_ns[short(r["Name"])] += int(float(r["TotalDurationNs"]))
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
        have = [k for k in MOVERS if (k, "FETCH_SIZE") in agg and (k, "WRITE_SIZE") in agg]
        mv = a.kernel or (max(have, key=lambda k: total_ns.get(k, 0)) if have else None)
        if mv and (mv, "FETCH_SIZE") in agg and (mv, "WRITE_SIZE") in agg:
            def med(v):    # (the median: a profile run's first launches differ - cold caches, k_res's shorter warm-up call)
                v = sorted(v)
                return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])
            fetch, write = med(agg[(mv, "FETCH_SIZE")]), med(agg[(mv, "WRITE_SIZE")])
            khash = hashes.get(mv)
            if not khash:
                raise SystemExit("pmc_summary: no hash for %s was recorded with these counters" % mv)
            js = {"kernel": mv, "config": a.config, "round": a.round, "kernel_hash": khash,
                  "ticks_per_launch": a.ticks_per_launch,
                  "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
                  "read_bytes": fetch * 1024 * 2, "write_bytes": write * 1024,
                  "k_move_hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
                  "hbm_bytes_per_tick": (fetch * 1024 * 2 + write * 1024) / a.ticks_per_launch,
                  "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests of wide streaming reads as 64 B); "
                                "WRITE_SIZE as reported",
                  # why x2 holds for THIS kernel's 8-byte-per-lane row loads and not only for the guide's 16-byte
                  # ones: the launch reads every live car once (16 N_live... 8 B each) and the raw counter comes out
                  # at half of that
                  "correction_basis": (("the kernel's read side is known exactly - 8 B per live car + the per-road words "
                                        "- and FETCH_SIZE x 1024 comes out at %.3f of it; " % (fetch * 1024 / a.read_bytes_expected))
                                       if a.read_bytes_expected else "") +
                                      "a stream of KNOWN size (tools/fetch_factor.sh over tools/copy_probe.hip, 8 and 16 bytes per "
                                      "lane) reads 0.500 x its bytes on FETCH_SIZE and 1.000 x on WRITE_SIZE "
                                      "(profiles/r03_fetch_size_factor.txt)",
                  "note": a.note}
            if (mv, "SQ_INSTS_VALU") in agg:
                v = med(agg[(mv, "SQ_INSTS_VALU")])
                js["valu_insts_per_launch"] = v
                js["valu_insts_per_tick"] = v / a.ticks_per_launch
            json.dump(js, open(os.path.join(out, "pmc_%s.json" % a.config), "w"), indent=1)
            print(json.dumps(js))


if __name__ == "__main__":
    main()
