#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: per-kernel time stats and PMC byte counters.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units (x1024 -> bytes); on gfx950 FETCH_SIZE
counts 128-B requests at 64 B, so the read side is DOUBLED here as MI355X_MICROARCH.md section HBM prescribes
for wide coalesced streams (4-B-per-lane row reads coalesce into the same 128-B requests)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


st = find("stats", "*kernel_stats.csv")
if st:
    print("== kernel stats (%s)" % os.path.relpath(st, out))
    for r in csv.DictReader(open(st)):
        if "pea" in r["Name"] or "k_" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f ns  min %10s  max %10s  %5s%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]),
                                                                             r["MinNs"], r["MaxNs"], r["Percentage"]))
for sub, scale in (("pmc_fetch", 2.0), ("pmc_write", 1.0), ("pmc_l2", None)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "k_" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s" % sub)
    for k, cs in acc.items():
        for c, v in cs.items():
            mean = sum(v) / len(v)
            if scale is None:
                print("%-60s %-14s mean %.4g over %d dispatches" % (k, c, mean, len(v)))
            else:
                print("%-60s %-11s raw mean %.1f -> %.1f MB per launch (x1024 x%.0f)" % (k, c, mean, mean * 1024 * scale / 1e6, scale))
