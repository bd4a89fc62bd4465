#!/usr/bin/env python3
"""profiles/make_traffic.py <key> <FETCH_SIZE pass dir> <WRITE_SIZE pass dir>: record the HBM-side bytes per launch of the forward
and backward kernels of one bench configuration in profiles/traffic.json (key = "c2" for the headline step loop of
profiles/pmc_step.sh, "c3" / "c4" / "c4n26" / "c5" / "c5f32" for profiles/pmc_cfg.sh), stamped with the hash of the kernel
sources: bench.py quotes an entry only when the hash matches the code it runs.  FETCH_SIZE is doubled as MI355X_MICROARCH.md
(HBM section) prescribes; both counters are in KiB."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("PEA_TRAFFIC_OUT") or os.path.join(ROOT, "profiles", "traffic.json")  # (round 6: the passes write a scratch file; it replaces the record only when all of them succeeded)


def sha16():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pixel-embedded-affinity_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def mean_counter(d, name):
    acc, dur = defaultdict(list), defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k].append(float(r["Counter_Value"]))
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: sum(v) / len(v) for k, v in dur.items()}


key = sys.argv[1]
(fetch, fdur), (write, _) = mean_counter(sys.argv[2], "FETCH_SIZE"), mean_counter(sys.argv[3], "WRITE_SIZE")
if not fetch or not write:
    sys.exit("make_traffic.py %s: no FETCH_SIZE / WRITE_SIZE rows under %s / %s (a pass failed)" % (key, sys.argv[2], sys.argv[3]))
doc = json.load(open(PATH)) if os.path.exists(PATH) else {}
sha = sha16()
if doc.get("src_sha16") != sha:  # entries of other sources are void
    doc = {}
doc["_note"] = ("HBM-side bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/pmc_step.sh for c2, "
                "profiles/pmc_cfg.sh for the other configs); counters are in KiB; FETCH doubled (MI355X_MICROARCH.md, HBM section)")
doc["src_sha16"] = sha
ent = {}
for which, pat in (("fwd", "k_fwd_"), ("bwd", "k_bwd_")):
    ks = [k for k in fetch if pat in k]
    if not ks:
        continue
    k = max(ks, key=lambda n: fetch[n])
    f, w = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    ent[which] = {"kernel": k, "fetch_bytes": round(f), "write_bytes": round(w), "bytes_per_launch": round(f + w),
                  "us_under_pmc": round(fdur[k], 1)}
doc[key] = ent
json.dump(doc, open(PATH, "w"), indent=1)
print(json.dumps({key: ent}, indent=1))
