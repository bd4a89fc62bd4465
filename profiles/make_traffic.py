#!/usr/bin/env python3
"""profiles/make_traffic.py <gpurun_out/pmcs_<tag>_g dir> <..._h dir>: FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_step.sh ->
profiles/traffic.json (HBM-side bytes per launch of the forward and backward kernels of the bench step, stamped with the hash of
the kernel sources; bench.py quotes it only when the hash matches).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha16():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pixel-embedded-affinity_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def mean_counter(d, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = mean_counter(sys.argv[1], "FETCH_SIZE"), mean_counter(sys.argv[2], "WRITE_SIZE")
out = {"_note": "HBM-side bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the bench step loop "
                "(profiles/pmc_step.sh); counters are in KB; FETCH doubled (MI355X_MICROARCH.md, HBM section)",
       "src_sha16": sha16()}
for key, pat in (("fwd_b8", "k_fwd_"), ("bwd_b8", "k_bwd_")):
    ks = [k for k in fetch if pat in k]
    if not ks:
        continue
    k = max(ks, key=lambda n: fetch[n])
    f, w = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    out[key] = {"kernel": k, "fetch_bytes": round(f), "write_bytes": round(w), "bytes_per_launch": round(f + w)}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
