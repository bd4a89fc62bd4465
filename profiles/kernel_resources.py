#!/usr/bin/env python3
"""VGPR / LDS / spill figures of every kernel in libpea_hip.so, from the code objects' metadata (no GPU needed).
The library is linked from several translation units, so its .hip_fatbin section holds one offload bundle per unit.
usage: kernel_resources.py [substring of the demangled name]"""
import os, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so, tmp):
    """paths of the gfx950 code objects unbundled from `so` (one per translation unit)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([shutil.which("objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    starts = []
    i = blob.find(MAGIC)
    while i >= 0:
        starts.append(i)
        i = blob.find(MAGIC, i + 1)
    out = []
    for n, s in enumerate(starts):
        e = starts[n + 1] if n + 1 < len(starts) else len(blob)
        part, co = os.path.join(tmp, "b%d.bin" % n), os.path.join(tmp, "k%d.co" % n)
        open(part, "wb").write(blob[s:e])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        if os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernels(so):
    """[{name, vgpr, sgpr, lds, spill, scratch}] over all code objects of `so`"""
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith("- .agpr_count:") or line.startswith("- .args:"):
                    cur = {}
                    res.append(cur)
                if cur is None or ":" not in line:
                    continue
                k, v = line.lstrip("- ").split(":", 1)
                v = v.strip()
                key = {".name": "name", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds",
                       ".vgpr_spill_count": "spill", ".private_segment_fixed_size": "scratch", ".agpr_count": "agpr"}.get(k)
                if key and key not in cur:
                    cur[key] = v if key == "name" else int(v)
    return [r for r in res if "name" in r and "vgpr" in r]


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    so = ge.load_package()._lib.SO_PATH
    ks = kernels(so)
    dem = subprocess.run(["c++filt"] + [k["name"] for k in ks], capture_output=True, text=True).stdout.splitlines() if shutil.which("c++filt") else [k["name"] for k in ks]
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%d kernels" % len(ks))
    for k, d in zip(ks, dem):
        if pat in d:
            print("vgpr %3d  spill %2d  scratch %3d  %s" % (k["vgpr"], k["spill"], k.get("scratch", 0), d[:150]))
