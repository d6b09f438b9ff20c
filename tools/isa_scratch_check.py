#!/usr/bin/env python
"""Where the scratch (spill) accesses of the fused-read kernels sit relative to their tile loop.

Compiles rmem_amd/csrc/read64.hip to gfx950 assembly with the flags of rmem_amd/build.py, and for every read64* kernel
finds the loops (a label and a later backward branch to it), ranks them by the MFMAs inside and reports the scratch
instructions inside the hot loops (the tile loop of each instantiated mode) and outside.  Any compiler-visible
vector-memory access inside the P.V cluster drains the hand-counted V ring (s_waitcnt vmcnt(0)), so the number that
matters is "scratch ops inside the tile loops" = 0.  Runs without a GPU.
    python tools/isa_scratch_check.py > profiles/rNN_read64_scratch_check.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rmem_amd.build import EXTRA_FLAGS
    src = os.path.join(ROOT, "rmem_amd", "csrc", "read64.hip")
    out = os.path.join(tempfile.mkdtemp(), "read64.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", out] \
        + EXTRA_FLAGS.get("read64.hip", [])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()
    # kernels: from the symbol label to its .end_amdhsa_kernel / next symbol
    starts = [(i, m.group(1)) for i, l in enumerate(text) for m in [re.match(r"^(_Z\d+read64[A-Za-z0-9_]*):", l)] if m]
    print("| kernel | instructions | scratch ops | loops with MFMAs (lines: MFMAs, scratch ops inside) | scratch ops inside MFMA loops | vgpr spill |")
    print("|---|---|---|---|---|---|")
    for k, (i0, name) in enumerate(starts):
        i1 = next((j for j in range(i0 + 1, len(text)) if text[j].startswith("\t.section") or text[j].startswith(".Lfunc_end")), len(text))
        body = text[i0:i1]
        labels = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for j, l in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < j:
                a, b = labels[m.group(1)], j
                mf = sum(1 for x in body[a:b] if "v_mfma" in x)
                sc = sum(1 for x in body[a:b] if re.match(r"\s+scratch_", x))
                if mf >= 48:
                    loops.append((a, b, mf, sc))
        # innermost MFMA loops only (drop loops that contain another listed loop)
        inner = [L for L in loops if not any(o is not L and L[0] <= o[0] and o[1] <= L[1] for o in loops)]
        n_ins = sum(1 for x in body if re.match(r"\s+[vs]_|\s+ds_|\s+global_|\s+scratch_|\s+buffer_", x))
        n_sc = sum(1 for x in body if re.match(r"\s+scratch_", x))
        spill = next((x.split(":")[1].strip() for x in text[i1:i1 + 400] if "vgpr_spill_count" in x), "?")
        desc = "; ".join(f"{a}-{b}: {mf}, {sc}" for a, b, mf, sc in inner)
        print(f"| `{name}` | {n_ins} | {n_sc} | {desc} | {sum(L[3] for L in inner)} | {spill} |")


if __name__ == "__main__":
    main()
