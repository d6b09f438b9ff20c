"""Build librmem_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels
with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librmem_hip.so")
SOURCES = ["linear.hip", "read64.hip", "mha.hip", "pointwise.hip", "postproc.hip", "batch.hip"]
HEADERS = ["rmem_common.h", "gemm_core.h", "linear_stream.h", "attn_common.h", "launch.h", os.path.join("..", "..", "include", "rmem_hip.h")]


# mha.hip: VGPR form of every MFMA (no AGPRs: the 32x32 score and output tiles are read and written by
# the softmax VALU code, so the accumulator form costs a v_accvgpr move per element and s_nops in
# front of each; 128 registers instead of 128 + 32 also give 4 waves per SIMD instead of 3)
EXTRA_FLAGS = {"mha.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # read64.hip: no SLP vectorisation -- it turns the eight per-lane softmax updates into <8 x float>
               # operations whose splat operands (the row reference m, eight copies) it then spills; every reload is a
               # compiler-visible vector-memory access that drains the hand-counted V-fragment ring (s_waitcnt vmcnt(0))
               "read64.hip": ["-fno-slp-vectorize"]}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
               os.path.join(CSRC, src), "-o", obj] + EXTRA_FLAGS.get(src, []) + os.environ.get("RMEM_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
