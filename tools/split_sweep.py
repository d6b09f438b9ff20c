#!/usr/bin/env python
"""Paired long-term + windowed read and its combine, isolated, for a list of split geometries
("long,win,self[,nfull,pf]", RMEM_KS syntax): one process per geometry (the splits are fixed at construction)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, json, ctypes as C, torch
sys.path.insert(0, %r)
from tools.kbench import timeit
from rmem_amd import hip
from rmem_amd.config import get_config
from rmem_amd.lstt import DeAOTLSTT
from rmem_amd.model import build_vos_model
from rmem_amd.synth import load_synthetic_weights
dev = torch.device("cuda:0")
cfg = get_config("r50_deaotl", 1, 3)
model = build_vos_model("deaot", cfg).eval(); load_synthetic_weights(model); model = model.to(dev)
L = DeAOTLSTT(model, 31, 54, dev)
T = 4
g = torch.Generator().manual_seed(0)
for pl in (L.bankK[0], L.bankV[0]):
    p = hip.Planes.from_f32(torch.randn(pl.hi.shape, generator=g).to(dev)); pl.hi.copy_(p.hi); pl.lo.copy_(p.lo)
q = hip.Planes.from_f32(torch.randn(L.Npad, 128, generator=g).to(dev)); L.Qpe.hi.copy_(q.hi); L.Qpe.lo.copy_(q.lo)
L.bank, L.short, L.cur, L._T = list(range(T)), T - 1, T, T
L.maps.copy_(torch.tensor(list(range(16)) + [T - 1] + [0] * 15, dtype=torch.int32))
L.Ucat0.normal_()
lib = hip.load()
def args():
    L._layer = 0
    A = L._read_args(L.ws_main, 0, T, L.bankK[0], L.bankV[0], L.maps.data_ptr(), L.Qpe, L.bias_pe, L.Ucat0, True, L.ks_long, uneven=True)
    B = L._read_args(L.ws_side, 1, 1, L.bankK[0], L.bankV[0], L.maps.data_ptr() + 64, hip.Planes(L.bankK[0].hi[T], L.bankK[0].lo[T]), None, L.Ucat0, False, L.ks_win)
    return A, B
A, B = args()
r = timeit(lambda: hip.check(lib.rmem_attn_read2(C.byref(A[0]), C.byref(B[0]), hip.stream_ptr()), "r2"), 20)
if B[0].gout:      # the windowed read is in one split and gated its own output: only the long-term splits are merged
    c = timeit(lambda: hip.check(lib.rmem_attn_read_combine(C.byref(A[1]), hip.stream_ptr()), "c1"), 20)
else:
    c = timeit(lambda: hip.check(lib.rmem_attn_read_combine2(C.byref(A[1]), C.byref(B[1]), hip.stream_ptr()), "c2"), 20)
print(json.dumps({"ks": [A[0].ksplits, B[0].ksplits], "nfull": A[0].nfull, "pf": A[0].pf, "read2_us": round(r, 2), "combine2_us": round(c, 2), "sum_us": round(r + c, 2)}))
""" % ROOT

for ks in sys.argv[1:]:
    env = dict(os.environ, RMEM_KS=ks) if ks != "default" else dict(os.environ)
    p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(ks, line[-1] if line else "ERR " + p.stderr[-300:])
