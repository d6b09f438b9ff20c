"""CPU study (no GPU): which operand formats of the attention contractions keep the 481x849
golden clip's integer label maps.  The oracle's attention functions are wrapped so that the
probabilities P and/or the values V (and optionally Q/K of the score GEMM) are rounded the way a
given MFMA operand plan would carry them; the clip is run teacher-forced like
tests/test_oracle_golden.py::test_480p_clip and mismatching pixels per frame are printed.

  python tests/probes/precision_study.py p16_v16x2 p16_v16 pbf_vbf p16@long p16_v16@self,win ...
  (@long / @self / @win restricts a plan to the long-term, self or windowed short-term read)

plan tokens:  p16 = P as one fp16 plane        pbf = P as one bf16 plane     pbfx2 = bf16 hi+lo
              v16 = V as one fp16 plane        v16x2 = V as fp16 hi+lo       vbf / vbfx2 likewise
              qk16x2 / qk16 / qkbf = both score operands
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import lstt_ref as R                      # noqa: E402
from oracle.engine_ref import OracleDeAOTEngine       # noqa: E402
from rmem_amd.config import get_config                # noqa: E402
from rmem_amd.model import build_vos_model            # noqa: E402
from rmem_amd.synth import load_synthetic_weights, synth_clip   # noqa: E402


def planes(x, dt, n):
    hi = x.to(dt).float()
    if n == 1:
        return hi
    return hi + (x - hi).to(dt).float()


def make_q(tok):
    if tok is None:
        return lambda x: x
    dt = torch.float16 if "16" in tok else torch.bfloat16
    n = 2 if tok.endswith("x2") else 1
    return lambda x: planes(x, dt, n)


def install(plan):
    """plan = seg[+seg...], seg = tokens[@reads]: e.g. p16_v16x2@long,self+p16x2_v16x2@win"""
    ident = lambda x: x
    table = {r: (ident, ident, ident, ident) for r in ("long", "self", "win")}
    for seg in plan.split("+"):
        where = "long,self,win"
        if "@" in seg:
            seg, where = seg.split("@")
        toks = seg.split("_")
        q3 = (make_q(next((t for t in toks if t.startswith("p")), None)),
              make_q(next((t for t in toks if t.startswith("v")), None)),
              make_q(next((t for t in toks if t.startswith("qk")), None)),
              make_q(next((t for t in toks if t.startswith("qonly")), None)))
        for r in where.split(","):
            table[r] = q3

    def core(Q, K, V, U, h, w, dw_w, proj_w, proj_b, d_att=128):
        qp, qv, qqk, qq = table["self" if Q is K else "long"]
        logits = qq(qqk(Q / (d_att ** 0.5))) @ qqk(K).t()
        m = logits.max(dim=-1, keepdim=True).values
        p = qp(torch.exp(logits - m))                 # the kernel stores exp(S - max), sums the stored values
        attn = p / p.sum(dim=-1, keepdim=True)
        out = (p @ qv(V)) / p.sum(dim=-1, keepdim=True) * U
        out = R.dwconv5x5(out, dw_w, h, w)
        out = R.linear(out, proj_w, proj_b)
        return out, attn, logits

    def local(q, k, v, u, h, w, rel_w, rel_b, dw_w, proj_w, proj_b, max_dis=7):
        n, d = q.shape
        qp, qv, qqk, qq = table["win"]
        idx, inside = R.local_window_index(h, w, max_dis)
        rel = q @ rel_w.view(rel_w.shape[0], d).t() + rel_b
        qs = qq(qqk(q / (d ** 0.5)))
        kg = qqk(k)[idx.clamp(min=0)] * inside.unsqueeze(-1)
        qk = torch.einsum("nc,noc->no", qs, kg) + rel
        qk = qk - (~inside).float() * 1e8
        m = qk.max(dim=1, keepdim=True).values
        p = qp(torch.exp(qk - m))
        s = p.sum(dim=1, keepdim=True)
        vg = qv(v)[idx.clamp(min=0)] * inside.unsqueeze(-1)
        agg = torch.einsum("no,noc->nc", p, vg) / s
        out = R.dwconv5x5(agg * u, dw_w, h, w)
        return R.linear(out, proj_w, proj_b), p / s

    R.gated_propagation_core = core
    R.local_gated_propagation = local


def install_conv(which):
    """Emulate split-bf16 (hi+lo planes, 16 significant bits per operand) in the encoder/decoder
    convolutions: which = 'c1' (1x1 only) or 'call' (every conv)."""
    orig = F.conv2d

    def conv2d(x, w, *a, **k):
        if which == "call" or tuple(w.shape[2:]) == (1, 1):
            x, w = planes(x, torch.bfloat16, 2), planes(w, torch.bfloat16, 2)
        return orig(x, w, *a, **k)

    torch.nn.functional.conv2d = conv2d


def install_linear(which):
    """Activations of the LSTT linears as ONE fp16 plane (weights exact): 'lin16' every linear,
    'lin16bf' activations as bf16 hi/lo (the shipped scheme: reference point)."""
    orig = R.linear

    def linear(x, w, b):
        if which == "lin16":
            x = planes(x, torch.float16, 1)
        elif which == "lin16bf":
            x, w = planes(x, torch.bfloat16, 2), planes(w, torch.bfloat16, 2)
        elif which == "lin16x2":
            x, w = planes(x, torch.float16, 2), planes(w, torch.float16, 2)
        elif which == "lin16x2a":       # activations fp16 hi/lo, weights bf16 hi/lo... (types must match on the MFMA: reference only)
            x, w = planes(x, torch.float16, 2), planes(w, torch.bfloat16, 2)
        elif which == "lin16w":
            x, w = planes(x, torch.float16, 1), planes(w, torch.bfloat16, 2)
        return orig(x, w, b)

    R.linear = linear


def run(plan):
    gd = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gd, "clip_480p.json")))
    gold = np.load(os.path.join(gd, "clip_480p.npz"))
    if plan in ("c1", "call"):
        install_conv(plan)
    elif plan.startswith("lin16"):
        install_linear(plan)
    elif plan != "fp32":
        install(plan)
    torch.manual_seed(0)
    model = build_vos_model("deaot", get_config("r50_deaotl", meta["former"], meta["latter"])).eval()
    load_synthetic_weights(model)
    eng = OracleDeAOTEngine(model, long_term_mem_gap=meta["gap"])
    imgs, lab = synth_clip(meta["seed"], meta["frames"], meta["H"], meta["W"], 3)
    out_hw = tuple(meta["out_hw"])
    eng.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
    mism, lerr = [], {}
    for t in range(1, meta["frames"]):
        logit = eng.match_propogate_one_frame(imgs[t], output_size=out_hw)
        pred = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True)
        mism.append(int((pred[0, 0].numpy().astype(np.uint8) != gold["labels"][t - 1]).sum()))
        if f"logits_{t}" in gold:
            lerr[t] = float(np.abs(eng.pred_id_logits.numpy() - gold[f"logits_{t}"].astype(np.float32)).max())
        fed = torch.from_numpy(gold["labels"][t - 1]).float()[None, None]
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
    ok = [list(i) for i in meta["indexes"]][-1] == list(eng.long_memories_indexes)
    print(f"{plan:24s} mismatching px/frame {mism}  sum {sum(mism)}  logit err {lerr}  evictions_ok {ok}", flush=True)


if __name__ == "__main__":
    for plan in sys.argv[1:]:
        run(plan)
