#!/usr/bin/env python
"""The LSTT forward pass (+ the memory update's three ID_V GEMMs) of one steady-state frame as ONE hipGraph,
replayed with nothing else on the GPU.  Under `rocprofv3 --kernel-trace` this gives every kernel's ISOLATED duration
without the Python launch cost that back-to-back eager launches (tools/kbench.py) add to the small kernels; run
plainly it prints the replay time per frame.  tools/prof_summary.py turns the trace into a table (frames are
delimited by the once-per-pass gn2_apply_kernel)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=31)
    ap.add_argument("--w", type=int, default=54)
    ap.add_argument("--cap", type=int, default=4)
    ap.add_argument("--nsplit", type=int, default=3)
    ap.add_argument("--replays", type=int, default=30)
    ap.add_argument("--no-update", action="store_true")
    args = ap.parse_args()
    from rmem_amd import hip
    from rmem_amd.config import get_config
    from rmem_amd.lstt import DeAOTLSTT
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    dev = torch.device("cuda:0")
    cfg = get_config("r50_deaotl", 1, args.cap - 1)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    L = DeAOTLSTT(model, args.h, args.w, dev, nsplit=args.nsplit)
    L.device_policy = False                    # host-side slot bookkeeping: the bank is filled by hand below
    N, T = L.N, args.cap
    g = torch.Generator(device="cpu").manual_seed(0)
    emb = torch.randn(N, 256, generator=g).to(dev)
    lab = torch.randint(0, 4, ((args.h - 1) * 16 + 1, (args.w - 1) * 16 + 1), generator=g).to(torch.uint8).to(dev)
    # reference frame + T - 1 propagated frames appended to the bank: a full bank of real activations
    L.assign_identity(lab, ignore=False)
    L.forward(emb, ref_frame=True)
    for t in range(1, T):
        L.forward(emb + 0.05 * t)
        L.assign_identity(lab)
        L.update_short_memories(True, t)
    torch.cuda.synchronize()
    assert len(L.bank) == T
    L.tgt.copy_(emb)
    L._prepare(False)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        L._forward_device(False)               # warm-up on the capture stream
        with torch.cuda.graph(graph, stream=s):
            L._forward_device(False)
            if not args.no_update:
                L._update_device(False)
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.replays):
        graph.replay()
    e1.record()
    e1.synchronize()
    print(json.dumps({"h": args.h, "w": args.w, "T": T, "ks": [L.ks_long, L.ks_win, L.ks_self],
                      "lstt_forward_plus_idv_us": round(1e3 * e0.elapsed_time(e1) / args.replays, 1),
                      "replays": args.replays}))


if __name__ == "__main__":
    main()
