#!/usr/bin/env python
"""Headline benchmark: frames/sec/GPU of the RMem hot path (BASELINE.json metric).

Workload (config.workload): R50-DeAOTL + RMem, 480p (481x849 -> 31x54 = 1674 tokens),
K=4 memory slots (FORMER_MEM_LEN=1, LATTER_MEM_LEN=3), one synthetic clip per GPU,
random-init (name-keyed synthetic) weights, fp32 I/O.  A "step" is one frame through the
reference's timing window (managers/evaluator.py:399-404,525-527):
match_propogate_one_frame -> label map (bilinear upsample + argmax; the evaluator's
softmax/argmax torch ops with --reference-postproc) -> nearest resize -> update_memory, with
the bank in steady state (T = K, one long-memory update + eviction every `gap` frames).
The next frames (engine.lookahead of them) are announced to the engine as `next_img` (encoder prefetch,
as rmem_amd.driver does).
Frames are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus N ...          # spawns N ranks itself (one per GPU), like the reference's
                                          # tools/eval.py:137-143 (mp.spawn over --gpu_num)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # same ranks, external launcher

Environment: RMEM_DIST_BACKEND (default "nccl" = RCCL; "gloo" lets N ranks share ONE GPU for a launcher
smoke test), RMEM_DEVICE_OVERRIDE=<index> (every rank uses that device instead of LOCAL_RANK), RMEM_FORCE_DIST=1
(world 1 still initialises the process group and goes through every collective of the N > 1 path).

Prints ONE JSON line on rank 0 (see the driver contract), with `roofline` for the
dominant kernel (HIP-event timed inside the timed region) and `cpu_baseline` (the
oracle's CPU restatement timed on a bounded sample of the same workload, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H_IN, W_IN = 481, 849          # 480x854 after MultiRestrictSize (dataloaders/video_transforms.py:604-622)
H_OUT, W_OUT = 480, 854
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 / fp16, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gap", type=int, default=5, help="long_term_mem_gap (evaluator rule gives 5 for clips <= 165 frames)")
    ap.add_argument("--nsplit", type=int, default=int(os.environ.get("RMEM_NSPLIT", "3")),
                    help="3 = split-fp16 (hi/lo planes, fp32-class), 1 = plain fp16 attention/linears")
    ap.add_argument("--config", choices=["480p_k4", "720p_k8", "clips64"], default="480p_k4",
                    help="480p_k4 = BASELINE.json configs[1] (the headline metric); 720p_k8 = configs[2] (stress); "
                         "clips64 = configs[3]: 8 clips per rank x 16 frames through the clip driver (reference frame "
                         "and bank fill inside the timed window), static shard + all-gather of masks, sha256 per clip")
    ap.add_argument("--clips-per-rank", type=int, default=8, help="clips64 mode: clips per rank (64 clips at 8 GPUs)")
    ap.add_argument("--clip-frames", type=int, default=16, help="clips64 mode: frames per clip")
    ap.add_argument("--ragged", action="store_true",
                    help="clips64 mode: clips of DIFFERENT lengths (uniform in [clip_frames/2, 2*clip_frames], 3 clips per "
                         "slot): longest-first assignment to ranks, and with --batched each rank's clips through the slot "
                         "queue of its BatchedClipDriver (a slot takes the next clip when its clip ends)")
    ap.add_argument("--model", choices=["r50_deaotl", "r50_aotl", "swinb_aotl"], default="r50_deaotl",
                    help="r50_deaotl = headline metric; r50_aotl = AOT block (BASELINE.json configs[0] on GPU)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="independent clips in flight per GPU, one engine + one HIP stream each (SURVEY.md 8f "
                         "rank 2; BASELINE.json configs[1] is 1, configs[3] runs 8 clips per rank)")
    ap.add_argument("--batched", action="store_true",
                    help="--clips-per-gpu B clips in lockstep through rmem_amd.batched: ONE launch per kernel for all "
                         "clips (encoder / decoder at batch B), instead of one engine + HIP stream per clip")
    ap.add_argument("--reference-postproc", action="store_true",
                    help="softmax/argmax/nearest-resize with the evaluator's torch ops on full-size logits "
                         "(managers/evaluator.py:424-441,518-523) instead of the driver's fused label kernels")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not hand the next frame to match_propogate_one_frame (its encoder pass then runs "
                         "in line instead of on a second stream beside this frame's LSTT/decoder)")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="frames announced ahead to the engine for encoder prefetch (0 = what the engine asks for: "
                         "3 with its default encoder batch of 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the second timed leg (drop-in flags)")
    ap.add_argument("--cpu-frames", type=int, default=4)
    return ap.parse_args()


# HBM traffic of the dominant kernel: from separate rocprofv3 --pmc passes of the same bench command
# (research/jobs/gpujob_profile_r03.sh: FETCH_SIZE x2 -- the gfx950 correction of MI355X_MICROARCH.md -- + WRITE_SIZE, mean per
# launch), committed under profiles/.  (model, config, clips) -> (file, kernel-name prefix in that file)
PMC_FILES = {("r50_deaotl", "480p_k4", "one"): ("r06_pmc_x3.json", "read64x2_kernel"),
             ("r50_deaotl", "720p_k8", "one"): ("r06_pmc_720p_k8.json", "read64x2_pull_kernel"),
             ("r50_deaotl", "480p_k4", "batched8"): ("r04m_pmc_batched8.json", "read64x2_many_pull_kernel"),
             ("r50_aotl", "480p_k4", "one"): ("r04m_pmc_aot.json", "mha_flash_kernel")}


def _pmc_path(name):
    """profiles/<name>, or the previous round's file of the same configuration while this round's is not collected yet."""
    if not name:
        return None, None
    for n in (name, name.replace("r06_", "r05_"), name.replace("r06_", "r04m_"), name.replace("r05_", "r04m_")):
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return p, n
    return None, None


def pmc_traffic(roofline: dict, key) -> None:
    name, kernel = PMC_FILES.get(key, (None, None))
    path, name = _pmc_path(name)
    if not path or not os.path.exists(path):
        return
    for k, v in json.load(open(path)).items():
        if k.startswith(kernel) and isinstance(v, dict) and "hbm_bytes_per_launch" in v:
            roofline["traffic"] = v["hbm_bytes_per_launch"]
            roofline["traffic_unit"] = f"bytes/launch (rocprofv3 PMC, profiles/{name})"
            return


def pmc_kernel_traffic(entry: dict, key) -> None:
    """HBM-side bytes per launch of a kernel class from the committed PMC file of this configuration (separate rocprofv3
    --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) and its ratio to the class's algorithmic bytes."""
    name, _ = PMC_FILES.get(key, (None, None))
    path, name = _pmc_path(name)
    if not path or not os.path.exists(path):
        return
    want = entry["kernel"].split(" ")[0].split("_kernel")[0]
    alias = {"read_combine2": "_ZN4rmem5k_oneI12Combine2Args", "read_combine": "_ZN4rmem5k_oneI22rmem_read_combine_args",
             "layernorm_red2": "_ZN4rmem5k_oneI10LnRed2Args"}
    for k, v in json.load(open(path)).items():
        if isinstance(v, dict) and "hbm_bytes_per_launch" in v and (k.startswith(want + "_kernel") or k.startswith(alias.get(want, "\0"))):
            entry["traffic"] = v["hbm_bytes_per_launch"]
            entry["traffic_source"] = f"profiles/{name}"
            # against the ALGORITHMIC bytes of SURVEY 8d (operands once + final outputs; no split partials): a class whose
            # algorithmic bytes are zero (the combine) moves nothing but overhead -- its ratio is reported as null
            if entry.get("algorithmic_mb_per_launch"):
                entry["traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / (entry["algorithmic_mb_per_launch"] * 1e6)
            else:
                entry["traffic_over_algorithmic"] = None
            return


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this script as N ranks of one node through
    torch.distributed.run (one process per GPU, LOCAL_RANK -> device), the role mp.spawn(main_worker,
    nprocs=cfg.TEST_GPU_NUM) plays in the reference (aot_plus/tools/eval.py:137-143).  Rank 0 prints the
    JSON line; the children's output passes through."""
    import socket
    import subprocess
    if "RMEM_DEVICE_OVERRIDE" not in os.environ and torch.cuda.is_available() and torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: only {torch.cuda.device_count()} GPU(s) visible "
                         "(RMEM_DEVICE_OVERRIDE=0 RMEM_DIST_BACKEND=gloo runs the ranks on one device)")
    with socket.socket() as sk:                      # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rank_device(local_rank: int) -> int:
    """Device index of this rank: LOCAL_RANK, or RMEM_DEVICE_OVERRIDE for every rank."""
    return int(os.environ["RMEM_DEVICE_OVERRIDE"]) if os.environ.get("RMEM_DEVICE_OVERRIDE", "") != "" else local_rank


def keep_stdout_for_the_json_line():
    """Native libraries print to the C-level stdout (librccl: "Librccl path : ..." when its buffer is flushed at exit,
    i.e. AFTER rank 0's JSON line; MIOpen warnings).  The driver reads ONE JSON line from stdout, so file descriptor 1 is
    pointed at stderr for everything native and Python's sys.stdout keeps the real stdout."""
    if getattr(keep_stdout_for_the_json_line, "done", False):
        return
    keep_stdout_for_the_json_line.done = True
    try:
        sys.stdout.flush()
        real = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real, "w", buffering=1)
    except OSError:
        pass


def init_dist(world: int):
    """None for one rank, else the initialised torch.distributed module (RCCL unless RMEM_DIST_BACKEND says gloo)."""
    if world <= 1 and os.environ.get("RMEM_FORCE_DIST") != "1":
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world <= 1:             # RMEM_FORCE_DIST=1: a one-rank group, so that one leased GPU runs every RCCL call of the N > 1 path
        os.environ.setdefault("MASTER_PORT", "29547")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend=os.environ.get("RMEM_DIST_BACKEND", "nccl"))
    return dist


def first_convolutions_in_turn(dist, fn):
    """Runs `fn` -- a warm-up that takes every convolution of the model through MIOpen for the first time -- on the rank
    with LOCAL_RANK 0 first and on the node's other ranks after it.  Ranks that start together on a cold MIOpen cache
    (kernel cache, user find-db) race for it, and a rank that loses a race can end up on another solver than the rank that
    filled the entry -- label hashes then differ between an 8-rank and a 1-rank run of the same clips
    (profiles/r05g_world_hash_probe.txt: 6 of 8 clips on the first run of a fresh box, none afterwards).  One barrier."""
    if dist is None or dist.get_world_size() == 1:
        return fn()
    first = int(os.environ.get("LOCAL_RANK", "0")) == 0
    if not first:
        dist.barrier()
    out = fn()
    torch.cuda.synchronize()
    if first:
        dist.barrier()
    return out


def gather_rank_vectors(dist, vec, dev):
    """Every rank's list of floats -> [world][len] (one all-gather; the values themselves are host-side statistics)."""
    if dist is None:
        return [list(map(float, vec))]
    on = dev if dist.get_backend() != "gloo" else torch.device("cpu")
    mine = torch.tensor(list(map(float, vec)), dtype=torch.float64, device=on)
    every = torch.zeros(dist.get_world_size() * len(vec), dtype=torch.float64, device=on)
    dist.all_gather_into_tensor(every, mine)
    return [[float(v) for v in row] for row in every.cpu().view(dist.get_world_size(), len(vec))]


def max_over_ranks(dist, elapsed: float, dev):
    """(max over ranks, list of every rank's value): the job takes as long as its slowest rank."""
    if dist is None:
        return elapsed, [elapsed]
    on = dev if dist.get_backend() != "gloo" else torch.device("cpu")
    mine = torch.tensor([elapsed], dtype=torch.float64, device=on)
    every = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=on)
    dist.all_gather_into_tensor(every, mine)
    vals = [float(v) for v in every.cpu()]
    return max(vals), vals


def main():
    args = parse()
    if not (args.gpus > 1 and "WORLD_SIZE" not in os.environ):   # (the spawning parent passes its children's output through)
        keep_stdout_for_the_json_line()
    if not args.batched:       # reproducible convolutions (rmem_amd/determinism.py): free for one clip per engine, so the
        from rmem_amd.determinism import reproducible_convolutions      # clip hashes of --config clips64 repeat run to run;
        reproducible_convolutions()                                     # MIOpen at batch B needs the solvers this disables
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # no launcher around us: become one
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:               # the launcher's world is what runs and what is reported
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus = {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # host placement: this rank's threads on its share of the NUMA node its GPU hangs off (rmem_amd/affinity.py), before
    # any thread pool exists; external launchers (torch.distributed.run sets OMP_NUM_THREADS=1 when it is unset) included
    from rmem_amd.affinity import pin_rank
    from rmem_amd import hip as _hip
    # (before anything initialises the HIP runtime -- pin_rank asks it for the GPU's PCI address)
    _hip.set_host_wait(rank_device(local_rank))       # RMEM_BLOCKING_WAIT=1 only (opt-in): see rmem_amd.hip.set_host_wait
    PIN = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), rank_device)
    args._pin = PIN
    local_rank = rank_device(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = init_dist(world)

    from rmem_amd import hip
    from rmem_amd.config import get_config
    from rmem_amd.engine import build_engine
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip

    global H_IN, W_IN, H_OUT, W_OUT
    if args.config == "clips64":
        return clips64(args, world, rank, local_rank, dev, dist)
    mem_k = 4
    if args.config == "720p_k8":      # 720x1280 -> 721x1281 -> 46x81 tokens, K=8, 3 objects
        H_IN, W_IN, H_OUT, W_OUT, mem_k = 721, 1281, 720, 1280, 8
    if args.model == "swinb_aotl" and args.config == "480p_k4":   # align_corners=False: 480x848 -> 30x53
        H_IN, W_IN, H_OUT, W_OUT = 480, 848, 480, 854
    cfg = get_config(args.model, 1, mem_k - 1)
    cpu_model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(cpu_model)
    model = copy.deepcopy(cpu_model).to(dev)
    C = max(1, args.clips_per_gpu)
    if args.batched:
        return batched_steady(args, world, rank, dev, dist, cfg, model, mem_k)
    engines = []
    for _ in range(C):
        e = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=model, gpu_id=local_rank,
                         long_term_mem_gap=args.gap, nsplit=args.nsplit)
        e.eval()
        engines.append(e)
    from rmem_amd.streams import concurrent_stream
    streams = [torch.cuda.current_stream(dev)] + [concurrent_stream(dev) for _ in range(C - 1)]

    # independent clips (seed = rank * C + i); a ring of 8 distinct frames per clip in HBM
    ring = 8
    clips = []
    for i in range(C):
        im, lb = synth_clip(rank * C + i, ring, H_IN, W_IN, 3)
        clips.append(([x.to(dev) for x in im], lb.to(dev)))

    PREFETCH = not args.no_prefetch
    LOOKAHEAD = args.lookahead if args.lookahead > 0 else engines[0].lookahead

    def frame_step(i, t, masks_out=None):
        engine = engines[i]
        # the next frames are announced to the engine (encoder prefetch, as rmem_amd.driver does)
        nxt = [clips[i][0][(t + d) % ring] for d in range(1, LOOKAHEAD + 1)] if PREFETCH else None
        if args.reference_postproc:
            logit = engine.match_propogate_one_frame(clips[i][0][t % ring], output_size=(H_OUT, W_OUT), next_img=nxt)
            prob = torch.softmax(logit, dim=1)
            pred = torch.argmax(prob, dim=1, keepdim=True).float()
            cur = F.interpolate(pred, size=engine.input_size_2d, mode="nearest")
            engine.update_memory(cur)
            if masks_out is not None:
                masks_out[i, t % masks_out.shape[1]] = pred[0, 0].to(torch.uint8)
            return
        # the clip driver's path (rmem_amd/driver.py): decoder logits -> uint8 label map at the
        # original size (bilinear upsample + argmax in one kernel, written straight into the
        # clip's mask tensor) -> nearest resize to the input size -> update_memory
        logit = engine.match_propogate_one_frame(clips[i][0][t % ring], output_size=None, next_img=nxt)
        lab = masks_out[i, t % masks_out.shape[1]] if masks_out is not None else None
        lab = hip.labels_from_logits([logit], [False], (H_OUT, W_OUT), cfg.MODEL_ALIGN_CORNERS, out=lab)
        buf = engine.aot_engines[0].label_buffer(engine.input_size_2d, lab.device) if len(engine.aot_engines) == 1 else None
        engine.update_memory(hip.label_resize_nearest(lab, engine.input_size_2d, out=buf)[None, None])

    def all_clips(t, masks_out=None):
        for i in range(C):
            with torch.cuda.stream(streams[i]):
                frame_step(i, t, masks_out)

    # ---- setup: reference frame + pre-roll until the bank holds K slots (steady state)
    for i, e in enumerate(engines):
        e.restart_engine()
        e.add_reference_frame(clips[i][0][0], clips[i][1], obj_nums=[3], frame_step=0)
    t = 1
    sub = engines[0].aot_engines[0]
    while len(sub.lstt.bank) < cfg.mem_cap:
        all_clips(t)
        t += 1
    # ... and until every hipGraph of the steady state exists (one per feature copy and free slot:
    # captures take tens of ms and must not land in the timed region, whatever --warmup is)
    def n_graphs():
        return sum(len(getattr(e.aot_engines[0], k)) for e in engines for k in ("_fg", "_ug", "_eg"))
    stable, last = 0, n_graphs()
    for _ in range(60):
        if stable >= 2 * cfg.mem_cap + 4:
            break
        all_clips(t)
        t += 1
        cur = n_graphs()
        stable, last = (stable + 1, last) if cur == last else (0, cur)
    # (with the opt-in early long-term read the front graph already holds layer 0's read: the sampled `tail` replay would
    # issue it a second time and time a schedule that is not the one that runs -- no sampling then)
    sampled = (hasattr(sub.lstt, "launch_read2_layer0") and sub.hoist_enabled and os.environ.get("RMEM_BENCH_EAGER_SAMPLE") != "1"
               and not getattr(sub.lstt, "early_long_read", False))
    if sampled:                  # the `tail` graphs of the sampled frames (all slot variants) are captured here, not in the timed region
        sub.lstt._sample_read = True
        all_clips(t)
        t += 1
    for _ in range(args.warmup):
        all_clips(t)
        t += 1
    torch.cuda.synchronize()

    if os.environ.get("RMEM_BENCH_NOSYNC"):      # experiment: no long-term updates (no D2H) in the timed region
        for e in engines:
            e.aot_engines[0].long_term_mem_gap = 10 ** 6
    masks = torch.zeros(C, args.steps, H_OUT, W_OUT, dtype=torch.uint8, device=dev)
    lstt = sub.lstt
    lstt.enable_kernel_timing(True)       # clears the event list
    lstt._timing = False
    if dist is not None:                  # warm-up of the exchange step too: the first all-gather of a shape sets up RCCL's
        from rmem_amd.driver import gather_masks      # channels and buffers (the timed region ends with the same call)
        del_me = gather_masks(masks, world)
        torch.cuda.synchronize()
        del del_me
    cpu0 = time.process_time()                # user + system time of every thread of this process (main + graph launcher)
    thr0 = thread_cpu_seconds()               # (reads /proc for every thread, ~1 ms: taken BEFORE the bracket, not inside it)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # Steady-state frames replay hipGraphs.  The dominant kernel is timed by HIP events on its launch stream INSIDE
    # replayed frames: on every fifth frame the LSTT is replayed as front graph | the fused read of layer 0 launched on
    # its own between two events | tail graph (engine._graphed_frame) -- same kernels, same order, the prefetched
    # encoder pass beside it as on every other frame.  (torch on ROCm refuses event nodes inside a graph; an eagerly
    # issued frame, rounds 1-3, is host-bound and leaves the kernel the GPU to itself: 106 us against 119 in the trace.)
    # Models without the front / tail split (AOT block) keep the eagerly issued frame, one in fifty.
    n_eager = max(1, args.steps // 50)
    eager_at = {(i * args.steps) // n_eager for i in range(n_eager)}
    step_ev = [] if os.environ.get("RMEM_BENCH_STEP_EVENTS") else None      # diagnosis: one event per step -> per-step ms on stderr
    if step_ev is not None:
        step_ev.append(torch.cuda.Event(enable_timing=True))
        step_ev[-1].record(streams[0])
        ev0_at = time.perf_counter() - t0
    for k in range(args.steps):
        if sampled:
            lstt._sample_read = (k % 5 == 2)
        else:
            lstt._timing = (k in eager_at) and not os.environ.get("RMEM_BENCH_NOSYNC")
        all_clips(t + k, masks)
        if step_ev is not None:
            step_ev.append(torch.cuda.Event(enable_timing=True))
            step_ev[-1].record(streams[0])
    lstt._timing = False
    host_issue = time.perf_counter() - t0     # host-side launch time (GPU work still in flight)
    host_cpu = time.process_time() - cpu0     # CPU seconds this rank burned while issuing (all threads)
    thr1 = thread_cpu_seconds()
    host_threads = sorted(((1e3 * (v - thr0.get(k, 0.0)) / args.steps, k) for k, v in thr1.items()), reverse=True)
    host_threads = {f"{name}#{i}": round(ms, 3) for i, (ms, name) in enumerate(host_threads[:6]) if ms > 0.005}
    for st in streams[1:]:
        streams[0].wait_stream(st)
    if dist is not None:                      # collect per-clip masks (the only exchange step)
        from rmem_amd.driver import gather_masks
        gathered = gather_masks(masks, world)          # [world*C, steps, H, W] uint8 over RCCL
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if step_ev is not None:
        print("per-step ms (stream events):", [round(a.elapsed_time(b), 3) for a, b in zip(step_ev[:-1], step_ev[1:])],
              "host issue ms %.2f, wall ms %.2f, first event at %.3f ms" % (host_issue * 1e3, elapsed * 1e3, ev0_at * 1e3), file=sys.stderr)
    elapsed, per_rank_s = max_over_ranks(dist, elapsed, dev)

    fps = world * C * args.steps / elapsed

    # ---- the two-import drop-in of INTEGRATION.md section 1: the reference evaluator announces no
    # frames and does its own softmax / argmax / nearest resize (managers/evaluator.py:424-441,
    # 518-523).  Same engines, same steady state, a shorter timed run; reported beside the headline.
    dropin = None
    if rank == 0 and world == 1 and C == 1 and not args.no_dropin and not (args.no_prefetch and args.reference_postproc):
        PREFETCH_SAVE, POST_SAVE = PREFETCH, args.reference_postproc
        PREFETCH, args.reference_postproc = False, True
        n_d = max(10, args.steps // 2)
        for k in range(8):
            all_clips(t + args.steps + k)
        torch.cuda.synchronize()
        td = time.perf_counter()
        for k in range(n_d):
            all_clips(t + args.steps + 8 + k)
        torch.cuda.synchronize()
        dropin = {"value": n_d / (time.perf_counter() - td), "unit": "frames/s", "steps": n_d,
                  "flags": "--no-prefetch --reference-postproc (match_propogate_one_frame(img, output_size) -> torch "
                           "softmax/argmax/nearest -> update_memory, nothing announced ahead)"}
        PREFETCH, args.reference_postproc = PREFETCH_SAVE, POST_SAVE
    # ---- per-kernel roofline sample (outside the timed window, same steady state and announcements): every fourth frame
    # of a short extra run issues the LSTT's `rest` part eagerly with HIP events around each launch
    # (DeAOTEngine._graphed_frame, lstt._ev); the host is frames ahead of the GPU, so the launches queue like replayed ones
    kernels = None
    if rank == 0 and C == 1 and sampled and os.environ.get("RMEM_BENCH_KERNELS", "1") == "1":
        tk = t + args.steps + 200
        lstt._kev_store, lstt._kev_frames = [], 0
        for k in range(8):
            all_clips(tk + k)
        for k in range(48):
            lstt._sample_kernels = (k % 4 == 2)
            all_clips(tk + 8 + k)
        lstt._sample_kernels = False
        torch.cuda.synchronize()
        kernels = lstt.kernel_report(MFMA_PEAK_TFLOPS)
    out = {
        "metric": "frames/sec/GPU (480p, K=4 memory) R50-DeAOTL+RMem; mask IoU vs ref"
        if (args.config == "480p_k4" and args.model == "r50_deaotl") else f"frames/sec/GPU ({args.config}) {args.model}+RMem"
        if args.model != "r50_deaotl"
        else "frames/sec/GPU (720p, K=8 memory, 3 objects) R50-DeAOTL+RMem",
        "value": fps, "unit": "frames/s (whole job)", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp16x3 (split-fp16 MFMA: hi/lo planes, 3 products, fp32 accumulate)"
        if args.nsplit == 3 else "fp16 linears (MFMA, fp32 accumulate), fp16x3 memory reads",
        "data": "synthetic",
        "config": {"workload": f"{ {'r50_deaotl': 'R50-DeAOTL', 'r50_aotl': 'R50-AOTL', 'swinb_aotl': 'SwinB-AOTL'}[args.model] } + RMem, {H_OUT}p ({H_IN}x{W_IN}, {lstt.N} tokens), K={mem_k} memory, "
                               f"batch={C} clip{'s' if C > 1 else ''} per GPU, long_term_mem_gap={args.gap}, steady-state bank (T={mem_k})",
                   "frames_per_sec_per_gpu": fps / world, "precision_nsplit": args.nsplit,
                   "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
                   "host_cpu_ms_per_step": 1e3 * host_cpu / args.steps, "host_cpu_ms_per_step_by_thread": host_threads,
                   "clips_per_gpu": C,
                   "parallelism": f"clips sharded {C}-per-GPU x{world}" + (" (one engine + HIP stream per clip)" if C > 1 else "")
                   + ", all-gather of masks"},
    }
    # every rank's host side (one small all-gather after the timed window): CPU ms per step of the process and of its
    # three busiest threads, where it is pinned
    top3 = (sorted(host_threads.values(), reverse=True) + [0.0, 0.0, 0.0])[:3]
    rows = gather_rank_vectors(dist, [1e3 * host_cpu / args.steps] + top3 +
                               [1.0 if PIN.get("pinned") else 0.0, float(PIN.get("numa_node") if PIN.get("numa_node") is not None else -1),
                                float(PIN.get("cpus", 0)), float(PIN.get("first_cpu", -1))], dev)
    out["config"]["per_rank_frames_per_sec"] = [C * args.steps / t_ for t_ in per_rank_s]
    out["config"]["per_rank_host"] = [{"host_cpu_ms_per_step": round(r[0], 3), "busiest_threads_ms_per_step": [round(v, 3) for v in r[1:4]],
                                       "pinned": bool(r[4]), "numa_node": int(r[5]), "cpus": int(r[6]), "first_cpu": int(r[7])} for r in rows]
    if dist is not None:
        import hashlib
        out["config"]["dist_backend"] = dist.get_backend()
        if rank == 0:      # outside the timed region: what the exchange step delivered, [world*C, steps, H, W] uint8
            out["config"]["gathered_masks_shape"] = list(gathered.shape)
            out["config"]["gathered_masks_sha256"] = hashlib.sha256(gathered.cpu().numpy().tobytes()).hexdigest()
    if rank == 0:
        out["roofline"] = lstt.roofline_report(MFMA_PEAK_TFLOPS)
        if out["roofline"]:
            out["roofline"]["how"] = ("HIP events around the layer-0 launch of every 5th REPLAYED frame of the timed region "
                                      "(front graph | launch | tail graph), beside the prefetched encoder pass") if sampled else \
                "HIP events inside one eagerly issued frame in fifty"
        if out["roofline"] and hasattr(lstt, "time_read_isolated"):
            # information only: the same launch with the GPU to itself (in the frame it shares the
            # CUs with the prefetched encoder pass); `achieved` / `frac` above are the in-frame figures
            iso = lstt.time_read_isolated()
            out["roofline"]["isolated_mean_us"] = iso
            out["roofline"]["frac_isolated"] = out["roofline"]["algorithmic_flops_per_launch"] / (iso * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS
        if out["roofline"] and C == 1:
            pmc_traffic(out["roofline"], (args.model, args.config, "one"))
        if out["roofline"]:
            # information only: what the peak means on this board under sustained matrix load
            out["roofline"]["context"] = ("split-fp16: 3 MFMAs issued per algorithmic product; under this kernel the board is "
                                          "power-limited (1300 W, shader clock 1.7-1.9 GHz of 2.4); hipBLASLt's fp16 8192^3 GEMM "
                                          "sustains 1275 TFLOP/s = 0.51 of `peak` on the same box (profiles/r03_s_power_probe.json)")
        if out["roofline"] and kernels:
            # the five largest kernel classes of the memory path by time per frame, each against the roofline that bounds it;
            # `roofline` itself stays the largest one.  traffic = HBM-side bytes per launch from the committed PMC passes
            for e in kernels[:5]:
                pmc_kernel_traffic(e, (args.model, args.config, "one"))
            how = (f"{lstt._kev_frames} sampled frames after the timed window (every 4th frame of 48, same announcements): the "
                   "LSTT's `rest` part issued eagerly with HIP events around every launch, queued behind the previous frames' "
                   "work; us_per_frame sums a class's launches")
            read_block = out["roofline"]
            top = kernels[0]
            if not top["kernel"].startswith("read64x2"):
                # `roofline` is the kernel with the largest time per frame.  Since round 5 that is measured, not assumed: the
                # projection kernel (one binary, four launch shapes, 11 launches per sampled pass) takes more of a frame than
                # the fused attention read.  The read's own block -- timed inside the timed region -- stays beside it.
                rl = dict(top)
                rl.update(algorithmic_flops_per_launch=top.get("algorithmic_gflop_per_launch", 0.0) * 1e9,
                          launches=int(round(top["launches_per_frame"] * lstt._kev_frames)), how=how,
                          why_this_kernel="largest time per frame among the kernels of the memory path (kernels[0]); "
                                          "the fused attention read, the largest single launch, is `attention_read`")
                rl.setdefault("traffic", None)
                rl["attention_read"] = read_block
                out["roofline"] = rl
            out["roofline"]["kernels"] = kernels[:5]
            out["roofline"]["kernels_how"] = how
            mp_us = sum(e["us_per_frame"] for e in kernels)
            out["roofline"]["memory_path_us_per_frame_sampled"] = mp_us
            # ONE number for the whole memory path: algorithmic GFLOP of a frame (the sampled launches' own counts; SURVEY 8d
            # gives 129.1 for 480p K=4 including the memory update's ID_V GEMMs, which run in the update graph and are not
            # sampled) over the kernel time of the memory path in the frame, against the MFMA peak
            mp_gf = sum(e.get("algorithmic_gflop_per_launch", 0.0) * e["launches_per_frame"] for e in kernels)
            out["roofline"]["memory_path_gflop_per_frame"] = mp_gf
            out["roofline"]["memory_path_frac"] = (mp_gf * 1e9 / (mp_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS) if mp_us > 0 else None
            out["roofline"]["memory_path_overhead_mb_per_frame"] = sum(e.get("overhead_mb_per_launch", 0.0) * e["launches_per_frame"]
                                                                        for e in kernels)
        if dropin is not None:
            out["dropin"] = dropin
        if world == 1 and not args.no_cpu_baseline:
            cb, par = cpu_baseline_and_parity(cpu_model, model, cfg, args, dev)
            out["cpu_baseline"] = cb
            out.update(par)
            fixture = {("r50_deaotl", "480p_k4"): "clip_480p_long", ("r50_deaotl", "720p_k8"): "clip_720p_k8",
                       ("r50_aotl", "480p_k4"): "clip_aot_480p", ("swinb_aotl", "480p_k4"): "clip_swin_480p"}.get((args.model, args.config))
            if fixture is not None and args.nsplit == 3:
                ref_par = parity_vs_reference_fixture(model, cfg, args, dev, fixture)
                if ref_par is not None:
                    out["parity_vs_reference"] = ref_par
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def batched_steady(args, world, rank, dev, dist, cfg, model, mem_k):
    """Steady-state frames of B = --clips-per-gpu clips per GPU served by shared launches
    (rmem_amd.batched.BatchedDeAOTEngine): same protocol as the default mode (reference frame,
    pre-roll until the bank holds K slots, timed steps), a step = one frame of every clip."""
    from rmem_amd import hip
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.synth import synth_clip
    B = max(1, args.clips_per_gpu)
    eng = BatchedDeAOTEngine(model, B, long_term_mem_gap=args.gap, nsplit=args.nsplit)
    ring = 8
    per = [synth_clip(rank * B + i, ring, H_IN, W_IN, 3) for i in range(B)]
    imgs = [torch.cat([per[i][0][t] for i in range(B)]).to(dev) for t in range(ring)]
    labs = torch.cat([per[i][1] for i in range(B)]).to(dev)
    eng.add_reference_frame(imgs[0], labs, obj_nums=[3] * B, frame_step=0)
    lab_in = eng.lstt.label_buffer(*eng.input_size_2d)

    def step(t, masks_out=None):
        logit = eng.match_propogate_one_frame(imgs[t % ring], output_size=None,
                                              next_imgs=None if args.no_prefetch else imgs[(t + 1) % ring])
        for i in range(B):
            out = masks_out[i, t % masks_out.shape[1]] if masks_out is not None else None
            lab = hip.labels_from_logits([logit[i:i + 1]], [False], (H_OUT, W_OUT), cfg.MODEL_ALIGN_CORNERS, out=out)
            hip.label_resize_nearest(lab, eng.input_size_2d, out=lab_in[i])
        eng.update_memory(lab_in)

    t = 1
    while len(eng.lstt.clips[0].bank) < cfg.mem_cap:
        step(t)
        t += 1
    for _ in range(2 * cfg.mem_cap + 6 + args.warmup):       # every (clip, slot) argument block recorded
        step(t)
        t += 1
    masks = torch.zeros(B, args.steps, H_OUT, W_OUT, dtype=torch.uint8, device=dev)
    if dist is not None:                  # warm-up of the exchange step (RCCL channels / buffers of this shape)
        from rmem_amd.driver import gather_masks
        del_me = gather_masks(masks, world)
        torch.cuda.synchronize()
        del del_me
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    launches0 = eng.lstt.launches
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(t + k, masks)
    host_issue = time.perf_counter() - t0
    if dist is not None:
        from rmem_amd.driver import gather_masks
        gathered = gather_masks(masks, world)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed, per_rank_s = max_over_ranks(dist, elapsed, dev)
    fps = world * B * args.steps / elapsed
    c0 = eng.lstt.clips[0]
    out = {"metric": "frames/sec/GPU (480p, K=4 memory) R50-DeAOTL+RMem; mask IoU vs ref" if args.config == "480p_k4"
           else "frames/sec/GPU (720p, K=8 memory, 3 objects) R50-DeAOTL+RMem",
           "value": fps, "unit": "frames/s (whole job)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "fp16x3 (split-fp16 MFMA: hi/lo planes, 3 products, fp32 accumulate)", "data": "synthetic",
           "config": {"workload": f"R50-DeAOTL + RMem, {H_OUT}p ({H_IN}x{W_IN}, {c0.N} tokens), K={mem_k} memory, {B} clips per GPU "
                                  f"in lockstep, ONE launch per kernel for all clips, long_term_mem_gap={args.gap}, steady-state bank",
                      "frames_per_sec_per_gpu": fps / world, "clips_per_gpu": B, "batched": True,
                      "memory_path_launches_per_step": (eng.lstt.launches - launches0) / args.steps,
                      "key_splits_long_win_self": [c0.ks_long, c0.ks_win, c0.ks_self],
                      "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
                      "parallelism": f"clips sharded {B}-per-GPU x{world}, batched launches, all-gather of masks"}}
    if dist is not None:
        import hashlib
        out["config"]["per_rank_frames_per_sec"] = [B * args.steps / t_ for t_ in per_rank_s]
        out["config"]["dist_backend"] = dist.get_backend()
        if rank == 0:
            out["config"]["gathered_masks_sha256"] = hashlib.sha256(gathered.cpu().numpy().tobytes()).hexdigest()
    if rank == 0:
        iso = eng.lstt.time_read_isolated()
        flops = B * c0.read_flops(len(c0.bank))
        ach = flops / (iso * 1e-6) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": f"read64x2_many[_pull]_kernel (fused long-term T={len(c0.bank)} + windowed memory read of {B} clips in one launch)",
                           "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                           "traffic": None, "mean_us": iso, "algorithmic_flops_per_launch": flops,
                           "note": "isolated launches (HIP events, back to back); includes the upload of the clips' argument blocks"}
        pmc_traffic(out["roofline"], ("r50_deaotl", "480p_k4", f"batched{B}"))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def thread_cpu_seconds():
    """user + system CPU seconds of every thread of this process, keyed by (tid, name) (Linux /proc; {} elsewhere)."""
    import threading
    names = {t.native_id: t.name for t in threading.enumerate()}
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            with open(f"/proc/self/task/{tid}/stat") as f:
                st = f.read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            fields = st[st.rindex(")") + 2:].split()
            out[f"{names.get(int(tid), comm)}:{tid}"] = (int(fields[11]) + int(fields[12])) / tick
    except (OSError, ValueError):
        return {}
    return out


def clips_ragged(args, world, rank, dev, dist, drv, D):
    """`--config clips64 --ragged`: a dataset of clips of unequal length -- what the reference's worker queue is for
    (managers/evaluator.py:276-295).  3 clips per slot, lengths uniform in [clip_frames/2, 2*clip_frames]; ranks get
    clips longest-first (driver.assign_clips_by_length), and a BatchedClipDriver serves its share through the slot queue
    (driver.run_sharded_dataset).  One untimed pass over a short dataset of the same geometry first; the timed window
    ends with the gathered masks in pinned host memory; their sha256 per clip is taken after it."""
    from rmem_amd.synth import synth_clip
    F_ = args.clip_frames
    n_clips = 3 * args.clips_per_rank * world
    rs = np.random.RandomState(7)
    lengths = [int(x) for x in rs.randint(max(2, F_ // 2), 2 * F_ + 1, size=n_clips)]

    def frames_of_len(cid, n):
        imgs, lab = synth_clip(cid, n, H_IN, W_IN, 3)
        lab0 = F.interpolate(lab, size=(H_OUT, W_OUT), mode="nearest").to(dev)
        return [D.make_samples(imgs[t].to(dev), lab0 if t == 0 else None, (H_OUT, W_OUT), 3, name=f"{t:05d}.jpg")
                for t in range(n)]
    # warm-up: every bank depth and slot state once (recordings, hipGraph captures, MIOpen search)
    wl = [F_, F_ // 2 + 1, F_, F_ // 2 + 2][:max(2, min(4, args.clips_per_rank + 1))]
    warm = {i: frames_of_len(10 ** 6 + i, n) for i, n in enumerate(wl)}
    D.run_sharded_dataset(drv, wl, 1, 0, lambda c: warm[c])
    assign = D.assign_clips_by_length(lengths, world)
    mine = assign[rank]
    cache = {c: frames_of_len(c, lengths[c]) for c in mine}
    per = max(1, max(len(a) for a in assign))
    host_pin = torch.empty((world * per, max(lengths) - 1, H_OUT, W_OUT), dtype=torch.uint8)     # destination of the masks: pinned,
    if dev.type == "cuda":                                                                       # allocated before the window
        host_pin = host_pin.pin_memory()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    allm, frames_per_rank = D.run_sharded_dataset(drv, lengths, world, rank, lambda c: cache[c], hashes=False, host_out=host_pin)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed, per_rank_s = max_over_ranks(dist, elapsed, dev)
    hashes = D.hash_dataset_masks(allm, lengths, world)       # (after the window, as in clips64)
    total = sum(lengths) - n_clips
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec (480p, K=4) R50-DeAOTL+RMem, clips of unequal length over ranks and slots, masks all-gathered",
            "value": total / elapsed, "unit": "frames/s (whole job)", "n_gpus": world, "steps": total // world, "warmup": 1,
            "ms_per_step": 1e3 * elapsed / max(1, total // world), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16x3 (split-fp16 MFMA)", "data": "synthetic",
            "config": {"workload": f"R50-DeAOTL + RMem, 480p, K=4, {n_clips} clips of {min(lengths)}-{max(lengths)} frames "
                                   f"({total} propagated), longest-first over {world} rank(s)"
                                   + (f", {args.clips_per_rank} slots per rank with the slot queue" if args.batched else ", one clip at a time"),
                       "lengths": lengths, "frames_per_rank": frames_per_rank, "per_rank_seconds": per_rank_s,
                       "batched": bool(args.batched), "queue_stats_rank0": getattr(drv, "queue_stats", None),
                       "dist_backend": dist.get_backend() if dist is not None else None},
            "clip_sha256": [h[:16] for h in hashes]}))
    if dist is not None:
        dist.destroy_process_group()


def clips64(args, world, rank, local_rank, dev, dist):
    """BASELINE.json configs[3]: R50-DeAOTL + RMem, 480p, K=4, `clips_per_rank` x world independent
    synthetic clips of `clip_frames` frames, clip i on rank i mod world, every clip through
    rmem_amd.driver.ClipDriver (the evaluator's per-clip protocol: gap rule, reference frame, bank
    fill, fused label post-processing), one RCCL all-gather of the uint8 masks.  The timed window
    covers everything from the first reference frame to the gathered masks."""
    import hashlib
    from rmem_amd import driver as D
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    load_synthetic_weights(model)
    model = model.to(dev)
    n_clips, F_ = args.clips_per_rank * world, args.clip_frames
    drv = D.ClipDriver(model, cfg, gpu_id=local_rank) if not args.batched else \
        D.BatchedClipDriver(model, args.clips_per_rank, cfg, gpu_id=local_rank)
    if args.ragged:
        return clips_ragged(args, world, rank, dev, dist, drv, D)

    def frames_of(cid):          # frames are resident in HBM before they are consumed; generated per clip
        imgs, lab = synth_clip(cid, F_, H_IN, W_IN, 3)
        lab0 = F.interpolate(lab, size=(H_OUT, W_OUT), mode="nearest").to(dev)
        return [D.make_samples(imgs[t].to(dev), lab0 if t == 0 else None, (H_OUT, W_OUT), 3, name=f"{t:05d}.jpg")
                for t in range(F_)]

    # warm-up clip (MIOpen solver search, hipGraph captures of the first geometry): not timed; the node's first rank before
    # the others (first_convolutions_in_turn)
    if args.batched:
        first_convolutions_in_turn(dist, lambda: drv.run_clips([frames_of(10 ** 6 + i) for i in range(args.clips_per_rank)], num_frames=F_))
    else:
        first_convolutions_in_turn(dist, lambda: drv.run_clip(frames_of(10 ** 6), num_frames=F_))
    # all clips of this rank are materialised first so that the timed window holds no host-side synthesis
    mine = D.shard_clips(n_clips, world, rank)
    cache = {c: frames_of(c) for c in mine}
    # the masks end in pinned host memory allocated before the window (a pageable destination adds 10-30 ms of page faults
    # and staging copies for 49 MB, varying from process to process)
    host_pin = torch.empty((n_clips, F_ - 1, H_OUT, W_OUT), dtype=torch.uint8)
    if dev.type == "cuda":
        host_pin = host_pin.pin_memory()
    if dist is not None:                  # warm-up of the exchange step (RCCL channels / buffers of this shape)
        del_me = D.gather_masks(torch.zeros((len(mine), F_ - 1, H_OUT, W_OUT), dtype=torch.uint8, device=dev), world)
        torch.cuda.synchronize()
        del del_me
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    if args.batched:       # this rank's clips in lockstep, one launch per kernel for all of them
        res = drv.run_clips([cache[c] for c in mine], num_frames=F_)
        t_issue = time.perf_counter() - t0             # host time to issue the clips (the GPU may still be running)
        allm = D.gather_masks(torch.stack([r.masks for r in res]), world)
    else:
        _, allm, frames_run = D.run_sharded_clips(drv, n_clips, world, rank, lambda c: cache[c], F_, hashes=False)
        t_issue = None
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0                   # every clip done, masks gathered in HBM
    host_pin.copy_(allm, non_blocking=True)            # the masks of every rank in (pinned) host memory: end of the job
    torch.cuda.synchronize()
    host = host_pin.numpy()
    sections = {"issued": round(t_issue, 4) if t_issue is not None else None, "masks_gathered": round(t_gpu, 4),
                "masks_on_host": round(time.perf_counter() - t0, 4)}
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    host_cpu = time.process_time() - cpu0
    elapsed, per_rank_s = max_over_ranks(dist, elapsed, dev)
    PIN = getattr(args, "_pin", {})
    rows = gather_rank_vectors(dist, [host_cpu, 1.0 if PIN.get("pinned") else 0.0,
                                      float(PIN.get("numa_node") if PIN.get("numa_node") is not None else -1),
                                      float(PIN.get("cpus", 0)), float(PIN.get("first_cpu", -1))], dev)
    hashes = D.hash_masks(host, n_clips, world)        # verification by-product, after the window (sha256 of 49 MB: 20-40 ms of host time)
    total_frames = n_clips * (F_ - 1)               # propagated frames (the reference frame is not a "frame/s" frame in evaluator.py:571-587)
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec (480p, K=4) R50-DeAOTL+RMem, independent clips sharded across GPUs, masks all-gathered",
            "value": total_frames / elapsed, "unit": "frames/s (whole job)", "n_gpus": world, "steps": total_frames // world,
            "warmup": 1, "ms_per_step": 1e3 * elapsed / max(1, total_frames // world), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16x3 (split-fp16 MFMA)", "data": "synthetic",
            "config": {"workload": f"R50-DeAOTL + RMem, 480p, K=4, {n_clips} clips x {F_} frames, {args.clips_per_rank} per rank "
                                   f"(clip i -> rank i mod {world}), evaluator gap rule (gap {D.memory_gap(F_)}), reference frame + bank fill + gather + copy of the masks to the host timed",
                       "clips": n_clips, "frames_per_clip": F_, "frames_per_sec_per_gpu": total_frames / elapsed / world,
                       "batched": bool(args.batched),
                       "per_rank_seconds": per_rank_s, "rank0_sections_s": sections,
                       "per_rank_frames_per_sec": [args.clips_per_rank * (F_ - 1) / t_ for t_ in per_rank_s],
                       "per_rank_host": [{"host_cpu_s": round(r[0], 3), "host_cpu_cores_busy": round(r[0] / t_, 2), "pinned": bool(r[1]),
                                          "numa_node": int(r[2]), "cpus": int(r[3]), "first_cpu": int(r[4])}
                                         for r, t_ in zip(rows, per_rank_s)],
                       "dist_backend": dist.get_backend() if dist is not None else None,
                       "parallelism": f"clips sharded {args.clips_per_rank}-per-GPU x{world}"
                                      + (" in lockstep (one launch per kernel for all clips of a rank)" if args.batched else "")
                                      + ", one all-gather of uint8 masks "
                                      f"({allm.numel() / 1e6:.1f} MB)"},
            "clip_sha256": [h[:16] for h in hashes],
            "masks_sha256": hashlib.sha256(host.tobytes()).hexdigest()}))
    if dist is not None:
        dist.destroy_process_group()


def db_eval_iou(annotation, segmentation):
    """Jaccard index of two binary maps, 1 when both are empty (restates
    evaluation/source/metrics.py:6-37 of the reference without its cv2 import)."""
    inter = float((annotation & segmentation).sum())
    union = float((annotation | segmentation).sum())
    return 1.0 if union == 0 else inter / union


def label_iou(a: np.ndarray, b: np.ndarray):
    """Mean / min Jaccard index over EVERY id present in either label map (background and all live object ids: the
    reference keeps all `max_obj_num` ids live, aot_engine.py:695-700), evaluation/source/metrics.py:db_eval_iou per id."""
    ids = sorted(set(np.unique(a).tolist()) | set(np.unique(b).tolist()))
    v = [db_eval_iou(a == o, b == o) for o in ids]
    return float(np.mean(v)), float(np.min(v)), [int(o) for o in ids]


def cpu_baseline_and_parity(cpu_model, gpu_model, cfg, args, dev):
    """(a) cpu_baseline: the oracle (CPU restatement of the reference path, fp32 PyTorch-CPU) timed
    on a bounded sample of the same workload: same clip generator, geometry and K; the bank is
    pre-filled (gap 1) to min(K, 4) slots, then `cpu_frames` frames are timed with the reference's
    timing window.  (b) parity, outside every timed region: a fresh HIP engine runs the same frames
    teacher-forced (both engines are fed the ORACLE's label, so the count is per frame, not the growth
    of a chaotic closed loop): mismatching label pixels and the IoU over every id present against the oracle.
    Serves every --model / --config of the one-clip mode (DeAOT and AOT oracle engines)."""
    from oracle.engine_ref import OracleAOTEngine, OracleDeAOTEngine
    from rmem_amd.engine import build_engine
    from rmem_amd.synth import synth_clip
    # all cores is slower than a few dozen on many-core hosts for these op sizes (measured:
    # 256 threads -> 112 s/frame); use at most 32 and report the count actually used.
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ora = (OracleDeAOTEngine if cfg.MODEL_VOS == "deaot" else OracleAOTEngine)(cpu_model, long_term_mem_gap=1)
    eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=gpu_model, gpu_id=dev.index or 0,
                       long_term_mem_gap=1, nsplit=args.nsplit)
    eng.eval()
    big = H_IN > 600
    n = min(args.cpu_frames, 2) if big else args.cpu_frames          # (720p: 8-11 s per frame on the CPU)
    pre = min(cfg.mem_cap, 4)
    imgs, lab = synth_clip(0, pre + n, H_IN, W_IN, 3)
    gimgs = [x.to(dev) for x in imgs]
    ora.add_reference_frame(imgs[0], lab, obj_nums=[3], frame_step=0)
    eng.add_reference_frame(gimgs[0], lab.to(dev), obj_nums=[3], frame_step=0)
    mism, ious, iou_min, ids_seen, lerr = [], [], [], set(), []

    def step(t, timed):
        logit = ora.match_propogate_one_frame(imgs[t], output_size=(H_OUT, W_OUT))
        pred = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True).float()
        cur = F.interpolate(pred, size=ora.input_size_2d, mode="nearest")
        ora.update_memory(cur)
        return pred, cur, ora.pred_id_logits.clone()

    def hip_step(t, pred, cur, ologits):
        lg = eng.match_propogate_one_frame(gimgs[t], output_size=(H_OUT, W_OUT))
        mine = torch.argmax(lg, dim=1, keepdim=True).cpu()
        a, b = pred.long().numpy()[0, 0], mine.numpy()[0, 0]
        mism.append(int((a != b).sum()))
        m, lo, ids = label_iou(a, b)
        ious.append(m), iou_min.append(lo), ids_seen.update(ids)
        lerr.append(float((eng.aot_engines[0].pred_id_logits.cpu() - ologits).abs().max()))
        eng.update_memory(cur.to(dev))

    fed = []
    for t in range(1, pre):
        fed.append((t,) + step(t, False))
    ora.long_term_mem_gap = args.gap
    t0 = time.perf_counter()
    for t in range(pre, pre + n):
        fed.append((t,) + step(t, True))
    el = time.perf_counter() - t0
    for t, pred, cur, ologits in fed:               # the HIP engine replays the same protocol
        if t == pre:
            eng.long_term_mem_gap = args.gap
            for e in eng.aot_engines:
                e.long_term_mem_gap = args.gap
        hip_step(t, pred, cur, ologits)
    idx_ok = list(eng.aot_engines[0].long_memories_indexes) == list(ora.long_memories_indexes)
    cb = {"value": n / el, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
          "sample": f"{n} frames (bank of {pre} slots) of the same {H_OUT}p K={cfg.mem_cap} workload ({args.model}), oracle/ on PyTorch-CPU fp32"}
    par = {"mask_mismatch_px": mism, "mask_pixels_per_frame": int(H_OUT * W_OUT),
           "iou_vs_oracle": float(np.mean(ious)), "iou_vs_oracle_min": float(np.min(iou_min)), "iou_ids": sorted(ids_seen),
           "logit_max_abs_err_vs_oracle": float(np.max(lerr)), "eviction_sequence_equal": bool(idx_ok),
           "parity_note": f"{len(mism)} teacher-forced frames (oracle labels fed to both), HIP engine vs CPU oracle, "
                          "outside the timed regions; IoU = evaluation/source/metrics.py:db_eval_iou per id, mean (and min) over "
                          "EVERY id present in either label map (iou_ids), background included"}
    return cb, par


def parity_vs_reference_fixture(gpu_model, cfg, args, dev, name="clip_480p_long"):
    """'mask IoU vs ref' on the benchmarked schedule, against the REFERENCE itself: tests/golden/clip_480p_long.* holds
    the reference's own closed-loop run of this workload (481x849, K = 4, the evaluator's gap 5, 46 frames: the bank is
    full from frame 15, six evictions; make_golden.py:gen_clip_480p_long), clip_720p_k8.* that of configs[2] (721x1281,
    K = 8, gap 1, 11 frames, three evictions), clip_aot_480p.* / clip_swin_480p.* those of configs[0] / configs[4] (R50-AOTL
    481x849 x 16 frames; SwinB-AOTL 480x848, gap 1, six evictions).  A fresh HIP engine runs the clip
    teacher-forced with the reference's labels, next frames announced as in the timed loop; per frame: pixels off the
    reference's label map, IoU over every id present; the kept-frame history must equal the reference's."""
    gd = os.path.join(ROOT, "tests", "golden")
    if not (os.path.exists(os.path.join(gd, name + ".json")) and os.path.exists(os.path.join(gd, name + ".npz"))):
        return None
    from rmem_amd.engine import build_engine
    from rmem_amd.synth import synth_clip
    meta = json.load(open(os.path.join(gd, name + ".json")))
    gold = np.load(os.path.join(gd, name + ".npz"))
    if (meta["H"], meta["W"], meta["former"] + meta["latter"]) != (H_IN, W_IN, cfg.mem_cap):
        return None
    eng = build_engine(cfg.MODEL_ENGINE, phase="eval", aot_model=gpu_model, gpu_id=dev.index or 0,
                       long_term_mem_gap=meta["gap"], nsplit=args.nsplit)
    eng.eval()
    imgs, lab = synth_clip(meta["seed"], meta["frames"], meta["H"], meta["W"], 3)
    imgs = [x.to(dev) for x in imgs]
    out_hw = tuple(meta["out_hw"])
    eng.add_reference_frame(imgs[0], lab.to(dev), obj_nums=[3], frame_step=0)
    mism, ious, iou_min, ids_seen, idx_ok = [], [], [], set(), True
    for t in range(1, meta["frames"]):
        nxt = imgs[t + 1:t + 1 + eng.lookahead] or None
        lg = eng.match_propogate_one_frame(imgs[t], output_size=out_hw, next_img=nxt)
        mine = torch.argmax(torch.softmax(lg, dim=1), dim=1)[0].cpu().numpy().astype(np.uint8)
        ref = gold["labels"][t - 1]
        mism.append(int((mine != ref).sum()))
        m, lo, ids = label_iou(ref, mine)
        ious.append(m), iou_min.append(lo), ids_seen.update(ids)
        fed = torch.from_numpy(ref).float()[None, None].to(dev)
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode="nearest"))
        idx_ok = idx_ok and list(eng.aot_engines[0].long_memories_indexes) == meta["indexes"][t - 1]
    return {"fixture": f"tests/golden/{name}.npz (the reference's own fp32 CPU run; fp64 near-tie lists in {name}_fp64.npz)",
            "frames": len(mism), "evictions": meta.get("evictions"), "gap": meta["gap"],
            "mask_mismatch_px": mism, "mask_mismatch_px_total": int(sum(mism)), "mask_pixels_per_frame": int(out_hw[0] * out_hw[1]),
            "iou_vs_reference": float(np.mean(ious)), "iou_vs_reference_min": float(np.min(iou_min)), "iou_ids": sorted(ids_seen),
            "bank_index_history_equal": bool(idx_ok),
            "note": "teacher-forced with the reference's labels; every differing pixel is an fp32 near-tie of the reference "
                    "itself (tests/test_hip_engine.py::test_480p_long_clip_gap5_vs_reference / test_720p_k8_vs_reference check "
                    "each against the fp64 list)"}


if __name__ == "__main__":
    main()
