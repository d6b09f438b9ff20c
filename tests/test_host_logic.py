"""CPU: host-side logic of the product path (no GPU compute): C-ABI exports, RMem policy
against the oracle and the reference's golden traces, slot bookkeeping, sharding."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rmem_amd.build import build_lib
    from rmem_amd import hip
    path = build_lib()
    lib = ctypes.CDLL(path)                       # symbol resolution only; no GPU calls
    header = open(os.path.join(ROOT, "include", "rmem_hip.h")).read()
    declared = set(re.findall(r"\bint\s+(rmem_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    every = set(re.findall(r"^[a-z][a-z0-9_ ]*[ *](rmem_[a-z0-9_]+)\s*\(", header, re.M))
    for name in every:
        assert hasattr(lib, name), f"{name} declared in include/rmem_hip.h but not exported"
    assert declared == set(hip.EXPORTS)
    assert every - declared == set(hip.EXPORTS_OTHER)
    assert lib.rmem_abi_version() == hip.ABI_VERSION == 18


def test_launch_recorder_records_without_a_gpu():
    """include/rmem_hip.h, "several clips' memory banks in one launch": while a thread records, the
    memory-path entry points validate and append their argument block instead of launching -- no HIP
    call is made, so this runs on the CPU.  Two recordings of the same calls with different buffers
    have the same signature and different blobs."""
    from rmem_amd import hip
    hip.load()
    recs = []
    for base in (0x10000, 0x90000):
        with hip.Recording() as r:
            lib = hip.load()
            rc = lib.rmem_layernorm_red(base, 256, None, 0, 0, 0, base + 0x1000, base + 0x2000, 100, 256, 1e-5,
                                        base + 0x3000, base + 0x4000, 256, None, 0, None)
            assert rc == 0
            assert lib.rmem_attn_mass_reduce(base, 100, 4, base + 64, base + 128, None) == 0
            assert lib.rmem_attn_mass_reduce(None, 100, 4, base + 64, base + 128, None) == -1   # still validated
        recs.append(r)
    a, b = recs
    assert a.count == b.count == 2 and len(a.blob) == len(b.blob) > 0
    assert a.signature == b.signature != 0 and a.blob != b.blob
    with hip.Recording() as c:
        assert hip.load().rmem_attn_mass_reduce(0x10000, 100, 4, 0x10040, 0x10080, None) == 0
    assert c.signature != a.signature
    with hip.Recording():
        with pytest.raises(hip.RmemError):
            with hip.Recording():
                pass


def test_ctypes_struct_sizes_match_header_layout():
    """The ctypes mirrors must have the C layout (catch drift between hip.py and the header)."""
    from rmem_amd import hip
    src = r'''
    #include "rmem_hip.h"
    #include <stdio.h>
    int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(rmem_linear_args), sizeof(rmem_mha_args),
                       sizeof(rmem_mha_combine_args), sizeof(rmem_read_args), sizeof(rmem_read_combine_args),
                       sizeof(rmem_bank_state)); return 0; }'''
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"),
                               "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [ctypes.sizeof(hip.LinearArgs), ctypes.sizeof(hip.MHAArgs),
                     ctypes.sizeof(hip.MHACombineArgs), ctypes.sizeof(hip.ReadArgs), ctypes.sizeof(hip.ReadCombineArgs),
                     ctypes.sizeof(hip.BankState)]


def test_configure_switches_and_no_getenv_in_the_library():
    """rmem_configure: known names inside their ranges are accepted, everything else is refused and changes nothing; and the
    C library itself never reads the environment (the RMEM_* variables are mapped by rmem_amd/hip.py when it loads it)."""
    from rmem_amd import hip
    lib = hip.load()
    assert lib.rmem_configure(b"dw_rows", 3) == 0 and lib.rmem_configure(b"dw_rows", 2) == 0
    assert lib.rmem_configure(b"dw_rows", 9) == -1 and lib.rmem_configure(b"no_such_switch", 1) == -1
    assert lib.rmem_configure(None, 1) == -1 and lib.rmem_configure(b"stream_var", 5) == -1
    with pytest.raises(hip.RmemError):
        hip.configure("linear_tiles", 2)
    hip.configure("linear_tiles", 0)
    csrc = os.path.join(ROOT, "rmem_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            code = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
            assert "getenv" not in code, f


def test_product_path_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rmem_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
                assert not re.search(r"sys\.path.*reference", txt), f


def test_missing_library_fails_loudly(monkeypatch):
    from rmem_amd import hip
    monkeypatch.setattr(hip, "_LIB", None)
    monkeypatch.setattr(hip, "_LIB_PATH", "/nonexistent/librmem_hip.so")
    with pytest.raises(hip.RmemError):
        hip.load()


def test_policy_matches_oracle_and_golden(golden_dir):
    from oracle import lstt_ref as R
    from rmem_amd.lstt import rmem_policy_step, temporal_pe_rows
    for T in range(1, 16):
        assert temporal_pe_rows(T) == R.temporal_pe_rows(T)
    rs = np.random.RandomState(0)
    for trial in range(200):
        n = rs.randint(1, 9)
        indexes = sorted(rs.choice(60, n + 1, replace=False).tolist())
        w = rs.rand(n).astype(np.float32)
        w /= w.sum()
        ema_prev = {i: float(rs.rand()) for i in indexes[:-1] if rs.rand() < 0.7}
        vis_prev = {i: int(rs.randint(1, 9)) for i in indexes[:-1] if rs.rand() < 0.7}
        d1, e1, v1 = rmem_policy_step(w, indexes, ema_prev, vis_prev, 1)
        d2, e2, v2, _ = R.rmem_policy_step([float(x) for x in w], indexes, ema_prev, vis_prev, 1)
        assert d1 == d2 and v1 == v2
        for k in e2:
            assert abs(float(e1[k]) - float(e2[k])) < 1e-6
    # replay the reference's golden EMA / visit traces through the product policy
    meta = json.load(open(os.path.join(golden_dir, "clip_small_k4_gap2.json")))
    for prev, cur in zip(meta["visits"][:-1], meta["visits"][1:]):
        if prev != cur:
            assert all(cur[k] >= prev.get(k, 0) for k in cur)


def test_slot_bookkeeping_never_overwrites_live_memory():
    """Ring-of-slots invariants of DeAOTLSTT (no GPU needed: exercise the pure bookkeeping)."""
    from rmem_amd.lstt import DeAOTLSTT

    class Stub:
        pass

    s = Stub()
    s.S, s.bank, s.short = 6, [], None
    free = lambda: DeAOTLSTT._free_slot(s)
    rs = np.random.RandomState(1)
    cur = free()
    s.bank, s.short = [cur], cur
    for step in range(200):
        cur = free()
        assert cur not in s.bank and cur != s.short
        s.short = cur
        if rs.rand() < 0.4:
            s.bank = s.bank + [cur]
            if len(s.bank) > 4:
                del s.bank[1 + rs.randint(0, len(s.bank) - 2)]
        assert len(set(s.bank)) == len(s.bank) <= 4


def test_shard_and_unshard():
    from rmem_amd.driver import shard_clips, unshard_order
    for n, world in [(64, 8), (16, 2), (8, 8), (5, 2)]:
        got = sorted(c for r in range(world) for c in shard_clips(n, world, r))
        assert got == list(range(n))
    assert unshard_order(8, 2) == [0, 2, 4, 6, 1, 3, 5, 7]


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from rmem_amd.driver import gather_masks, shard_clips
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    clips = shard_clips(4, world, rank)
    local = torch.stack([torch.full((3, 5, 7), c, dtype=torch.uint8) for c in clips])
    allm = gather_masks(local, world)
    if rank == 0:
        q.put(allm[:, 0, 0, 0].tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_masks_world2_gloo():
    """N>1 path on CPU: two processes, gloo backend, static shard + all-gather of masks."""
    import torch.multiprocessing as mp
    from rmem_amd.driver import unshard_order
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == unshard_order(4, 2)


def test_c_abi_rejects_invalid_arguments():
    """Argument validation happens before any launch, so it is testable without a GPU: every
    entry point returns RMEM_ERR_INVALID (-1) instead of launching on bad arguments."""
    import ctypes as C
    from rmem_amd import hip
    lib = hip.load()
    a = hip.LinearArgs()
    assert lib.rmem_linear(None, None) == -1
    assert lib.rmem_linear(C.byref(a), None) == -1                      # M = N = K = 0
    a.M, a.N, a.K, a.xh, a.yh, a.nsplit = 64, 64, 100, 1 << 20, 1 << 20, 1
    assert lib.rmem_linear(C.byref(a), None) == -1                      # K not a multiple of 64
    a.K, a.nsplit = 128, 2
    assert lib.rmem_linear(C.byref(a), None) == -1                      # nsplit must be 1 or 3
    a.nsplit = 3
    assert lib.rmem_linear(C.byref(a), None) == -1                      # nsplit 3 needs lo planes
    r = hip.ReadArgs()
    r.N, r.Npad, r.T, r.ksplits, r.ncols = 100, 100, 1, 1, 1024
    assert lib.rmem_attn_read(C.byref(r), None) == -1                   # Npad not a multiple of 128
    assert lib.rmem_attn_read2(C.byref(r), C.byref(r), None) == -1
    rc = hip.ReadCombineArgs()
    assert lib.rmem_attn_read_combine(C.byref(rc), None) == -1
    m = hip.MHAArgs()
    assert lib.rmem_mha_flash(C.byref(m), None) == -1
    assert lib.rmem_layernorm_split(None, 0, None, None, 0, 256, 1e-5, None, None, 0, None, 0, None) == -1
    assert lib.rmem_id_assign(None, 0, 0, None, None, 12, 17, 16, 8, 1, 1, 256, None, None, 1e-5,
                              None, None, 0, None, 0, 1, None) == -1
    assert lib.rmem_set_ints(None, None, 0, None) == -1


# ------------------------------------------------------------------ config 4: rank-count invariance
def _clip_frames(cid, frames, H, W, device="cpu"):
    from rmem_amd import driver as D
    from rmem_amd.synth import synth_clip
    imgs, lab = synth_clip(100 + cid, frames, H, W, 3)
    return [D.make_samples(imgs[t].to(device), lab.to(device) if t == 0 else None, (H, W), 3, name=f"{t:05d}.jpg")
            for t in range(frames)]


def _sharded_oracle_worker(rank, world, port, q, n_clips, frames, H, W):
    import torch.distributed as dist
    from oracle.engine_ref import OracleDeAOTEngine
    from rmem_amd import driver as D
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model.cfg = cfg
    drv = D.ClipDriver(model, cfg, engine_factory=lambda m: OracleDeAOTEngine(m), fixed_gap=2)
    hashes, allm, _ = D.run_sharded_clips(drv, n_clips, world, rank, lambda c: _clip_frames(c, frames, H, W), frames)
    if rank == 0:
        q.put(hashes)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_clips_hashes_do_not_depend_on_world_size():
    """BASELINE.json configs[3] (64 clips over 8 ranks) in miniature, oracle engines injected into
    the clip driver: 4 clips x 6 frames as world = 1 and as two gloo processes give the same sha256
    per clip (static shard clip i -> rank i mod world, one all-gather of uint8 masks)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 31500 + os.getpid() % 2000 + world
        procs = [ctx.Process(target=_sharded_oracle_worker, args=(r, world, port, q, 4, 6, 49, 65)) for r in range(world)]
        for p in procs:
            p.start()
        out[world] = q.get(timeout=600)
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
    assert out[1] == out[2] and len(set(out[1])) == 4      # same per clip, and the clips differ


def test_assign_clips_by_length():
    """Length-aware sharding (missing item 2; reference: work queue, evaluator.py:276-295)."""
    from rmem_amd.driver import assign_clips_by_length, shard_clips, shard_clips_by_length
    lengths = [100, 20, 20, 20, 20, 20, 90, 10]
    a = assign_clips_by_length(lengths, 2)
    assert sorted(c for r in a for c in r) == list(range(8))              # a partition
    loads = [sum(lengths[c] for c in r) for r in a]
    assert max(loads) == 150                                              # optimal here (300 frames over 2 ranks)
    rr = [sum(lengths[c] for c in shard_clips(8, 2, r)) for r in range(2)]
    assert max(rr) == 230                                                 # round-robin puts 100 and 90 on one rank
    assert a == assign_clips_by_length(lengths, 2)                        # deterministic: every rank computes the same
    assert shard_clips_by_length(lengths, 2, 1) == a[1]
    assert a[0][0] == 0 and a[1][0] == 6                                  # longest first on each rank
    # equal clips: as balanced as round-robin
    eq = assign_clips_by_length([16] * 64, 8)
    assert all(len(r) == 8 for r in eq)
    # more ranks than clips: the surplus ranks get nothing
    assert assign_clips_by_length([5, 7], 4) == [[1], [0], [], []]
    import random
    rnd = random.Random(3)
    for _ in range(50):                                                   # 4/3 bound of longest-first greedy
        ls = [rnd.randint(2, 120) for _ in range(rnd.randint(1, 40))]
        w = rnd.randint(1, 8)
        mk = max(sum(ls[c] for c in r) for r in assign_clips_by_length(ls, w))
        lower = max(max(ls), -(-sum(ls) // w))
        assert mk <= (4 / 3) * lower + max(ls) / 3 + 1e-9 or mk <= lower * (4 / 3 - 1 / (3 * w)) + max(ls)
    with pytest.raises(ValueError):
        assign_clips_by_length([3, 0], 2)
    with pytest.raises(ValueError):
        shard_clips_by_length([3, 4], 2, 2)


def _sharded_dataset_worker(rank, world, port, q, lengths, H, W):
    import torch.distributed as dist
    from oracle.engine_ref import OracleDeAOTEngine
    from rmem_amd import driver as D
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    model.cfg = cfg
    drv = D.ClipDriver(model, cfg, engine_factory=lambda m: OracleDeAOTEngine(m), fixed_gap=2)
    hashes, frames = D.run_sharded_dataset(drv, lengths, world, rank, lambda c: _clip_frames(c, lengths[c], H, W))
    if rank == 0:
        q.put((hashes, frames))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_dataset_of_unequal_clips_world_invariant():
    """Clips of unequal length (7, 3, 4, 5, 3 frames), longest-first assignment, padded all-gather: the sha256 of
    every clip's own frames as world = 1, 2 and 3 gloo processes (3: one rank ends up with a single clip)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    lengths = [7, 3, 4, 5, 3]
    out = {}
    for world in (1, 2, 3):
        q = ctx.Queue()
        port = 33500 + os.getpid() % 2000 + world
        procs = [ctx.Process(target=_sharded_dataset_worker, args=(r, world, port, q, lengths, 49, 65)) for r in range(world)]
        for p in procs:
            p.start()
        out[world] = q.get(timeout=600)
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
    assert out[1][0] == out[2][0] == out[3][0] and len(set(out[1][0])) == 5
    assert out[2][1] == [8, 9] and out[3][1] == [6, 6, 5]       # propagated frames per rank (22 frames over 2 / 3 ranks)


def test_sharded_dataset_with_fewer_clips_than_ranks_and_a_one_frame_clip():
    """run_sharded_dataset when a rank gets NO clip (one clip, two ranks: that rank still takes part in the size
    agreement and contributes a zero block) and when a clip has a single frame (nothing propagated: masks None) -- both
    used to fail (device taken from a mask that does not exist; `None.shape`).  Same hashes as one rank; and
    gather_masks refuses a world that is not the group's size instead of hanging in the collective."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for lengths in ([4], [1, 3]):
        out = {}
        for world in (1, 2):
            q = ctx.Queue()
            port = 35500 + os.getpid() % 2000 + world + 10 * len(lengths)
            procs = [ctx.Process(target=_sharded_dataset_worker, args=(r, world, port, q, lengths, 49, 65)) for r in range(world)]
            for p in procs:
                p.start()
            out[world] = q.get(timeout=600)
            for p in procs:
                p.join(timeout=600)
                assert p.exitcode == 0
        assert out[1][0] == out[2][0] and len(out[1][0]) == len(lengths)
        assert sum(out[2][1]) == sum(n - 1 for n in lengths)


def test_batched_clip_driver_validates_its_input_on_the_host():
    """BatchedClipDriver refuses what it cannot run in lockstep before anything is launched (no GPU
    needed): wrong clip count, clips whose lengths give different memory gaps, flip augmentation, more
    objects than one engine holds."""
    from rmem_amd import driver as D
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    cfg = get_config("r50_deaotl", 1, 3)
    model = build_vos_model("deaot", cfg).eval()
    drv = D.BatchedClipDriver(model, 2, cfg)
    img = torch.zeros(1, 3, 33, 49)
    lab = torch.zeros(1, 1, 33, 49)
    clip = lambda n, aug=False, mid=False: [D.make_samples(img, lab if (t == 0 or (mid and t == 2)) else None, (33, 49), 3,
                                                           flip_aug=aug, name=f"{t:05d}.jpg") for t in range(n)]
    with pytest.raises(ValueError):
        drv.run_clips([clip(4)], num_frames=4)
    with pytest.raises(ValueError, match="share the memory gap"):      # evaluator.py:327-331: 4 frames -> gap 5, 200 -> 7
        drv.run_clips([clip(4), clip(200)])
    with pytest.raises(ValueError):
        drv.run_clips([clip(4, aug=True), clip(4, aug=True)], num_frames=4)
    # more objects than one engine holds (meta['obj_num'], or the label map when the meta has none): the
    # reference spawns a sub-engine per 10 ids (engines/aot_engine.py:675-702); the batched engine has none
    many = lambda n: [D.make_samples(img, lab if t == 0 else None, (33, 49), 12, name=f"{t:05d}.jpg") for t in range(n)]
    with pytest.raises(NotImplementedError, match="12 objects"):
        drv.run_clips([clip(4), many(4)], num_frames=4)
    lab13 = lab.clone()
    lab13[0, 0, :4, :4] = 13
    lab13[0, 0, 5:, 5:] = 255                              # the ignore id is not an object
    nometa = [D.make_samples(img, lab13 if t == 0 else None, (33, 49), 3, name=f"{t:05d}.jpg") for t in range(4)]
    for fr in nometa:
        fr[0]["meta"].pop("obj_num")
    with pytest.raises(NotImplementedError, match="13 objects"):
        drv.run_clips([nometa, clip(4)], num_frames=4)


def test_plan_ragged_batches():
    """Which clips of a mixed dataset share a lockstep batch (driver.plan_ragged_batches): groups by (gap from the
    clip's own length -- managers/evaluator.py:327-331 --, network size, original size), longest first, B per batch,
    a remainder of >= 2 padded with -1 slots, everything else (TTA, > 10 objects, a lone clip) to the one-clip driver;
    a clip with mid-clip labels batches like any other (its slot re-references itself); every clip exactly once."""
    from rmem_amd import driver as D
    mk = lambda n, size=(465, 465), ori=(480, 480), **kw: dict(num_frames=n, size=size, ori_size=ori, **kw)
    info = [mk(40), mk(100), mk(165), mk(166),               # gaps 5, 5, 6 (round(5.5) = 6: Python rounds half to even, as the reference does), 6
            mk(60, n_aug=2), mk(70, mid_labels=True), mk(80, obj_num=12),
            mk(90, size=(481, 849), ori=(480, 848)), mk(30), mk(20), mk(50, size=(481, 849), ori=(480, 848)),
            mk(400), mk(10)]
    gaps = [D.memory_gap(c["num_frames"]) for c in info]
    assert gaps[:4] == [5, 5, 6, 6] and gaps[11] == 13
    plan = D.plan_ragged_batches(info, 4)
    seen = sorted([i for b in plan["batches"] for i in b if i >= 0] + plan["singles"])
    assert seen == list(range(len(info)))                     # every clip exactly once
    for b in plan["batches"]:
        real = [i for i in b if i >= 0]
        assert len(b) == 4 and len(real) >= 2 and b[:len(real)] == real       # padding slots last
        assert len({(gaps[i], info[i]["size"], info[i]["ori_size"]) for i in real}) == 1
        lens = [info[i]["num_frames"] for i in real]
        assert lens == sorted(lens, reverse=True)             # longest first: the batch runs for b[0]'s length
    # gap 5 at 465x465: clips 1, 5 (mid-clip labels), 0, 8 fill a batch, 9 and 12 a padded one; gap 6: clips 3, 2 padded;
    # the two 481x849 clips (gap 5 both) padded; TTA / 12 objects / the lone gap-13 clip run alone
    assert sorted(plan["batches"]) == sorted([[1, 5, 0, 8], [9, 12, -1, -1], [3, 2, -1, -1], [7, 10, -1, -1]])
    assert plan["singles"] == [4, 6, 11]
    # a fixed gap removes the length from the key
    plan2 = D.plan_ragged_batches(info, 4, gap_of=lambda n: 3)
    assert any(11 in b for b in plan2["batches"])
    with pytest.raises(ValueError):
        D.plan_ragged_batches([mk(0)], 4)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_multi_engine_wrapper_vs_reference_vectors(which, golden_dir, deaot_model):
    """SURVEY 8f-4 pinned: separate_mask (label branch, 4-d and 3-d masks) and soft_logit_aggregation of
    the > 10-object wrapper against the outputs of the reference's OWN functions
    (engines/aot_engine.py:604-618, 650-673; tests/golden/make_golden.py:gen_multiengine), for 2 and 3
    sub-engines, ids beyond the last engine's range, the ignore id 255 and logits that hit the
    clamp(1e-5, 1 - 1e-5).  Checked for the oracle's wrapper and for the product's (host functions)."""
    import json
    meta = json.load(open(os.path.join(golden_dir, "multiengine_wrapper.json")))
    gold = np.load(os.path.join(golden_dir, "multiengine_wrapper.npz"))
    if which == "oracle":
        from oracle.engine_ref import OracleDeAOTInferEngine
        w = OracleDeAOTInferEngine(deaot_model, long_term_mem_gap=5)
        assert w.max_obj == meta["max_aot_obj_num"]
        attr = "engines"
    else:
        from rmem_amd.engine import DeAOTInferEngine
        w = DeAOTInferEngine(deaot_model, gpu_id=0, long_term_mem_gap=5, fold_bn=False)
        assert w.max_aot_obj_num == meta["max_aot_obj_num"]
        attr = "aot_engines"
    for case in meta["cases"]:
        n = int(case[1:])
        setattr(w, attr, [object()] * n)
        mask = torch.from_numpy(gold[f"{case}_mask"])
        for tag, m in (("sep", mask), ("sep3d", mask[0])):
            got = w.separate_mask(m)
            assert len(got) == n
            for i in range(n):
                assert np.array_equal(got[i].numpy(), gold[f"{case}_{tag}{i}"]), (case, tag, i)
        agg = w.soft_logit_aggregation([torch.from_numpy(gold[f"{case}_logit{i}"]) for i in range(n)])
        assert agg.shape[1] == 1 + n * meta["max_aot_obj_num"]
        assert np.array_equal(agg.numpy(), gold[f"{case}_agg"]), case          # same torch ops, same order: bit-equal
    setattr(w, attr, [object()])                                              # single-engine fast paths: identity
    m1, l1 = torch.zeros(1, 1, 4, 4), torch.zeros(1, 11, 4, 4)
    assert w.separate_mask(m1)[0] is m1 and w.soft_logit_aggregation([l1]) is l1
    setattr(w, attr, [])


def test_uneven_split_chooser_returns_valid_geometries():
    """DeAOTLSTT.choose_uneven (off by default, RMEM_UNEVEN=1): whatever it proposes must be a geometry rmem_attn_read2
    accepts -- at least one short split, the full pieces leave key tiles for the short ones, no empty split -- and the
    windowed units' tile counts it models must be what read64.hip computes (15 x 15 window, clipped rows)."""
    from rmem_amd.lstt import DeAOTLSTT as D
    for (h, w, cap) in ((31, 54, 4), (46, 81, 8), (12, 17, 4), (30, 53, 4), (7, 9, 4)):
        N = h * w
        tv = (N + 63) // 64
        kl, kw = D.choose_splits(N, h, w, cap)
        wt = D.window_unit_tiles(N, h, w, kw)
        assert len(wt) == tv * kw and all(0 <= t <= tv for t in wt)
        # every key tile a query tile can see is covered by exactly its kw units
        for qt in range(tv):
            q_lo, q_hi = qt * 64, min(qt * 64 + 63, N - 1)
            y_lo, y_hi = max(q_lo // w - 7, 0), min(q_hi // w + 7, h - 1)
            assert sum(wt[qt * kw:(qt + 1) * kw]) == ((y_hi + 1) * w + 63) // 64 - (y_lo * w) // 64
        for T in range(1, cap + 2):
            u = D.choose_uneven(N, h, w, T, min(kl, T * tv), kw)
            if u is None:
                continue
            ks, nfull, pf = u
            tiles = T * tv
            assert 0 < nfull < ks <= 16 and pf >= 1 and nfull * pf < tiles
            rest, ns = tiles - nfull * pf, ks - nfull
            pr = -(-rest // ns)
            assert (ns - 1) * pr < rest and pr < pf


def test_slot_queue_plan_serves_every_frame_once_without_idle_slots():
    """plan_slot_queue (the reference's worker queue, managers/evaluator.py:276-295, for the slots of a batch)."""
    from rmem_amd.driver import plan_slot_queue, plan_ragged_batches
    import numpy as np
    rs = np.random.RandomState(0)
    for B in (1, 2, 3, 8):
        for trial in range(20):
            lens = [int(x) for x in rs.randint(1, 40, size=rs.randint(1, 30))]
            for order in ("longest_first", "given"):
                steps = plan_slot_queue(lens, B, order)
                seen = {}
                for k, row in enumerate(steps):
                    assert len(row) == B
                    for s, e in enumerate(row):
                        if e is not None:
                            assert e not in seen
                            seen[e] = (k, s)
                assert sorted(seen) == [(c, t) for c in range(len(lens)) for t in range(lens[c])]
                for c, n in enumerate(lens):               # a clip stays in its slot, one frame per step
                    k0, s0 = seen[(c, 0)]
                    assert all(seen[(c, t)] == (k0 + t, s0) for t in range(n))
                for s in range(B):                          # a slot that went idle never works again (queue empty)
                    col = [row[s] is None for row in steps]
                    assert col == sorted(col)
                if order == "given":                        # clips start in queue order
                    starts = [seen[(c, 0)][0] for c in range(len(lens))]
                    assert starts == sorted(starts)
                # never longer than lockstep batches of the same clips (longest first, B per batch)
                ordered = sorted(lens, reverse=True)
                lockstep = sum(ordered[i] for i in range(0, len(ordered), B))
                if order == "longest_first":
                    assert len(steps) <= lockstep
    assert plan_slot_queue([3], 2) == [[(0, 0), None], [(0, 1), None], [(0, 2), None]]
    import pytest
    with pytest.raises(ValueError):
        plan_slot_queue([3, 0], 2)


def test_wait_event_polls_without_the_runtime_wait(monkeypatch):
    """hip.wait_event: the engine thread's long wait (resolve_policy) polls event.query() with a short sleep instead of
    the runtime's spinning hipEventSynchronize; RMEM_SPIN_WAIT=1 selects the runtime's wait."""
    from rmem_amd import hip

    class Ev:
        def __init__(self, ready_after):
            self.n, self.ready_after, self.synced = 0, ready_after, 0

        def query(self):
            self.n += 1
            return self.n > self.ready_after

        def synchronize(self):
            self.synced += 1

    monkeypatch.delenv("RMEM_SPIN_WAIT", raising=False)
    e = Ev(0)
    hip.wait_event(e)
    assert e.n == 1 and e.synced == 0                # already complete: one query, no wait
    e = Ev(5)
    hip.wait_event(e)
    assert e.n == 6 and e.synced == 0
    monkeypatch.setenv("RMEM_SPIN_WAIT", "1")
    e = Ev(5)
    hip.wait_event(e)
    assert e.synced == 1 and e.n == 1


def test_hash_masks_follows_the_gather_order():
    """driver.hash_masks: sha256 per clip in clip-id order from masks in gather order (rank-major, unshard_order)."""
    import hashlib
    import numpy as np
    import torch
    from rmem_amd import driver as D
    n, world = 6, 3
    rs = np.random.RandomState(0)
    per_clip = [rs.randint(0, 4, (2, 5, 7)).astype(np.uint8) for _ in range(n)]
    gathered = np.stack([per_clip[c] for r in range(world) for c in D.shard_clips(n, world, r)])
    want = [hashlib.sha256(m.tobytes()).hexdigest() for m in per_clip]
    assert D.hash_masks(gathered, n, world) == want
    assert D.hash_masks(torch.from_numpy(gathered), n, world) == want


def test_set_host_wait_rejects_a_device_that_does_not_exist_and_is_opt_in(monkeypatch):
    """rmem_set_host_wait (include/rmem_hip.h): an out-of-range device is refused without touching anything; the Python
    host calls it only when RMEM_BLOCKING_WAIT=1 (two processes sharing a GPU hung in MIOpen under the flag)."""
    import ctypes as C
    from rmem_amd import hip
    lib = hip.load()
    lib.rmem_set_host_wait.argtypes = [C.c_int32, C.c_int32]
    assert lib.rmem_set_host_wait(9999, 1) == -1           # RMEM_ERR_INVALID
    assert lib.rmem_set_host_wait(-1, 1) == -1
    monkeypatch.delenv("RMEM_BLOCKING_WAIT", raising=False)
    assert hip.set_host_wait(0) is False                   # opt-in: nothing is called without the switch


def test_slot_queue_of_nothing_and_of_one_frame_clips():
    from rmem_amd.driver import plan_slot_queue
    assert plan_slot_queue([], 4) == []
    # clips of one frame (a reference frame only): every step restarts slots, nothing is propagated
    assert plan_slot_queue([1, 1, 1], 2) == [[(0, 0), (1, 0)], [(2, 0), None]]


def test_rank_pinning_plan_follows_the_numa_topology():
    """rmem_amd/affinity.py: the ranks whose GPUs hang off one NUMA node split that node's physical cores (SMT siblings
    together) into contiguous equal shares in rank order -- the 8-GPU node of profiles/r04_numa_topo.txt (two sockets of 64
    cores, GPUs 0-2 and 7 on node 0, 3-6 on node 1); unknown topology or too few cores -> no pin."""
    from rmem_amd.affinity import parse_cpulist, plan_node_shares
    assert parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    node0 = [[c, c + 128] for c in range(0, 64)]
    node1 = [[c, c + 128] for c in range(64, 128)]
    shares = plan_node_shares([0, 0, 0, 1, 1, 1, 1, 0], {0: node0, 1: node1})
    assert all(len(s) == 32 for s in shares)                                  # 16 cores x 2 threads each
    assert shares[0][:3] == [0, 1, 2] and shares[0][16:19] == [128, 129, 130]
    assert shares[7][0] == 48 and shares[3][0] == 64 and shares[6][0] == 112
    flat = [c for s in shares for c in s]
    assert len(flat) == len(set(flat)) == 256                                  # disjoint, everything used
    assert plan_node_shares([None, 0], {0: node0}) == [None, sorted(c for g in node0 for c in g)]
    assert plan_node_shares([0, 0, 0], {0: node0[:2]}) == [None, None, None]


def _undefined_names(path):
    """Names a function of the file loads that nothing in the function, the module or builtins binds (what pyflakes calls
    an undefined name) -- legs of bench.py that only run with particular flags get no other check on the CPU."""
    import ast
    import builtins
    tree = ast.parse(open(path).read())

    def bound(node):
        out = set()
        for n in ast.walk(node):
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                out.add(n.id)
            elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                out.add(n.name)
                if not isinstance(n, ast.ClassDef):
                    a = n.args
                    out.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
                    out.update(x.arg for x in (a.vararg, a.kwarg) if x is not None)
            elif isinstance(n, ast.Lambda):
                a = n.args
                out.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
                out.update(x.arg for x in (a.vararg, a.kwarg) if x is not None)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                out.update((al.asname or al.name).split(".")[0] for al in n.names)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                out.add(n.name)
            elif isinstance(n, (ast.Global, ast.Nonlocal)):
                out.update(n.names)
        return out

    top = set()                                  # module scope: what top-level statements bind, not the functions' locals
    for st in tree.body:
        if isinstance(st, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            top.add(st.name)
        else:
            top |= bound(st)
    mod = top | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []
    for fn in [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        have = mod | bound(fn)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in have:
                bad.append((fn.name, n.id, n.lineno))
    return bad


def test_no_undefined_names_in_bench_and_package():
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + \
        sorted(glob.glob(os.path.join(root, "rmem_amd", "*.py"))) + sorted(glob.glob(os.path.join(root, "rmem_amd", "nets", "*.py")))
    bad = {os.path.relpath(f, root): _undefined_names(f) for f in files}
    assert not any(bad.values()), {k: v for k, v in bad.items() if v}
