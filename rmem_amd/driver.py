"""Clip-level driver: the caller side of the engine API.

Restates the per-clip loop of the reference evaluator (managers/evaluator.py:337-568) on top
of ``rmem_amd.engine``: one engine per test-time augmentation, reference frame, then per frame
match -> (un-flip, softmax, mean, argmax) -> label -> (flip, nearest resize) -> update_memory,
mid-clip new objects, the evaluator's memory-gap rule and its CUDA-event timing window, the
palette-PNG writer (utils/image.py:90-105) and the input-size rule of the test transform
(dataloaders/video_transforms.py:575-621).

Post-processing runs on device: when every engine has a single sub-engine (<= 10 objects) the
decoder logits of all augmentations go through ONE kernel (``rmem_labels_from_logits``) that
upsamples, un-flips, averages and arg-maxes straight to a uint8 label map, and one
``rmem_label_resize_nearest`` per engine produces the map fed back to ``update_memory`` -- the
18 MB fp32 probability volume the reference materialises per augmentation never exists.  The
generic path (several sub-engines, whose logits are aggregated at output resolution:
engines/aot_engine.py:650-673,704-712) uses the same torch ops as the reference.

Also here: static clip sharding across ranks (round-robin for equal clips, longest-first for a dataset of
unequal clips: assign_clips_by_length) and the one exchange step (all-gather of masks).
The reference distributes clips through an mp.Queue work queue and funnels statistics through
a second queue (tools/eval.py:137-143, managers/evaluator.py:276-295,589-613); clips are
independent, so the shard is static (clip i -> rank i mod world) and the masks are collected
with one all-gather (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import contextlib
import os
import threading
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ sharding / exchange
def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Clip ids owned by `rank` (round-robin, as balanced as the reference's queue for
    equal-length clips)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_clips, world))


def gather_masks(local_masks: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all-gather uint8 masks [clips_per_rank, F, H, W] -> [world*clips_per_rank, F, H, W]
    ordered by rank.  Every rank must contribute the same shape (pad short shards)."""
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_masks
    # (a one-rank group that IS initialised goes through the collective: bench.py with RMEM_FORCE_DIST=1 runs the
    # whole RCCL path -- communicator, uint8 all-gather, device-side timing exchange -- on a single leased GPU)
    gsize = dist.get_world_size(group)
    if world == 1 and gsize > 1:           # a local run inside a multi-rank job: nothing to exchange
        return local_masks
    if world != gsize:
        raise ValueError(f"gather_masks: world={world} but the process group has {gsize} ranks")
    if local_masks.dtype != torch.uint8:
        raise TypeError("masks are exchanged as uint8 label maps")
    src = local_masks.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":      # (tests: two processes sharing one GPU)
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=torch.uint8, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(local_masks.device)


def run_sharded_clips(driver, n_clips: int, world: int, rank: int, frames_of: Callable[[int], list],
                      num_frames: int, group=None, hashes: bool = True):
    """BASELINE.json configs[3]: `n_clips` independent clips, clip i on rank i mod world
    (tools/eval.py:137-143 and managers/evaluator.py:276-295 hand clips to processes through a queue;
    equal-length clips make the static shard as balanced), each through `driver.run_clip` (reference
    frame and bank fill included), then ONE all-gather of the uint8 masks
    (evaluator.py:589-613 funnels results through a queue instead).  Returns (sha256 per clip in
    clip-id order, gathered masks [n_clips, F-1, H0, W0] in gather order, frames run on this rank).
    The hashes do not depend on `world`: what rank a clip runs on changes nothing it computes.  hashes=False returns
    None in their place (bench.py hashes after its timed window: hash_masks)."""
    if n_clips % world:
        raise ValueError("n_clips must be a multiple of the world size (pad the clip list)")
    local, frames_run = [], 0
    for cid in shard_clips(n_clips, world, rank):
        res = driver.run_clip(frames_of(cid), num_frames=num_frames)
        local.append(res.masks)
        frames_run += int(res.masks.shape[0])
    allm = gather_masks(torch.stack(local), world, group)
    return (hash_masks(allm, n_clips, world) if hashes else None), allm, frames_run


def hash_masks(allm, n_clips: int, world: int) -> List[str]:
    """sha256 per clip, in clip-id order, of gathered masks [n_clips, F-1, H0, W0] (a device tensor in gather order
    -- rank-major, see unshard_order -- or its host copy as a numpy array)."""
    import hashlib
    host = allm.cpu().numpy() if isinstance(allm, torch.Tensor) else allm
    out: List[Optional[str]] = [None] * n_clips
    for pos, cid in enumerate(unshard_order(n_clips, world)):
        out[cid] = hashlib.sha256(host[pos].tobytes()).hexdigest()
    return out


def assign_clips_by_length(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Clips of UNEQUAL length -> ranks, longest first: clips in order of (-frames, id), each to the rank that holds
    the fewest frames so far (ties: the lowest rank).  The reference hands clips to its workers from a queue as they
    become free (managers/evaluator.py:276-295), which for per-frame costs that do not depend on the clip IS this
    greedy rule when the queue is sorted longest first; the frame counts of a dataset are known before the run, so the
    assignment is computed up front, identically on every rank, and no work queue (no control-plane exchange) is
    needed.  Greedy longest-first is within 4/3 - 1/(3 world) of the optimal makespan; round-robin is not bounded."""
    if world <= 0:
        raise ValueError("world must be positive")
    if any(int(n) <= 0 for n in lengths):
        raise ValueError("every clip needs at least one frame")
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for cid in sorted(range(len(lengths)), key=lambda c: (-int(lengths[c]), c)):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(cid)
        load[r] += int(lengths[cid])
    return out


def shard_clips_by_length(lengths: Sequence[int], world: int, rank: int) -> List[int]:
    """Clip ids `rank` runs under assign_clips_by_length, in the order it runs them (longest first)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return assign_clips_by_length(lengths, world)[rank]


def hash_dataset_masks(allm, lengths: Sequence[int], world: int) -> List[str]:
    """sha256 per clip over ITS OWN frames, in clip-id order, from the padded gather of run_sharded_dataset
    (`allm`: numpy [world * clips per rank, max frames - 1, H0, W0] in rank-major order)."""
    import hashlib
    assign = assign_clips_by_length(lengths, world)
    per = max(1, max(len(a) for a in assign))
    out: List[Optional[str]] = [None] * len(lengths)
    for r, ids in enumerate(assign):
        for j, cid in enumerate(ids):
            out[cid] = hashlib.sha256(allm[r * per + j, :int(lengths[cid]) - 1].tobytes()).hexdigest()
    return out


def run_sharded_dataset(driver, lengths: Sequence[int], world: int, rank: int, frames_of: Callable[[int], list],
                        group=None, hashes: bool = True, host_out: Optional[torch.Tensor] = None):
    """Clips of unequal length over `world` ranks: length-aware static assignment (assign_clips_by_length), every
    clip through `driver.run_clip`, ONE all-gather of uint8 masks padded to [max clips per rank, max frames - 1, H0, W0]
    (all clips of a call share the output size; a dataset with several sizes is one call per size).  Returns
    (sha256 per clip over ITS OWN frames, in clip-id order; frames run per rank, as assigned).  Like
    run_sharded_clips the hashes do not depend on `world`.  hashes=False: returns the gathered masks on the host (numpy,
    rank-major, padded) in place of the hashes -- hash_dataset_masks() turns them into the same list later (bench.py
    hashes after its timed window); `host_out`: a (pinned) uint8 tensor of the gathered shape to copy them into."""
    assign = assign_clips_by_length(lengths, world)
    per = max(1, max(len(a) for a in assign))
    fmax = max(int(n) for n in lengths) - 1
    local: List[Optional[torch.Tensor]] = []
    if hasattr(driver, "run_dataset"):
        # BatchedClipDriver: this rank's clips share its B slots through the clip queue (run_dataset -> run_queue: a
        # slot takes the next clip when its clip ends) -- the reference's worker queue twice over, ranks x slots
        clips = [frames_of(cid)[:int(lengths[cid])] for cid in assign[rank]]
        results = driver.run_dataset(clips) if clips else []
    elif isinstance(driver, InFlightClipDriver):
        # this rank's clips in flight on its lanes (any model: the AOT block, Swin, augmentation)
        clips = [frames_of(cid)[:int(lengths[cid])] for cid in assign[rank]]
        results = driver.run_clips(clips) if clips else []
    else:
        results = [driver.run_clip(frames_of(cid), num_frames=int(lengths[cid])) for cid in assign[rank]]
    dev_of_driver = torch.device("cuda", int(driver.gpu_id)) if (getattr(driver, "gpu_id", None) is not None
                                                                 and torch.cuda.is_available()) else torch.device("cpu")
    for cid, res in zip(assign[rank], results):
        n_run = 0 if res.masks is None else int(res.masks.shape[0])       # (a one-frame clip propagates nothing)
        if n_run != int(lengths[cid]) - 1:
            raise ValueError(f"clip {cid}: {n_run + 1} frames run, {int(lengths[cid])} announced")
        local.append(res.masks)
    shape = None
    have = [m for m in local if m is not None]
    for m in have:
        if shape is not None and tuple(m.shape[1:]) != shape:
            raise ValueError("clips of one run_sharded_dataset call must share the output size")
        shape = tuple(m.shape[1:])
    # the device of the exchange is the DRIVER's (a rank without clips -- fewer clips than ranks -- has no mask to take it
    # from, and an RCCL group cannot exchange host tensors); gloo groups exchange on the host
    devs = have[0].device if have else dev_of_driver
    if world > 1:          # a rank without clips still contributes a (zero) block: agree on the size
        import torch.distributed as dist
        hw = torch.tensor(list(shape) if shape else [0, 0], dtype=torch.int64)
        if devs.type == "cuda" and dist.get_backend(group) != "gloo":
            hw = hw.to(devs)
        dist.all_reduce(hw, op=dist.ReduceOp.MAX, group=group)
        shape = tuple(int(v) for v in hw.cpu())
    block = torch.zeros((per, max(fmax, 1)) + tuple(shape or (0, 0)), dtype=torch.uint8, device=devs)
    for j, m in enumerate(local):
        if m is not None:
            block[j, :m.shape[0]] = m
    gathered = gather_masks(block, world, group)
    if host_out is not None and tuple(host_out.shape) == tuple(gathered.shape):
        host_out.copy_(gathered, non_blocking=True)
        if gathered.is_cuda:
            torch.cuda.synchronize()
        allm = host_out.numpy()
    else:
        allm = gathered.cpu().numpy()
    frames_run = [sum(int(lengths[c]) - 1 for c in ids) for ids in assign]
    return (hash_dataset_masks(allm, lengths, world) if hashes else allm), frames_run


def unshard_order(n_clips: int, world: int) -> List[int]:
    """Position in the gathered tensor -> clip id (inverse of shard_clips + rank-major gather),
    assuming n_clips % world == 0."""
    per = n_clips // world
    return [r + world * j for r in range(world) for j in range(per)]


# ------------------------------------------------------------------ evaluator rules
def memory_gap(num_frames: int, no_memory_gap: bool = False) -> int:
    """long_term_mem_gap chosen per clip (managers/evaluator.py:327-331)."""
    gap = max(int(round(num_frames / 30)), 5)
    if no_memory_gap:
        gap = int(round(gap / 4))
    return gap


def restrict_size(h: int, w: int, max_size: Optional[int] = 800, min_size: Optional[int] = None,
                  scale: float = 1.0, align_corners: bool = True, max_stride: int = 16):
    """Network input size for an h x w frame (MultiRestrictSize,
    dataloaders/video_transforms.py:575-621): cap the long (or short) edge, apply the
    multi-scale factor, snap to stride (+1 when align_corners)."""
    if (min_size is not None) and (max_size is not None):
        raise ValueError("give min_size or max_size, not both")
    sc = None
    if min_size is not None:
        short = w if h > w else h
        if short > min_size:
            sc = float(min_size) / short
    else:
        long_edge = h if h > w else w
        if long_edge > max_size:
            sc = float(max_size) / long_edge
    new_h, new_w = (h, w) if sc is None else (sc * h, sc * w)
    new_h, new_w = int(new_h * scale), int(new_w * scale)
    if align_corners:
        if (new_h - 1) % max_stride != 0:
            new_h = int(np.around((new_h - 1) / max_stride) * max_stride + 1)
        if (new_w - 1) % max_stride != 0:
            new_w = int(np.around((new_w - 1) / max_stride) * max_stride + 1)
    else:
        if new_h % max_stride != 0:
            new_h = int(np.around(new_h / max_stride) * max_stride)
        if new_w % max_stride != 0:
            new_w = int(np.around(new_w / max_stride) * max_stride)
    return new_h, new_w


def mask_palette() -> List[int]:
    """The 256-entry palette of the result PNGs (utils/image.py `_palette`): the PASCAL-VOC
    bit-interleave colour map for ids 0..21 with 192 written as 191, a grey ramp above."""
    pal: List[int] = []
    for i in range(256):
        if i >= 22:
            pal += [i, i, i]
            continue
        rgb, c = [0, 0, 0], i
        for j in range(8):
            for ch in range(3):
                rgb[ch] |= ((c >> ch) & 1) << (7 - j)
            c >>= 3
        pal += [191 if v == 192 else v for v in rgb]
    return pal


def _write_png(mask: np.ndarray, path: str, squeeze_idx):
    from PIL import Image
    if squeeze_idx is not None:                                # utils/image.py:91-97
        un = np.zeros_like(mask)
        for idx in range(1, len(squeeze_idx)):
            un += ((mask == idx) * squeeze_idx[idx]).astype(np.uint8)
        mask = un
    im = Image.fromarray(mask).convert("P")
    im.putpalette(mask_palette())
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    im.save(path)


def save_mask(mask: torch.Tensor, path: str, squeeze_idx=None, background: bool = True):
    """Palette PNG of a uint8 label map (utils/image.py:90-105; the reference also writes from
    a background thread).  Returns the thread (or None)."""
    m = mask.detach().cpu().numpy().astype(np.uint8)
    if not background:
        _write_png(m, path, squeeze_idx)
        return None
    th = threading.Thread(target=_write_png, args=(m, path, squeeze_idx))
    th.start()
    return th


# ------------------------------------------------------------------ the clip loop
class ClipResult:
    """masks: uint8 [F-1, H0, W0] (frames 1..F-1) on the engines' device; frame_ms: per-frame
    time of the reference's window (match .. update_memory); names / obj_idx as in the samples."""

    def __init__(self):
        self.masks: Optional[torch.Tensor] = None
        self.frame_ms: List[float] = []
        self.names: List[str] = []
        self.obj_idx = None
        self.gap = None
        self.batched: Optional[bool] = None            # BatchedClipDriver.run_dataset: shared a lockstep batch / ran alone
        self.handed_over_at: Optional[int] = None      # batched test-time augmentation -> per-augmentation engines at this frame
        self.aug_groups: Optional[List[List[int]]] = None   # batched test-time augmentation: augmentation indexes per image size (one batched engine each)

    @property
    def fps(self) -> float:
        return 1e3 * len(self.frame_ms) / max(sum(self.frame_ms), 1e-9)


class ClipDriver:
    """Runs clips through one engine per augmentation (managers/evaluator.py:337-353).

    `samples` for one frame is what the reference's dataloader yields: a list (one entry per
    augmentation) of dicts with 'current_img' [1,3,H,W] float, optionally 'current_label'
    [1,1,H0,W0], and 'meta' = {'flip', 'obj_num', 'height', 'width', 'obj_idx',
    'current_name'} (dataloaders/eval_datasets.py).  All augmentations share the model weights
    (the reference deep-copies the model per augmentation because its LSTT state lives in the
    module, managers/evaluator.py:348; here the state lives in the engine)."""

    def __init__(self, model, cfg=None, gpu_id: int = 0, engine_factory: Optional[Callable] = None,
                 fused_post: Optional[bool] = None, no_memory_gap: Optional[bool] = None,
                 fixed_gap: Optional[int] = None):
        self.model = model
        self.cfg = cfg if cfg is not None else model.cfg
        self.gpu_id = gpu_id
        self.engines: list = []
        self._factory = engine_factory
        self.fused_post = fused_post
        self.no_memory_gap = bool(getattr(self.cfg, "NO_MEMORY_GAP", False)) if no_memory_gap is None \
            else no_memory_gap
        self.fixed_gap = fixed_gap
        self.align_corners = bool(self.cfg.MODEL_ALIGN_CORNERS)
        self._aug_bat: Dict[tuple, object] = {}        # (size group, augmentations in it) -> BatchedDeAOTEngine (see _aug_batched_ok)

    # -- engines
    def _aug_batched_ok(self, samples) -> bool:
        """Test-time augmentation (managers/evaluator.py:337-353: one engine per augmentation -- the flipped copy, the
        multi-scale copies) as the slots of one BatchedDeAOTEngine per image size: the augmented images of one size are
        one encoder / decoder batch and every launch of the memory path serves all of them -- the augmentations of a
        clip are in lockstep by construction.  DeAOT block on the GPU, product engines, at least two augmentations of
        one size (flip, or flip x multi-scale: the flip pair of every scale), <= 10 objects; RMEM_TTA=serial keeps one
        engine after the other."""
        if len(samples) < 2 or self._factory is not None or self.fused_post is False:
            return False
        if os.environ.get("RMEM_TTA", "batched") == "serial" or getattr(self.cfg, "MODEL_VOS", "") != "deaot":
            return False
        if not all(s["current_img"].is_cuda for s in samples) or len(_aug_groups(samples)) == len(samples):
            return False
        n = samples[0]["meta"]["obj_num"]
        n = int(n[0] if isinstance(n, (list, tuple)) else n)
        return n <= int(self.model.max_obj_num)

    def _engine(self, aug_idx: int):
        while len(self.engines) <= aug_idx:
            if self._factory is not None:
                eng = self._factory(self.model)
            else:
                from .engine import build_engine
                eng = build_engine(self.cfg.MODEL_ENGINE, phase="eval", aot_model=self.model,
                                   gpu_id=self.gpu_id,
                                   long_term_mem_gap=getattr(self.cfg, "TEST_LONG_TERM_MEM_GAP", 9999))
            eng.eval()
            self.engines.append(eng)
        return self.engines[aug_idx]

    def _can_fuse(self, engines, sample0) -> bool:
        """Fused device post-processing needs the un-aggregated decoder logits of every
        augmentation: single sub-engine (<= 10 objects) engines on a GPU."""
        if self.fused_post is False or not sample0["current_img"].is_cuda:
            return False
        return all(len(getattr(e, "aot_engines", ())) == 1 for e in engines)

    # -- post-processing, generic path: the reference's torch ops (evaluator.py:424-441)
    @staticmethod
    def _labels_generic(logits: Sequence[torch.Tensor], flips: Sequence[bool]) -> torch.Tensor:
        probs = []
        for lg, fl in zip(logits, flips):
            if fl:
                lg = lg.flip(3)
            probs.append(torch.softmax(lg, dim=1))
        prob = torch.mean(torch.cat(probs, dim=0), dim=0, keepdim=True)
        return torch.argmax(prob, dim=1, keepdim=True).to(torch.uint8)[0, 0].contiguous()

    @staticmethod
    def _resize_generic(label: torch.Tensor, size, flip: bool) -> torch.Tensor:
        lab = label[None, None].float()
        if flip:
            lab = lab.flip(3)
        return F.interpolate(lab, size=size, mode="nearest")

    @torch.no_grad()
    def run_clip(self, frames: Iterable[List[Dict]], num_frames: Optional[int] = None,
                 save_dir: Optional[str] = None, on_frame: Optional[Callable] = None) -> ClipResult:
        """One clip.  `frames` yields the per-frame sample lists; `num_frames` (len of the clip)
        sets the memory gap (evaluator.py:327-331) -- required unless `frames` has a len().
        `on_frame(frame_idx, label_u8, engines)` is called after each propagated frame."""
        gen = self.clip_steps(frames, num_frames, save_dir, on_frame)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    def clip_steps(self, frames: Iterable[List[Dict]], num_frames: Optional[int] = None,
                   save_dir: Optional[str] = None, on_frame: Optional[Callable] = None):
        """run_clip() as a generator: yields after every frame it has ISSUED (nothing is waited for) and returns the
        ClipResult -- what lets one host thread keep several clips in flight, each on a HIP stream of its own
        (InFlightClipDriver).  Every step of one clip must run under the same current stream."""
        if num_frames is None:
            num_frames = len(frames)        # type: ignore[arg-type]
        gap = self.fixed_gap if self.fixed_gap is not None else memory_gap(num_frames, self.no_memory_gap)
        res = ClipResult()
        res.gap = gap
        for e in self.engines:
            e.restart_engine()
        labels_out: List[torch.Tensor] = []
        timers = []
        writers = []
        it = iter(frames)
        from .engine import default_lookahead
        depth = default_lookahead()                       # look-ahead (encoder prefetch)
        ahead = []
        for _ in range(depth):
            f = next(it, None)
            if f is not None:
                ahead.append(f)
        if ahead and self._aug_batched_ok(ahead[0]):
            return self._run_clip_batched_aug(ahead, it, gap, res, save_dir, on_frame)
        return (yield from self._serial_frames(ahead, it, gap, res, save_dir, on_frame, -1, labels_out, timers, writers))

    def _serial_loop(self, *args):
        """_serial_frames() run to its end (the batched-augmentation path hands a clip over mid-way)."""
        gen = self._serial_frames(*args)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    def _serial_frames(self, ahead, it, gap, res, save_dir, on_frame, frame_idx, labels_out, timers, writers):
        """The evaluator's per-frame loop (managers/evaluator.py:384-527) over the frames still in `ahead` / `it`, one
        engine per augmentation; a generator that yields after every frame.  frame_idx = index of the last frame already
        handled (-1: the clip starts here; the batched-augmentation path hands a clip over mid-way when a new label brings
        more objects than one engine holds)."""
        on_cuda = False
        while ahead:
            samples, ahead = ahead[0], ahead[1:]
            f = next(it, None)
            if f is not None:
                ahead.append(f)
            frame_idx += 1
            engines = [self._engine(i) for i in range(len(samples))]
            flips = [bool(s["meta"]["flip"]) for s in samples]
            meta0 = samples[0]["meta"]
            ori_hw = (int(meta0["height"]), int(meta0["width"]))
            obj_nums = [int(v) for v in meta0["obj_num"]] if isinstance(meta0["obj_num"], (list, tuple)) \
                else [int(meta0["obj_num"])]
            on_cuda = samples[0]["current_img"].is_cuda
            if frame_idx == 0:
                res.obj_idx = meta0.get("obj_idx")
                for e, s in zip(engines, samples):
                    e.long_term_mem_gap = gap
                    img = s["current_img"]
                    lab = F.interpolate(s["current_label"].float(), size=img.shape[2:], mode="nearest").int()
                    e.add_reference_frame(img, lab, frame_step=0, obj_nums=obj_nums)
                yield frame_idx
                continue
            if on_cuda:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
            fuse = self._can_fuse(engines, samples[0])
            logits, new_obj_label = [], None
            for i, (e, s, fl) in enumerate(zip(engines, samples, flips)):
                kw = {}
                if getattr(e, "supports_prefetch", False) and ahead and all(len(a) > i for a in ahead):
                    kw["next_img"] = [a[i]["current_img"] for a in ahead]
                lg = e.match_propogate_one_frame(s["current_img"], output_size=None if fuse else ori_hw, **kw)
                logits.append(lg)
                if (not fl) and s.get("current_label") is not None and new_obj_label is None:
                    new_obj_label = s["current_label"].to(lg.device).float()
            if fuse:
                from . import hip
                label = hip.labels_from_logits([lg.contiguous() for lg in logits], flips, ori_hw,
                                               self.align_corners)
            else:
                label = self._labels_generic(logits, flips)
            if new_obj_label is not None:                        # evaluator.py:484-508
                new = new_obj_label[0, 0].to(torch.uint8)
                label = torch.where(new == 0, label, new)
                new_nums = [int(label.max().item())]
                for e, s, fl in zip(engines, samples, flips):
                    cur = self._resize_generic(label, e.input_size_2d, fl)
                    e.add_reference_frame(s["current_img"], cur, obj_nums=new_nums, frame_step=frame_idx)
            else:                                                # evaluator.py:509-523
                for e, fl in zip(engines, flips):
                    if fuse:
                        from . import hip
                        subs = getattr(e, "aot_engines", [])
                        buf = subs[0].label_buffer(e.input_size_2d, label.device) \
                            if len(subs) == 1 and hasattr(subs[0], "label_buffer") else None
                        cur = hip.label_resize_nearest(label, e.input_size_2d, fl, out=buf)[None, None]
                    else:
                        cur = self._resize_generic(label, e.input_size_2d, fl)
                    e.update_memory(cur)
            if on_cuda:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
                timers.append((t0, t1))
            labels_out.append(label)
            if on_frame is not None:
                on_frame(frame_idx, label, engines)
            name = meta0.get("current_name", f"{frame_idx:05d}")
            name = name[0] if isinstance(name, (list, tuple)) else name
            res.names.append(str(name))
            if save_dir is not None:
                writers.append(save_mask(label, os.path.join(save_dir, str(name).split(".")[0] + ".png"),
                                         res.obj_idx))
            yield frame_idx
        if on_cuda:
            # (the clip's own stream -- its engines' side streams have been joined by their last frame --, not the device:
            # other clips may be in flight on streams of their own)
            torch.cuda.current_stream(labels_out[0].device if labels_out else None).synchronize()
            res.frame_ms = [a.elapsed_time(b) for a, b in timers]
        for th in writers:
            if th is not None:
                th.join()
        res.masks = torch.stack(labels_out) if labels_out else None
        return res


class InFlightClipDriver:
    """`lanes` clips in flight on one GPU: one ClipDriver (its engines, their graphs) and one HIP stream per lane, ONE host
    thread that issues the lanes' frames in turn (hipGraph capture is process-global: host threads would collide), a lane
    taking the next clip of the list when its clip ends -- the reference's worker processes sharing a GPU
    (managers/evaluator.py:276-295) without the processes.  A single clip leaves the machine idle between the launches
    of its chain; a second clip fills the gaps: two R50-AOTL clips run at 628 frames/s against 525 for one, R50-DeAOTL
    555-569 against 538, a third lane adds nothing (profiles/r06ab_aot_clips_in_flight.txt).  Works for every model and
    everything ClipDriver.run_clip takes (augmentation, mid-clip objects, more than ten objects); per clip the label
    maps are those of ClipDriver.run_clip."""

    def __init__(self, model, lanes: int = 2, cfg=None, gpu_id: int = 0, **kw):
        self.lanes = [ClipDriver(model, cfg, gpu_id=gpu_id, **kw) for _ in range(max(1, int(lanes)))]
        self.gpu_id = gpu_id
        self._streams: list = []

    def _lane_streams(self, device):
        from .streams import concurrent_stream
        main = torch.cuda.current_stream(device)
        if not self._streams:
            self._streams = [main] + [concurrent_stream(device) for _ in self.lanes[1:]]
        self._streams[0] = main
        return self._streams

    @torch.no_grad()
    def run_clips(self, clips: Sequence[Sequence[List[Dict]]], save_dirs: Optional[Sequence[Optional[str]]] = None,
                  on_frame: Optional[Callable] = None) -> List[ClipResult]:
        """clips[i][t] = the sample list of frame t of clip i (any lengths, sizes, augmentations) -> one ClipResult per
        clip, in order."""
        results: List[Optional[ClipResult]] = [None] * len(clips)
        if not clips:
            return []
        dev = clips[0][0][0]["current_img"].device
        on_cuda = dev.type == "cuda"
        streams = self._lane_streams(dev) if on_cuda else [None] * len(self.lanes)
        main = streams[0]
        nxt = 0
        running: List[Optional[tuple]] = [None] * len(self.lanes)         # (clip id, generator)

        def on(i):
            if not on_cuda or i == 0:
                return contextlib.nullcontext()
            return torch.cuda.stream(streams[i])

        if on_cuda:
            for st in streams[1:]:
                st.wait_stream(main)                      # the clips' tensors were made on the caller's stream
        while True:
            busy = False
            for i, lane in enumerate(self.lanes):
                if running[i] is None and nxt < len(clips):
                    cid, nxt = nxt, nxt + 1
                    sd = save_dirs[cid] if save_dirs is not None else None
                    running[i] = (cid, lane.clip_steps(clips[cid], len(clips[cid]), sd, on_frame))
                if running[i] is None:
                    continue
                busy = True
                cid, gen = running[i]
                with on(i):
                    try:
                        next(gen)
                    except StopIteration as stop:
                        results[cid] = stop.value
                        running[i] = None
            if not busy:
                break
        if on_cuda:
            for st in streams[1:]:
                main.wait_stream(st)
        return results


class _AugEngineView:
    """One augmentation's engine as `on_frame` callbacks see it (the multi-object wrapper's attributes) when the
    augmentations are the slots of one BatchedDeAOTEngine."""

    def __init__(self, bat, i: int):
        from .engine import _SubEngineView
        self.bat, self.i = bat, i
        self.aot_engines = [_SubEngineView(bat, i)]

    long_memories_indexes = property(lambda self: self.bat.long_memories_indexes[self.i])
    input_size_2d = property(lambda self: self.bat.input_size_2d)
    enc_size_2d = property(lambda self: self.bat.enc_size_2d)


def _aug_groups(samples) -> List[List[int]]:
    """The augmentations of a frame grouped by image size, in order of first appearance: [[aug indices], ...].  Flip
    keeps the size, a multi-scale factor changes it (dataloaders/video_transforms.py MultiRestrictSize: for every scale
    the restricted size, then the flipped copy), so the flip pair of every scale is one group."""
    groups: Dict[tuple, List[int]] = {}
    for i, s_ in enumerate(samples):
        groups.setdefault(tuple(s_["current_img"].shape[1:]), []).append(i)
    return list(groups.values())


def _run_clip_batched_aug(self, ahead, it, gap, res, save_dir, on_frame):
    """ClipDriver.run_clip for a clip with test-time augmentation through ONE BatchedDeAOTEngine PER IMAGE SIZE whose
    slots are the augmentations of that size (ClipDriver._aug_batched_ok; flip only: one engine, flip x multi-scale:
    one engine per scale).  Same protocol per frame as the loop of run_clip: logits of every augmentation -> bilinear to
    the original size, un-flip, softmax, mean, argmax (rmem_labels_from_logits takes sources of different sizes) -> per
    augmentation flip + nearest resize -> update_memory; a mid-clip label re-references every slot
    (evaluator.py:484-508)."""
    from . import hip
    from .batched import BatchedDeAOTEngine
    A = len(ahead[0])
    groups = _aug_groups(ahead[0])
    engs = []
    for gi, idx in enumerate(groups):
        key = (gi, len(idx))
        e = self._aug_bat.get(key)
        if e is None:
            e = self._aug_bat[key] = BatchedDeAOTEngine(self.model, len(idx), gpu_id=self.gpu_id,
                                                        long_term_mem_gap=getattr(self.cfg, "TEST_LONG_TERM_MEM_GAP", 9999))
        e.restart_engine()
        e.long_term_mem_gap = gap
        engs.append(e)
    maxo = int(self.model.max_obj_num)
    where = {a: (g, j) for g, idx in enumerate(groups) for j, a in enumerate(idx)}
    views = [_AugEngineView(engs[where[a][0]], where[a][1]) for a in range(A)]   # what on_frame callbacks read per augmentation
    labels_out, timers, writers = [], [], []
    cat = lambda samples: [torch.cat([samples[a]["current_img"] for a in idx]) for idx in groups]
    pending = None                                            # (frame's sample list, its per-group image batches): the prefetched batches
    frame_idx = -1
    while ahead:
        samples, ahead = ahead[0], ahead[1:]
        f = next(it, None)
        if f is not None:
            ahead.append(f)
        frame_idx += 1
        if len(samples) != A or _aug_groups(samples) != groups:
            raise ValueError("the augmentations changed inside a clip")
        flips = [bool(s_["meta"]["flip"]) for s_ in samples]
        meta0 = samples[0]["meta"]
        ori_hw = (int(meta0["height"]), int(meta0["width"]))
        imgs = pending[1] if pending is not None and pending[0] is samples else cat(samples)
        pending = None
        if frame_idx == 0:
            res.obj_idx = meta0.get("obj_idx")
            for e, idx, im in zip(engs, groups, imgs):
                labs = torch.cat([F.interpolate(samples[a]["current_label"].float(), size=im.shape[2:], mode="nearest").int()
                                  for a in idx])
                e.add_reference_frame(im, labs, obj_nums=[maxo] * len(idx), frame_step=0)
            continue
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        nxt = None
        if ahead and len(ahead[0]) == A and _aug_groups(ahead[0]) == groups:
            nxt = cat(ahead[0])
            pending = (ahead[0], nxt)
        lgs = [None] * A
        for g, (e, idx) in enumerate(zip(engs, groups)):
            lg = e.match_propogate_one_frame(imgs[g], output_size=None, next_imgs=None if nxt is None else nxt[g])
            for j, a in enumerate(idx):
                lgs[a] = lg[j:j + 1]
        label = hip.labels_from_logits(lgs, flips, ori_hw, self.align_corners)
        new_obj_label = next((s_["current_label"] for s_, fl in zip(samples, flips)
                              if (not fl) and s_.get("current_label") is not None), None)
        if new_obj_label is not None:                          # evaluator.py:484-508
            new = new_obj_label.to(label.device).float()[0, 0].to(torch.uint8)
            label = torch.where(new == 0, label, new)
            n_new = int(label.max().item())
            if n_new > maxo:
                # More objects than one engine holds (aot_engine.py:675-702 grows sub-engines): the batched engines'
                # slots are the augmentations, so the clip is handed to the per-augmentation multi-object engines AT
                # THIS FRAME -- a mid-clip add_reference_frame re-initialises the memory from the frame anyway
                # (aot_engine.py:241-325), so nothing of the batched engines' state is needed -- and the evaluator's
                # loop continues there (ClipDriver._serial_loop).
                engines = [self._engine(i) for i in range(A)]
                for e, s_, fl in zip(engines, samples, flips):
                    e.restart_engine()
                    e.long_term_mem_gap = gap
                    if hasattr(e, "adopt_frame_steps"):            # the one sub-engine per augmentation that ran so far keeps counting
                        e.adopt_frame_steps([frame_idx])
                    e.add_reference_frame(s_["current_img"],       # (a fresh engine learns its input size from this call)
                                          ClipDriver._resize_generic(label, tuple(s_["current_img"].shape[2:]), fl),
                                          obj_nums=[n_new], frame_step=frame_idx)
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
                timers.append((t0, t1))
                labels_out.append(label)
                if on_frame is not None:
                    on_frame(frame_idx, label, engines)
                name = meta0.get("current_name", f"{frame_idx:05d}")
                name = name[0] if isinstance(name, (list, tuple)) else name
                res.names.append(str(name))
                if save_dir is not None:
                    writers.append(save_mask(label, os.path.join(save_dir, str(name).split(".")[0] + ".png"), res.obj_idx))
                res.handed_over_at = frame_idx
                return self._serial_loop(ahead, it, gap, res, save_dir, on_frame, frame_idx, labels_out, timers, writers)
            for e, idx, im in zip(engs, groups, imgs):
                cur = torch.cat([ClipDriver._resize_generic(label, e.input_size_2d, flips[a]) for a in idx])
                e.add_reference_frame(im, cur, obj_nums=[maxo] * len(idx), frame_step=frame_idx)
        else:                                                  # evaluator.py:509-523
            for e, idx in zip(engs, groups):
                lab_in = e.lstt.label_buffer(*e.input_size_2d)
                for j, a in enumerate(idx):
                    hip.label_resize_nearest(label, e.input_size_2d, flips[a], out=lab_in[j])
                e.update_memory(lab_in)
        t1 = torch.cuda.Event(enable_timing=True)
        t1.record()
        timers.append((t0, t1))
        labels_out.append(label)
        if on_frame is not None:
            on_frame(frame_idx, label, views)
        name = meta0.get("current_name", f"{frame_idx:05d}")
        name = name[0] if isinstance(name, (list, tuple)) else name
        res.names.append(str(name))
        if save_dir is not None:
            writers.append(save_mask(label, os.path.join(save_dir, str(name).split(".")[0] + ".png"), res.obj_idx))
    torch.cuda.synchronize()
    res.frame_ms = [a.elapsed_time(b) for a, b in timers]
    for th in writers:
        if th is not None:
            th.join()
    res.masks = torch.stack(labels_out) if labels_out else None
    res.batched = True
    res.aug_groups = [list(idx) for idx in groups]
    return res


ClipDriver._run_clip_batched_aug = _run_clip_batched_aug


def plan_ragged_batches(clips_info: Sequence[Dict], B: int, no_memory_gap: bool = False,
                        max_obj_num: int = 10, gap_of: Optional[Callable[[int], int]] = None) -> Dict[str, list]:
    """Which clips of a dataset can share a lockstep batch (host logic only, no tensors).

    clips_info[i]: {"num_frames", "size" (network H, W), "ori_size" (H0, W0), "n_aug", "mid_labels" (bool),
    "obj_num"}.  The reference hands clips of any length to its worker processes one at a time
    (managers/evaluator.py:276-295) and picks the memory gap per clip from its length (:327-331); a
    lockstep batch needs one gap schedule, one geometry and one augmentation, so clips are grouped by
    (gap, size, ori_size), longest first inside a group (a shorter clip idles -- its last frame is
    repeated and the output dropped -- until the longest of its batch ends), B per batch.  A group's
    remainder of >= 2 clips runs as a batch padded with repeats of its first clip (-1 marks a padding
    slot); a single left-over clip, and every clip with test-time augmentation or more than max_obj_num
    objects, goes to the one-clip driver (a clip with mid-clip labels batches like any other: its slot turns the
    frame into a reference frame, BatchedDeAOTEngine.add_reference_slots).  `gap_of` (optional) replaces the gap rule (a
    driver with a fixed gap).  Returns {"batches": [[clip ids (or -1)] * B], "singles": [clip ids]}; every
    clip id appears exactly once."""
    B = int(B)
    if B < 1:
        raise ValueError("B must be >= 1")
    groups: Dict[tuple, List[int]] = {}
    singles: List[int] = []
    for i, c in enumerate(clips_info):
        n = int(c["num_frames"])
        if n < 1:
            raise ValueError(f"clip {i} is empty")
        if int(c.get("n_aug", 1)) != 1 or int(c.get("obj_num", 1)) > max_obj_num:
            singles.append(i)
            continue
        gap = gap_of(n) if gap_of is not None else memory_gap(n, no_memory_gap)
        key = (gap, tuple(c["size"]), tuple(c["ori_size"]))
        groups.setdefault(key, []).append(i)
    batches: List[List[int]] = []
    for key in sorted(groups):
        ids = sorted(groups[key], key=lambda i: (-int(clips_info[i]["num_frames"]), i))
        for k in range(0, len(ids), B):
            chunk = ids[k:k + B]
            if len(chunk) == B:
                batches.append(chunk)
            elif len(chunk) >= 2:
                batches.append(chunk + [-1] * (B - len(chunk)))
            else:
                singles.extend(chunk)
    return {"batches": batches, "singles": sorted(singles)}


def plan_slot_queue(lengths: Sequence[int], B: int, order: str = "longest_first") -> List[List[Optional[tuple]]]:
    """Schedule of a clip queue served by B slots (host logic only).  The reference's workers take the next clip of
    the queue when theirs ends (managers/evaluator.py:276-295); here a SLOT of the batch does: steps[k][s] =
    (clip id, frame index) of slot s at step k, or None when the queue has run dry for that slot.  Frame 0 of a clip is
    its reference frame (the slot restarts inside that step), so a slot never idles between clips.
    order: "longest_first" (shortest makespan: the long clips cannot end up alone at the tail) or "given"."""
    B = int(B)
    if B < 1:
        raise ValueError("B must be >= 1")
    if any(int(n) < 1 for n in lengths):
        raise ValueError("empty clip")
    if order not in ("longest_first", "given"):
        raise ValueError("order: longest_first | given")
    ids = list(range(len(lengths)))
    if order == "longest_first":
        ids.sort(key=lambda i: (-int(lengths[i]), i))
    slot_clip: List[Optional[int]] = [None] * B
    pos = [0] * B
    steps: List[List[Optional[tuple]]] = []
    while True:
        row: List[Optional[tuple]] = []
        for s_ in range(B):
            if slot_clip[s_] is None or pos[s_] >= int(lengths[slot_clip[s_]]):
                slot_clip[s_] = ids.pop(0) if ids else None
                pos[s_] = 0
            if slot_clip[s_] is None:
                row.append(None)
            else:
                row.append((slot_clip[s_], pos[s_]))
                pos[s_] += 1
        if all(e is None for e in row):
            return steps
        steps.append(row)


class BatchedClipDriver:
    """B clips of one frame size and one memory-gap schedule in lockstep through
    rmem_amd.batched.BatchedDeAOTEngine: ONE launch per kernel of the memory path for all clips,
    encoder / decoder at batch B (BASELINE.json configs[3] runs 8 such clips per GPU).
    Per clip the protocol is ClipDriver.run_clip's for one augmentation
    (managers/evaluator.py:344-523): gap rule, reference frame, per frame decoder logits -> label map at
    the original size -> nearest resize to the network size -> update_memory.  run_clips(): B clips in
    lockstep (lengths may differ as long as the gap rule gives them the same gap); run_queue(): any number
    of clips of one frame size through the B slots, a slot taking the next clip when its clip ends, every
    clip on its own gap schedule; run_dataset(): any list of clips -- one queue per frame size, the rest
    (augmentation, > 10 objects, a lone clip) through the one-clip driver.  A frame that brings a label with
    new objects (managers/evaluator.py:484-508) is handled per slot in all three: the new ids go over the
    prediction and the frame becomes that slot's reference frame while the other slots update as usual."""

    def __init__(self, model, B: int, cfg=None, gpu_id: int = 0, no_memory_gap: Optional[bool] = None,
                 fixed_gap: Optional[int] = None):
        from .batched import BatchedDeAOTEngine
        self.cfg = cfg if cfg is not None else model.cfg
        self.model, self.gpu_id = model, gpu_id
        self.B = int(B)
        self.engine = BatchedDeAOTEngine(model, self.B, gpu_id=gpu_id,
                                         long_term_mem_gap=getattr(self.cfg, "TEST_LONG_TERM_MEM_GAP", 9999))
        self.no_memory_gap = bool(getattr(self.cfg, "NO_MEMORY_GAP", False)) if no_memory_gap is None \
            else no_memory_gap
        self.fixed_gap = fixed_gap
        self.align_corners = bool(self.cfg.MODEL_ALIGN_CORNERS)
        self._single: Optional[ClipDriver] = None

    def _gap_of(self, n: int) -> int:
        return self.fixed_gap if self.fixed_gap is not None else memory_gap(n, self.no_memory_gap)

    @torch.no_grad()
    def run_clips(self, clips: Sequence[Sequence[List[Dict]]], num_frames: Optional[int] = None) -> List[ClipResult]:
        """clips[i][t] = the sample list (one augmentation) of frame t of clip i.  Lengths may differ when
        the gap rule gives every clip the same gap: the batch runs for the longest clip, a finished clip
        is fed its last frame again and those outputs are dropped (clips are independent: what the other
        slots compute changes nothing a clip computes).  `num_frames` (optional) cuts every clip to that
        many frames."""
        from . import hip
        B, eng = self.B, self.engine
        if len(clips) != B:
            raise ValueError(f"{B} clips per batch")
        lens = [len(c) if num_frames is None else min(len(c), int(num_frames)) for c in clips]
        if min(lens) < 1:
            raise ValueError("empty clip")
        if any(len(c[0]) != 1 for c in clips):
            raise ValueError("batched clips take one augmentation: use ClipDriver for test-time augmentation")
        gaps = [self._gap_of(n) for n in lens]
        if any(g != gaps[0] for g in gaps):
            raise ValueError(f"batched clips must share the memory gap (lengths {lens} give gaps {gaps}): group them with "
                             "plan_ragged_batches / run_dataset")
        gap, nmax = gaps[0], max(lens)
        eng.restart_engine()
        eng.long_term_mem_gap = gap
        meta = [c[0][0]["meta"] for c in clips]
        ori_hw = (int(meta[0]["height"]), int(meta[0]["width"]))
        if any((int(m["height"]), int(m["width"])) != ori_hw for m in meta):
            raise ValueError("batched clips must share the frame size")
        stack = lambda t: torch.cat([c[min(t, n - 1)][0]["current_img"] for c, n in zip(clips, lens)])
        imgs = stack(0)
        labs = torch.cat([F.interpolate(c[0][0]["current_label"].float(), size=imgs.shape[2:], mode="nearest")
                          for c in clips]).int()
        # Object count per clip (meta['obj_num'], else the label map's largest id): a clip with more than
        # max_obj_num objects needs one sub-engine per 10 ids (engines/aot_engine.py:675-702) -- the batched
        # engine has none, and rmem_id_assign drops ids above max_obj_num, so such a clip must not run here.
        maxo = int(eng.AOT.max_obj_num)
        for i, m in enumerate(meta):
            n = m.get("obj_num")
            n = int(n[0] if isinstance(n, (list, tuple)) else n) if n is not None else None
            if n is None:
                ids = labs[i][labs[i] != 255]
                n = int(ids.max().item()) if ids.numel() else 0
            if n > maxo:
                raise NotImplementedError(f"clip {i} holds {n} objects (> {maxo}): use ClipDriver (one sub-engine per "
                                          f"{maxo} objects, engines/aot_engine.py:675-702)")
        # the per-clip engine wrapper hands max_obj_num to its engine whatever the clip holds
        # (engines/aot_engine.py:675-690): same here
        eng.add_reference_frame(imgs, labs, obj_nums=[maxo] * B, frame_step=0)
        lab_in = eng.lstt.label_buffer(*eng.input_size_2d)
        out = torch.zeros(B, max(nmax - 1, 1), ori_hw[0], ori_hw[1], dtype=torch.uint8, device=imgs.device)
        nxt = stack(1) if nmax > 1 else None
        for t in range(1, nmax):
            cur, nxt = nxt, (stack(t + 1) if t + 1 < nmax else None)
            logit = eng.match_propogate_one_frame(cur, output_size=None, next_imgs=nxt)
            refs = {}
            for i in range(B):
                # (a finished clip's slot keeps running on its last frame; its rows of `out` past the clip's end are cut below)
                lab = hip.labels_from_logits([logit[i:i + 1]], [False], ori_hw, self.align_corners, out=out[i, t - 1])
                new = clips[i][t][0].get("current_label") if t < lens[i] else None
                if new is not None:
                    lab = self._paste_new_objects(lab, new, i)
                    refs[i] = (lab_in[i], maxo)
                hip.label_resize_nearest(lab, eng.input_size_2d, False, out=lab_in[i])
            eng.add_reference_slots(refs)            # (evaluator.py:484-508: new objects -> the frame is a reference frame)
            eng.update_memory(lab_in)
        results = []
        for i, c in enumerate(clips):
            r = ClipResult()
            r.gap, r.masks, r.obj_idx = gap, out[i, :lens[i] - 1], meta[i].get("obj_idx")
            r.names = [str(s[0]["meta"].get("current_name", "")) for s in c[1:lens[i]]]
            results.append(r)
        return results

    def _paste_new_objects(self, lab: torch.Tensor, new: torch.Tensor, i: int) -> torch.Tensor:
        """managers/evaluator.py:484-496: the ids of a mid-clip label (a map at the original size, 0 where it says nothing)
        go over the prediction, in place.  More ids than one engine holds need the one-clip driver's sub-engines."""
        new = new.to(lab.device)[0, 0].to(torch.uint8)
        if tuple(new.shape) != tuple(lab.shape):
            raise ValueError(f"clip {i}: a mid-clip label must have the frames' original size {tuple(lab.shape)}")
        lab.copy_(torch.where(new == 0, lab, new))
        maxo = int(self.engine.AOT.max_obj_num)
        ids = lab[lab != 255]
        n = int(ids.max().item()) if ids.numel() else 0
        if n > maxo:
            raise NotImplementedError(f"clip {i} grows to {n} objects (> {maxo}) with a mid-clip label: use ClipDriver (one "
                                      f"sub-engine per {maxo} objects, engines/aot_engine.py:675-702)")
        return lab

    def _check_objects(self, clip, lab_net: torch.Tensor, i: int) -> None:
        """A clip with more than max_obj_num objects needs one sub-engine per 10 ids (engines/aot_engine.py:675-702):
        the batched engine has none, and rmem_id_assign drops ids above max_obj_num."""
        maxo = int(self.engine.AOT.max_obj_num)
        n = clip[0][0]["meta"].get("obj_num")
        n = int(n[0] if isinstance(n, (list, tuple)) else n) if n is not None else None
        if n is None:
            ids = lab_net[lab_net != 255]
            n = int(ids.max().item()) if ids.numel() else 0
        if n > maxo:
            raise NotImplementedError(f"clip {i} holds {n} objects (> {maxo}): use ClipDriver (one sub-engine per "
                                      f"{maxo} objects, engines/aot_engine.py:675-702)")

    @torch.no_grad()
    def run_queue(self, clips: Sequence[Sequence[List[Dict]]], order: str = "longest_first") -> List[ClipResult]:
        """Any number of clips of ONE frame size through the B slots of the batch, each slot taking the next clip of
        the queue the moment its clip ends (plan_slot_queue; the reference's worker queue,
        managers/evaluator.py:276-295).  Lengths and memory gaps may differ per clip: every slot follows its own
        clip's schedule, and the launches of the memory path are shared by the slots that are in the same state
        (BatchedLSTT._run groups them) -- a slot that has just restarted (reference frame, bank still filling) costs a
        few extra launches, not an idle slot until the longest clip of a lockstep batch ends.  Per clip the results
        are those of run_clips / ClipDriver.run_clip: nothing a clip computes depends on its neighbours."""
        from . import hip
        B, eng = self.B, self.engine
        if not clips:
            return []
        if any(len(c) < 1 for c in clips):
            raise ValueError("empty clip")
        if any(len(c[0]) != 1 for c in clips):
            raise ValueError("batched clips take one augmentation: use ClipDriver for test-time augmentation")
        meta = [c[0][0]["meta"] for c in clips]
        ori_hw = (int(meta[0]["height"]), int(meta[0]["width"]))
        size = tuple(clips[0][0][0]["current_img"].shape[2:])
        if any((int(m["height"]), int(m["width"])) != ori_hw for m in meta) or \
                any(tuple(c[0][0]["current_img"].shape[2:]) != size for c in clips):
            raise ValueError("the clips of a queue must share the frame size (run_dataset groups them)")
        lens = [len(c) for c in clips]
        gaps = [self._gap_of(n) for n in lens]
        maxo = int(eng.AOT.max_obj_num)
        # reference labels at the network size (nearest, as run_clips) and the object-count check, before the loop
        ref_lab = []
        for i, c in enumerate(clips):
            lab = F.interpolate(c[0][0]["current_label"].float(), size=size, mode="nearest").to(torch.uint8)[0, 0]
            self._check_objects(c, lab, i)
            ref_lab.append(lab)
        steps = plan_slot_queue(lens, B, order)
        dev = clips[0][0][0]["current_img"].device
        out = [torch.zeros(max(n - 1, 0), ori_hw[0], ori_hw[1], dtype=torch.uint8, device=dev) for n in lens]
        filler = clips[steps[0][0][0]][0][0]["current_img"]        # an idle slot still holds a row of the encoder batch

        def stack(row):
            return torch.cat([filler if e is None else clips[e[0]][e[1]][0]["current_img"] for e in row])

        eng.restart_engine()
        lab_in = None
        nxt = stack(steps[0])
        self.queue_stats = {"steps": len(steps), "slot_steps": B * len(steps),
                            "busy_slot_steps": sum(e is not None for row in steps for e in row), "launch_groups": 0}
        for k, row in enumerate(steps):
            cur, nxt = nxt, (stack(steps[k + 1]) if k + 1 < len(steps) else None)
            ref = {s_: (ref_lab[e[0]], maxo, gaps[e[0]]) for s_, e in enumerate(row) if e is not None and e[1] == 0}
            idle = [s_ for s_, e in enumerate(row) if e is None]
            logit = eng.match_propogate_one_frame(cur, output_size=None, next_imgs=nxt, ref_slots=ref, idle_slots=idle)
            self.queue_stats["launch_groups"] += eng.lstt.groups_last
            if lab_in is None:
                lab_in = eng.lstt.label_buffer(*eng.input_size_2d)
            refs = {}
            for s_, e in enumerate(row):
                if e is None or e[1] == 0:
                    continue
                lab = hip.labels_from_logits([logit[s_:s_ + 1]], [False], ori_hw, self.align_corners, out=out[e[0]][e[1] - 1])
                new = clips[e[0]][e[1]][0].get("current_label")
                if new is not None:
                    lab = self._paste_new_objects(lab, new, e[0])
                    refs[s_] = (lab_in[s_], maxo)
                hip.label_resize_nearest(lab, eng.input_size_2d, False, out=lab_in[s_])
            # a frame that brought new objects becomes its slot's reference frame (evaluator.py:484-508), the others update
            eng.add_reference_slots(refs)
            eng.update_memory(lab_in)
        results = []
        for i, c in enumerate(clips):
            r = ClipResult()
            r.gap, r.masks, r.obj_idx = gaps[i], out[i], meta[i].get("obj_idx")
            r.names = [str(s[0]["meta"].get("current_name", "")) for s in c[1:]]
            r.batched = True
            results.append(r)
        return results

    @staticmethod
    def clip_info(clip: Sequence[List[Dict]]) -> Dict:
        """The facts plan_ragged_batches() needs about one clip (see there)."""
        s0 = clip[0][0]
        m = s0["meta"]
        n = m.get("obj_num")
        n = int(n[0] if isinstance(n, (list, tuple)) else n) if n is not None else 1
        return {"num_frames": len(clip), "size": tuple(int(x) for x in s0["current_img"].shape[2:]),
                "ori_size": (int(m["height"]), int(m["width"])), "n_aug": len(clip[0]),
                "mid_labels": any(s[0].get("current_label") is not None for s in clip[1:]), "obj_num": n}

    @torch.no_grad()
    def run_dataset(self, clips: Sequence[Sequence[List[Dict]]], mode: str = "queue") -> List[ClipResult]:
        """Any list of clips -> one ClipResult per clip, in order.

        mode "queue" (default): the clips of one frame size share the B slots through run_queue() -- a slot takes
        the next clip when its clip ends, whatever the lengths and gaps.  mode "lockstep": plan_ragged_batches()
        groups clips by (gap, size) into batches that run for their longest clip (run_clips).  Either way test-time
        augmentation and > 10 objects run through ClipDriver.run_clip -- the reference's one-clip loop
        (mode "lockstep": a lone clip of its group too).  ClipResult.batched tells which way a clip went."""
        if mode not in ("queue", "lockstep"):
            raise ValueError("mode: queue | lockstep")
        info = [self.clip_info(c) for c in clips]
        results: List[Optional[ClipResult]] = [None] * len(clips)
        singles: List[int] = []
        if mode == "lockstep":
            plan = plan_ragged_batches(info, self.B, max_obj_num=int(self.engine.AOT.max_obj_num), gap_of=self._gap_of)
            for batch in plan["batches"]:
                real = [i for i in batch if i >= 0]
                res = self.run_clips([clips[i if i >= 0 else real[0]] for i in batch])
                for slot, i in enumerate(batch):
                    if i >= 0:
                        res[slot].batched = True
                        results[i] = res[slot]
            singles = plan["singles"]
        else:
            maxo = int(self.engine.AOT.max_obj_num)
            groups: Dict[tuple, List[int]] = {}
            for i, c in enumerate(info):
                if int(c.get("n_aug", 1)) != 1 or int(c.get("obj_num", 1)) > maxo:
                    singles.append(i)
                else:
                    groups.setdefault((tuple(c["size"]), tuple(c["ori_size"])), []).append(i)
            for key in sorted(groups):
                ids = groups[key]
                # (a LONE clip of its size also goes through the slot queue, the other slots idle: a clip's label maps
                # must not depend on how many clips of its size the rank happens to hold -- the one-clip engine splits
                # the projections' K range differently from the recorded launches, so a near-tie pixel could differ and
                # with it the clip hash that run_sharded_dataset promises to be independent of the world size)
                for i, r in zip(ids, self.run_queue([clips[i] for i in ids])):
                    results[i] = r
        if singles:
            if self._single is None:
                self._single = ClipDriver(self.model, self.cfg, gpu_id=self.gpu_id, no_memory_gap=self.no_memory_gap,
                                          fixed_gap=self.fixed_gap)
            for i in sorted(singles):
                r = self._single.run_clip(clips[i], num_frames=len(clips[i]))
                r.batched = False
                results[i] = r
        return results


def make_samples(img: torch.Tensor, label: Optional[torch.Tensor], ori_hw, obj_num: int, flip_aug: bool = False,
                 name: str = "", obj_idx=None, scaled_imgs: Sequence[torch.Tensor] = ()) -> List[Dict]:
    """Sample list for one frame from a network-sized image [1,3,H,W] (and an original-sized
    label [1,1,H0,W0] or None), with the flipped copy when `flip_aug`; `scaled_imgs` = the same frame at the further
    network sizes of multi-scale testing (restrict_size(..., scale=s)), each followed by its flipped copy -- the order
    MultiRestrictSize emits (dataloaders/video_transforms.py:563-652: per scale, the copy, then its flip)."""
    out = []
    for im in [img] + list(scaled_imgs):
        for fl in ([False, True] if flip_aug else [False]):
            s = {"current_img": im.flip(3) if fl else im,
                 "meta": {"flip": fl, "obj_num": [obj_num], "height": int(ori_hw[0]), "width": int(ori_hw[1]),
                          "obj_idx": obj_idx, "current_name": name}}
            if label is not None:
                s["current_label"] = label.flip(3) if fl else label
            out.append(s)
    return out
