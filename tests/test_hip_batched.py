"""GPU: several clips per launch (rmem_amd.batched).

The reference has nothing to compare with here -- its attention asserts batch 1
(aot_plus/networks/layers/transformer.py:641,1190) -- so the statement is: B clips served by shared
launches give, per clip, exactly what B single-clip runs give (bit for bit on the LSTT, whose
kernels are the same code reading its arguments from device memory; within MIOpen's
batch-size-dependent rounding on whole frames), and the single-clip path is pinned to the oracle
and the reference's golden vectors elsewhere (tests/test_hip_engine.py)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(former=1, latter=3):
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights
    cfg = get_config("r50_deaotl", former, latter)
    model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(model)
    return cfg, copy.deepcopy(model).to(DEV)


@pytest.mark.parametrize("h,w,B", [(12, 17, 3), (20, 23, 4), (31, 54, 2)])
def test_batched_lstt_equals_single_clips_bit_for_bit(h, w, B):
    """Reference frame + 9 frames with long-term updates every second frame (K = 4: evictions from
    frame 6 on), different token features and label maps per clip.  Every frame: LSTT output, the
    attention mass, the eviction decision and the bank map of clip i from the shared launches ==
    those of a single-clip DeAOTLSTT with the same key splits."""
    from rmem_amd.batched import BatchedLSTT
    from rmem_amd.lstt import DeAOTLSTT
    cfg, model = _model()
    N = h * w
    H, W = (h - 1) * 16 + 1, (w - 1) * 16 + 1
    bat = BatchedLSTT(model, h, w, DEV, B)
    singles = [DeAOTLSTT(model, h, w, DEV, 3, clips_per_launch=B) for _ in range(B)]
    assert (bat.clips[0].ks_long, bat.clips[0].ks_win, bat.clips[0].ks_self) == \
        (singles[0].ks_long, singles[0].ks_win, singles[0].ks_self)
    rs = np.random.RandomState(1)
    idx_b = [[0] for _ in range(B)]
    idx_s = [[0] for _ in range(B)]
    drops = []
    for t in range(10):
        emb = torch.from_numpy(rs.standard_normal((B, N, 256)).astype(np.float32)).to(DEV)
        lab = torch.from_numpy(rs.randint(0, 4, (B, 1, H // 8 + 1, W // 8 + 1)).astype(np.float32))
        lab = F.interpolate(lab, size=(H, W), mode="nearest")[:, 0].to(torch.uint8).to(DEV).contiguous()
        fg = torch.from_numpy(rs.uniform(0.05, 1.0, (B, N)).astype(np.float32)).to(DEV)
        if t == 0:
            bat.assign_identity(lab, ignore=False)
            out = bat.forward(emb, ref_frame=True)
            for i, s in enumerate(singles):
                s.assign_identity(lab[i], ignore=False)
                so = s.forward(emb[i], ref_frame=True)
                assert torch.equal(out[i], so), (t, i)
            continue
        launches0 = bat.launches
        out = bat.forward(emb)
        n_fwd = bat.launches - launches0
        upd = (t % 2 == 0)
        bat.assign_identity(lab, ignore=True)
        bat.update_short_memories(upd)
        if upd:
            for i in range(B):
                idx_b[i].append(t)
            db = bat.restrict_long_memories(idx_b, fg, wait=True)     # (device-side rule: wait for the decisions)
        for i, s in enumerate(singles):
            so = s.forward(emb[i])
            assert torch.equal(out[i], so), (t, i, (out[i] - so).abs().max().item())
            T = s.mass_T
            assert torch.equal(bat.clips[i].mass.flatten()[:N * T], s.mass.flatten()[:N * T]), (t, i)
            s.assign_identity(lab[i], ignore=True)
            s.update_short_memories(upd)
            if upd:
                idx_s[i].append(t)
                ds = s.restrict_long_memories(idx_s[i], fg[i])
                assert ds == db[i] and idx_s[i] == idx_b[i], (t, i, ds, db[i])
                drops.append(ds)
            assert bat.clips[i].bank == s.bank and bat.clips[i].short == s.short
            for l in range(s.L):
                assert torch.equal(bat.clips[i].bankV[l].hi[s.cur], s.bankV[l].hi[s.cur]), (t, i, l)
                assert torch.equal(bat.clips[i].bankK[l].lo[s.cur], s.bankK[l].lo[s.cur]), (t, i, l)
    assert any(d is not None for d in drops), "no eviction happened"
    print(f"B={B} {h}x{w}: {n_fwd} launches per batched forward pass (every one serves {B} clips); drops {drops}")


def test_batched_engine_vs_single_engines_small_clips():
    """BatchedDeAOTEngine (encoder, LSTT, decoder at batch 3) against three single-clip engines on
    97x129 clips, teacher-forced with the single engines' label maps: eviction sequences equal,
    label maps equal up to MIOpen's batch-size-dependent rounding (<= 2 pixels of 12.5k)."""
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.engine import DeAOTEngine
    from rmem_amd.synth import synth_clip
    cfg, model = _model()
    model.optimize_for_inference(True)
    B, frames, Hh, Ww = 3, 9, 97, 129
    clips = [synth_clip(200 + i, frames, Hh, Ww, 3) for i in range(B)]
    objs = [3, 2, 3]
    labs = []
    for i in range(B):
        lab = clips[i][1].clone()
        lab[lab > objs[i]] = 0
        labs.append(lab)
    singles = []
    for i in range(B):
        # the per-clip engine itself (the multi-object wrapper would pass obj_nums = [10] whatever the clip holds)
        e = DeAOTEngine(model, 0, long_term_mem_gap=2)
        e.eval()
        e.add_reference_frame(clips[i][0][0].to(DEV), labs[i].to(DEV), obj_nums=[objs[i]], frame_step=0)
        singles.append(e)
    bat = BatchedDeAOTEngine(model, B, long_term_mem_gap=2)
    imgs0 = torch.cat([clips[i][0][0] for i in range(B)]).to(DEV)
    bat.add_reference_frame(imgs0, torch.cat(labs).to(DEV), obj_nums=objs, frame_step=0)
    worst = 0
    for t in range(1, frames):
        imgs = torch.cat([clips[i][0][t] for i in range(B)]).to(DEV)
        lg_b = bat.match_propogate_one_frame(imgs, output_size=(Hh, Ww))
        lab_b = lg_b.argmax(1)
        fed = []
        for i, e in enumerate(singles):
            lg = e.match_propogate_one_frame(clips[i][0][t].to(DEV), output_size=(Hh, Ww))
            lab = lg.argmax(1)
            mism = int((lab[0] != lab_b[i]).sum())
            worst = max(worst, mism)
            assert mism <= 2, (t, i, mism)
            assert float((lg[0, :objs[i] + 1] - lg_b[i, :objs[i] + 1]).abs().max()) < 2e-3, (t, i)
            cur = F.interpolate(lab[None].float(), size=e.input_size_2d, mode="nearest")
            e.update_short_term_memory(cur)
            fed.append(cur)
        bat.update_memory(torch.cat(fed))
        for i, e in enumerate(singles):
            assert bat.long_memories_indexes[i] == list(e.long_memories_indexes), (t, i)
    assert len(bat.long_memories_indexes[0]) == cfg.mem_cap
    print("batched engine vs single engines: worst label mismatch per frame", worst)


@pytest.mark.parametrize("policy", ["device", "host"])
def test_batched_slots_in_different_states_bit_for_bit(policy, monkeypatch):
    """Slots of one batch in DIFFERENT states (VERDICT r3 missing #4: a finished slot takes the next clip while the
    others carry on): slot 0 runs one clip of 13 frames with a long-term update every second frame; slot 1 runs a
    clip of 5 frames, then restarts (reference frame at step 5) on a clip that updates its bank EVERY frame; slot 2 is
    idle for two steps, starts at step 2 and goes idle again after step 9.  BatchedLSTT groups the recordings by
    signature -- the slots in the same state share a launch, the others get their own -- and every slot's LSTT output,
    attention mass, eviction decision and bank must equal, bit for bit, a single-clip DeAOTLSTT fed the same
    sequence.  policy: the clips' eviction rule on the device (default: rmem_bank_policy_step per clip, no D2H) or on
    the host (RMEM_HOST_POLICY=1); the single-clip comparison runs the host rule either way."""
    from rmem_amd.batched import BatchedLSTT
    from rmem_amd.lstt import DeAOTLSTT
    if policy == "host":
        monkeypatch.setenv("RMEM_HOST_POLICY", "1")
    cfg, model = _model()
    h, w, B = 12, 17, 3
    N = h * w
    H, W = (h - 1) * 16 + 1, (w - 1) * 16 + 1
    bat = BatchedLSTT(model, h, w, DEV, B)
    assert bat.device_policy == (policy == "device") and all(c._dev_policy == bat.device_policy for c in bat.clips)
    singles = [DeAOTLSTT(model, h, w, DEV, 3, clips_per_launch=B) for _ in range(B)]
    assert not singles[0]._dev_policy
    # per slot: step -> ("ref" | "prop" | None, update_long)
    def plan(slot, t):
        if slot == 0:
            return ("ref", False) if t == 0 else ("prop", t % 2 == 0)
        if slot == 1:
            if t < 5:
                return ("ref", False) if t == 0 else ("prop", t % 2 == 0)
            return ("ref", False) if t == 5 else ("prop", True)
        if t < 2 or t > 9:
            return (None, False)
        return ("ref", False) if t == 2 else ("prop", t % 3 == 0)
    rs = np.random.RandomState(3)
    idx_b = [[] for _ in range(B)]
    idx_s = [[] for _ in range(B)]
    step_of = [0] * B
    groups_seen, drops = set(), []
    for t in range(13):
        emb = torch.from_numpy(rs.standard_normal((B, N, 256)).astype(np.float32)).to(DEV)
        lab = torch.from_numpy(rs.randint(0, 4, (B, 1, H // 8 + 1, W // 8 + 1)).astype(np.float32))
        lab = F.interpolate(lab, size=(H, W), mode="nearest")[:, 0].to(torch.uint8).to(DEV).contiguous()
        fg = torch.from_numpy(rs.uniform(0.05, 1.0, (B, N)).astype(np.float32)).to(DEV)
        kind = [plan(i, t)[0] for i in range(B)]
        upd = [plan(i, t)[1] for i in range(B)]
        on = [k is not None for k in kind]
        ref = [k == "ref" for k in kind]
        prop = [k == "prop" for k in kind]
        for i in range(B):
            if ref[i]:
                bat.clear_memory(i)
                singles[i].clear_memory()
                idx_b[i], idx_s[i], step_of[i] = [0], [0], 0
            elif prop[i]:
                step_of[i] += 1
        if any(ref):
            bat.assign_identity(lab, ignore=False, active=ref)
        out = bat.forward(emb, ref_frame=ref, active=on)
        groups_seen.add(bat.groups_last)
        states = {(kind[i], bat.clips[i]._T) for i in range(B) if on[i]}
        assert bat.groups_last == len(states), (t, bat.groups_last, states)
        for i, s in enumerate(singles):
            if not on[i]:
                continue
            if ref[i]:
                s.assign_identity(lab[i], ignore=False)
            so = s.forward(emb[i], ref_frame=ref[i])
            assert torch.equal(out[i], so), (t, i, (out[i] - so).abs().max().item())
        if any(prop):
            bat.assign_identity(lab, ignore=True, active=prop)
            bat.update_short_memories(upd, active=prop)
            do = [prop[i] and upd[i] for i in range(B)]
            for i in range(B):
                if do[i]:
                    idx_b[i].append(step_of[i])
            db = bat.restrict_long_memories(idx_b, fg, active=do, wait=True) if any(do) else [None] * B
            for i, s in enumerate(singles):
                if not prop[i]:
                    continue
                T = s.mass_T
                assert torch.equal(bat.clips[i].mass.flatten()[:N * T], s.mass.flatten()[:N * T]), (t, i)
                s.assign_identity(lab[i], ignore=True)
                s.update_short_memories(upd[i])
                if upd[i]:
                    idx_s[i].append(step_of[i])
                    ds = s.restrict_long_memories(idx_s[i], fg[i])
                    assert ds == db[i] and idx_s[i] == idx_b[i], (t, i, ds, db[i])
                    drops.append((i, ds))
        for i, s in enumerate(singles):
            if on[i]:
                assert bat.clips[i].bank == s.bank and bat.clips[i].short == s.short, (t, i)
                for l in range(s.L):
                    assert torch.equal(bat.clips[i].bankV[l].hi[s.cur], s.bankV[l].hi[s.cur]), (t, i, l)
    assert groups_seen >= {1, 2}, groups_seen
    assert any(d is not None for i, d in drops if i == 1) and any(d is not None for i, d in drops if i == 0), drops
    print("launch groups per forward pass seen:", sorted(groups_seen), "drops", drops)


def test_queue_refill_equals_lockstep_runs_per_clip():
    """BatchedClipDriver.run_queue: five clips of 7 / 5 / 6 / 4 / 3 frames (97x129) through TWO slots, each slot taking
    the next clip when its clip ends, every clip with the gap its own length asks for (here forced to differ: odd
    lengths 1, even lengths 2).  Per clip the label maps must EQUAL those of the same clip run in a lockstep batch of
    the same driver (run_clips with the clip in both slots): what the neighbouring slot holds, and when a slot started,
    changes nothing a clip computes.  The queue needs 14 steps for 25 frames; lockstep batches of the same clips
    (longest first, plan_ragged_batches by gap) need more and leave slots idle."""
    from rmem_amd import driver as D
    from rmem_amd.synth import synth_clip
    cfg, model = _model()
    Hh, Ww = 97, 129
    lens = [7, 5, 6, 4, 3]

    def clip(cid, n):
        imgs, lab = synth_clip(700 + cid, n, Hh, Ww, 3)
        return [D.make_samples(imgs[t].to(DEV), lab.to(DEV) if t == 0 else None, (Hh, Ww), 3, name=f"{t:05d}.jpg")
                for t in range(n)]
    clips = [clip(i, n) for i, n in enumerate(lens)]
    drv = D.BatchedClipDriver(model, 2, cfg)
    drv._gap_of = lambda n: 1 if n % 2 else 2
    res = drv.run_queue(clips)
    st = drv.queue_stats
    print("queue:", st)
    assert st["steps"] == 14 and st["busy_slot_steps"] == 25     # (lockstep batches by gap, longest first: 7 + 6 + 4 = 17 steps)
    assert st["launch_groups"] > st["steps"]            # some steps had slots in different states
    for i, c in enumerate(clips):
        ref = drv.run_clips([c, c])
        assert res[i].gap == ref[0].gap == drv._gap_of(lens[i])
        assert tuple(res[i].masks.shape) == (lens[i] - 1, Hh, Ww) and res[i].names == ref[0].names
        assert torch.equal(ref[0].masks, ref[1].masks)
        assert torch.equal(res[i].masks, ref[0].masks), (i, [int((res[i].masks[t] != ref[0].masks[t]).sum()) for t in range(lens[i] - 1)])
        assert int((res[i].masks != 0).sum()) > 0
    # a queue shorter than the batch: the second slot idles, the clip's result is unchanged
    lone = drv.run_queue([clips[3]])
    assert torch.equal(lone[0].masks, res[3].masks)
    # a clip that is its reference frame only (nothing to propagate) beside a real one, and an empty queue
    tiny = drv.run_queue([clip(9, 1), clips[4]])
    assert tuple(tiny[0].masks.shape) == (0, Hh, Ww) and tiny[0].names == [] and torch.equal(tiny[1].masks, res[4].masks)
    assert drv.run_queue([]) == []


def test_mid_clip_new_objects_in_the_slots_of_a_batch():
    """A frame that brings a label with NEW objects (managers/evaluator.py:484-508) inside batched clips: the new ids go
    over the prediction and the frame becomes that slot's reference frame (BatchedDeAOTEngine.add_reference_slots) while
    the other slots update as usual.  Four clips of 7 / 6 / 5 / 6 frames through two slots, two of them with a mid-clip
    label at different frames: (a) the queue's label maps EQUAL the lockstep runs of the same driver per clip (what the
    neighbour does changes nothing); (b) the frame that carries the label holds the pasted rectangle verbatim, and the first
    propagated frame agrees with the one-clip driver (ClipDriver: the evaluator's loop) up to MIOpen's batch-size rounding
    (later frames of the closed loop are reported); (c) the slot's bank restarts with that frame: its index list is
    [frame] afterwards and grows from there on the slot's own gap schedule; (d) a label that takes a clip past ten
    objects is refused with a pointer to the one-clip driver."""
    from rmem_amd import driver as D
    from rmem_amd.synth import synth_clip
    cfg, model = _model()
    Hh, Ww = 97, 129
    lens, new_at = [7, 6, 5, 6], {0: 3, 3: 2}

    def new_label(obj_id):
        lab = torch.zeros(1, 1, Hh, Ww)
        lab[:, :, Hh // 8: Hh // 3, Ww // 2: Ww * 3 // 4] = obj_id
        return lab

    def clip(cid, n, obj_id=4):
        imgs, lab = synth_clip(900 + cid, n, Hh, Ww, 3)
        lab_of = lambda t: lab if t == 0 else (new_label(obj_id) if new_at.get(cid) == t else None)
        return [D.make_samples(imgs[t].to(DEV), None if lab_of(t) is None else lab_of(t).to(DEV), (Hh, Ww), 3,
                               name=f"{t:05d}.jpg") for t in range(n)]
    clips = [clip(i, n) for i, n in enumerate(lens)]
    drv = D.BatchedClipDriver(model, 2, cfg, fixed_gap=2)
    res = drv.run_queue(clips)
    assert drv.queue_stats["launch_groups"] > drv.queue_stats["steps"]
    one = D.ClipDriver(model, cfg, fixed_gap=2)
    rect = new_label(4)[0, 0].numpy() == 4
    for i, c in enumerate(clips):
        ref = drv.run_clips([c, c])
        assert torch.equal(ref[0].masks, ref[1].masks)
        assert torch.equal(res[i].masks, ref[0].masks), (i, [int((res[i].masks[t] != ref[0].masks[t]).sum()) for t in range(lens[i] - 1)])
        single = one.run_clip(c, num_frames=lens[i])
        mism = [int((res[i].masks[t] != single.masks[t]).sum()) for t in range(lens[i] - 1)]
        print("clip", i, "pixels off the one-clip driver per frame (of %d):" % (Hh * Ww), mism)
        if i in new_at:
            t = new_at[i]
            assert (res[i].masks[t - 1].cpu().numpy()[rect] == 4).all()
            # (against the one-clip driver only the first propagated frame is asserted: MIOpen rounds the encoder at batch 2
            # differently from batch 1, and a closed loop with synthetic weights amplifies one flipped near-tie pixel --
            # 0 pixels on every frame on one box, [0, 1, 30, 185, ...] on another; test_batched_clip_driver_vs_clip_driver)
            assert mism[0] <= 3, mism
            assert int((res[i].masks[t:] == 4).sum()) > 0          # the new object is propagated
        else:
            assert mism[0] <= 3, mism                       # (synthetic weights predict every id anywhere: no statement about id 4)
    # (c) engine level: the slot's bank and index list restart with the frame that carried the label
    eng = drv.engine
    eng.restart_engine()
    imgs = torch.cat([clips[0][0][0]["current_img"], clips[1][0][0]["current_img"]])
    labs = torch.cat([F.interpolate(c[0][0]["current_label"].float(), size=imgs.shape[2:], mode="nearest") for c in clips[:2]]).int()
    eng.long_term_mem_gap = 1
    eng.add_reference_frame(imgs, labs, obj_nums=[10, 10], frame_step=0)
    lab_in = eng.lstt.label_buffer(*eng.input_size_2d)
    for t in range(1, 5):
        cur = torch.cat([clips[0][t][0]["current_img"], clips[1][t][0]["current_img"]])
        logit = eng.match_propogate_one_frame(cur)
        lab_in.copy_(torch.argmax(F.interpolate(logit, size=eng.input_size_2d, mode="bilinear", align_corners=True), 1).to(torch.uint8))
        if t == 3:
            lab_in[1, 5:20, 5:20] = 4
            eng.add_reference_slots({1: (lab_in[1], 10)})
        eng.update_memory(lab_in)
        idx = eng.long_memories_indexes
        assert idx[0][0] == 0 and idx[0][-1] == t         # (the neighbour carries on: reference frame kept, this frame appended)
        if t >= 3:
            assert idx[1] == list(range(3, t + 1)), (t, idx)
            assert eng.lstt.clips[1].bank and len(eng.lstt.clips[1].bank) == len(idx[1])
        else:
            assert idx[1][0] == 0
    eng.match_propogate_one_frame(torch.cat([clips[0][5][0]["current_img"], clips[1][5][0]["current_img"]]))
    eng.add_reference_slots({1: (lab_in[1], 10)})
    with pytest.raises(ValueError):                       # the slot has just been re-referenced in this step
        eng.add_reference_slots({1: (lab_in[1], 10)})
    # (d) a mid-clip label with id 12: more objects than the batched engine holds
    new_at[1] = 2
    big = clip(1, 6, obj_id=12)
    with pytest.raises(NotImplementedError, match="12 objects"):
        drv.run_queue([big, clips[2]])


def test_batched_clip_driver_vs_clip_driver():
    """BatchedClipDriver (2 clips of 6 frames, 97x129, closed loop) against ClipDriver per clip:
    same gap, same shapes, first propagated frame equal up to MIOpen's batch-size rounding; later
    frames of a closed loop with synthetic weights amplify a single flipped pixel
    (tests/test_oracle_golden.py), so they are reported, not asserted."""
    from rmem_amd import driver as D
    from rmem_amd.synth import synth_clip
    cfg, model = _model()
    frames, Hh, Ww = 6, 97, 129

    def frames_of(cid):
        imgs, lab = synth_clip(300 + cid, frames, Hh, Ww, 3)
        return [D.make_samples(imgs[t].to(DEV), lab.to(DEV) if t == 0 else None, (Hh, Ww), 3, name=f"{t:05d}.jpg")
                for t in range(frames)]
    clips = [frames_of(0), frames_of(1)]
    bat = D.BatchedClipDriver(model, 2, cfg).run_clips(clips, num_frames=frames)
    drv = D.ClipDriver(model, cfg)
    for i in range(2):
        one = drv.run_clip(clips[i], num_frames=frames)
        assert one.gap == bat[i].gap and one.masks.shape == bat[i].masks.shape and one.names == bat[i].names
        mism = [int((one.masks[t] != bat[i].masks[t]).sum()) for t in range(frames - 1)]
        print("clip", i, "mismatching pixels per frame (closed loop)", mism)
        assert mism[0] <= 2, mism
    with pytest.raises(ValueError):
        D.BatchedClipDriver(model, 2, cfg).run_clips(clips[:1], num_frames=frames)


def test_ragged_batch_and_run_dataset():
    """Clips of different lengths in one lockstep batch, and a mixed dataset through run_dataset (VERDICT r2 missing #4;
    the reference feeds clips of any length one by one, managers/evaluator.py:276-295,327-331).
    (a) prefix property: a 7- and a 5-frame clip in one batch give, frame for frame, the label maps of the same two
    clips both cut to 5 frames -- the idle slot (last frame repeated) changes nothing the other clip computes, and a
    clip's own result does not depend on how long its batch runs;  (b) run_dataset sends the two equal-gap clips
    through the batch, the flip-augmented clip and the left-over clip through ClipDriver, in the callers' order."""
    from rmem_amd import driver as D
    from rmem_amd.synth import synth_clip
    cfg, model = _model()
    Hh, Ww = 97, 129

    def clip(cid, n, aug=False):
        imgs, lab = synth_clip(500 + cid, n, Hh, Ww, 3)
        return [D.make_samples(imgs[t].to(DEV), lab.to(DEV) if t == 0 else None, (Hh, Ww), 3, flip_aug=aug, name=f"{t:05d}.jpg")
                for t in range(n)]
    a, b, c, d = clip(0, 7), clip(1, 5), clip(2, 6, aug=True), clip(3, 4)
    drv = D.BatchedClipDriver(model, 2, cfg)
    rag = drv.run_clips([a, b])
    assert [tuple(r.masks.shape) for r in rag] == [(6, Hh, Ww), (4, Hh, Ww)] and [len(r.names) for r in rag] == [6, 4]
    cut = drv.run_clips([a, b], num_frames=5)
    assert torch.equal(rag[1].masks, cut[1].masks) and torch.equal(rag[0].masks[:4], cut[0].masks)
    assert int((rag[0].masks != 0).sum()) > 0
    with pytest.raises(ValueError, match="share the memory gap"):
        drv.run_clips([clip(4, 170)[:170], b])               # 170 frames -> gap 6
    res = drv.run_dataset([a, b, c, d], mode="lockstep")
    assert [r.batched for r in res] == [True, True, False, False]
    # the default mode: the three single-augmentation clips share the two slots through the queue (d takes the slot b frees)
    resq = drv.run_dataset([a, b, c, d])
    assert [r.batched for r in resq] == [True, True, False, True]
    assert torch.equal(resq[0].masks, rag[0].masks) and torch.equal(resq[1].masks, rag[1].masks)
    assert torch.equal(resq[3].masks, drv.run_clips([d, d])[0].masks)
    assert torch.equal(res[0].masks, rag[0].masks) and torch.equal(res[1].masks, rag[1].masks)
    one = D.ClipDriver(model, cfg)
    assert torch.equal(res[3].masks, one.run_clip(d, num_frames=4).masks)
    # a group whose remainder is two clips of a three-slot driver: one padded batch (the third slot repeats the first clip)
    drv3 = D.BatchedClipDriver(model, 3, cfg)
    res3 = drv3.run_dataset([a, b], mode="lockstep")
    assert [r.batched for r in res3] == [True, True]
    assert torch.equal(res3[0].masks, rag[0].masks) and torch.equal(res3[1].masks, rag[1].masks)
    # the flip-augmented clip: same path (ClipDriver, two engines), but a driver's engines carry their history (which
    # hipGraphs exist, what MIOpen tuned first), so a second driver is compared within the near-tie bound of the suite
    ref_c = D.ClipDriver(model, cfg).run_clip(c, num_frames=6).masks
    mism = [int((res[2].masks[t] != ref_c[t]).sum()) for t in range(5)]
    print("TTA clip through run_dataset vs a fresh ClipDriver: mismatching pixels per frame", mism)
    assert res[2].masks.shape[0] == 5 and mism[0] <= 4, mism


def test_batched_engine_vs_oracle():
    """Two clips in lockstep through BatchedDeAOTEngine against one CPU oracle engine per clip,
    teacher-forced with the oracle's label maps over 20 frames (K = 4, gap 2: 7 evictions per clip):
    kept-frame indexes equal every frame, label maps within 2 of 12.5k pixels."""
    from oracle.engine_ref import OracleDeAOTInferEngine
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    cfg = get_config("r50_deaotl", 1, 3)
    cpu_model = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(cpu_model)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    B, frames, Hh, Ww = 2, 21, 97, 129
    clips = [synth_clip(400 + i, frames, Hh, Ww, 3) for i in range(B)]
    oras = [OracleDeAOTInferEngine(cpu_model, long_term_mem_gap=2) for _ in range(B)]
    for i, o in enumerate(oras):
        o.add_reference_frame(clips[i][0][0], clips[i][1], obj_nums=[3], frame_step=0)
    bat = BatchedDeAOTEngine(gpu_model, B, long_term_mem_gap=2)
    bat.add_reference_frame(torch.cat([clips[i][0][0] for i in range(B)]).to(DEV),
                            torch.cat([clips[i][1] for i in range(B)]).to(DEV), obj_nums=[10] * B, frame_step=0)
    worst = 0
    for t in range(1, frames):
        lg = bat.match_propogate_one_frame(torch.cat([clips[i][0][t] for i in range(B)]).to(DEV), output_size=(Hh, Ww))
        pb = lg.argmax(1).cpu()
        fed = []
        for i, o in enumerate(oras):
            po = o.match_propogate_one_frame(clips[i][0][t], output_size=(Hh, Ww)).argmax(1, keepdim=True).float()
            mism = int((po[0, 0] != pb[i]).sum())
            worst = max(worst, mism)
            assert mism <= 2, (t, i, mism)
            cur = F.interpolate(po, size=o.input_size_2d, mode="nearest")
            o.update_memory(cur)
            fed.append(cur)
        bat.update_memory(torch.cat(fed).to(DEV))
        for i, o in enumerate(oras):
            assert bat.long_memories_indexes[i] == list(o.engines[0].long_memories_indexes), (t, i)
    print("batched engine vs oracle: worst label mismatch", worst, "final indexes", bat.long_memories_indexes)


@pytest.mark.parametrize("kind", ["batched", "single"])
def test_load_network_after_engine_build_repacks_weights_and_graphs(kind):
    """load_network() on a model an engine already holds (utils/checkpoint.py:75-101 in the evaluator
    happens before the engines exist; here it may happen after): the packed LSTT / ID-bank planes, the
    folded encoder and every captured hipGraph are stale.  After restart_engine + a new reference frame
    the engine must give what an engine BUILT after the load gives (same process, same kernels: bit for
    bit), and not what the old weights gave."""
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.checkpoint import load_network
    from rmem_amd.engine import build_engine
    from rmem_amd.config import get_config
    from rmem_amd.model import build_vos_model
    from rmem_amd.synth import load_synthetic_weights, synth_clip
    cfg, model = _model()
    B, frames, Hh, Ww = 2, 6, 97, 129
    clips = [synth_clip(600 + i, frames, Hh, Ww, 3) for i in range(B)]

    def make(m):
        if kind == "batched":
            return BatchedDeAOTEngine(m, B, long_term_mem_gap=2)
        e = build_engine("deaotengine", phase="eval", aot_model=m, gpu_id=0, long_term_mem_gap=2)
        e.eval()
        return e

    def run(eng):
        outs = []
        eng.restart_engine()
        if kind == "batched":
            eng.add_reference_frame(torch.cat([c[0][0] for c in clips]).to(DEV), torch.cat([c[1] for c in clips]).to(DEV),
                                    obj_nums=[3] * B, frame_step=0)
        else:
            eng.add_reference_frame(clips[0][0][0].to(DEV), clips[0][1].to(DEV), obj_nums=[3], frame_step=0)
        for t in range(1, frames):
            img = torch.cat([c[0][t] for c in clips]).to(DEV) if kind == "batched" else clips[0][0][t].to(DEV)
            lg = eng.match_propogate_one_frame(img, output_size=(Hh, Ww))
            outs.append(lg.clone())
            lab = lg.argmax(1, keepdim=True).float()
            eng.update_memory(F.interpolate(lab, size=eng.input_size_2d, mode="nearest"))
        return outs

    eng = make(model)
    old = run(eng)
    again = run(eng)                                    # graphs replayed: same engine, same weights
    assert all(torch.equal(a, b) for a, b in zip(old, again))
    donor = build_vos_model("deaot", cfg).eval()
    load_synthetic_weights(donor, salt=1)
    load_network(model, {"state_dict": donor.state_dict()})
    new = run(eng)
    fresh = run(make(model))
    assert max(float((a - b).abs().max()) for a, b in zip(new, old)) > 1e-2          # the new weights are in use
    for t, (a, b) in enumerate(zip(new, fresh)):
        assert torch.equal(a, b), (t, float((a - b).abs().max()))


def test_batched_engine_480p_vs_golden_and_unbatched(golden_dir):
    """BatchedDeAOTEngine at the BASELINE.json configs[3] geometry (481x849, K = 4), B = 4 clips in lockstep:
    slots 0 and 2 run the reference's golden clip, slots 1 and 3 two other synthetic clips, all
    teacher-forced (golden labels for 0 / 2, the unbatched engine's own labels for 1 / 3).
    (a) slots 0 / 2 reproduce the reference's golden label maps up to fp64 near-ties -- every pixel that differs is in
    the fixture's near-tie list (fp64 margin < 2e-5) and got one of the tie's two classes (tests/ties.py) -- and its
    kept-frame history exactly;
    (b) every slot against an UNBATCHED engine fed the same labels: equal eviction histories, decoder logits within 2e-3,
    and every pixel whose label differs is a near-tie in the unbatched engine's own logits (margin below twice the logit
    difference of that frame).  What is bit-identical between the two is the
    memory path (test_batched_lstt_equals_single_clips_bit_for_bit, incl. 31x54 tokens); MIOpen at
    batch 4 and at batch 1-2 picks different algorithms for the encoder / decoder convolutions, which
    moves near-tie pixels."""
    import json
    import os
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.engine import DeAOTEngine
    from rmem_amd.synth import synth_clip
    from ties import Fp64Ties, oracle_margin_check
    meta = json.load(open(os.path.join(golden_dir, "clip_480p.json")))
    gold = np.load(os.path.join(golden_dir, "clip_480p.npz"))
    ties = Fp64Ties(np.load(os.path.join(golden_dir, "clip_480p_fp64.npz")))
    cfg, model = _model(meta["former"], meta["latter"])
    model.optimize_for_inference(True)
    B, frames, H, W = 4, meta["frames"], meta["H"], meta["W"]
    out_hw = tuple(meta["out_hw"])
    seeds = [meta["seed"], 901, meta["seed"], 902]
    clips = [synth_clip(s, frames, H, W, 3) for s in seeds]
    singles = []
    for i in range(B):
        e = DeAOTEngine(model, 0, long_term_mem_gap=meta["gap"])
        e.eval()
        e.add_reference_frame(clips[i][0][0].to(DEV), clips[i][1].to(DEV), obj_nums=[10], frame_step=0)   # as the wrapper does
        singles.append(e)
    bat = BatchedDeAOTEngine(model, B, long_term_mem_gap=meta["gap"])
    bat.add_reference_frame(torch.cat([c[0][0] for c in clips]).to(DEV), torch.cat([c[1] for c in clips]).to(DEV),
                            obj_nums=[10] * B, frame_step=0)
    vs_gold, vs_single, lerr = [], [], 0.0
    for t in range(1, frames):
        lg_b = bat.match_propogate_one_frame(torch.cat([c[0][t] for c in clips]).to(DEV), output_size=out_hw)
        lab_b = lg_b.argmax(1)
        fed = []
        for i, e in enumerate(singles):
            lg = e.match_propogate_one_frame(clips[i][0][t].to(DEV), output_size=out_hw)
            lab = lg.argmax(1)
            le = float((e.pred_id_logits[0] - bat.pred_id_logits[i]).abs().max())
            lerr = max(lerr, le)
            vs_single.append(oracle_margin_check(lab_b[i].cpu().numpy().astype(np.uint8), lg[0].cpu(), 2 * le + 1e-7,
                                                 f"frame {t} slot {i} batched vs unbatched"))
            if seeds[i] == meta["seed"]:
                g = torch.from_numpy(gold["labels"][t - 1]).to(DEV)
                n32, _, _ = ties.check(t, lab_b[i].cpu().numpy().astype(np.uint8), gold["labels"][t - 1], 2e-5)
                vs_gold.append(n32)
                cur = g[None, None].float()
            else:
                cur = lab[None].float()
            cur = F.interpolate(cur, size=e.input_size_2d, mode="nearest")
            e.update_short_term_memory(cur)
            fed.append(cur)
        bat.update_memory(torch.cat(fed))
        for i, e in enumerate(singles):
            assert bat.long_memories_indexes[i] == list(e.long_memories_indexes), (t, i)
        assert bat.long_memories_indexes[0] == meta["indexes"][t - 1] == bat.long_memories_indexes[2], t
    print("batched B=4 at 481x849: mismatching pixels vs golden", vs_gold, "vs unbatched", vs_single, "logit err", lerr)
    assert lerr < 2e-3, (vs_gold, vs_single, lerr)


def test_batched_every_slot_vs_reference_fixture_480p(golden_dir):
    """BASELINE.json configs[3]'s per-rank shape (8 clips per launch, 481x849, K = 4) with EVERY slot running the
    reference's golden clip, teacher-forced with its labels.  MIOpen's batch-8 convolutions round a sample differently
    depending on its position in the batch (1.5e-5 ... 3.6e-5 on the encoder features, profiles/r04_slot_position_probe.md),
    so the slots' label maps are not hash-equal; the stated and tested bound instead: for every slot and frame, every pixel
    off the reference's fp32 map is an fp64 near-tie (margin < 2e-5) that got one of the tie's two classes
    (tests/ties.py), the kept-frame history equals the reference's, and no slot is further from the fp64 maps than the
    fp32 reference itself + slack; the spread of the decoder logits across slots is reported and bounded."""
    import json
    import os
    from ties import Fp64Ties
    from rmem_amd.batched import BatchedDeAOTEngine
    from rmem_amd.synth import synth_clip
    meta = json.load(open(os.path.join(golden_dir, "clip_480p.json")))
    gold = np.load(os.path.join(golden_dir, "clip_480p.npz"))
    ties = Fp64Ties(np.load(os.path.join(golden_dir, "clip_480p_fp64.npz")))
    cfg, model = _model(meta["former"], meta["latter"])
    model.optimize_for_inference(True)
    B, frames, H, W = 8, meta["frames"], meta["H"], meta["W"]
    out_hw = tuple(meta["out_hw"])
    imgs, lab = synth_clip(meta["seed"], frames, H, W, 3)
    bat = BatchedDeAOTEngine(model, B, long_term_mem_gap=meta["gap"])
    bat.add_reference_frame(torch.cat([imgs[0]] * B).to(DEV), torch.cat([lab] * B).to(DEV), obj_nums=[10] * B, frame_step=0)
    off32 = [[0] * (frames - 1) for _ in range(B)]
    off64 = [0] * B
    spread, worst = 0.0, 0.0
    for t in range(1, frames):
        lg = bat.match_propogate_one_frame(torch.cat([imgs[t]] * B).to(DEV), output_size=out_hw)
        labs = lg.argmax(1).cpu().numpy().astype(np.uint8)
        spread = max(spread, float((bat.pred_id_logits - bat.pred_id_logits[:1]).abs().max()))
        for i in range(B):
            n32, n64, w = ties.check(t, labs[i], gold["labels"][t - 1], 2e-5)
            off32[i][t - 1], off64[i], worst = n32, off64[i] + n64, max(worst, w)
        g = torch.from_numpy(gold["labels"][t - 1]).to(DEV)[None, None].float()
        bat.update_memory(torch.cat([F.interpolate(g, size=bat.input_size_2d, mode="nearest")] * B))
        for i in range(B):
            assert bat.long_memories_indexes[i] == meta["indexes"][t - 1], (t, i)
    ref64 = sum(ties.n_ref32_vs_64(t, gold["labels"][t - 1]) for t in range(1, frames))
    print("8 slots x golden 480p clip: pixels off the reference's fp32 maps per slot and frame:", off32)
    print("  off the fp64 maps per slot:", off64, "(fp32 reference itself:", ref64, "); largest fp64 margin of a moved pixel",
          f"{worst:.2e}; decoder-logit spread across slots {spread:.2e}")
    assert max(off64) <= ref64 + 4 and spread < 2e-4
