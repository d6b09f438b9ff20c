"""Parity mode: the reference's ``--fix_random`` switch (aot_plus/tools/eval.py:21-37) for this build.

The reference fixes its seeds and asks cuDNN for deterministic algorithms
(``torch.backends.cudnn.deterministic = True``, ``benchmark = False``, ``CUDNN_DETERMINISTIC=1``).  On
PyTorch-ROCm the same two flags steer MIOpen: ``benchmark = False`` selects convolutions through MIOpen's
immediate mode (no timing-based search whose winner can differ between processes) and
``deterministic = True`` excludes solvers marked non-deterministic.  ``fix_random()`` sets them, plus the
MIOpen find-mode environment (only effective before the first convolution of the process).

Whether two PROCESSES then produce bit-identical encoder features on this ROCm is measured by
``tools/parity_mode_probe.py`` (result under ``profiles/``); the hot path itself (rmem_amd/csrc) has no
floating-point atomics and is bit-reproducible with or without this switch.

``RMEM_DETERMINISTIC=1`` in the environment applies it when ``rmem_amd.engine`` builds its first engine.

``reproducible_convolutions()`` is the part that matters on this ROCm (measured, tools/encoder_race_probe.py,
profiles/r03_i_encoder_race_probe_97x129.json): MIOpen's implicit-GEMM solver family is switched off for the
process (MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0, only effective before the first convolution).  At small frame
sizes (97x129) MIOpen picks an implicit-GEMM solver for the encoder's stride-2 1x1 downsample convolutions
whose output differs from call to call (~1e-5, kernel launches serialised or not); with the family disabled
encoder and decoder are bit-reproducible call to call and process to process at 97x129 and at 481x849
(profiles/r03_parity_mode_probe.json).  One clip per GPU does not slow down (466.7 vs 463.8 frames/s,
profiles/r03_j_bench_ab_*.json); several clips per launch DO (MIOpen at batch 8: 14.6 -> 21.9 ms per
step with the switch on and off, measured on one box in round 3; profiles/r03g_bench_batched8.json is without it), so the switch is applied by fix_random(), by the test
suite (tests/conftest.py) and by bench.py's one-clip-per-engine modes -- not by the library on import.
"""
from __future__ import annotations

import os
import random

_applied = False


def reproducible_convolutions() -> None:
    """MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 unless the environment already says otherwise (see the module text)."""
    os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM", "0")


def fix_random(seed: int = 1) -> None:
    """Mirror of tools/eval.py:21-37 (same seed offsets), with the MIOpen counterparts of the cuDNN flags."""
    global _applied
    import numpy as np
    import torch
    os.environ["CUDNN_DETERMINISTIC"] = "1"
    os.environ["PYTHONHASHSEED"] = str(seed)
    reproducible_convolutions()
    # MIOpen: "fast" find mode (2) = find-db hit or the immediate-mode fallback, never a timed search
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    os.environ.setdefault("MIOPEN_FIND_ENFORCE", "1")          # NONE: never (re)search / update the find-db
    # ... and an EMPTY user find-db of this process's own.  In MIOpen's default (hybrid) mode the first use of a convolution
    # runs a timed search and records the winner in ~/.config/miopen; what a later process computes then depends on what
    # earlier processes happened to time -- round 6 traced every clip-hash difference between runs of one command to that
    # state (profiles/r06m_world_hash_matrix.txt: 1 or 8 ranks, cold or warm caches: one set of hashes; another find mode:
    # another set; profiles/r05g: the first 8-rank run of a fresh box, searching under contention, differed from all later
    # ones).  With no entries to find and no search allowed, the solver of every convolution is MIOpen's heuristic choice.
    if "MIOPEN_USER_DB_PATH" not in os.environ:
        import tempfile
        os.environ["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="rmem_miopen_userdb_")
    random.seed(seed + 1)
    np.random.seed(seed + 2)
    torch.manual_seed(seed + 3)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed + 4)
        torch.cuda.manual_seed_all(seed + 5)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    _applied = True


def maybe_fix_random() -> bool:
    """Apply fix_random() once if RMEM_DETERMINISTIC=1 (called when an engine is built)."""
    if not _applied and os.environ.get("RMEM_DETERMINISTIC", "") == "1":
        fix_random()
    return _applied
