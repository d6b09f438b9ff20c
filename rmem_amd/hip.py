"""ctypes binding of librmem_hip.so (include/rmem_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this
module raises.  Tensors are passed as raw device pointers (``tensor.data_ptr()``) and
the launch stream is torch's current HIP stream, so calls are capturable in a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_LIB = None
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librmem_hip.so")

c_p = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float


class RmemError(RuntimeError):
    pass


class LinearArgs(C.Structure):
    _fields_ = [
        ("xh", c_p), ("xl", c_p), ("ldx", i64),
        ("xh2", c_p), ("xl2", c_p), ("ldx2", i64), ("kx_split", i32),
        ("yh", c_p), ("yl", c_p), ("ldy", i64),
        ("yh2", c_p), ("yl2", c_p), ("ldy2", i64), ("ky_split", i32),
        ("M", i32), ("N", i32), ("K", i32),
        ("bias", c_p), ("bias_per_row", i32), ("act", i32),
        ("d0", c_p), ("ldd0", i64), ("d1", c_p), ("ldd1", i64),
        ("csplit", i32), ("accumulate", i32),
        ("pah", c_p), ("pal", c_p), ("ldpa", i64),
        ("pbh", c_p), ("pbl", c_p), ("ldpb", i64), ("addvec", c_p),
        ("nbatch", i32), ("bsx", i64), ("bsy", i64), ("bsd", i64), ("bsbias", i64), ("bspa", i64),
        ("nsplit", i32), ("tile", i32),
        ("ksplits", i32), ("parts", c_p), ("part_stride", i64),
        ("pa_blocked", i32), ("d0_cs", i32),
    ]


class ReadArgs(C.Structure):
    _fields_ = [
        ("mode", i32),
        ("qh", c_p), ("ql", c_p),
        ("kh", c_p), ("kl", c_p), ("k_slot_stride", i64),
        ("vh", c_p), ("vl", c_p), ("v_slot_stride", i64),
        ("slot_map", c_p), ("T", i32), ("N", i32), ("Npad", i32), ("ncols", i32),
        ("scale", f32), ("bias", c_p), ("R", c_p), ("ldr", i32), ("h", i32), ("w", i32), ("rcs", i32),
        ("ksplits", i32), ("part", c_p), ("ml", c_p), ("lslot", c_p), ("sched", c_p),
        ("nfull", i32), ("pf", i32), ("gate", c_p), ("ldgate", i64), ("gout", c_p), ("ldgout", i64),
        ("dbg_logits", c_p), ("dbg_ld", i64),
    ]


class LnArgs(C.Structure):
    _fields_ = [
        ("x", c_p), ("ldx", i64), ("x2", c_p), ("ldx2", i64), ("sum_out", c_p), ("ldsum", i64), ("gamma", c_p), ("beta", c_p),
        ("post", c_p), ("ldpost", i64), ("oh", c_p), ("ol", c_p), ("ldo", i64), ("of32", c_p), ("ldof", i64),
    ]


class AddArgs(C.Structure):
    _fields_ = [("a", c_p), ("b", c_p), ("dst", c_p), ("oh", c_p), ("ol", c_p)]


class ReadCombineArgs(C.Structure):
    _fields_ = [
        ("T", i32), ("N", i32), ("Npad", i32), ("ncols", i32), ("ksplits", i32),
        ("part", c_p), ("ml", c_p), ("lslot", c_p),
        ("U", c_p), ("ldu", i64), ("G", c_p), ("ldg", i64), ("mass", c_p),
    ]


class MHAArgs(C.Structure):
    _fields_ = [
        ("qh", c_p), ("ql", c_p), ("ldq", i64),
        ("kh", c_p), ("kl", c_p), ("k_slot_stride", i64), ("ldk", i64),
        ("vh", c_p), ("vl", c_p), ("v_slot_stride", i64), ("ldv", i64),
        ("slot_map", c_p), ("T", i32), ("N", i32), ("Npad", i32), ("heads", i32),
        ("scale", f32), ("bias", c_p), ("ksplits", i32),
        ("opart", c_p), ("ml", c_p), ("slot_ml", c_p), ("nsplit", i32),
    ]


class MHACombineArgs(C.Structure):
    _fields_ = [
        ("N", i32), ("Npad", i32), ("heads", i32), ("T", i32), ("ksplits", i32),
        ("opart", c_p), ("ml", c_p), ("slot_ml", c_p),
        ("oh", c_p), ("ol", c_p), ("of32", c_p), ("ldo", i64), ("mass", c_p),
    ]


class BankState(C.Structure):
    _fields_ = [("T", i32), ("index", i32 * 16), ("visits", i32 * 16), ("has_ema", i32 * 16), ("ema", f32 * 16),
                ("last_drop", i32), ("steps", i32)]


class LabelSrc(C.Structure):
    _fields_ = [("logits", c_p), ("h", i32), ("w", i32), ("flip", i32)]


EXPORTS = [
    "rmem_abi_version", "rmem_configure", "rmem_set_host_wait", "rmem_linear", "rmem_linear_trace",
    "rmem_layernorm_split", "rmem_dwconv5x5_split", "rmem_groupnorm2",
    "rmem_id_assign", "rmem_attn_mass_reduce", "rmem_fg_weights", "rmem_bank_reset", "rmem_bank_append",
    "rmem_bank_policy_step", "rmem_split_planes", "rmem_groupnorm_nchw",
    "rmem_mha_flash", "rmem_mha_flash2", "rmem_mha_combine", "rmem_mha_combine2", "rmem_layernorm_ex", "rmem_layernorm_multi", "rmem_transpose_planes", "rmem_add_split", "rmem_add_split_multi",
    "rmem_gn_gelu_tokens", "rmem_pe_bias_heads", "rmem_linear_grouped", "rmem_layernorm_red", "rmem_bias_act_nchw", "rmem_set_ints",
    "rmem_labels_from_logits", "rmem_label_resize_nearest", "rmem_upsample_add_nchw", "rmem_groupnorm_nchw_bias",
    "rmem_upsample_add_nchw_out", "rmem_layernorm_red2", "rmem_layernorm_cn",
    "rmem_bias_act_nchw_batched", "rmem_dwconv5x5_split2",
    "rmem_attn_read", "rmem_attn_read_trace", "rmem_attn_read2", "rmem_attn_read_combine", "rmem_attn_read_combine2",
    "rmem_rec_end", "rmem_launch_recorded", "rmem_groupnorm2_fold",
]
# exports with a non-int return type
EXPORTS_OTHER = ["rmem_rec_begin", "rmem_rec_free", "rmem_rec_count", "rmem_rec_size", "rmem_rec_data",
                 "rmem_rec_signature"]


def lib_path() -> str:
    return _LIB_PATH


ABI_VERSION = 18          # rmem_abi_version() of the library these ctypes structures describe (include/rmem_hip.h)


def load():
    """Load librmem_hip.so (built in-tree by rmem_amd.build).  Raises if missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise RmemError(f"{_LIB_PATH} not found: run `python -m rmem_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback for the RMem hot path.")
    lib = C.CDLL(_LIB_PATH)
    lib.rmem_abi_version.restype = C.c_int
    if lib.rmem_abi_version() != ABI_VERSION:
        raise RmemError(f"{_LIB_PATH} has ABI {lib.rmem_abi_version()}, this binding expects {ABI_VERSION}: "
                        "stale library -- rebuild with `python -m rmem_amd.build --force`")
    for name in EXPORTS:
        getattr(lib, name).restype = C.c_int
    lib.rmem_configure.argtypes = [C.c_char_p, i64]
    lib.rmem_linear.argtypes = [C.POINTER(LinearArgs), c_p]
    lib.rmem_linear_grouped.argtypes = [C.POINTER(LinearArgs), i32, c_p]
    lib.rmem_linear_trace.argtypes = [C.POINTER(LinearArgs), i32, c_p, c_p]
    lib.rmem_layernorm_red.argtypes = [c_p, i64, c_p, i32, i64, i64, c_p, c_p, i32, i32, f32, c_p, c_p, i64,
                                       c_p, i64, c_p]
    lib.rmem_bias_act_nchw.argtypes = [c_p, c_p, c_p, i32, i64, i32, c_p]
    lib.rmem_bias_act_nchw_batched.argtypes = [c_p, c_p, c_p, i32, i32, i64, i32, c_p]
    lib.rmem_layernorm_red2.argtypes = [c_p, c_p, i64, c_p, c_p, i32, i64, i64, c_p, c_p, c_p, c_p, i32, i32, f32,
                                        c_p, c_p, i64, c_p, c_p, i64, c_p]
    lib.rmem_layernorm_cn.argtypes = [c_p, i64, c_p, c_p, c_p, c_p, i32, i32, f32, c_p, c_p, i64, c_p]
    lib.rmem_set_ints.argtypes = [c_p, C.POINTER(i32), i32, c_p]
    lib.rmem_dwconv5x5_split2.argtypes = [c_p, c_p, i64, c_p, c_p, i32, i32, i32, c_p, c_p, c_p, c_p, i64, c_p]
    lib.rmem_attn_read.argtypes = [C.POINTER(ReadArgs), c_p]
    lib.rmem_attn_read_trace.argtypes = [C.POINTER(ReadArgs), c_p, c_p]
    lib.rmem_attn_read2.argtypes = [C.POINTER(ReadArgs), C.POINTER(ReadArgs), c_p]
    lib.rmem_attn_read_combine.argtypes = [C.POINTER(ReadCombineArgs), c_p]
    lib.rmem_attn_read_combine2.argtypes = [C.POINTER(ReadCombineArgs), C.POINTER(ReadCombineArgs), c_p]
    lib.rmem_layernorm_split.argtypes = [c_p, i64, c_p, c_p, i32, i32, f32, c_p, c_p, i64, c_p, i64, c_p]
    lib.rmem_dwconv5x5_split.argtypes = [c_p, i64, c_p, i32, i32, i32, c_p, c_p, i64, c_p]
    lib.rmem_groupnorm2.argtypes = [c_p, c_p, i32, i32, c_p, c_p, f32, c_p, c_p, i64, c_p]
    lib.rmem_groupnorm2_fold.argtypes = [c_p, c_p, c_p, i32, i64, i64, i32, i32, c_p, c_p, f32, c_p, c_p, i64, c_p]
    lib.rmem_id_assign.argtypes = [c_p, i32, i32, c_p, c_p, i32, i32, i32, i32, i32, i32, i32,
                                   c_p, c_p, f32, c_p, c_p, i64, c_p, i64, i32, c_p]
    lib.rmem_attn_mass_reduce.argtypes = [c_p, i32, i32, c_p, c_p, c_p]
    lib.rmem_fg_weights.argtypes = [c_p, i32, i32, i32, i32, i32, c_p, c_p]
    lib.rmem_bank_reset.argtypes = [c_p, c_p, i32, i32, c_p]
    lib.rmem_bank_append.argtypes = [c_p, c_p, i32, i32, c_p]
    lib.rmem_bank_policy_step.argtypes = [c_p, c_p, c_p, i32, i32, i32, c_p, c_p]
    lib.rmem_split_planes.argtypes = [c_p, i64, c_p, c_p, c_p]
    lib.rmem_groupnorm_nchw.argtypes = [c_p, c_p, i32, i64, i32, c_p, c_p, f32, i32, c_p, c_p]
    lib.rmem_groupnorm_nchw_bias.argtypes = [c_p, c_p, c_p, i32, i64, i32, c_p, c_p, f32, i32, c_p, c_p]
    lib.rmem_mha_flash.argtypes = [C.POINTER(MHAArgs), c_p]
    lib.rmem_mha_combine.argtypes = [C.POINTER(MHACombineArgs), c_p]
    lib.rmem_mha_flash2.argtypes = [C.POINTER(MHAArgs), C.POINTER(MHAArgs), c_p]
    lib.rmem_mha_combine2.argtypes = [C.POINTER(MHACombineArgs), C.POINTER(MHACombineArgs), c_p]
    lib.rmem_layernorm_ex.argtypes = [c_p, i64, c_p, i64, c_p, c_p, i32, i32, f32, c_p, i64, c_p, c_p, i64,
                                      c_p, i64, c_p]
    lib.rmem_layernorm_multi.argtypes = [C.POINTER(LnArgs), i32, i32, i32, f32, c_p]
    lib.rmem_transpose_planes.argtypes = [c_p, c_p, i64, i32, i32, c_p, c_p, i64, c_p]
    lib.rmem_add_split_multi.argtypes = [C.POINTER(AddArgs), i32, i64, c_p]
    lib.rmem_add_split.argtypes = [c_p, c_p, i64, c_p, c_p, c_p, c_p]
    lib.rmem_gn_gelu_tokens.argtypes = [c_p, i32, i32, i32, c_p, c_p, f32, c_p, c_p, c_p]
    lib.rmem_pe_bias_heads.argtypes = [c_p, i64, c_p, c_p, C.POINTER(i32), i32, i32, i32, c_p, c_p]
    lib.rmem_labels_from_logits.argtypes = [C.POINTER(LabelSrc), i32, i32, i32, i32, i32, c_p, c_p]
    lib.rmem_upsample_add_nchw.argtypes = [c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, c_p]
    lib.rmem_upsample_add_nchw_out.argtypes = [c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, c_p]
    lib.rmem_label_resize_nearest.argtypes = [c_p, i32, i32, c_p, i32, i32, i32, c_p]
    lib.rmem_rec_begin.restype, lib.rmem_rec_begin.argtypes = c_p, []
    lib.rmem_rec_end.argtypes = [c_p]
    lib.rmem_rec_free.restype, lib.rmem_rec_free.argtypes = None, [c_p]
    lib.rmem_rec_count.restype, lib.rmem_rec_count.argtypes = i32, [c_p]
    lib.rmem_rec_size.restype, lib.rmem_rec_size.argtypes = i64, [c_p]
    lib.rmem_rec_data.restype, lib.rmem_rec_data.argtypes = c_p, [c_p]
    lib.rmem_rec_signature.restype, lib.rmem_rec_signature.argtypes = C.c_uint64, [c_p]
    lib.rmem_launch_recorded.argtypes = [c_p, c_p, i64, i32, c_p]
    _LIB = lib
    _configure_from_environment(lib)
    return lib


# RMEM_* tuning variables -> rmem_configure() names.  The C library reads no environment variable; this host layer does, once,
# when it loads the library (tests and tools switch kernels at run time with configure()).
_ENV_SWITCHES = (
    ("RMEM_LINEAR", lambda v: [("linear_tiles", 1 if v.startswith("t") else 0)]),
    ("RMEM_DW", lambda v: list(zip(("dw_rx", "dw_v"), (int(x) for x in v.split(","))))),
    ("RMEM_DW_ROWS", lambda v: [("dw_rows", int(v))]),
    ("RMEM_DW_ORDER", lambda v: [("dw_grid_order", 1 if v.startswith("g") else 0)]),
    ("RMEM_IDA", lambda v: list(zip(("ida_tokens", "ida_unroll"), (int(x) for x in (v.split(",") + ["16"])[:2])))),
)


def _configure_from_environment(lib) -> None:
    for var, parse in _ENV_SWITCHES:
        val = os.environ.get(var)
        if val:
            for name, value in parse(val):
                if lib.rmem_configure(name.encode(), int(value)) != 0:
                    raise RmemError(f"{var}={val}: rmem_configure({name!r}, {value}) refused")


def configure(name: str, value: int) -> None:
    """rmem_configure: a process-wide tuning / debug switch of the library (names and ranges in include/rmem_hip.h)."""
    check(load().rmem_configure(name.encode(), int(value)), f"rmem_configure({name!r}, {value})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


_HOST_WAIT: dict = {}


def set_host_wait(device_index: int = 0, blocking: Optional[bool] = None) -> bool:
    """rmem_set_host_wait (include/rmem_hip.h): every host wait for `device_index` sleeps on the interrupt
    (hipDeviceScheduleBlockingSync) instead of spinning.  OPT-IN (blocking=None: only with RMEM_BLOCKING_WAIT=1):
    with two processes sharing one GPU the first MIOpen convolution never returns under this flag
    (profiles/r04_host_cpu_blocking_wait.md), and the one long wait of the frame loop is handled without it
    (wait_event below).  Returns whether the flag was set."""
    if blocking is None:
        blocking = os.environ.get("RMEM_BLOCKING_WAIT") == "1"
        if not blocking:
            return False
    key = int(device_index)
    if _HOST_WAIT.get(key) == bool(blocking):
        return True
    lib = load()
    lib.rmem_set_host_wait.argtypes = [i32, i32]
    ok = lib.rmem_set_host_wait(key, int(bool(blocking))) == 0
    if ok:
        _HOST_WAIT[key] = bool(blocking)
    return ok


def wait_event(ev) -> None:
    """Wait for a HIP event WITHOUT burning a core: poll + sleep 0.2 ms.  The engine thread runs up to
    `long_term_mem_gap` frames ahead of the GPU and waits for it once per long-term update (resolve_policy); the
    runtime's hipEventSynchronize spins for the whole wait -- 0.88 of every 1.12 s of the bench loop -- and
    torch.cuda.Event(blocking=True) does not change that on this ROCm.  The sleep's latency (< 0.3 ms) is hidden: the
    host is frames ahead.  RMEM_SPIN_WAIT=1: the runtime's wait."""
    if ev.query():
        return
    if os.environ.get("RMEM_SPIN_WAIT") == "1":
        ev.synchronize()
        return
    import time
    while not ev.query():
        time.sleep(2e-4)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def check(rc: int, what: str):
    if rc != 0:
        raise RmemError(f"{what} failed with status {rc}")


class Planes:
    """An fp32 tensor carried as two fp16 planes (hi, lo): 22 significant bits, values beyond
    +-65504 saturate (rmem_common.h: split-fp16)."""

    __slots__ = ("hi", "lo")

    def __init__(self, hi: torch.Tensor, lo: torch.Tensor):
        self.hi, self.lo = hi, lo

    @staticmethod
    def empty(shape, device):
        return Planes(torch.zeros(shape, dtype=torch.float16, device=device),
                      torch.zeros(shape, dtype=torch.float16, device=device))

    @staticmethod
    def from_f32(x: torch.Tensor):
        x = x.float().clamp(-65504.0, 65504.0)
        hi = x.to(torch.float16)
        lo = (x - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
        return Planes(hi.contiguous(), lo.contiguous())

    def float(self):
        return self.hi.float() + self.lo.float()

    def __getitem__(self, idx):
        return Planes(self.hi[idx], self.lo[idx])


# ------------------------------------------------------------------ thin call wrappers
def linear(x: Planes, y: Planes, M, N, K, *, ldx, ldy, bias=None, bias_per_row=False, act=0,
           d0=None, ldd0=0, d1=None, ldd1=0, csplit=0, accumulate=False,
           pa: Planes = None, ldpa=0, pb: Planes = None, ldpb=0, addvec=None,
           x2: Planes = None, ldx2=0, kx_split=0, y2: Planes = None, ldy2=0, ky_split=0,
           nbatch=1, bsx=0, bsy=0, bsd=0, bsbias=0, bspa=0, nsplit=3, tile=0,
           x_off=0, y_off=0, ksplits=1, parts=None, part_stride=0, launch=True, pa_blocked=False,
           pa_off=0, d0_cs=0):
    """x_off / y_off: element offsets into the plane tensors (column windows).
    launch=False returns the filled argument struct (for linear_grouped)."""
    a = LinearArgs()
    eb = 2  # bytes per plane element
    a.xh, a.xl, a.ldx = x.hi.data_ptr() + x_off * eb, x.lo.data_ptr() + x_off * eb, ldx
    if x2 is not None:
        a.xh2, a.xl2, a.ldx2, a.kx_split = x2.hi.data_ptr(), x2.lo.data_ptr(), ldx2, kx_split
    a.yh, a.yl, a.ldy = y.hi.data_ptr() + y_off * eb, y.lo.data_ptr() + y_off * eb, ldy
    if y2 is not None:
        a.yh2, a.yl2, a.ldy2, a.ky_split = y2.hi.data_ptr(), y2.lo.data_ptr(), ldy2, ky_split
    a.M, a.N, a.K = M, N, K
    a.bias, a.bias_per_row, a.act = ptr(bias), int(bias_per_row), act
    a.d0, a.ldd0, a.d1, a.ldd1 = d0, ldd0, d1, ldd1
    a.d0_cs = d0_cs
    a.csplit, a.accumulate = csplit, int(accumulate)
    if pa is not None:
        a.pah, a.pal, a.ldpa = pa.hi.data_ptr() + pa_off * eb, pa.lo.data_ptr() + pa_off * eb, ldpa
        a.pa_blocked = int(bool(pa_blocked))
    if pb is not None:
        a.pbh, a.pbl, a.ldpb, a.addvec = pb.hi.data_ptr(), pb.lo.data_ptr(), ldpb, ptr(addvec)
    a.nbatch, a.bsx, a.bsy, a.bsd, a.bsbias, a.bspa = nbatch, bsx, bsy, bsd, bsbias, bspa
    a.nsplit, a.tile = nsplit, tile
    a.ksplits, a.parts, a.part_stride = ksplits, ptr(parts), part_stride
    if not launch:
        return a
    check(load().rmem_linear(C.byref(a), stream_ptr()), "rmem_linear")


def linear_grouped(args):
    """One launch for up to 8 problems built with linear(..., launch=False)."""
    arr = (LinearArgs * len(args))(*args)
    check(load().rmem_linear_grouped(arr, len(args), stream_ptr()), "rmem_linear_grouped")


def groupnorm_nchw(x: torch.Tensor, gn: torch.nn.GroupNorm, relu: bool, conv_bias=None) -> torch.Tensor:
    """GroupNorm(+ReLU) of an NCHW fp32 tensor through rmem_groupnorm_nchw (one pair of launches per
    sample: the statistics are per sample anyway); with `conv_bias`, x is the output of a bias-free
    convolution and the bias is added in the kernel."""
    x = x.contiguous()
    n, c, h, w = x.shape
    if x.dtype != torch.float32 or ((c // gn.num_groups) * h * w) % 4:
        if conv_bias is not None:
            x = x + conv_bias.view(1, -1, 1, 1)
        y = torch.nn.functional.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
        return torch.relu_(y) if relu else y
    # per-call workspace (stream-ordered caching allocator / graph pool): engines running
    # concurrently on different streams must not share it; the stats pass overwrites all of it
    ws = torch.empty(n, 2 * 32 * gn.num_groups, dtype=torch.float64, device=x.device)
    y = torch.empty_like(x)
    lib, st = load(), stream_ptr()
    for i in range(n):
        if conv_bias is not None:
            check(lib.rmem_groupnorm_nchw_bias(x[i].data_ptr(), conv_bias.data_ptr(), y[i].data_ptr(), c, h * w,
                                               gn.num_groups, gn.weight.data_ptr(), gn.bias.data_ptr(), gn.eps,
                                               int(relu), ws[i].data_ptr(), st), "rmem_groupnorm_nchw_bias")
        else:
            check(lib.rmem_groupnorm_nchw(x[i].data_ptr(), y[i].data_ptr(), c, h * w, gn.num_groups,
                                          gn.weight.data_ptr(), gn.bias.data_ptr(), gn.eps, int(relu),
                                          ws[i].data_ptr(), st), "rmem_groupnorm_nchw")
    return y


def bias_act_nchw_(x: torch.Tensor, bias: torch.Tensor, residual=None, relu: bool = True) -> torch.Tensor:
    """In-place x = act(x + bias[c] (+ residual)) for a contiguous NCHW fp32 tensor."""
    n, c, h, w = x.shape
    if not x.is_contiguous() or x.dtype != torch.float32 or h * w < 4 or \
            (residual is not None and not residual.is_contiguous()):
        y = x + bias.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return torch.relu_(y) if relu else y
    if n != 1:
        check(load().rmem_bias_act_nchw_batched(x.data_ptr(), bias.data_ptr(), ptr(residual), n, c, h * w,
                                                int(relu), stream_ptr()), "rmem_bias_act_nchw_batched")
        return x
    check(load().rmem_bias_act_nchw(x.data_ptr(), bias.data_ptr(), ptr(residual), c, h * w, int(relu),
                                    stream_ptr()), "rmem_bias_act_nchw")
    return x


def upsample_add_nchw_(y: torch.Tensor, bias, x: torch.Tensor, align_corners: bool, inplace: bool = True) -> torch.Tensor:
    """y = (y + bias[c]) + bilinear(x -> y's size) for contiguous NCHW fp32 maps (one launch per
    sample); in place, or into a new tensor (inplace=False: y is left untouched)."""
    n, c, H, W = y.shape
    if x.shape[0] != n or x.shape[1] != c or not y.is_contiguous() or y.dtype != torch.float32 or c * H > 65535:
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        if x.shape[-2:] != y.shape[-2:]:
            x = torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=align_corners)
        return y + x
    x = x.contiguous()
    lib, st = load(), stream_ptr()
    out = y if inplace else torch.empty_like(y)
    for i in range(n):
        if not inplace:
            check(lib.rmem_upsample_add_nchw_out(y[i].data_ptr(), out[i].data_ptr(), ptr(bias), x[i].data_ptr(), c, H, W,
                                                 x.shape[2], x.shape[3], int(bool(align_corners)), st),
                  "rmem_upsample_add_nchw_out")
        else:
            check(lib.rmem_upsample_add_nchw(y[i].data_ptr(), ptr(bias), x[i].data_ptr(), c, H, W, x.shape[2],
                                             x.shape[3], int(bool(align_corners)), st), "rmem_upsample_add_nchw")
    return out


def set_ints(dst: torch.Tensor, values, offset: int = 0, count: int = None):
    """dst[offset : offset + count] = values (int32 device tensor; count defaults to all 32 - offset entries,
    zero-filled past the values), stream-ordered, no host-blocking copy."""
    n = (32 - offset) if count is None else int(count)
    arr = (i32 * 32)(*(list(values) + [0] * (32 - len(values))))
    check(load().rmem_set_ints(dst.data_ptr() + 4 * offset, arr, n, stream_ptr()), "rmem_set_ints")


def labels_from_logits(logits_list, flips, out_hw, align_corners: bool, out: torch.Tensor = None) -> torch.Tensor:
    """uint8 label map [H0,W0] = argmax_c mean_aug softmax(unflip(bilinear(logits -> out_hw))).
    logits_list: batch-1 fp32 [1,C,h,w] (or [C,h,w]) contiguous device tensors, one per augmentation."""
    H0, W0 = int(out_hw[0]), int(out_hw[1])
    srcs = (LabelSrc * len(logits_list))()
    Cn = None
    for i, (lg, fl) in enumerate(zip(logits_list, flips)):
        if lg.dtype != torch.float32 or not lg.is_contiguous() or not lg.is_cuda:
            raise RmemError("labels_from_logits needs contiguous fp32 device logits")
        c, h, w = lg.shape[-3:]
        if lg.numel() != c * h * w or (Cn is not None and c != Cn):
            raise RmemError("labels_from_logits: batch 1 and equal channel counts only")
        Cn = c
        srcs[i].logits, srcs[i].h, srcs[i].w, srcs[i].flip = lg.data_ptr(), h, w, int(bool(fl))
    if out is None:
        out = torch.empty((H0, W0), dtype=torch.uint8, device=logits_list[0].device)
    check(load().rmem_labels_from_logits(srcs, len(logits_list), Cn, int(bool(align_corners)), H0, W0,
                                         out.data_ptr(), stream_ptr()), "rmem_labels_from_logits")
    return out


def label_resize_nearest(src: torch.Tensor, size, flip: bool = False, out: torch.Tensor = None) -> torch.Tensor:
    """F.interpolate(mode="nearest") (+ horizontal flip first) on a uint8 [H,W] label map."""
    if src.dtype != torch.uint8 or src.dim() != 2 or not src.is_contiguous():
        raise RmemError("label_resize_nearest needs a contiguous uint8 [H,W] map")
    Hd, Wd = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((Hd, Wd), dtype=torch.uint8, device=src.device)
    check(load().rmem_label_resize_nearest(src.data_ptr(), src.shape[0], src.shape[1], out.data_ptr(), Hd, Wd,
                                           int(bool(flip)), stream_ptr()), "rmem_label_resize_nearest")
    return out


class Recording:
    """A recorded launch sequence of the memory path (include/rmem_hip.h, "several clips' memory
    banks in one launch"): `with Recording() as r: <launch code>` launches nothing and leaves the
    argument blob in r.blob (bytes), the op count in r.count and the geometry hash in r.signature."""

    def __init__(self):
        self.handle = None
        self.blob = b""
        self.count = 0
        self.signature = 0

    def __enter__(self):
        self.handle = load().rmem_rec_begin()
        if not self.handle:
            raise RmemError("rmem_rec_begin: this thread is already recording")
        return self

    def __exit__(self, et, ev, tb):
        lib = load()
        check(lib.rmem_rec_end(self.handle), "rmem_rec_end")
        if et is None:
            n = lib.rmem_rec_size(self.handle)
            self.blob = C.string_at(lib.rmem_rec_data(self.handle), n) if n > 0 else b""
            self.count = lib.rmem_rec_count(self.handle)
            self.signature = lib.rmem_rec_signature(self.handle)
        return False

    def launch(self, dev_args: torch.Tensor, clip_stride: int, B: int):
        check(load().rmem_launch_recorded(self.handle, dev_args.data_ptr(), clip_stride, B, stream_ptr()),
              "rmem_launch_recorded")

    def __del__(self):
        if self.handle and _LIB is not None:
            _LIB.rmem_rec_free(self.handle)
            self.handle = None
