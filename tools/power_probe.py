"""Is the bank read clock-throttled?  Runs one read configuration in a loop for a few seconds while a thread samples the
board's power and shader clock (amdgpu hwmon / rocm-smi), beside a plain fp16 GEMM (hipBLASLt) as the known
MFMA-heavy load.  Prints one JSON object.  GPU only; measurement tool, not part of the product path."""
import argparse, ctypes as C, glob, json, math, os, subprocess, sys, threading, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmem_amd import hip


def sensors():
    out = {}                                          # every card of the host is listed; the busy one stands out
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        card = hw.split("/")[4]
        for name in ("power1_input", "freq1_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out[card + ":" + name] = p
    return out


class Sampler(threading.Thread):
    def __init__(self, paths):
        super().__init__(daemon=True)
        self.paths, self.rows, self.stop = paths, [], False

    def run(self):
        while not self.stop:
            row = {}
            for k, p in self.paths.items():
                try:
                    row[k] = int(open(p).read().strip())
                except Exception:
                    pass
            if not self.paths:
                try:
                    txt = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
                    row["smi"] = txt.strip()[:400]
                except Exception as e:
                    row["smi"] = repr(e)
            self.rows.append(row)
            time.sleep(0.05 if self.paths else 0.5)


def summarize(rows):
    res = {"samples": len(rows)}
    keys = sorted({k for r in rows for k in r if k != "smi"})
    for k in keys:
        v = [r[k] / 1e6 for r in rows if k in r]
        v = v[len(v) // 3:]
        if v:
            res[k + ("_W" if "power" in k else "_MHz") + "_mean_min_max"] = [round(sum(v) / len(v), 1), round(min(v), 1), round(max(v), 1)]
    smi = [r["smi"] for r in rows if "smi" in r]
    if smi:
        res["smi_last"] = smi[-1]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = hip.load()
    paths = sensors()
    res = {"sensors": paths}
    h, w, T = 31, 54, 4
    N = h * w
    Np = (N + 127) // 128 * 128
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    K = hip.Planes.from_f32((rnd(T, Np, 128) * 1.5).to(dev))
    V = hip.Planes.from_f32(rnd(T, Np // 16, 1024, 16).to(dev))
    Q = hip.Planes.from_f32((rnd(Np, 128) * 1.5).to(dev))
    bias = (rnd(N, T) * 3).to(dev)
    smap = torch.arange(16, dtype=torch.int32, device=dev)
    part = torch.zeros(9, Np, 1024, device=dev)
    ml = torch.zeros(9, Np, 2, device=dev)
    st = hip.stream_ptr()

    def reader(ks):
        ra = hip.ReadArgs()
        ra.mode, ra.qh, ra.ql = 0, Q.hi.data_ptr(), Q.lo.data_ptr()
        ra.kh, ra.kl, ra.k_slot_stride = K.hi.data_ptr(), K.lo.data_ptr(), Np * 128
        ra.vh, ra.vl, ra.v_slot_stride = V.hi.data_ptr(), V.lo.data_ptr(), 1024 * Np
        ra.slot_map = smap.data_ptr()
        ra.T, ra.N, ra.Npad, ra.ncols, ra.scale = T, N, Np, 1024, 1.0 / math.sqrt(128)
        ra.bias, ra.R, ra.ldr = bias.data_ptr(), None, 232
        ra.h, ra.w, ra.ksplits = h, w, ks
        ra.part, ra.ml, ra.lslot = part.data_ptr(), ml.data_ptr(), None
        return lambda: hip.check(lib.rmem_attn_read(C.byref(ra), st), "read")

    A = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    B = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    loads = {"idle": None, "read_ks9_243_units": reader(9), "read_ks6_162_units": reader(6), "read_ks3_81_units": reader(3),
             "gemm_fp16_8192": lambda: torch.mm(A, B)}
    for name, fn in loads.items():
        s = Sampler(paths)
        s.start()
        t0 = time.time()
        n = 0
        if fn is None:
            time.sleep(1.0)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.time() - t0 < args.seconds:
                for _ in range(50):
                    fn()
                n += 50
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
        s.stop = True
        s.join()
        ent = summarize(s.rows)
        if fn is not None:
            ent["us_per_call"] = round(e0.elapsed_time(e1) * 1e3 / n, 2)
        res[name] = ent
    if "gemm_fp16_8192" in res:
        res["gemm_fp16_8192"]["TFLOPs"] = round(2 * 8192 ** 3 / (res["gemm_fp16_8192"]["us_per_call"] * 1e-6) / 1e12, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
