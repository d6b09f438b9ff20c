#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite `*_results.db` or `*_kernel_stats.csv`) into a
markdown table for profiles/.  Usage: tools/prof_summary.py <db-or-csv> <frames> > out.md"""
import csv
import sqlite3
import sys


def short_name(n: str) -> str:
    """rmem::k_one<Args, &body, 256> / rmem::k_many<...> (csrc/launch.h) -> `body [one clip]` /
    `body [clips batched]`; rocprofv3 reports these template instances mangled."""
    import re
    m = re.match(r"_ZN4rmem(5k_one|6k_many)I.*?EXadL_Z\d+([A-Za-z0-9_]+?)(I[A-Za-z0-9_]*?Ev)?RK", n)
    if m:
        tp = m.group(3) or ""
        tp = "<" + ",".join(re.findall(r"Li(\d+)E", tp)) + ">" if tp else ""
        return f"{m.group(2)}{tp} [{'one clip' if m.group(1) == '5k_one' else 'clips batched'}; rmem::{m.group(1)[1:]}]"
    return n


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4]))
            for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def from_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return rows


def trace_rows_from_db(path):
    """Dispatch rows of a rocpd database (rocprofv3's default output of this ROCm: `kernels` view)."""
    cur = sqlite3.connect(path).cursor()
    return [{"Kernel_Name": r[0], "Start_Timestamp": r[1], "End_Timestamp": r[2]}
            for r in cur.execute("select name, start, end from kernels")]


def from_trace(path, frames):
    """Steady-state summary from a rocprofv3 kernel trace (CSV or rocpd .db): only the dispatches of the last
    `frames` frames (delimited by the once-per-frame gn2_apply_kernel; AOT block: id_assign_kernel) are counted, which
    excludes MIOpen find-mode / first-call kernels of the warm-up."""
    rows = trace_rows_from_db(path) if path.endswith(".db") else list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "gn2_apply_kernel" in r["Kernel_Name"]]
    if not marks:          # the AOT block has no final GroupNorm: its frames are delimited by the once-per-frame ID assignment
        marks = [i for i, r in enumerate(rows) if "id_assign_kernel" in r["Kernel_Name"]]
    start = marks[-frames - 1] + 1 if len(marks) > frames else 0
    end = marks[-1] + 1
    agg = {}
    for r in rows[start:end]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in agg.values())
    # wall-clock span and busy union of the window (kernels overlap across streams / graph branches)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[start:end])
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    global SPAN, PV_LONG
    SPAN = ((iv[-1][1] - iv[0][0]) / 1e3 / frames, busy / 1e3 / frames)
    # the dominant kernel of bench.py's roofline entry: the long-term P.V launches are the first
    # pv_kernel launch of every layer on the long chain = the longest third of the pv_kernel dispatches
    pv = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[start:end]
                if r["Kernel_Name"].startswith(("read64x2_kernel", "read64x2_pull_kernel", "read2_kernel")))
    if pv:
        PV_LONG = (len(pv), sum(pv) / len(pv), min(pv), max(pv))
        # per layer: the launches of a frame in time order are layers 0, 1, 2 (bench.py's HIP events sample layer 0)
        global PV_LAYER
        seq = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[start:end]
               if r["Kernel_Name"].startswith(("read64x2_kernel", "read64x2_pull_kernel", "read2_kernel"))]
        seq.sort()
        if len(seq) % 3 == 0:
            PV_LAYER = [sum(d for _, d in seq[l::3]) / (len(seq) // 3) for l in range(3)]
    return [(k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot) for k, v in agg.items()]


SPAN = None
PV_LONG = None
PV_LAYER = None


def main():
    path, frames = sys.argv[1], int(sys.argv[2])
    if path.endswith("kernel_trace.csv") or (path.endswith(".db") and "--top" not in sys.argv):
        rows = from_trace(path, frames)
    else:
        rows = from_csv(path) if path.endswith(".csv") else from_db(path)
    rows.sort(key=lambda r: -r[2])
    rows = [(short_name(r[0]),) + tuple(r[1:]) for r in rows]
    tot = sum(r[2] for r in rows)
    ours = ("read64", "fg_weights", "bank_policy", "bank_edit", "read2_kernel", "read_kernel", "read_combine", "linear_kernel", "linear_grouped", "linear_stream", "pv_kernel", "pv16_kernel", "scores_kernel", "scores2_kernel",
            "combine_kernel", "combine2_kernel", "dwconv5x5",
            "layernorm_", "gn2_", "gn_nchw", "gn_tok", "id_assign", "pe_bias", "mass_reduce", "split_planes",
            "bias_act_nchw", "upsample_add", "labels_kernel", "label_resize", "set_ints", "mha_", "transpose_planes",
            "add_split")
    mine = sum(r[2] for r in rows if any(k in r[0] for k in ours))
    print(f"source: {path}\n\nframes profiled: {frames}; total kernel time {tot/1e3:.2f} ms "
          f"= {tot/frames:.1f} us/frame; rmem_amd kernels {mine/frames:.1f} us/frame "
          f"({100*mine/tot:.1f} %), PyTorch/MIOpen/rocBLAS (encoder, decoder, glue) "
          f"{(tot-mine)/frames:.1f} us/frame\n")
    if SPAN:
        print(f"wall-clock span {SPAN[0]:.1f} us/frame, GPU busy (union over streams) {SPAN[1]:.1f} us/frame: "
              f"{tot/frames - SPAN[1]:.1f} us/frame of kernel time runs concurrently with other kernels\n")
    if PV_LONG:
        print(f"`read64x2_kernel` launches (bench.py roofline kernel: fused long-term + windowed memory read of a layer): "
              f"{PV_LONG[0]} launches, mean {PV_LONG[1]:.1f} us (min {PV_LONG[2]:.1f}, max {PV_LONG[3]:.1f}) "
              f"-- inside a frame, beside the encoder stream"
              + (f"; per layer (launch order within a frame) {PV_LAYER[0]:.1f} / {PV_LAYER[1]:.1f} / {PV_LAYER[2]:.1f} us -- bench.py's events "
                 f"bracket the layer-0 launch of sampled frames" if PV_LAYER else "") + "\n")
    print("| kernel | calls | calls/frame | total us | avg us | % |")
    print("|---|---|---|---|---|---|")
    for n, c, t, a, p in rows[:40]:
        print(f"| `{n[:100]}` | {c} | {c/frames:.1f} | {t:.0f} | {a:.2f} | {100*t/tot:.2f} |")


if __name__ == "__main__":
    main()
