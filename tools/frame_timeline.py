#!/usr/bin/env python
"""Timeline of one steady-state frame from a rocprofv3 kernel_trace CSV: kernels in start order with
their hardware queue, per-queue busy time, and the gaps on the critical (LSTT) chain.
Usage: tools/frame_timeline.py <kernel_trace.csv> [frame_index_from_end=4] [--brief]"""
import csv
import sys


def main():
    path = sys.argv[1]
    fr = -int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else -4
    brief = "--brief" in sys.argv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("gn2_apply_kernel")]
    a, b = marks[fr - 1] + 1, marks[fr] + 1
    t0 = int(rows[a]["Start_Timestamp"])
    seg = rows[a:b]
    print(len(seg), "kernels", (int(seg[-1]["End_Timestamp"]) - t0) / 1e3, "us")
    qs = {}
    for r in seg:
        q = r["Queue_Id"]
        qs[q] = qs.get(q, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("busy per queue (us):", {k: round(v, 1) for k, v in qs.items()})
    for r in seg:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3
        e = (int(r["End_Timestamp"]) - t0) / 1e3
        n = r["Kernel_Name"]
        if brief and not any(k in n for k in ("pe_bias", "pv_kernel", "linear_grouped", "gn2", "labels_kernel", "id_assign")):
            continue
        print(f"{s:8.1f} {e - s:7.1f} q{r['Queue_Id']} {n[:70]}")


if __name__ == "__main__":
    main()
