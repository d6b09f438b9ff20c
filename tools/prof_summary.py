#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite `*_results.db` or `*_kernel_stats.csv`) into a
markdown table for profiles/.  Usage: tools/prof_summary.py <db-or-csv> <frames> > out.md"""
import csv
import sqlite3
import sys


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


def main():
    path, frames = sys.argv[1], int(sys.argv[2])
    rows = from_csv(path) if path.endswith(".csv") else from_db(path)
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    ours = ("linear_kernel", "pv_kernel", "scores_kernel", "combine_kernel", "dwconv5x5", "layernorm_split",
            "gn2_", "id_assign", "pe_bias", "mass_reduce", "split_planes", "frame_", "aot_")
    mine = sum(r[2] for r in rows if any(k in r[0] for k in ours))
    print(f"source: {path}\n\nframes profiled: {frames}; total kernel time {tot/1e3:.2f} ms "
          f"= {tot/frames:.1f} us/frame; rmem_amd kernels {mine/frames:.1f} us/frame "
          f"({100*mine/tot:.1f} %), PyTorch/MIOpen/rocBLAS (encoder, decoder, glue) "
          f"{(tot-mine)/frames:.1f} us/frame\n")
    print("| kernel | calls | calls/frame | total us | avg us | % |")
    print("|---|---|---|---|---|---|")
    for n, c, t, a, p in rows[:40]:
        print(f"| `{n[:100]}` | {c} | {c/frames:.1f} | {t:.0f} | {a:.2f} | {100*t/tot:.2f} |")


if __name__ == "__main__":
    main()
