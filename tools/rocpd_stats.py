#!/usr/bin/env python3
"""Kernel statistics (+ optional per-step timeline) from a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace`).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--timeline N]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--timeline", type=int, default=0, help="print the last N dispatches with start offsets (us)")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(name, [])
        d.append(e - s)
    total = sum(sum(v) for v in stats.values())
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}')
    out = "\n".join(lines)
    if a.csv:
        open(a.csv, "w").write(out + "\n")
    print(out)
    if a.timeline:
        t0 = rows[-a.timeline][1]
        for name, s, e in rows[-a.timeline:]:
            print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:7.2f} us  {name[:90]}")


if __name__ == "__main__":
    main()
