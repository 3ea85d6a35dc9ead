#!/usr/bin/env python3
"""Turn rocprofv3 output (kernel_stats.csv / kernel_trace.csv / rocpd .db) into a small text
summary that is committed under profiles/.

    python tools/prof_summary.py <kernel_stats.csv | kernel_trace.csv | results.db> [--top 40]
        [--last-ms T]      only dispatches that started in the last T ms of a kernel trace
"""
import argparse
import collections
import csv
import sqlite3
import sys


def from_stats(path):
    rows = list(csv.DictReader(open(path)))
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows]


def from_trace(path, last_ms):
    ev = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    if last_ms:
        t1 = ev[-1][1]
        ev = [e for e in ev if e[0] >= t1 - last_ms * 1e6]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in ev:
        agg[n][0] += 1
        agg[n][1] += e - s
    span = (ev[-1][1] - ev[0][0]) if ev else 0
    return [(n, c, t) for n, (c, t) in agg.items()], span


def from_db(path):
    c = sqlite3.connect(path)
    return [(r[0], int(r[1]), float(r[2]) * 1e3) for r in c.execute(
        "select name, total_calls, total_duration from top_kernels")]   # total_duration in us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--last-ms", type=float, default=0.0)
    a = ap.parse_args()
    span = 0
    if a.path.endswith(".db"):
        rows = from_db(a.path)
    elif a.path.endswith("kernel_trace.csv"):
        rows, span = from_trace(a.path, a.last_ms)
    else:
        rows = from_stats(a.path)
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"# source: {a.path}" + (f" (last {a.last_ms:.0f} ms window)" if a.last_ms else ""))
    print(f"# kernels: {len(rows)}  dispatches: {sum(r[1] for r in rows)}  total kernel time: {tot / 1e6:.2f} ms"
          + (f"  window span: {span / 1e6:.2f} ms  busy: {tot / span:.1%}" if span else ""))
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10}  name")
    for n, c, t in rows[:a.top]:
        print(f"{t / 1e6:10.2f} {100 * t / tot:5.1f}% {c:7d} {t / c / 1e3:10.1f}  {n[:160]}")


if __name__ == "__main__":
    sys.exit(main())
