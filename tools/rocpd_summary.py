#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace --stats) into a small text/CSV table.

usage: python tools/rocpd_summary.py <results.db> [out.csv]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        f"group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"\"{n}\",{c},{s},{a:.1f},{mn},{mx},{100.0 * s / total:.2f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
