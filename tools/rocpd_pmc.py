#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (separate --pmc passes).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'evr::', '', name)
    name = re.sub(r'\(.*\)$', '', name)
    return name if len(name) <= 90 else name[:87] + '...'


def table(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    agg = {}
    for k, c, v, d in rows:
        a = agg.setdefault((short(k), c), [0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += d
    return agg


if __name__ == '__main__':
    agg = table(sys.argv[1])
    print("| kernel | counter | launches | avg per launch | avg us |")
    print("|---|---|---:|---:|---:|")
    for (k, c), (n, v, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {c} | {n} | {v / n:.1f} | {d / n / 1e3:.1f} |")
