#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace as a per-kernel table, like `--stats` prints.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--md] > profiles/...
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'evr::', '', name)
    name = re.sub(r'\(.*\)$', '', name)          # drop the argument list
    return name if len(name) <= 90 else name[:87] + '...'


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count "
                      "from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, lds, vg, ag, sg in rows:
        a = agg.setdefault(short(name), dict(n=0, tot=0, mn=1 << 62, mx=0, lds=lds, vgpr=vg, agpr=ag, sgpr=sg, wg=wx))
        a['n'] += 1; a['tot'] += dur; a['mn'] = min(a['mn'], dur); a['mx'] = max(a['mx'], dur)
    total = sum(a['tot'] for a in agg.values()) or 1
    print(f"# rocprofv3 kernel-trace stats from {path.split('/')[-1]}: {len(rows)} dispatches, "
          f"{total / 1e6:.3f} ms of kernel time\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | wg | VGPR | AGPR | SGPR | LDS B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['tot']):
        print(f"| `{k}` | {a['n']} | {a['tot'] / 1e6:.3f} | {a['tot'] / a['n'] / 1e3:.2f} | {a['mn'] / 1e3:.2f} | "
              f"{a['mx'] / 1e3:.2f} | {100 * a['tot'] / total:.1f} | {a['wg']} | {a['vgpr']} | {a['agpr']} | "
              f"{a['sgpr']} | {a['lds']} |")


if __name__ == '__main__':
    main()
