#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats table (like --stats CSV).
usage: summarize_rocpd.py <results.db> [<out.md>]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, grid_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, duration from kernels").fetchall()
    agg = {}
    for name, gx, vg, ag, sg, lds, scr, dur in rows:
        key = (name, vg, ag, sg, lds, scr)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds | scratch |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for (name, vg, ag, sg, lds, scr), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{name[:110]}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / tot:.1f} | {vg} | {ag} | {sg} | {lds} | {scr} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(txt + "\n")


if __name__ == "__main__":
    main()
