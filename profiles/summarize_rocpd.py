#!/usr/bin/env python3
"""Summarises a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / avg / min / max / total.
Usage: python profiles/summarize_rocpd.py gpurun_out/prof/<name>_results.db [--skip-first N] > profiles/<round>_kernel_stats.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, start, end from kernels order by start"))
stats = {}
for name, s, e in rows:
  d = (e - s) / 1e3
  name = name.split('(')[0]
  st = stats.setdefault(name, [0, 0.0, 1e18, 0.0])
  st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
tot = sum(v[1] for v in stats.values())
print('| kernel | calls | avg us | min us | max us | total ms | % |')
print('|---|---|---|---|---|---|---|')
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
  print(f'| {k[:70]} | {v[0]} | {v[1] / v[0]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {v[1] / 1e3:.3f} | {100 * v[1] / tot:.1f} |')
if rows:
  print(f'\nwall span of the trace: {(rows[-1][2] - rows[0][1]) / 1e6:.2f} ms, kernel time {tot / 1e3:.2f} ms')
