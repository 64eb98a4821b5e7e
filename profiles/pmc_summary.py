#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 --pmc counter from the rocpd sqlite output.  Usage: pmc_summary.py <db> [<db> ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
  db = sqlite3.connect(path)
  cur = db.cursor()
  cols = [r[1] for r in cur.execute('pragma table_info(pmc_events)')]
  names = [r[1] for r in cur.execute('pragma table_info(kernels)')]
  try:
    rows = list(cur.execute('select name, counter_name, counter_value from pmc_events'))
  except Exception as e:
    print('# schema:', cols, names, e)
    continue
  agg = {}
  for name, counter, value in rows:
    a = agg.setdefault((name.split('(')[0], counter), [0, 0.0])
    a[0] += 1; a[1] += float(value)
  print(f'## {path}\n| kernel | counter | launches | avg per launch |\n|---|---|---|---|')
  for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if not k.startswith('void at::') and not k.startswith('__amd'):
      print(f'| {k[:60]} | {c} | {n} | {tot / n:.3f} |')
