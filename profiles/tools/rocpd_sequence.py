#!/usr/bin/env python3
"""The LAST n kernel dispatches of a rocprofv3 rocpd trace in start order: name, duration, gap to the previous dispatch's end (one update of a captured plan, launch by launch).
Usage: python profiles/tools/rocpd_sequence.py <results.db> [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
prev = None
for name, s, e in rows:
  print(f'{name.split("(")[0][:40]:40s} {(e - s) / 1e3:8.2f} us   gap {((s - prev) / 1e3 if prev else 0):7.2f} us')
  prev = e
print(f'span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us')
