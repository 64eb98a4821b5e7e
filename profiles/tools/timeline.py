"""Timeline of ONE steady-state update from a rocprofv3 kernel trace (rocpd sqlite): start offset, duration and idle gap per kernel, so the
critical path through the two-stream graph can be read off.  Usage: python profiles/tools/timeline.py <results.db> [update index from the end] [anchor kernel prefix]
The anchor (default k_sample, the first kernel of an update in the round-1 schedules; k_sac_chain for the resident-sampler schedule) marks where an update starts."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = [(n.split('(')[0], s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_sample'
starts = [i for i, r in enumerate(rows) if r[0].startswith(anchor)]
i0, i1 = starts[-back], starts[-back + 1]
t0 = rows[i0][1]
print(f'update = kernels {i0}..{i1 - 1}; period {(rows[i1][1] - t0) / 1e3:.2f} us')
print(f'{"kernel":28s} {"start":>8s} {"end":>8s} {"dur":>7s}')
for n, s, e in rows[i0:i1]:
  print(f'{n[:28]:28s} {(s - t0) / 1e3:8.2f} {(e - t0) / 1e3:8.2f} {(e - s) / 1e3:7.2f}')
# averages over the last 200 updates
import collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for a, b in zip(starts[-201:-1], starts[-200:]):
  base = rows[a][1]
  seen = collections.Counter()
  for n, s, e in rows[a:b]:
    seen[n] += 1
    key = f'{n}#{seen[n]}'
    acc[key][0] += (s - base) / 1e3; acc[key][1] += (e - base) / 1e3; acc[key][2] += 1
print('\naverage over 200 updates (start, end offsets in us):')
for k, (s, e, c) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
  print(f'{k[:30]:30s} {s / c:8.2f} {e / c:8.2f} {(e - s) / c:7.2f}')
