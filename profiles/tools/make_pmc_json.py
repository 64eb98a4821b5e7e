"""Builds profiles/pmc_latest.json (what bench.py reports as roofline.traffic) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite).
Usage: IL_COMMIT=<short sha> python profiles/tools/make_pmc_json.py <FETCH db> <WRITE db> "<profiled command>" > profiles/pmc_latest.json
Kernel names are the launched kernels' own (k_sac_chain_pair, ...): bench.py refuses a file that does not list the kernels of the schedule it timed."""
import datetime, json, os, sqlite3, sys

def per_kernel(path, counter):
  agg = {}
  for name, c, v in sqlite3.connect(path).execute('select name, counter_name, counter_value from pmc_events'):
    if c != counter: continue
    k = name.split('(')[0]
    if k.startswith('k_'):
      a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(v)
  return {k: tot / n for k, (n, tot) in agg.items()}

fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), eager launches of bench.py --no-graph; per-launch averages in KB; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: '
               'gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md HBM section). k_dw_adam averages the critic and the actor launch; '
               'k_policy_critic includes the policy backward that runs as its tail.',
       'collected': datetime.datetime.utcnow().strftime('%Y-%m-%d %H:%M UTC'), 'commit': os.environ.get('IL_COMMIT', 'unknown'), 'command': sys.argv[3] if len(sys.argv) > 3 else None,
       'kernels': {}}
for k in sorted(set(fetch) | set(write)):
  f, w = fetch.get(k, 0.0), write.get(k, 0.0)
  out['kernels'][k] = {'FETCH_SIZE_KB': round(f, 3), 'WRITE_SIZE_KB': round(w, 3), 'traffic_bytes': int((2 * f + w) * 1024)}
print(json.dumps(out, indent=1))
