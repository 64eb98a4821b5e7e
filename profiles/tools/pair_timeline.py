"""Developer tool: workgroup-level timeline of ONE steady-state SAC+GAIL update of the headline schedule in PAIR MODE (k_sac_chain_pair / k_policy_critic_pair), from a
-DIL_TIMELINE build (see update_timeline.py for the build lines and the other schedule):

  bash profiles/tools/build_variants.sh tl:"-DIL_TIMELINE -w"
  IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/pair_timeline.py [replays]

Times are microseconds relative to the moment the sampler workgroup signals [IL_SYNC_INDICES] for the LAST replayed update; min / median / max over the workgroups of a role."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from imitation_learning_amd import _lib

K, W, S = 12, 512, 8   # IL_TL_K, IL_TL_WGS, IL_TL_SLOTS (csrc/il_common.hpp)
dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
plan.capture(warmup=3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(n): plan.replay()
torch.cuda.synchronize()
assert plan.sync_timeouts() == 0
raw = C.CDLL(_lib.LIB_PATH)


def read(fn):
  buf = (C.c_ulonglong * (K * W * S))()
  assert getattr(raw, fn)(buf) == 0
  return np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.int64)


sac, gail = read('il_debug_timeline_sac'), read('il_debug_timeline_gail')
B, nt = plan.B, plan.B // 16
t0 = gail[0, nt, 3]
us = lambda a: (np.asarray(a, np.float64) - t0) / 100.0


def row(name, a):
  a = np.asarray(a, np.float64)
  a = a[np.isfinite(a)]
  if a.size == 0:
    print(f'  {name:58s} (no workgroups)'); return
  print(f'  {name:58s} {a.min():8.2f} {np.median(a):8.2f} {a.max():8.2f}   (n = {a.size})')


print(f'B = {B}; PAIR MODE; all times in us after the sampler signalled [IL_SYNC_INDICES]; min / median / max over workgroups')
print('discriminator branch (side stream)')
sam = gail[0, nt]
row('sampler: previous update over ([IL_SYNC_MAIN_EPOCH] seen)', us([sam[2]]))
row('sampler: indices drawn (signal issued)', us([sam[3]]))
gw = np.array([x + (nt + 1) * y for y in range(3) for x in range(nt)])
row('k_gail_grad: launched', us(gail[0, gw, 0]))
row('k_gail_grad: indices seen', us(gail[0, gw, 2]))
row('k_gail_grad: done', us(gail[0, gw, 7]))
nr = int((gail[1, :, 7] >= t0).sum())
row('k_gail_reduce: launched', us(gail[1, :nr, 0]))
row('k_gail_reduce: done ([IL_SYNC_PARAMS] signalled)', us(gail[1, :nr, 7]))
print('SAC branch (main stream)')
c = sac[10]
live = c[:, 0] >= t0 - 2000   # workgroups of the last replay
nwg = int(live.sum())
relabel = 1 if nwg >= 10 * nt else 0
seg = {'a1': (0, nt), 'a0': (nt, 2 * nt), 't1': (2 * nt, 4 * nt), 't0': (4 * nt, 6 * nt)}
o = 6 * nt
if relabel: seg['r'] = (o, o + nt); o += nt
seg['c'] = (o, o + 2 * nt); o += 2 * nt
seg['s'] = (o, o + nt); o += nt
seg['g'] = (o, nwg)
sl = lambda k: c[seg[k][0]:seg[k][1]]
row('k_sac_chain_pair: launched', us(c[:nwg, 0]))
for k, name in (('a1', "actor(s') half 1"), ('a0', "actor(s') half 0")):
  a = sl(k)
  row(f'  {name}: prologue done', us(a[:, 1])); row(f'  {name}: rows in LDS', us(a[:, 2])); row(f'  {name}: layer 1 done', us(a[:, 3])); row(f'  {name}: layer 2 done', us(a[:, 4]))
a = sl('a1'); row("  actor(s') half 1: published", us(a[:, 7]))
a = sl('a0'); row("  actor(s') half 0: partner's half received", us(a[:, 5])); row("  actor(s') half 0: head + sample done", us(a[:, 6])); row("  actor(s') half 0: arrival signalled", us(a[:, 7]))
for k, name in (('t1', 'targets half 1'), ('t0', 'targets half 0')):
  a = sl(k)
  row(f"  {name}: actor(s') of the tile seen", us(a[:, 2])); row(f'  {name}: layer 1 done', us(a[:, 3])); row(f'  {name}: layer 2 done', us(a[:, 4]))
a = sl('t1'); row('  targets half 1: published', us(a[:, 7]))
a = sl('t0'); row("  targets half 0: partner's half received", us(a[:, 5])); row('  targets half 0: Q done', us(a[:, 6])); row('  targets half 0: arrival signalled', us(a[:, 7]))
if relabel:
  a = sl('r'); row('  relabel: rows loaded', us(a[:, 1])); row('  relabel: discriminator step seen ([IL_SYNC_PARAMS])', us(a[:, 2])); row('  relabel: rewards written', us(a[:, 6])); row('  relabel: arrival signalled', us(a[:, 7]))
a = sl('c'); row('  critics: forward done', us(a[:, 2])); row('  critics: backward GEMM done', us(a[:, 3])); row('  critics: targets + rewards of the tile seen', us(a[:, 5])); row('  critics: done', us(a[:, 7]))
row('  actor(s): done', us(sl('s')[:, 7]))
row('  row-copy workgroups: done', us(sl('g')[:, 7]))


def dw_rows(kid, name, n_blocks):
  m = sac[kid, :, 7] >= t0
  row(f'{name}: launched', us(sac[kid, m, 0]))
  row(f'{name}: done', us(sac[kid, m, 7]))
  idx = np.where(m)[0]
  for lo, hi, what in ((0, n_blocks, '32 x 32 LDS block jobs (every layer + biases)'), (n_blocks, 10 ** 6, 'tail blocks (polyak, alpha)')):
    sel = idx[(idx >= lo) & (idx < hi)]
    if sel.size:
      row(f'    {what}: done', us(sac[kid, sel, 7]))


H = 256
jobs = lambda IN, OUT: (H // 32) ** 2 + (H // 32) * ((IN + 31) // 32) + ((OUT + 31) // 32) * (H // 32)
dw_rows(1, 'k_dw_adam (critic)', 2 * jobs(bench.S + bench.A, 1))
p = sac[11]
for lo, hi, name in ((0, 2 * nt, 'half 1'), (2 * nt, 4 * nt, 'half 0')):
  a = p[lo:hi]
  row(f'k_policy_critic_pair {name}: launched', us(a[:, 0])); row(f'  {name}: rows + panel requested', us(a[:, 1])); row(f'  {name}: layer 1 done', us(a[:, 2])); row(f'  {name}: layer 2 done, half published', us(a[:, 3]))
  row(f"  {name}: partner's half received", us(a[:, 4])); row(f'  {name}: Q + mask done', us(a[:, 5])); row(f'  {name}: layer 2 backward done', us(a[:, 6])); row(f'  {name}: {"published" if lo == 0 else "dQ/da done, arrival signalled"}', us(a[:, 7]))
h = sac[3][4 * nt:8 * nt]
h = h[h[:, 7] >= t0]
row('  helpers: own work done, waiting for the critics', us(h[:, 1]))
row('  helpers: both critics of the tile seen', us(h[:, 2]))
row('  helpers: done', us(h[:, 7]))
dw_rows(2, 'k_dw_adam (actor) + tail', jobs(bench.S, 2 * bench.A))
