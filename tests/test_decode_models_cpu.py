"""Python models of two workgroup re-labellings of the HIP kernels (no GPU): what must hold for ANY launch shape is a bijection plus an ordering property the
device-side waits rely on. The formulas are restated from the kernels (imitation-learning_amd/csrc/mlp_tile.hpp pop_ids, csrc/sac.hip chain_decode_xcd and the
IL_PC_XCD_NETS decode of k_policy_critic); the -m gpu tests check the kernels themselves bit for bit."""
import pytest


def pop_ids(bx, by, nx, L):
  """Population launches, grid (nx workgroups per learner, L learners), dispatched in linear order g = by * nx + bx round-robin over 8 XCDs: learner l -> XCD l % 8."""
  g, full = by * nx + bx, (L >> 3) * 8 * nx
  if g < full:
    grp, r = divmod(g, 8 * nx)
    return r >> 3, grp * 8 + (r & 7)
  return bx, by


@pytest.mark.parametrize('nx,L', [(32, 8), (32, 18), (96, 64), (41, 16), (7, 9), (33, 3), (64, 1)])
def test_pop_ids_is_a_bijection_that_pins_a_learner_to_one_xcd_and_keeps_its_block_order(nx, L):
  seen, last_bx = set(), {}
  for g in range(nx * L):          # dispatch order
    by, bx = divmod(g, nx)
    nbx, nby = pop_ids(bx, by, nx, L)
    assert 0 <= nbx < nx and 0 <= nby < L and (nbx, nby) not in seen
    seen.add((nbx, nby))
    if nby < (L >> 3) * 8:
      assert g % 8 == nby % 8, 'a learner of a full group of 8 lives on XCD learner % 8'
    assert last_bx.get(nby, -1) < nbx, "a learner's workgroups are dispatched in increasing block order (a role only waits for lower-numbered workgroups of its learner)"
    last_bx[nby] = nbx
  assert len(seen) == nx * L


def chain_decode_xcd(bid):
  x, tile = bid & 7, bid >> 3
  if x == 0: return 0, 0, tile      # actor(s')
  if x <= 2: return 1, x - 1, tile  # targets
  if x <= 4: return 2, x - 3, tile  # critics
  if x == 5: return 3, 0, tile      # actor(s)
  return None                        # XCDs 6, 7: row-copy workgroups / idle


def pop_dw_ids(bx, by, nx, L, nb64):
  """csrc/sac.hip pop_dw_ids (IL_POP_DW_BIG_FIRST=1 builds; an A/B switch, off by default): the 64 x 64 block workgroups of ALL full groups of 8 learners first, then every
  other job; learners behind the last full group keep the natural decode."""
  lf, g = (L >> 3) * 8, by * nx + bx
  if 0 < nb64 < nx and g < lf * nx:
    nbig = lf * nb64
    if g < nbig:
      q = g >> 3
      return q % nb64, (q // nb64) * 8 + (g & 7)
    r = g - nbig
    q, nr = r >> 3, nx - nb64
    return nb64 + q % nr, (q // nr) * 8 + (r & 7)
  if g >= lf * nx:
    return bx, by
  return pop_ids(bx, by, nx, L)


@pytest.mark.parametrize('nx,L,nb64', [(72, 32, 32), (69, 18, 16), (69, 9, 16), (72, 3, 32), (69, 64, 16), (40, 16, 0)])
def test_pop_dw_ids_is_a_bijection_that_keeps_a_learner_on_its_xcd_and_puts_the_blocks_first(nx, L, nb64):
  seen, lf, first_small = set(), (L >> 3) * 8, None
  for g in range(nx * L):
    by, bx = divmod(g, nx)
    nbx, nby = pop_dw_ids(bx, by, nx, L, nb64)
    assert 0 <= nbx < nx and 0 <= nby < L and (nbx, nby) not in seen
    seen.add((nbx, nby))
    if nby < lf:
      assert g % 8 == nby % 8
      if nb64 and nbx >= nb64 and first_small is None: first_small = g
      if nb64 and nbx < nb64: assert first_small is None, 'every block workgroup of the full groups is dispatched before the first small job'
  assert len(seen) == nx * L


@pytest.mark.parametrize('nt', [1, 5, 16, 32])
def test_chain_decode_xcd_one_network_per_xcd_and_waits_only_on_lower_blocks(nt):
  where = {}
  for bid in range(8 * nt):
    d = chain_decode_xcd(bid)
    if d is None: continue
    role, net, tile = d
    assert (role, net, tile) not in where and tile < nt
    where[(role, net, tile)] = bid
    assert bid % 8 == {(0, 0): 0, (1, 0): 1, (1, 1): 2, (2, 0): 3, (2, 1): 4, (3, 0): 5}[(role, net)], 'every tile of a role-network sits on the same XCD'
  assert len(where) == 6 * nt
  for tile in range(nt):
    for net in (0, 1):
      assert where[(0, 0, tile)] < where[(1, net, tile)], 'a target waits for actor(s′) of its tile'
      for tnet in (0, 1):
        assert where[(1, tnet, tile)] < where[(2, net, tile)], 'a critic waits for both targets of its tile'


@pytest.mark.parametrize('nt,helpers', [(16, 4), (5, 4), (16, 6)])
def test_policy_critic_xcd_decode(nt, helpers):
  crit, help_ = {}, {}
  for bx in range(8 * nt):
    x, q = bx & 7, bx >> 3
    if x >= 2 + helpers: continue
    if x < 2: crit[(x, q)] = bx
    else:
      h = (x - 2) * nt + q
      tile, part = h % nt, h // nt
      assert tile == q and part == x - 2 and (tile, part) not in help_
      help_[(tile, part)] = bx
  assert len(crit) == 2 * nt and len(help_) == helpers * nt
  for (tile, part), bx in help_.items():
    assert crit[(0, tile)] < bx and crit[(1, tile)] < bx, 'a helper only waits for lower-numbered workgroups (both critics of its tile)'


def test_dw_block_xcd_rectangles():
  """csrc/sac.hip dw_block_job (IL_DW_XCD_BLOCKS): the 64 blocks of an H = 256 layer's dW re-labelled so that XCD x (= workgroup index % 8; a network's job list starts at a
  multiple of 8) owns the 2 x 4 rectangle n in {2 (x / 2), + 1}, k in {4 (x % 2) .. + 3}: a bijection, 2 + 4 operand panels per XCD instead of 8 + 1."""
  seen, panels = set(), {x: (set(), set()) for x in range(8)}
  for job in range(64):
    x, slot = job & 7, job >> 3
    nb, kb = 2 * (x >> 1) + (slot >> 2), 4 * (x & 1) + (slot & 3)
    seen.add((nb, kb))
    panels[x][0].add(nb); panels[x][1].add(kb)
  assert len(seen) == 64 and seen == {(n, k) for n in range(8) for k in range(8)}
  assert all(len(dz) == 2 and len(xp) == 4 for dz, xp in panels.values())
  # the job counts the host and the kernel agree on (dw_block_jobs): per network nbh^2 + nbh * ceil(IN / 32) + ceil(OUT / 32) * nbh, a multiple of 8 at H = 256
  jobs = lambda IN, H, OUT: (H // 32) ** 2 + (H // 32) * ((IN + 31) // 32) + ((OUT + 31) // 32) * (H // 32)
  assert jobs(24, 256, 1) == 80 and jobs(18, 256, 12) == 80 and jobs(120, 256, 1) % 8 == 0


def test_pwil_wave_ranking_equals_full_rank_counting():
  import numpy as np
  """csrc/pwil.hip pw_rank256: a key's rank among the 256 (distance, index) keys of a chunk = its rank inside its wave of 64 + per other wave the number of smaller keys
  found by a branch-free binary search (steps 64 .. 1) over that wave's sorted run padded to 128 slots with the largest key. Must equal counting every key - ties on the
  distance (consumed atoms all carry FLT_MAX), exhausted lists and pad keys included."""
  rng = np.random.RandomState(5)
  for case in range(6):
    dist = rng.uniform(0, 4, 256).astype(np.float32)
    if case >= 1: dist[rng.choice(256, 90, replace=False)] = np.finfo(np.float32).max   # consumed atoms / rows past the end of the set
    if case >= 2: dist[rng.choice(256, 40, replace=False)] = dist[3]                      # exact ties: the index decides
    if case == 5: dist[:] = 1.5
    key = (dist.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(256, dtype=np.uint64)
    want = np.array([(key < k).sum() for k in key])
    runs = [np.concatenate([np.sort(key[64 * w:64 * w + 64]), np.full(64, np.uint64(2 ** 64 - 1))]) for w in range(4)]
    got = np.empty(256, np.int64)
    for t in range(256):
      w = t // 64
      r = int((key[64 * w:64 * w + 64] < key[t]).sum())   # the v_readlane loop of the wave
      for q in (1, 2, 3):
        a, pos = runs[(w + q) & 3], 0
        for step in (64, 32, 16, 8, 4, 2, 1):
          if a[pos + step - 1] < key[t]: pos += step
        r += pos
      got[t] = r
    np.testing.assert_array_equal(got, want)
    assert sorted(got) == list(range(256))
