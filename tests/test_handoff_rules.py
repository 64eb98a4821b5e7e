"""The rules the device-side hand-offs follow since round 6 (profiles/r06_soak_under_load.md), checked on the SOURCES so that a new hand-off cannot quietly bring the
round-5 forms back - they pass every functional test on a quiet GPU and fail about once per 10^5 updates beside a busy neighbour process.

  release side: a workgroup's agent-scope release-add is preceded, within a few lines, by every wave draining its stores (sync_drain_stores / s_waitcnt vmcnt(0)) or by a
                fence executed by every thread (__threadfence) - never by a bare workgroup barrier;
  acquire side: the only agent-scope acquire fences in the kernels are sync_acquire_all (every wave, behind the barrier) and the leader form inside sync_wait<true>; nothing
                else spells one out (the instrumented build's verification pass excepted);
  producers of data that a RESIDENT launch of another stream consumes write it through (wstore1 / wstore4): the discriminator's stepped parameters, the index arrays, the
                rewards, the gathered rows, the Philox counter."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'imitation-learning_amd', 'csrc')
FILES = [f for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.hpp'))]


def lines(name):
  return open(os.path.join(CSRC, name)).read().splitlines()


def test_every_workgroup_release_is_preceded_by_a_drain_of_every_wave():
  seen = 0
  for name in FILES:
    if name == 'peer_device.hpp': continue   # (its releases are fences executed by every thread: checked below)
    src = lines(name)
    for i, line in enumerate(src):
      if '__ATOMIC_RELEASE' not in line or 'wavefront' in line or line.lstrip().startswith('//'): continue
      if '__hip_atomic_store' in line and 'SCOPE_SYSTEM' in line or 'version, next' in line:   # acting mailbox / snapshot publish: __threadfence by every thread in front
        before = '\n'.join(src[max(0, i - 14):i])
        assert '__threadfence' in before, f'{name}:{i + 1}: release store without a fence of every thread in front'
        seen += 1
        continue
      before = '\n'.join(src[max(0, i - 8):i + 1])
      assert 'sync_drain_stores()' in before or 'vmcnt(0)' in before, f'{name}:{i + 1}: agent-scope release behind a bare workgroup barrier (every wave must drain its stores first)'
      seen += 1
  assert seen >= 10   # the sites of round 6 (a refactoring that moves them all into helpers may lower this on purpose)


def test_peer_exchange_releases_and_acquires_with_every_thread():
  src = '\n'.join(lines('peer_device.hpp'))
  for m in re.finditer(r'__builtin_amdgcn_fence\(__ATOMIC_(RELEASE|ACQUIRE), ""\)', src):
    head = src[max(0, m.start() - 40):m.start()]
    assert 'tid == 0' not in head and 'threadIdx.x == 0' not in head, 'peer_device.hpp: a system-scope fence executed by one thread only'


def test_no_kernel_spells_out_its_own_agent_acquire():
  allowed = {('il_common.hpp', 'sync_acquire_all'), ('il_common.hpp', 'sync_wait'), ('sac.hip', 'relabel_verify')}
  for name in FILES:
    src = lines(name)
    fn = None
    for i, line in enumerate(src):
      m = re.match(r'\s*(?:template <[^>]*>\s*)?__device__ __forceinline__ \w[\w:<> ]*?\b(\w+)\(', line)
      if m: fn = m.group(1)
      m = re.match(r'__global__ .*?\b(k_\w+)\(', line)
      if m: fn = m.group(1)
      if '__ATOMIC_ACQUIRE' in line and '"agent"' in line and not line.lstrip().startswith('//'):
        assert (name, fn) in allowed, f'{name}:{i + 1} ({fn}): an agent-scope acquire outside sync_acquire_all / sync_wait_leader - use the helpers (every wave acquires behind the barrier)'


def test_cross_stream_producers_write_through():
  gail = '\n'.join(lines('gail.hip'))
  assert 'if (close_epoch) wstore1(d.params, e, pp);' in gail and 'wstore1(d.u1, i, o[i])' in gail, 'k_gail_reduce: stepped parameters / u, v for the inline relabel must be written through'
  assert 'if (d.sync) wstore1(d.params, e, pp);' in gail, 'k_disc_adam (data-parallel apply): stepped parameters must be written through under the device hand-off'
  assert 'if (d.sync) wstore1(out_r, row0 + r, reward);' in gail, 'k_gail_reward: rewards behind [IL_SYNC_REWARDS] must be written through'
  mt = '\n'.join(lines('mt_device.hpp'))
  assert 'wstore1(reinterpret_cast<float*>(out), count + before' in mt, 'the resident sampler must write the index arrays through'
  rp = '\n'.join(lines('replay.hip'))
  assert 'wstore4<true>(rows_a' in rp and 'wstore4<true>(rows_b' in rp, 'k_gather2 must write the gathered rows through under [IL_SYNC_ROWS]'
  sac = '\n'.join(lines('sac.hip'))
  assert 'wstore1(reinterpret_cast<float*>(a.noise_counter), 0' in sac, "the update's tail must write the bumped Philox counter through under il_sync"
  assert sac.count('disc_reward_tile<3, true>(') >= 2, 'the inline relabel must read the stepped parameters below the caches (disc_reward_tile<.., COH = true>)'
