"""smoke(): one tiny GAIL discriminator step + reward + SAC update on cuda:0 (Hopper dims, B=32, H=64), checked against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run_smoke(gi, np, torch):
  import imitation_learning_amd as il
  from gpu_util import N, T, close, close_params, crit_from_flat, make_disc, make_sac, make_sac_oracle, tbatch
  from oracle import gail as ogail
  from oracle import sac as osac
  c = gi.sac_case(1, 'hopper', 64, 32, 1)
  g = gi.gail_case(2, env='hopper', hidden=64, batch=32, steps=1)
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  st = make_sac_oracle(c)
  d, ods, icfg = make_disc(g)
  icfg.update(loss_function='BCE', grad_penalty=1.0, entropy_bonus=0.0)
  do = il.AdamW(d, lr=3e-5, weight_decay=10)
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  b, e = c['batches'][0], g['expert'][0]
  tb = tbatch(b)
  il.adversarial_imitation_update(actor, d, tb, tbatch(e), do, icfg, eps_gp=T(g['eps'][0]))
  tb['rewards'] = d.predict_reward(tb['states'], tb['actions'])
  logp, q = il.sac_update(actor, critic, log_alpha, target, tb, ao, co, to, c['discount'], c['entropy_target'], c['polyak'], eps_next=T(c['eps_next'][0]), eps_cur=T(c['eps_cur'][0]))
  torch.cuda.synchronize()
  ogail.gail_update(ods, cat(b), b['weights'], cat(e), e['weights'], g['eps'][0], lr=3e-5, weight_decay=10, grad_penalty=1.0)
  ob = dict(b); ob['rewards'] = ogail.predict_reward(ods, cat(b))
  ologp, oq = osac.sac_update(st, ob, c['eps_next'][0], c['eps_cur'][0], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'], lr=c['lr'])
  close(N(d.flat), ods.pack(), 'smoke disc params', atol_scale=4e-6); close(N(tb['rewards']), ob['rewards'], 'smoke rewards', rtol=1e-4, atol_scale=1e-5)
  close(N(logp), ologp, 'smoke logp', atol_scale=4e-6); close(N(q), oq, 'smoke q', atol_scale=4e-6)
  close_params(N(actor.flat), st.actor, 'smoke actor', c['lr']); close_params(crit_from_flat(critic, critic.flat), st.critic, 'smoke critic', c['lr'])
  print('smoke ok: SAC+GAIL update on', torch.cuda.get_device_name(0), 'matches the oracle')
