#!/usr/bin/env python3
"""Entry point with the reference's command line:  python train.py algorithm=<ALG> env=<ENV> [key.sub=value ...]

Drives the loop of reference train.py:26-243 (act -> store -> [update block] -> evaluate -> save) on the MI355X path:
the update block (train.py:171-203) is `UpdatePlan` (one captured hipGraph replay per step, every algorithm) or the per-function HIP entry
points (GAIL variants with per-update host inputs, batch sizes that are not a multiple of 16).  Hydra is replaced by `imitation_learning_amd.config.compose`
(same keys, same precedence).  Supported on the HIP path: SAC, GAIL (BCE loss), GMMIL, PWIL, AdRIL (and SQIL via update_freq=0), RED, DRIL, BC - every algorithm= of the reference.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import imitation_learning_amd as il  # noqa: E402
from imitation_learning_amd import _lib  # noqa: E402
from imitation_learning_amd import config as il_config  # noqa: E402
from imitation_learning_amd.environments import make_env  # noqa: E402
from imitation_learning_amd.evaluation import evaluate_agent  # noqa: E402
from imitation_learning_amd.models import default_device  # noqa: E402
from imitation_learning_amd.utils import cycle, lineplot  # noqa: E402


def expert_batches(cfg, expert_memory, state_size, action_size, count):
  """`count` shuffled, drop-last minibatches of the expert memory, reshuffled every epoch (the cycle(DataLoader(...)) of train.py:94,116)."""
  B, n = cfg.training.batch_size, expert_memory.size
  assert n >= B, f'pretraining needs at least one full batch of expert data ({n} < {B})'
  shuffle = torch.Generator().manual_seed(cfg.seed)
  while count > 0:
    order = torch.randperm(n, generator=shuffle).to(torch.int32)
    for lo in range(0, min(n - B + 1, count * B), B):
      yield il.memory.batch_views(expert_memory.gather(order[lo:lo + B]), state_size, action_size, cfg.imitation.absorbing)
      count -= 1


def pretrain_bc(cfg, actor, expert_memory, state_size, action_size):
  """cfg.bc_pretraining.iterations steps of il_bc_step (train.py:93-100)."""
  optimiser = il.AdamW(actor, lr=cfg.bc_pretraining.learning_rate, weight_decay=cfg.bc_pretraining.weight_decay)
  for batch in expert_batches(cfg, expert_memory, state_size, action_size, cfg.bc_pretraining.iterations):
    il.behavioural_cloning_update(actor, batch, optimiser)


def check_timeouts_seen(plan, step, last_good):
  """Every update step, no synchronisation: the pinned host words a device-side wait stores into when it gives up (UpdatePlan.watch_timeouts). Raises within the pipeline
  depth of the expiry (one or two updates), naming the last step at which the words still read zero - checkpoints are only ever written after a clean check."""
  handoff, exchange = plan.timeouts_seen()
  if handoff or exchange:
    what = (f'{exchange} wait(s) of the peer-window gradient exchange (a rank did not deliver its gradients within the bound: IL_PEER_EXCHANGE=0 selects RCCL all-reduces)' if exchange else
            f'{handoff} device-side hand-off wait(s) (the discriminator and SAC branches did not run concurrently: IL_DEVICE_SYNC=0 selects stream dependencies)')
    raise RuntimeError(f'step {step}: {what} expired; updates since step {last_good} (the last one checked clean) may have trained on stale rewards / gradients. Nothing was saved after that step.')


def check_handoff(plan, step, runner=None):
  """The two branches of a captured GAIL update hand over through device counters with BOUNDED waits (include/il_hip.h il_sync): a wait that expires lets the
  update proceed on stale rewards / discriminator weights and bumps a counter. That can only happen if something stops the two streams from running
  concurrently after `capture()` validated them (a profiler attached mid-run, a CU mask, a co-tenant process). Training on such updates silently is worse
  than stopping: raise, naming the switch that trades the hand-off for plain stream dependencies."""
  n = runner.exchange_timeouts() if hasattr(runner, 'exchange_timeouts') else 0
  if n:
    raise RuntimeError(f'step {step}: {n} device-side waits of the peer-window gradient exchange expired since set-up - a rank did not deliver its gradients within the bound, '
                       'so some update averaged stale slabs and the replicas may have diverged. Re-run with IL_PEER_EXCHANGE=0 (RCCL all-reduces).')
  if plan is None or not getattr(plan, 'device_sync', False): return
  n = plan.sync_timeouts()
  if n:
    raise RuntimeError(f'step {step}: {n} device-side hand-off waits expired since the last check - the discriminator and SAC branches of the update did not run concurrently, '
                       'so updates in this interval used stale rewards. Re-run with IL_DEVICE_SYNC=0 (stream dependencies instead of device counters).')


def train(cfg, file_prefix: str = '') -> float:
  il_config.validate(cfg)
  # ---- data parallelism (BASELINE.json configs[4]): one process per GPU, own environment + replay shard, replicated networks, three gradient
  # all-reduces per update. Launched by torch.distributed.run; rank r seeds everything with seed + r (its data must differ), parameters come from rank 0.
  world = int(cfg.distributed.world_size)
  rank = 0
  if world > 1:
    from imitation_learning_amd import parallel
    assert torch.cuda.is_available(), 'train.py needs a GPU: the update path has no CPU fallback'
    rank, _, dev = parallel.init_from_env(world, cfg.distributed.backend)
    dog = parallel.Watchdog(float(cfg.distributed.timeout_s), what=f'train.py (rank {rank} of {world})', armed=False)   # a rank that dies inside a collective must not hang the others; armed at the first captured update (the set-up phases before it have no bound of their own)
  else:
    dev, dog = default_device(), None
  assert _lib.on_device(torch.empty(0, device=dev)), 'train.py needs a GPU: the update path has no CPU fallback'
  lead = rank == 0                # evaluation, plots and checkpoints are rank 0's job
  seed = cfg.seed + rank
  il.seed(seed)                   # replay index stream (np.random.seed in the reference, train.py:51)
  np.random.seed(seed)
  torch.manual_seed(seed)

  env_kw = dict(cfg.get('synthetic_env', {}) or {})
  env, eval_env = make_env(cfg.env, cfg.imitation.absorbing, load_data=True, **env_kw), make_env(cfg.env, cfg.imitation.absorbing, **env_kw)
  env.seed(seed); eval_env.seed(seed)
  score_lo, score_hi = env.env.ref_min_score, env.env.ref_max_score
  normalise = lambda returns: (np.asarray(returns) - score_lo) / (score_hi - score_lo)  # D4RL normalised score
  expert_memory = env.get_dataset(trajectories=cfg.imitation.trajectories, subsample=cfg.imitation.subsample, device=dev)
  state_size, action_size = env.observation_space.shape[0], env.action_space.shape[0]

  actor, critic, log_alpha = il.SoftActor(state_size, action_size, cfg.reinforcement.actor), il.TwinCritic(state_size, action_size, cfg.reinforcement.critic), torch.zeros(1, device=dev)
  target_critic, entropy_target = il.create_target_network(critic), cfg.reinforcement.target_temperature * action_size
  actor_optimiser = il.AdamW(actor, lr=cfg.training.learning_rate, weight_decay=cfg.training.weight_decay)
  critic_optimiser = il.AdamW(critic, lr=cfg.training.learning_rate, weight_decay=cfg.training.weight_decay)
  temperature_optimiser = il.Adam(log_alpha, lr=cfg.training.learning_rate)
  memory = il.ReplayMemory(cfg.memory.size, state_size, action_size, cfg.imitation.absorbing)

  discriminator = discriminator_optimiser = None
  if cfg.algorithm == 'AdRIL':
    discriminator = il.RewardRelabeller(cfg.imitation.update_freq, cfg.imitation.balanced)
  elif cfg.algorithm == 'GAIL':
    discriminator = il.GAILDiscriminator(state_size, action_size, cfg.imitation, cfg.reinforcement.discount)
    discriminator_optimiser = il.AdamW(discriminator, lr=cfg.imitation.learning_rate, weight_decay=cfg.imitation.weight_decay)
  elif cfg.algorithm == 'GMMIL':
    discriminator = il.GMMILDiscriminator(state_size, action_size, cfg.imitation)
  elif cfg.algorithm == 'PWIL':
    discriminator = il.PWILDiscriminator(state_size, action_size, cfg.imitation, expert_memory, env.max_episode_steps)
  elif cfg.algorithm == 'DRIL':
    discriminator = il.DropoutSoftActor(state_size, action_size, cfg.imitation.discriminator)   # dropout policy ensemble (train.py:73 builds it as SoftActor(...))
    discriminator_optimiser = il.AdamW(discriminator, lr=cfg.imitation.learning_rate, weight_decay=cfg.imitation.weight_decay)
  elif cfg.algorithm == 'RED':
    discriminator = il.REDDiscriminator(state_size, action_size, cfg.imitation)
    discriminator_optimiser = il.AdamW(discriminator, lr=cfg.imitation.learning_rate, weight_decay=cfg.imitation.weight_decay)

  if world > 1:   # replicas start from rank 0's initialisation (the per-rank torch seeds differ)
    parallel.broadcast_parameters(parallel.replica_tensors(actor, critic, target_critic, log_alpha, discriminator))
  metrics = dict(train_steps=[], train_returns=[], test_steps=[], test_returns=[], test_returns_normalized=[], update_steps=[], predicted_rewards=[], alphas=[], entropies=[], Q_values=[])
  score = []
  if cfg.check_time_usage: start_time = time.time()
  B = cfg.training.batch_size

  # ---- behavioural cloning pretraining (train.py:93-112)
  if cfg.bc_pretraining.iterations > 0:
    pretrain_bc(cfg, actor, expert_memory, state_size, action_size)
    if cfg.algorithm == 'BC':
      if cfg.check_time_usage: metrics['pre_training_time'] = time.time() - start_time
      episode_returns = evaluate_agent(actor, eval_env, cfg.evaluation.episodes)
      normalised = normalise(episode_returns)
      metrics.update(test_steps=[0], test_returns=[episode_returns], test_returns_normalized=[list(normalised)])
      if world > 1:   # algorithm=BC has no update block to parallelise: every rank clones on its own, rank 0 reports
        torch.distributed.barrier(); torch.distributed.destroy_process_group(); dog.stop()
      if not lead: return float(np.mean(normalised))
      torch.save(dict(actor=actor.state_dict()), f'{file_prefix}agent.pth')
      torch.save(metrics, f'{file_prefix}metrics.pth')
      return float(np.mean(normalised))

  if cfg.algorithm in ('DRIL', 'RED'):  # train.py:114-134: pretrain the "discriminator" on expert data, then fix its reward threshold / bandwidth
    for batch in expert_batches(cfg, expert_memory, state_size, action_size, cfg.imitation.pretraining.iterations):
      if cfg.algorithm == 'DRIL': il.behavioural_cloning_update(discriminator, batch, discriminator_optimiser)
      else: il.target_estimation_update(discriminator, batch, discriminator_optimiser)
    if cfg.algorithm == 'DRIL': discriminator.set_uncertainty_threshold(expert_memory['states'][:expert_memory.size], expert_memory['actions'][:expert_memory.size], cfg.imitation.quantile_cutoff)
    else: discriminator.set_sigma(expert_memory['states'][:B], expert_memory['actions'][:B])
    if cfg.check_time_usage: metrics['pre_training_time'], start_time = time.time() - start_time, time.time()
    if cfg.imitation.mix_expert_data == 'prefill_memory': memory.transfer_transitions(expert_memory)

  if world > 1 and (cfg.bc_pretraining.iterations > 0 or cfg.algorithm in ('DRIL', 'RED')):   # every rank pretrained on its own shuffles: keep rank 0's result
    parallel.broadcast_parameters(parallel.replica_tensors(actor, critic, target_critic, log_alpha, discriminator))
    if cfg.algorithm == 'DRIL': discriminator.q, = parallel.broadcast_scalars([discriminator.q])
    if cfg.algorithm == 'RED': discriminator.sigma_1, = parallel.broadcast_scalars([discriminator.sigma_1])

  if cfg.algorithm == 'PWIL' and cfg.imitation.mix_expert_data != 'none':  # train.py:135-141
    for i in range(expert_memory.size):
      tr = expert_memory[i]
      expert_memory.rewards[i] = discriminator.compute_reward(tr['states'].unsqueeze(0), tr['actions'].unsqueeze(0))
      if tr['terminals'] or tr['timeouts']: discriminator.reset()
  if cfg.algorithm in ('PWIL', 'GMMIL') and cfg.imitation.mix_expert_data == 'prefill_memory':
    memory.transfer_transitions(expert_memory)

  # ---- the update block as ONE captured graph per step (UpdatePlan) whenever its inputs are device-resident; otherwise the per-function entry points
  plan = None
  mixed = cfg.imitation.mix_expert_data == 'mixed_batch'
  fusable = B % 16 == 0
  if cfg.algorithm == 'GAIL' and (mixed or cfg.imitation.bc_aux_loss):
    fusable = False   # a mix between the discriminator step and the relabel / an auxiliary actor step on the expert batch: per-function path. (Every discriminator variant -
    # loss functions, finite PUGAIL margin, subtract_log_policy, reward shaping, depth 2 / tanh, Mixup with any alpha - is captured by UpdatePlan.)
  general = bool(getattr(actor, 'general', False) or getattr(critic, 'general', False))   # reinforcement.actor / critic outside depth 2 / relu / hidden <= 256 (csrc/general.hip): the plan runs them on one stream, captured as one graph
  if fusable:
    plan = il.UpdatePlan(cfg.algorithm, actor, critic, log_alpha, target_critic, memory, actor_optimiser, critic_optimiser, temperature_optimiser, B, cfg.reinforcement.discount,
                         entropy_target, cfg.reinforcement.polyak_factor, expert_memory=expert_memory, discriminator=discriminator, discriminator_optimiser=discriminator_optimiser,
                         imitation_cfg=cfg.imitation if cfg.algorithm == 'GAIL' else None, mix_expert=mixed and cfg.algorithm in ('DRIL', 'GMMIL', 'RED'),
                         bc_aux=bool(cfg.imitation.bc_aux_loss))
  captured = False
  runner = plan
  if world > 1:
    if plan is None:
      raise NotImplementedError('distributed.world_size > 1 needs the captured update plan (a GAIL variant with per-update host inputs, or a batch size that is not a multiple of 16, has no data-parallel path)')
    runner = parallel.DataParallelUpdate(plan)   # grads-only kernels -> all-reduce(mean) of one flat bucket per sync point -> apply kernels

  # acting (train.py:151-168): il_act_step through a pinned mailbox; PWIL computes its reward per step on the device and keeps the per-function path
  schedule = (cfg.get('acting', {}) or {}).get('schedule', 'exact')  # `+acting.schedule=fused|overlap`: see imitation_learning_amd/acting.py (behaviour policy lags 1-2 updates)
  assert schedule in ('exact', 'fused', 'overlap', 'per_function')
  worker = il.ActingWorker(actor, memory, mirror=schedule == 'overlap') if cfg.algorithm != 'PWIL' and schedule != 'per_function' and not general else None
  if worker is None: schedule = 'per_function'
  if schedule == 'overlap' and world > 1:
    raise NotImplementedError('+acting.schedule=overlap with distributed.world_size > 1: the overlap schedule captures the append in the update plan\'s hooks, which the '
                              'data-parallel runner does not run (use exact or fused)')
  if schedule == 'overlap' and plan is not None: worker.attach(plan)   # append + parameter snapshot ride in the update's hipGraph
  elif plan is not None: plan.main_feeds_ring = True   # appends are enqueued on this stream between updates: a resident index draw on the other stream must come after them
  if cfg.algorithm in ('GAIL', 'RED'): discriminator.eval()   # train.py:147: from here on the RED predictor's dropout is off (DRIL keeps its dropout on purpose)
  t, state, terminal, train_return = 0, env.reset(), False, 0
  action = worker.act(state) if schedule in ('fused', 'overlap') else None
  last_good = 0   # the last update step whose time-out words read zero (check_timeouts_seen)
  for step in range(1, cfg.steps + 1):
    if dog is not None: dog.beat(f'step {step}')
    update_due = step >= cfg.training.start and step % cfg.training.interval == 0
    if schedule == 'per_function':
      with torch.inference_mode():
        action = actor(state).sample()
        next_state, reward, terminal = env.step(action)
        t += 1
        reward_stored = discriminator.compute_reward(state, action) if cfg.algorithm == 'PWIL' else reward
        memory.append(step, state, action, reward_stored, next_state, terminal and t != env.max_episode_steps, t == env.max_episode_steps)
        if terminal and cfg.imitation.absorbing and t != env.max_episode_steps: memory.wrap_for_absorbing_states()
    else:
      if schedule == 'exact': action = worker.act(state)
      next_state, reward, terminal = env.step(action)
      t += 1
      timed_out = t == env.max_episode_steps
      following = env.reset() if terminal else next_state   # the observation the next action is for
      if schedule == 'exact':
        worker.append(step, next_state, reward, terminal and not timed_out, timed_out)
      elif schedule == 'fused':
        action = worker.step(step, next_state, reward, terminal and not timed_out, timed_out, obs=following)
      else:
        worker.post(step, state, action, next_state, reward, terminal and not timed_out, timed_out)
        if not (update_due and plan is not None):
          if plan is not None: plan.launcher_wait()   # (an update handed to the launcher thread is issued before this append)
          worker.enqueue_append()   # otherwise the update graph carries it
    train_return += reward
    if terminal:
      if cfg.algorithm == 'PWIL': discriminator.reset()
      metrics['train_steps'].append(step); metrics['train_returns'].append([train_return])
      t, train_return = 0, 0
      state = env.reset() if schedule == 'per_function' else following
    else:
      state = next_state

    act_early = schedule == 'overlap' and update_due and plan is not None and captured   # (round 6) the act launch ahead of the update's host work (ActingWorker.act_begin)
    if act_early: worker.act_begin(state)
    if update_due:
      if plan is not None:
        if cfg.algorithm == 'AdRIL': plan.relabel_args(step, memory.num_trajectories)   # the relabeller's per-update scalars -> device buffer (models.py:300-318)
        if not captured:
          runner.run()   # first update eagerly (loads code objects; GMMIL: fixes the kernel bandwidths), then capture
          if world > 1 and getattr(runner, 'handoff', False) and not runner.agree_on_handoff():   # collective decision (see DataParallelUpdate.agree_on_handoff)
            print(f'[train] rank {rank}: a bounded device-side wait expired during the first data-parallel update on some rank: every rank continues with stream dependencies '
                  '(the affected rank skipped that update - an expired wait poisons its learner -; every rank has been reset to rank 0\'s replica)', file=sys.stderr)
          if world > 1 and cfg.algorithm == 'GMMIL':   # one reward function on every rank: rank 0's bandwidths (models.py:193-195 freezes the first batch's)
            discriminator.gamma_1, discriminator.gamma_2 = parallel.broadcast_scalars([discriminator.gamma_1, discriminator.gamma_2])
          if world > 1 and cfg.distributed.backend != 'nccl' and getattr(runner, 'peer', None) is None:
            step_update = runner.run   # gloo collectives synchronise the host: not capturable (and a failed capture poisons the stream) - eager launches
          elif runner is plan and plan.direct_launch_ok() and os.environ.get('IL_TRAIN_LAUNCH', 'direct') != 'graph':
            # one GPU, the two-branch schedule: the update's six launches issued directly (two library calls per update). A hipGraph replay costs ~4.5 us more between
            # two updates than the launch boundary of the same kernels (profiles/r05_launch_ab.txt); IL_TRAIN_LAUNCH=graph keeps the graphs
            plan.record_direct()
            # (round 6) IL_TRAIN_LAUNCH_THREAD=1 with +acting.schedule=overlap: the recorded launches go out from the library's launcher thread (UpdatePlan.launch_async) while
            # this thread posts the next observation and steps the environment; every host-side read / launch of this learner below drains the launcher first. Opt-in: it pays
            # when an environment step costs more than the act launch's turn-around (a real simulator); on the synthetic environment it measured 10.3k against 11.5k
            # env-steps/s (profiles/r06_acting.json: the loop is bound by that turn-around, which then has nothing to hide behind).
            use_thread = schedule == 'overlap' and not plan._direct_overlap and os.environ.get('IL_TRAIN_LAUNCH_THREAD', '0') == '1'
            step_update = plan.launch_async if use_thread else plan.launch_direct
          elif runner is not plan and runner.direct_launch_ok() and os.environ.get('IL_TRAIN_LAUNCH', 'direct') != 'graph':
            runner.record_direct()   # (round 6) data parallel with the exchanges inside the optimiser launches: the launch sequence of one GPU, issued the same way
            step_update = runner.launch_direct
          else:
            runner.capture(warmup=0)   # the gradient exchange (peer-window kernels, or RCCL collectives) is captured with the kernels: one graph replay per data-parallel update
            step_update = runner.replay
          captured = True
          plan.watch_timeouts(runner.peer.status if getattr(runner, 'peer', None) is not None else None)   # expired device-side waits raise a host-visible flag from now on
          if world > 1:   # graph capture takes different times on different ranks: meet again before the first replay, whose device-side waits are bounded
            torch.cuda.synchronize(); torch.distributed.barrier()
          if dog is not None: dog.arm(f'step {step}')
        else:
          step_update()
        check_timeouts_seen(plan, step, last_good)   # host read of pinned words: every step, independent of logging.interval
        last_good = step
        rewards, log_probs, Q_values = plan.transitions['rewards'], plan.logp, plan.q
      else:
        transitions, expert_transitions = memory.sample(B), expert_memory.sample(B)
        if cfg.algorithm == 'GAIL':
          discriminator.train()
          il.adversarial_imitation_update(actor, discriminator, transitions, expert_transitions, discriminator_optimiser, cfg.imitation)
          discriminator.eval()
        if cfg.imitation.mix_expert_data == 'mixed_batch' and cfg.algorithm in ('DRIL', 'GAIL', 'GMMIL', 'RED'):   # train.py:175,183: only inside the imitation block, never for SAC / PWIL / AdRIL
          il.mix_expert_agent_transitions(transitions, expert_transitions)
        if cfg.algorithm == 'AdRIL':
          discriminator.resample_and_relabel(transitions, expert_transitions, step, memory.num_trajectories, expert_memory.num_trajectories)
        if cfg.algorithm == 'GAIL':
          transitions['rewards'] = discriminator.predict_reward(**il.make_gail_input(transitions['states'], transitions['actions'], transitions['next_states'], transitions['terminals'], actor,
                                                                                     cfg.imitation.discriminator.reward_shaping, cfg.imitation.discriminator.subtract_log_policy))
        elif cfg.algorithm in ('DRIL', 'RED'):
          transitions['rewards'].copy_(discriminator.predict_reward(transitions['states'], transitions['actions']))
        elif cfg.algorithm == 'GMMIL':
          transitions['rewards'] = discriminator.predict_reward(transitions['states'], transitions['actions'], expert_transitions['states'], expert_transitions['actions'],
                                                                transitions['weights'].contiguous(), expert_transitions['weights'].contiguous())
        if cfg.imitation.bc_aux_loss: il.behavioural_cloning_update(actor, expert_transitions, actor_optimiser)
        log_probs, Q_values = il.sac_update(actor, critic, log_alpha, target_critic, transitions, actor_optimiser, critic_optimiser, temperature_optimiser, cfg.reinforcement.discount,
                                            entropy_target, cfg.reinforcement.polyak_factor)
        rewards = transitions['rewards']
      if schedule == 'overlap' and plan is None: worker.enqueue_publish()
      if cfg.logging.interval > 0 and step % cfg.logging.interval == 0:  # the only D2H reads of the update path (train.py:205-210)
        if plan is not None: plan.join()   # the logged tensors may have been written by the plan's second stream
        check_handoff(plan, step, runner)
        metrics['update_steps'].append(step); metrics['predicted_rewards'].append(rewards.cpu().numpy())
        metrics['alphas'].append(log_alpha.exp().cpu().numpy()); metrics['entropies'].append((-log_probs).cpu().numpy()); metrics['Q_values'].append(Q_values.cpu().numpy())

    if schedule == 'overlap': action = worker.act_end() if act_early else worker.act(state)   # own stream, published snapshot: returns while the update is still running

    if dog is not None and step % cfg.evaluation.interval == 0 and not cfg.check_time_usage:   # every rank: rank 0 evaluates, its peers wait for it inside their next collective
      dog.grace(0.005 * cfg.evaluation.episodes * env.max_episode_steps, 'evaluation')
    if step % cfg.evaluation.interval == 0 and not cfg.check_time_usage and lead:
      if plan is not None: plan.launcher_wait()
      episode_returns = evaluate_agent(actor, eval_env, cfg.evaluation.episodes)
      normalised = normalise(episode_returns)
      score.append(float(normalised.mean()))
      for key, value in (('test_steps', step), ('test_returns', episode_returns), ('test_returns_normalized', list(normalised))): metrics[key].append(value)
      lineplot(metrics['test_steps'], metrics['test_returns'], filename=f'{file_prefix}test_returns', title=f'{cfg.algorithm}: {cfg.env} Test Returns')
      if len(metrics['train_returns']) > 0:
        lineplot(metrics['train_steps'], metrics['train_returns'], filename=f'{file_prefix}train_returns', title=f'Training {cfg.algorithm}: {cfg.env} Train Returns')
      if cfg.logging.interval > 0 and len(metrics['update_steps']) > 0:   # train.py:224-228
        if cfg.algorithm != 'SAC': lineplot(metrics['update_steps'], metrics['predicted_rewards'], filename=f'{file_prefix}predicted_rewards', yaxis='Predicted Reward', title=f'{cfg.algorithm}: {cfg.env} Predicted Rewards')
        lineplot(metrics['update_steps'], metrics['alphas'], filename=f'{file_prefix}sac_alpha', yaxis='Alpha', title=f'{cfg.algorithm}: {cfg.env} Alpha')
        lineplot(metrics['update_steps'], metrics['entropies'], filename=f'{file_prefix}sac_entropy', yaxis='Entropy', title=f'{cfg.algorithm}: {cfg.env} Entropy')
        lineplot(metrics['update_steps'], metrics['Q_values'], filename=f'{file_prefix}Q_values', yaxis='Q-value', title=f'{cfg.algorithm}: {cfg.env} Q-values')
    if world > 1 and step % cfg.evaluation.interval == 0 and not cfg.check_time_usage:
      # rank 0 has just spent seconds evaluating: align the hosts here, so that no BOUNDED device-side wait of the next update (the peer-window exchange, the hand-off on
      # the all-reduced discriminator step) has to span that gap
      torch.cuda.synchronize(); torch.distributed.barrier()
      if plan is not None:   # replicas apply the same averaged gradients with the same kernels: anything but identical bits means an exchange went wrong - stop before training on
        check_handoff(plan, step, runner)
        same, digests = parallel.replicas_bit_identical(runner.replica_state())
        if not same:
          raise RuntimeError(f'step {step}: the data-parallel replicas are no longer bit-identical (sha256 per rank: {[d[:12] for d in digests]}; exchange: {runner.exchange_name()}). '
                             'Re-run with IL_PEER_EXCHANGE=0 (RCCL all-reduces).')
        metrics.setdefault('replica_checks', []).append((step, digests[0][:16], runner.exchange_name()))

  if plan is not None: plan.join()   # the discriminator is stepped on the plan's second stream: order the checkpoint reads after it
  check_handoff(plan, cfg.steps, runner)   # never save a learner whose last updates ran on expired device-side waits
  if world > 1:
    import torch.distributed as dist
    torch.cuda.synchronize(); dist.barrier()
    if plan is not None:
      same, digests = parallel.replicas_bit_identical(runner.replica_state())
      if not same:
        raise RuntimeError(f'end of training: the data-parallel replicas are not bit-identical (sha256 per rank: {[d[:12] for d in digests]}); nothing saved')
    dog.stop()
    if not lead:   # replicas are identical: rank 0 writes the checkpoint; the others return their (empty) score
      dist.destroy_process_group()
      return float('nan')
  if cfg.check_time_usage: metrics['training_time'] = time.time() - start_time
  if cfg.save_trajectories:   # train.py:231-234
    _, trajectories = evaluate_agent(actor, eval_env, cfg.evaluation.episodes, return_trajectories=True, render=cfg.render)
    torch.save(trajectories, f'{file_prefix}trajectories.pth')
  torch.save(dict(actor=actor.state_dict(), critic=critic.state_dict(), log_alpha=log_alpha), f'{file_prefix}agent.pth')
  if cfg.algorithm in ('DRIL', 'GAIL', 'RED'): torch.save(discriminator.state_dict(), f'{file_prefix}discriminator.pth')   # train.py:238
  torch.save(metrics, f'{file_prefix}metrics.pth')
  if world > 1:
    dist.destroy_process_group()
  return float(np.mean(score)) if score else float('nan')


def main(argv):
  cfg = il_config.compose(argv)
  out = os.path.join('outputs', f'{cfg.algorithm}_{cfg.env}', time.strftime('%m-%d_%H-%M-%S'))
  if int(os.environ.get('RANK', '0')) == 0:   # data-parallel runs: rank 0 owns the output directory (the other ranks write nothing)
    os.makedirs(out, exist_ok=True)
    os.chdir(out)  # hydra.job.chdir=true in the reference
  score = train(cfg)
  print(f'{cfg.algorithm} {cfg.env}: mean normalised score {score:.4f} (outputs in {out})')
  return score


if __name__ == '__main__':
  main(sys.argv[1:])
