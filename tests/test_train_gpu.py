"""`-m gpu` integration: the train.py entry point end to end on the synthetic D4RL-shaped environment, every supported algorithm."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = ['steps=260', 'training.start=120', 'evaluation.interval=130', 'evaluation.episodes=2', 'logging.interval=50', '+synthetic_env.max_episode_steps=60',
          '+synthetic_env.dataset_trajectories=6', 'training.batch_size=64']


@pytest.mark.parametrize('args', [
    ['algorithm=SAC', 'env=halfcheetah'],
    ['algorithm=GAIL', 'env=halfcheetah'],
    ['algorithm=GAIL', 'env=hopper', 'imitation.mix_expert_data=mixed_batch', 'imitation.bc_aux_loss=true'],   # un-fused path (host-visible batch edits)
    ['algorithm=GMMIL', 'env=ant'],
    ['algorithm=PWIL', 'env=walker2d'],
    ['algorithm=AdRIL', 'env=hopper', 'imitation.update_freq=100'],
    ['algorithm=AdRIL', 'env=walker2d', 'imitation.update_freq=0', 'imitation.balanced=false'],   # SQIL
    ['algorithm=GAIL', 'env=hopper', 'imitation.loss_function=PUGAIL'],
    ['algorithm=GAIL', 'env=hopper', 'imitation.discriminator.depth=2', 'imitation.discriminator.activation=tanh'],   # general discriminator kernels (gail_deep.hip)
    ['algorithm=GAIL', 'env=walker2d', 'imitation.discriminator.depth=2', 'imitation.loss_function=Mixup', 'imitation.spectral_norm=false'],
    ['algorithm=GAIL', 'env=walker2d', 'imitation.loss_function=Mixup', 'imitation.discriminator.reward_function=FAIRL'],
    ['algorithm=GAIL', 'env=hopper', 'imitation.loss_function=Mixup', 'imitation.entropy_bonus=0.24', 'imitation.grad_penalty=0.28', 'imitation.weight_decay=8.5'],   # GAIL_5_trajectories.yaml's shape: fused plan
    ['algorithm=GAIL', 'env=hopper', 'imitation.loss_function=Mixup', 'imitation.mixup_alpha=0.5'],   # Beta(0.5, 0.5) coefficients drawn on the device inside the captured plan (il_noise_fill_beta)
    ['algorithm=GAIL', 'env=halfcheetah', 'imitation.discriminator.subtract_log_policy=true'],
    ['algorithm=GAIL', 'env=hopper', 'imitation.discriminator.reward_shaping=true', 'imitation.discriminator.subtract_log_policy=true'],
    ['algorithm=GAIL', 'env=hopper', 'imitation.discriminator.reward_shaping=true', 'imitation.discriminator.depth=2', 'imitation.discriminator.activation=tanh',
     'imitation.discriminator.hidden_size=64', 'imitation.loss_function=PUGAIL', 'imitation.nonnegative_margin=0.05'],   # a depth-2 tanh shaping potential (gail_shaped_deep.hip), finite PUGAIL margin
    ['algorithm=RED', 'env=hopper', 'imitation.pretraining.iterations=50'],
    ['algorithm=RED', 'env=walker2d', 'imitation.pretraining.iterations=50', 'imitation.discriminator.depth=2', 'imitation.discriminator.activation=tanh',
     'imitation.discriminator.hidden_size=64', 'imitation.discriminator.input_dropout=0.05', 'imitation.discriminator.dropout=0.4'],   # conf/optimised_hyperparameters/RED_25_trajectories.yaml's shape
    ['algorithm=DRIL', 'env=hopper', 'imitation.pretraining.iterations=50'],
    ['algorithm=DRIL', 'env=walker2d', 'imitation.pretraining.iterations=50', 'imitation.discriminator.depth=2', 'imitation.discriminator.activation=relu',
     'imitation.discriminator.hidden_size=32', 'imitation.discriminator.input_dropout=0.4', 'imitation.discriminator.dropout=0.55'],   # conf/optimised_hyperparameters/DRIL_25_trajectories.yaml's shape
    ['algorithm=SAC', 'env=hopper', '+acting.schedule=overlap'],
    ['algorithm=GAIL', 'env=halfcheetah', '+acting.schedule=overlap'],
    ['algorithm=AdRIL', 'env=hopper', '+acting.schedule=overlap'],
    ['algorithm=SAC', 'env=walker2d', '+acting.schedule=fused'],
    ['algorithm=GAIL', 'env=walker2d', '+acting.schedule=per_function'],
    ['algorithm=BC', 'env=hopper', 'bc_pretraining.iterations=60'],
    # actor / critic shapes outside the fused kernels (models.py:48-69: any depth, relu / tanh / sigmoid; reinforcement.actor and reinforcement.critic configured separately): csrc/general.hip
    ['algorithm=SAC', 'env=hopper', 'reinforcement.actor.depth=3', 'reinforcement.actor.activation=tanh', 'reinforcement.actor.hidden_size=48', 'reinforcement.critic.depth=1',
     'reinforcement.critic.activation=sigmoid', 'reinforcement.critic.hidden_size=80'],
    ['algorithm=GAIL', 'env=halfcheetah', 'reinforcement.actor.hidden_size=320', 'reinforcement.critic.hidden_size=320', 'bc_pretraining.iterations=20'],
])
def test_train_runs(tmp_path, args):
  sys.path.insert(0, ROOT)
  import train
  from imitation_learning_amd import config
  os.chdir(tmp_path)
  cfg = config.compose(args + COMMON)
  score = train.train(cfg)
  assert np.isfinite(score)
  agent = torch.load(tmp_path / 'agent.pth', weights_only=False)
  assert 'actor' in agent and all(torch.isfinite(v).all() for v in agent['actor'].values())
  assert set(agent['actor']) == {f'actor.{2 * l}.{p}' for l in range(cfg.reinforcement.actor.depth + 1) for p in ('weight', 'bias')}   # Sequential slots: Linear, activation, Linear, ...
  metrics = torch.load(tmp_path / 'metrics.pth', weights_only=False)
  if cfg.algorithm != 'BC':
    assert 'critic_1.critic.0.weight' in agent['critic'] and len(metrics['update_steps']) >= 2
    assert all(np.isfinite(q).all() for q in metrics['Q_values'])
  if cfg.algorithm == 'GAIL':
    disc = torch.load(tmp_path / 'discriminator.pth', weights_only=False)
    if cfg.imitation.discriminator.reward_shaping:
      last = 2 * cfg.imitation.discriminator.depth
      assert 'g.parametrizations.weight.original' in disc and 'h.0.parametrizations.weight.original' in disc and f'h.{last}.parametrizations.weight.0._v' in disc
    elif (cfg.imitation.discriminator.depth, cfg.imitation.discriminator.activation) != (1, 'relu'):
      assert ('g.4.parametrizations.weight.original' if cfg.imitation.spectral_norm else 'g.4.weight') in disc
    else:
      assert 'g.0.parametrizations.weight.original' in disc and 'g.2.parametrizations.weight.0._v' in disc


def test_unsupported_configurations_fail_loudly():
  """Every algorithm= of the reference runs; option combinations without a kernel raise instead of silently running something else."""
  sys.path.insert(0, ROOT)
  import train
  from imitation_learning_amd import config
  for extra in (['algorithm=GAIL', 'imitation.discriminator.depth=3'], ['algorithm=GAIL', 'imitation.discriminator.hidden_size=256', 'imitation.discriminator.activation=tanh'],
                ['algorithm=SAC', 'reinforcement.actor.depth=9'],   # (depth 1-8 run: csrc/general.hip)
                ['algorithm=RED', 'imitation.discriminator.depth=3'], ['algorithm=RED', 'imitation.discriminator.activation=sigmoid']):
    with pytest.raises(NotImplementedError):
      train.train(config.compose(extra + ['env=hopper', 'steps=10'] + COMMON[5:7]))
