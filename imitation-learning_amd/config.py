"""Configuration surface of the reference (`conf/train_config.yaml` + `conf/algorithm/*.yaml`) without Hydra.

`compose(['algorithm=GAIL', 'env=halfcheetah', 'training.batch_size=512'])` applies, in Hydra's order,
  base defaults  <-  algorithm overlay (`# @package _global_` files)  <-  optimised_hyperparameters overlay  <-  dotted CLI overrides
and returns an attribute/`.get` dict like omegaconf's DictConfig.  The base and per-algorithm defaults are stated here (same
keys and values as the reference's YAML, reference conf/train_config.yaml:7-52 and conf/algorithm/*.yaml); the 21 tuned
`optimised_hyperparameters=<ALG>_<N>_trajectories` overlays come from `tuned.py` (their values as flat tables, generated from the reference's data
files by tests/golden/make_tuned_table.py) or, with `config_dir=<path>/conf`, from a reference-style YAML tree (which then also supplies the algorithm
overlays). `write_conf_tree(path)` writes such a tree - train_config.yaml, algorithm/*.yaml, optimised_hyperparameters/*.yaml - from this module's tables.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, List

ALGORITHMS = ['AdRIL', 'BC', 'DRIL', 'GAIL', 'GMMIL', 'PWIL', 'RED', 'SAC']
ENVS = ['ant', 'halfcheetah', 'hopper', 'walker2d']


class Config(dict):
  """dict with attribute access; nested dicts are wrapped on read (like omegaconf.DictConfig for this code base's needs)."""

  def __getattr__(self, k):
    try:
      v = self[k]
    except KeyError:
      raise AttributeError(k)
    if isinstance(v, dict) and not isinstance(v, Config):
      v = Config(v); self[k] = v
    return v

  def __setattr__(self, k, v):
    self[k] = v

  def __deepcopy__(self, memo):
    return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _mlp(hidden_size=256, depth=2, activation='relu', **kw):
  return dict(hidden_size=hidden_size, depth=depth, activation=activation, **kw)


BASE: Dict[str, Any] = dict(
    seed=0, steps=1000000, env='ant', algorithm='SAC',
    bc_pretraining=dict(iterations=0, learning_rate=0.00025, weight_decay=0),
    training=dict(start=1000, interval=1, batch_size=256, learning_rate=0.0003, weight_decay=0),
    evaluation=dict(interval=10000, episodes=30),
    logging=dict(interval=1000),
    reinforcement=dict(actor=_mlp(), critic=_mlp(), discount=0.99, target_temperature=-1, polyak_factor=0.995),
    memory=dict(size=1000000),
    imitation=dict(trajectories=0, subsample=1, state_only=False, absorbing=True, mix_expert_data='none', bc_aux_loss=False),
    check_time_usage=False, save_trajectories=False, render=False,
    distributed=dict(world_size=1, backend='nccl', timeout_s=600),   # data-parallel SAC over RCCL: launch with torch.distributed.run --nproc-per-node <world_size>; not a key of the reference
)

_DISC = dict(reward_shaping=False, subtract_log_policy=False, reward_function='AIRL')
ALGORITHM_OVERLAYS: Dict[str, Dict[str, Any]] = {
    'SAC': {}, 'GMMIL': {},
    'BC': dict(bc_pretraining=dict(iterations=50000)),
    'PWIL': dict(imitation=dict(reward_scale=5, reward_bandwidth_scale=5)),
    'GAIL': dict(
        reinforcement=dict(discount=0.97, target_temperature=-0.5, polyak_factor=0.99),
        imitation=dict(absorbing=True, discriminator=_mlp(64, 1, 'relu', input_dropout=0.5, dropout=0.75, **_DISC), learning_rate=0.00003, weight_decay=10, grad_penalty=1,
                       spectral_norm=True, entropy_bonus=0, loss_function='BCE', mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf'))),
    'AdRIL': dict(reinforcement=dict(discount=0.98, polyak_factor=0.98), imitation=dict(mix_expert_data='mixed_batch', balanced=True, update_freq=1250)),
    'DRIL': dict(imitation=dict(bc_aux_loss=True, discriminator=_mlp(64, 1, 'tanh', input_dropout=0.1, dropout=0.1), pretraining=dict(iterations=100000),
                                learning_rate=0.00003, weight_decay=0, quantile_cutoff=0.98)),
    'RED': dict(imitation=dict(discriminator=_mlp(32, 1, 'relu', input_dropout=0, dropout=0), reward_bandwidth_scale=None, pretraining=dict(iterations=100000),
                               learning_rate=0.00003, weight_decay=0)),
}


def _merge(dst: dict, src: dict):
  for k, v in src.items():
    if isinstance(v, dict) and isinstance(dst.get(k), dict):
      _merge(dst[k], v)
    else:
      dst[k] = copy.deepcopy(v)


def _parse_value(text: str):
  import yaml
  return yaml.safe_load(text)


def _set_dotted(cfg: dict, key: str, value):
  parts = key.lstrip('+').split('.')
  for p in parts[:-1]:
    cfg = cfg.setdefault(p, {})
  cfg[parts[-1]] = value


def compose(overrides: List[str], config_dir: str = None) -> Config:
  import yaml
  groups, dotted = {}, []
  for o in overrides:
    if o in ('-m', '--multirun'):
      raise NotImplementedError('multirun / sweeper plugins are orchestration outside this hot-path rebuild (SURVEY.md §2)')
    k, _, v = o.partition('=')
    if k in ('algorithm', 'optimised_hyperparameters', 'config_dir'):
      groups[k] = v
    else:
      dotted.append((k, v))
  config_dir = groups.get('config_dir', config_dir)
  cfg = copy.deepcopy(BASE)
  algorithm = groups.get('algorithm', 'SAC')
  if algorithm not in ALGORITHM_OVERLAYS:
    raise ValueError(f'algorithm={algorithm}: expected one of {ALGORITHMS}')
  if config_dir and os.path.exists(os.path.join(config_dir, 'algorithm', f'{algorithm}.yaml')):
    _merge(cfg, yaml.safe_load(open(os.path.join(config_dir, 'algorithm', f'{algorithm}.yaml'))) or {})
  else:
    _merge(cfg, ALGORITHM_OVERLAYS[algorithm])
  cfg['algorithm'] = algorithm
  oh = groups.get('optimised_hyperparameters')
  if oh and oh != 'null':
    if config_dir:   # an explicit tree wins (and must then hold the overlay)
      path = os.path.join(config_dir, 'optimised_hyperparameters', f'{oh}.yaml')
      if not os.path.exists(path):
        raise FileNotFoundError(f'optimised_hyperparameters={oh}: {path} not found')
      _merge(cfg, yaml.safe_load(open(path)) or {})
    else:
      from .tuned import TUNED
      if oh not in TUNED:
        raise FileNotFoundError(f'optimised_hyperparameters={oh}: not one of the reference\'s tuned overlays ({", ".join(sorted(TUNED))}); pass config_dir=<path>/conf for your own')
      for k, v in TUNED[oh].items():
        _set_dotted(cfg, k, copy.deepcopy(v))
  for k, v in dotted:
    _set_dotted(cfg, k, _parse_value(v))
  return Config(cfg)


def validate(cfg: Config):
  """The assertions of reference train.py:28-48."""
  assert cfg.algorithm in ALGORITHMS
  assert cfg.env in ENVS
  cfg.memory.size = min(cfg.steps, cfg.memory.size)
  assert cfg.bc_pretraining.iterations >= 0
  assert cfg.imitation.trajectories >= 0
  assert cfg.imitation.subsample >= 1
  assert cfg.imitation.mix_expert_data in ['none', 'mixed_batch', 'prefill_memory']
  if cfg.algorithm == 'AdRIL':
    assert cfg.imitation.mix_expert_data == 'mixed_batch' and cfg.imitation.update_freq >= 0
  elif cfg.algorithm == 'DRIL':
    assert 0 <= cfg.imitation.quantile_cutoff <= 1
  elif cfg.algorithm == 'GAIL':
    assert cfg.imitation.mix_expert_data != 'prefill_memory'
    assert cfg.imitation.discriminator.reward_function in ['AIRL', 'FAIRL', 'GAIL']
    assert cfg.imitation.grad_penalty >= 0 and cfg.imitation.entropy_bonus >= 0
    assert cfg.imitation.loss_function in ['BCE', 'Mixup', 'PUGAIL']
    if cfg.imitation.loss_function == 'Mixup': assert cfg.imitation.mixup_alpha > 0
    if cfg.imitation.loss_function == 'PUGAIL': assert 0 <= cfg.imitation.pos_class_prior <= 1 and cfg.imitation.nonnegative_margin >= 0
  assert cfg.logging.interval >= 0
  assert int(cfg.distributed.world_size) >= 1 and cfg.distributed.backend in ('nccl', 'gloo')
  return cfg


def write_conf_tree(path: str):
  """A reference-style `conf/` tree (train_config.yaml, algorithm/<ALG>.yaml with the `# @package _global_` header, optimised_hyperparameters/<name>.yaml) written from this
  module's tables: what `config_dir=` reads back, and a starting point for one's own overlays. `distributed` is this repository's key and is left out."""
  import yaml
  from .tuned import TUNED

  def nest(flat):
    out = {}
    for k, v in flat.items(): _set_dotted(out, k, v)
    return out

  def dump(d):
    return yaml.safe_dump(d, sort_keys=False, default_flow_style=False)
  os.makedirs(os.path.join(path, 'algorithm'), exist_ok=True)
  os.makedirs(os.path.join(path, 'optimised_hyperparameters'), exist_ok=True)
  base = {k: copy.deepcopy(v) for k, v in BASE.items() if k not in ('algorithm', 'distributed')}
  open(os.path.join(path, 'train_config.yaml'), 'w').write('defaults:\n  - _self_\n  - algorithm: SAC\n\n' + dump(base))
  for alg, overlay in ALGORITHM_OVERLAYS.items():
    open(os.path.join(path, 'algorithm', f'{alg}.yaml'), 'w').write('# @package _global_\n\n' + dump(dict(algorithm=alg, **copy.deepcopy(overlay))))
  for name, flat in TUNED.items():
    open(os.path.join(path, 'optimised_hyperparameters', f'{name}.yaml'), 'w').write('# @package _global_\n\n' + dump(nest(flat)))
  return path
