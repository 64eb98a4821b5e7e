"""ctypes binding of libil_hip.so (the C ABI declared in include/il_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
The structures mirror include/il_hip.h field for field.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IL_HIP_LIBRARY') or os.path.join(_HERE, 'libil_hip.so')   # IL_HIP_LIBRARY: developer A/B builds of the same ABI

IL_FLAG_GRADS_ONLY, IL_FLAG_TICK, IL_FLAG_SAC_FORWARD_ONLY, IL_FLAG_SAC_SKIP_FORWARD, IL_FLAG_SAC_PREPARED = 1, 2, 4, 8, 16
IL_FLAG_GAIL_CLOSE_EPOCH = 32
IL_FLAG_SAC_WAIT_INDICES = 64
IL_FLAG_SAC_STAGED_ROWS = 0x400
c_f32p, c_i32p, c_u32p, c_i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)


class Batch(C.Structure):
  _fields_ = [(k, C.c_void_p) for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights', 'absorbing')] + \
             [('ld_' + k, C.c_int32) for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights', 'absorbing')] + [('n', C.c_int32)] + \
             [('gather', C.c_void_p), ('gather_capacity', C.c_int64)]   # optional row indirection (include/il_hip.h il_batch)


class Adam(C.Structure):
  _fields_ = [('m', C.c_void_p), ('v', C.c_void_p), ('step', C.c_void_p), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
              ('weight_decay', C.c_double)]


class Sac(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32),
              ('actor', C.c_void_p), ('critic', C.c_void_p), ('target', C.c_void_p), ('log_alpha', C.c_void_p),
              ('actor_grad', C.c_void_p), ('critic_grad', C.c_void_p), ('alpha_grad', C.c_void_p),
              ('actor_opt', Adam), ('critic_opt', Adam), ('alpha_opt', Adam),
              ('discount', C.c_float), ('entropy_target', C.c_float), ('polyak', C.c_double),
              ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('noise_seed', C.c_uint64), ('noise_counter', C.c_void_p),
              ('out_logp', C.c_void_p), ('out_q', C.c_void_p), ('sync', C.c_void_p), ('debug_masks', C.c_void_p)]


class Disc(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32),
              ('spectral_norm', C.c_int32), ('state_only', C.c_int32), ('reward_function', C.c_int32),
              ('params', C.c_void_p), ('u1', C.c_void_p), ('v1', C.c_void_p), ('u2', C.c_void_p), ('v2', C.c_void_p), ('grad', C.c_void_p),
              ('opt', Adam), ('grad_penalty', C.c_float), ('entropy_bonus', C.c_float),
              ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('noise_seed', C.c_uint64), ('noise_counter', C.c_void_p), ('sync', C.c_void_p), ('loss_function', C.c_int32), ('pos_class_prior', C.c_float), ('pu_clamped', C.c_int32), ('nonnegative_margin', C.c_float)]


class DiscShaped(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32),
              ('spectral_norm', C.c_int32), ('state_only', C.c_int32), ('reward_function', C.c_int32), ('loss_function', C.c_int32),
              ('params', C.c_void_p), ('ug', C.c_void_p), ('vg', C.c_void_p), ('u1', C.c_void_p), ('v1', C.c_void_p), ('u2', C.c_void_p), ('v2', C.c_void_p), ('grad', C.c_void_p),
              ('opt', Adam), ('grad_penalty', C.c_float), ('entropy_bonus', C.c_float), ('pos_class_prior', C.c_float), ('discount', C.c_float),
              ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('noise_seed', C.c_uint64), ('noise_counter', C.c_void_p), ('pu_clamped', C.c_int32), ('nonnegative_margin', C.c_float)]


class DiscDeep(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32),
              ('spectral_norm', C.c_int32), ('state_only', C.c_int32), ('reward_function', C.c_int32), ('loss_function', C.c_int32),
              ('depth', C.c_int32), ('activation', C.c_int32), ('params', C.c_void_p), ('sn', C.c_void_p), ('grad', C.c_void_p), ('opt', Adam),
              ('grad_penalty', C.c_float), ('entropy_bonus', C.c_float), ('pos_class_prior', C.c_float), ('reserved', C.c_float),
              ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('noise_seed', C.c_uint64), ('noise_counter', C.c_void_p), ('pu_clamped', C.c_int32), ('nonnegative_margin', C.c_float)]


class DiscShapedDeep(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32),
              ('spectral_norm', C.c_int32), ('state_only', C.c_int32), ('reward_function', C.c_int32), ('loss_function', C.c_int32),
              ('depth', C.c_int32), ('activation', C.c_int32), ('params', C.c_void_p), ('sn', C.c_void_p), ('grad', C.c_void_p), ('opt', Adam),
              ('grad_penalty', C.c_float), ('entropy_bonus', C.c_float), ('pos_class_prior', C.c_float), ('discount', C.c_float),
              ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('noise_seed', C.c_uint64), ('noise_counter', C.c_void_p), ('pu_clamped', C.c_int32), ('nonnegative_margin', C.c_float)]


class GailExtra(C.Structure):
  _fields_ = [('eps_mix', C.c_void_p), ('logit_offset_policy', C.c_void_p), ('logit_offset_expert', C.c_void_p), ('logit_offset_mix', C.c_void_p)]


class Pwil(C.Structure):
  _fields_ = [('n_atoms', C.c_int32), ('dim', C.c_int32), ('state_dim', C.c_int32), ('action_dim', C.c_int32),
              ('atoms', C.c_void_p), ('weights', C.c_void_p), ('dists', C.c_void_p), ('scale', C.c_void_p), ('offset', C.c_void_p),
              ('reward_scale', C.c_double), ('reward_bandwidth', C.c_double), ('agent_weight', C.c_double)]


class Red(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32), ('state_only', C.c_int32), ('depth', C.c_int32),
              ('predictor', C.c_void_p), ('target', C.c_void_p), ('grad', C.c_void_p), ('opt', Adam), ('workspace', C.c_void_p),
              ('sigma_1', C.c_float), ('activation', C.c_int32), ('out_pred', C.c_void_p), ('out_target', C.c_void_p),
              ('p_in', C.c_float), ('p', C.c_float), ('noise_seed', C.c_uint64)]


class Dril(C.Structure):
  _fields_ = [('state_dim', C.c_int32), ('action_dim', C.c_int32), ('hidden', C.c_int32), ('batch', C.c_int32), ('p_in', C.c_float), ('p', C.c_float),
              ('params', C.c_void_p), ('grad', C.c_void_p), ('opt', Adam), ('workspace', C.c_void_p), ('noise_seed', C.c_uint64), ('q', C.c_float), ('activation', C.c_int32),
              ('depth', C.c_int32), ('reserved', C.c_int32), ('noise_counter', C.c_void_p)]


class SampleArgs(C.Structure):
  _fields_ = [('state', C.c_void_p),
              ('ring_state_a', C.c_void_p), ('ring_a', C.c_void_p), ('capacity_a', C.c_int64), ('row_floats_a', C.c_int32), ('idx_a', C.c_void_p), ('rows_a', C.c_void_p),
              ('ring_state_b', C.c_void_p), ('ring_b', C.c_void_p), ('capacity_b', C.c_int64), ('row_floats_b', C.c_int32), ('idx_b', C.c_void_p), ('rows_b', C.c_void_p)]


IL_PEER_MAX_RANKS, IL_PEER_HANDLE_BYTES, IL_PEER_CHUNK_FLOATS, IL_PEER_WRITE_THROUGH = 16, 64, 2048, 1
IL_PEER_SPIN_LIMIT = 1 << 23   # include/il_hip.h: default bound (polls) of a device-side wait for the peers


class PeerBucket(C.Structure):
  _fields_ = [('rank', C.c_int32), ('world', C.c_int32), ('n', C.c_int64), ('window_offset', C.c_int64), ('windows', C.c_void_p * IL_PEER_MAX_RANKS),
              ('epoch', C.c_void_p), ('status', C.c_void_p), ('spin_limit', C.c_int32), ('flags', C.c_int32), ('n_jobs', C.c_int32), ('reserved', C.c_int32)]


_P = C.c_void_p
_SIGNATURES = {
    'il_peer_region_bytes': (C.c_int64, [C.c_int32, C.c_int64]),
    'il_peer_job_region_bytes': (C.c_int64, [C.c_int32, C.c_int64, C.c_int32]),
    'il_sac_peer_bucket_floats': (C.c_int64, [C.POINTER(Sac), C.c_int32]),
    'il_sac_peer_jobs': (C.c_int32, [C.POINTER(Sac), C.c_int32]),
    'il_peer_window_alloc': (C.c_int, [C.c_int64, C.POINTER(C.c_void_p), C.c_char_p, C.POINTER(C.c_int32)]),
    'il_peer_window_open': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'il_peer_window_close': (C.c_int, [_P]),
    'il_peer_window_free': (C.c_int, [_P]),
    'il_peer_allreduce_mean': (C.c_int, [C.POINTER(PeerBucket), _P, _P]),
    'il_sac_dp_phase_peer': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.c_int32, _P, _P, C.c_uint32, C.POINTER(PeerBucket), _P]),
    'il_last_error': (C.c_char_p, []),
    'il_abi_version': (C.c_int, []),
    'il_device_info': (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    'il_sync_layout': (None, [C.POINTER(C.c_int32)]),
    'il_trace_enable': (C.c_int, [C.c_int]),
    'il_trace_report': (C.c_int, [C.c_char_p, C.c_int]),
    'il_kernel_stamp_ids': (C.c_int32, []), 'il_kernel_stamps': (C.c_int, [C.POINTER(C.c_uint64)]), 'il_kernel_stamps_clear': (C.c_int, []),
    'il_stream_create_cu_mask': (C.c_int, [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]), 'il_stream_destroy': (C.c_int, [_P]),
    'il_kernel_stamp_workgroups': (C.c_int32, []), 'il_kernel_stamp_rows': (C.c_int, [C.c_int32, C.POINTER(C.c_uint64)]),
    'il_ring_row_floats': (C.c_int32, [C.c_int32, C.c_int32]),
    'il_replay_write_rows': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, _P, C.c_int32, _P]),
    'il_replay_wrap_absorbing': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P]),
    'il_replay_gather': (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int32, _P, _P]),
    'il_mt19937_seed': (C.c_int, [c_u32p, C.c_uint32]),
    'il_mt19937_sample_indices': (C.c_int, [c_u32p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, c_i32p]),
    'il_mt19937_randint': (C.c_int, [c_u32p, C.c_int64, C.c_int32, c_i32p]),
    'il_mt19937_sample_indices_device': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'il_struct_size': (C.c_int32, [C.c_int32]),
    'il_launcher_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'il_launcher_destroy': (C.c_int, [C.c_void_p]),
    'il_launcher_clear': (C.c_int, [C.c_void_p]),
    'il_launcher_add': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]),
    'il_launcher_submit': (C.c_int, [C.c_void_p]),
    'il_launcher_wait': (C.c_int, [C.c_void_p]),
    'il_launcher_pending': (C.c_int64, [C.c_void_p]),
    'il_noise_fill': (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, _P, _P]),
    'il_noise_fill_beta': (C.c_int, [C.c_uint64, _P, C.c_float, C.c_int64, _P, _P]),
    'il_sync_probe': (C.c_int, [_P, C.c_int32, _P]),
    'il_replay_gather_workgroups': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    'il_replay_sample_device': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    'il_replay_draw_resident': (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    'il_replay_sample_population': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    'il_sac_update_population': (C.c_int, [_P, _P, C.c_int32, C.POINTER(Sac), C.c_uint32, _P]),
    'il_gail_step_population': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(Disc), _P]),
    'il_adam_step': (C.c_int, [_P, _P, C.POINTER(Adam), C.c_int64, C.c_uint32, _P]),
    'il_polyak': (C.c_int, [_P, _P, C.c_int64, C.c_double, _P]),
    'il_mlp_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_mlp_stride': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_sac_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_sac_critic_step': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), _P, C.c_uint32, _P]),
    'il_sac_actor_step': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), _P, _P, _P, C.c_uint32, _P]),
    'il_sac_apply_actor_grads': (C.c_int, [C.POINTER(Sac), _P]),
    'il_sac_apply_critic_grads': (C.c_int, [C.POINTER(Sac), _P]),
    'il_sac_prepare': (C.c_int, [C.POINTER(Sac), _P]),
    'il_sac_dp_phase': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.c_int32, _P, _P, C.c_uint32, _P]),
    'il_sac_update': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), _P, _P, _P, _P, C.c_uint32, _P]),
    'il_sac_update_gather': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(Disc), _P, _P, _P, _P, _P, C.c_uint32, _P]),
    'il_sac_update_gather_overlap': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(Disc), _P, _P, _P, _P, _P, C.c_uint32, _P, _P]),
    'il_sac_overlap_enter': (C.c_int, [C.POINTER(Sac), _P]),
    'il_sync_clear_poison': (C.c_int, [_P, _P]),
    'il_sync_layout_ex': (None, [C.POINTER(C.c_int32), C.c_int32]),
    'il_sac_update_gather_peer': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(Disc), _P, _P, _P, _P, _P, C.c_uint32, C.POINTER(PeerBucket), C.POINTER(PeerBucket), _P]),
    'il_gail_step_workgroups': (C.c_int32, [C.POINTER(Disc)]),
    'il_sac_chain_gather_workgroups': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    'il_sac_handoff_timeouts': (C.c_int, [C.POINTER(Sac), C.POINTER(C.c_uint32)]),
    'il_bc_step': (C.c_int, [_P, _P, C.POINTER(Adam), C.c_int32, C.c_int32, C.c_int32, C.POINTER(Batch), _P, C.c_int64, _P, C.c_uint32, _P]),
    'il_actor_act': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_uint64, C.c_uint32, C.c_int32, _P, _P, _P]),
    'il_batch_mix_relabel': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_int64, _P]),
    'il_batch_mix_relabel_dyn': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P]),
    'il_act_mailbox_floats': (C.c_int32, [C.c_int32, C.c_int32]),
    'il_act_step': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_uint64, C.c_uint32, _P, C.c_int64, _P]),
    'il_act_publish': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P]),
    'il_disc_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_gail_disc_step': (C.c_int, [C.POINTER(Disc), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(GailExtra), C.c_uint32, _P]),
    'il_disc_deep_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_deep_sn_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_deep_lds_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_deep_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_gail_deep_step': (C.c_int, [C.POINTER(DiscDeep), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(GailExtra), C.c_uint32, _P]),
    'il_gail_deep_reward': (C.c_int, [C.POINTER(DiscDeep), C.POINTER(Batch), _P, _P, _P, _P]),
    'il_disc_shaped_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_shaped_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_gail_shaped_step': (C.c_int, [C.POINTER(DiscShaped), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(GailExtra), C.c_uint32, _P]),
    'il_gail_shaped_reward': (C.c_int, [C.POINTER(DiscShaped), C.POINTER(Batch), _P, _P, _P, _P]),
    'il_disc_shaped_deep_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_shaped_deep_sn_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_shaped_deep_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_disc_shaped_deep_lds_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_gail_shaped_deep_step': (C.c_int, [C.POINTER(DiscShapedDeep), C.POINTER(Batch), C.POINTER(Batch), _P, C.POINTER(GailExtra), C.c_uint32, _P]),
    'il_gail_shaped_deep_reward': (C.c_int, [C.POINTER(DiscShapedDeep), C.POINTER(Batch), _P, _P, _P, _P]),
    'il_actor_log_prob': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    # general shapes (csrc/general.hip)
    'il_mlp_numel_general': (C.c_int64, [C.c_int32] * 4), 'il_mlp_stride_general': (C.c_int64, [C.c_int32] * 4),
    'il_sac_workspace_floats_general': (C.c_int64, [C.c_int32] * 7), 'il_actor_workspace_floats_general': (C.c_int64, [C.c_int32] * 5),
    'il_sac_update_general': (C.c_int, [C.POINTER(Sac), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_uint32, _P]),
    'il_actor_act_general': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_uint64, C.c_uint32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'il_actor_log_prob_general': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    'il_bc_step_general': (C.c_int, [_P, _P, C.POINTER(Adam), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Batch), _P, C.c_int64, _P, C.c_uint32, _P]),
    'il_gail_disc_step_draw': (C.c_int, [C.POINTER(Disc), C.POINTER(Batch), C.POINTER(Batch), _P, _P, _P, _P, _P, C.c_uint32, _P]),
    'il_gail_disc_step_draw_staged': (C.c_int, [C.POINTER(Disc), C.POINTER(Batch), C.POINTER(Batch), _P, _P, _P, _P, _P, _P, C.c_uint32, _P]),
    'il_gail_disc_step_draw_peer': (C.c_int, [C.POINTER(Disc), C.POINTER(Batch), C.POINTER(Batch), _P, _P, _P, _P, _P, C.c_uint32, C.POINTER(PeerBucket), _P]),
    'il_gail_apply_grads': (C.c_int, [C.POINTER(Disc), _P]),
    'il_gail_reward': (C.c_int, [C.POINTER(Disc), C.POINTER(Batch), _P, _P, _P, _P]),
    'il_gmmil_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_gmmil_reward': (C.c_int, [C.POINTER(Batch), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, C.c_int64, _P]),
    'il_gmmil_sqdist': (C.c_int, [C.POINTER(Batch), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    'il_pwil_reset': (C.c_int, [C.POINTER(Pwil), _P]),
    'il_pwil_scratch_floats': (C.c_int64, [C.c_int32, C.c_double]),
    'il_pwil_reward': (C.c_int, [C.POINTER(Pwil), _P, _P, _P, _P]),
    'il_dril_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_dril_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_dril_bc_step': (C.c_int, [C.POINTER(Dril), C.POINTER(Batch), _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    'il_dril_uncertainty': (C.c_int, [C.POINTER(Dril), C.POINTER(Batch), _P, _P, _P, C.c_uint32, _P, _P, _P]),
    'il_red_numel': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'il_red_workspace_floats': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'il_red_step': (C.c_int, [C.POINTER(Red), C.POINTER(Batch), _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    'il_red_forward': (C.c_int, [C.POINTER(Red), C.POINTER(Batch), C.c_int32, _P, _P, _P, C.c_uint32, _P, _P, _P, _P]),
}


class HipLibraryMissing(ImportError):
  pass


_lib = None


def lib():
  """Loads libil_hip.so once; raises HipLibraryMissing (no CPU fallback exists) when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise HipLibraryMissing(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              f'(hipcc --offload-arch=gfx950). There is no CPU fallback for the update path.')
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
      fn = getattr(handle, name)  # AttributeError here = header / library out of sync
      fn.restype, fn.argtypes = res, args
    _lib = handle
  return _lib


_SYNC_LAYOUT = None


def sync_layout():
  """(slots, timeouts index, gather-workgroups index, stride, spin-limit index, host-flag index) of the il_sync counter buffer, as the loaded library lays it out (include/il_hip.h IL_SYNC_*)."""
  global _SYNC_LAYOUT
  if _SYNC_LAYOUT is None:
    out = (C.c_int32 * 6)()
    lib().il_sync_layout(out)
    _SYNC_LAYOUT = tuple(int(v) for v in out)
  return _SYNC_LAYOUT


def sync_layout_ex():
  """sync_layout() followed by (poison index, first stage-epoch index, first stage-ticket index, main-epoch index): include/il_hip.h IL_SYNC_POISON / IL_SYNC_OV_EPOCH / IL_SYNC_OV_TICKET."""
  out = (C.c_int32 * 10)()
  lib().il_sync_layout_ex(out, 10)
  return tuple(int(v) for v in out)


STAMP_KERNELS = ('k_gail_grad', 'k_gail_reduce', 'k_sac_chain_pair', 'k_dw_adam_critic', 'k_policy_critic_pair', 'k_dw_adam_actor', 'k_gmmil_direct', 'k_pwil')   # IL_STAMP_* (include/il_hip.h)


def kernel_stamps(handle=None) -> dict:
  """il_kernel_stamps as {kernel: dict(begin_us, last_begin_us, first_end_us, end_us, duration_us, workgroups)} for the kernels that ran since the last clear; times are
  microseconds on the device-wide 100 MHz counter (comparable between kernels: begin/end of different kernels give the launch boundaries of one update)."""
  L = handle or lib()
  n = int(L.il_kernel_stamp_ids())
  buf = (C.c_uint64 * (5 * n))()
  check(L.il_kernel_stamps(buf))
  out = {}
  for k in range(n):
    b0, b1, e0, e1, wgs = (int(buf[5 * k + i]) for i in range(5))
    if wgs:
      out[STAMP_KERNELS[k]] = dict(begin_us=b0 / 100.0, last_begin_us=b1 / 100.0, first_end_us=e0 / 100.0, end_us=e1 / 100.0, duration_us=(e1 - b0) / 100.0, workgroups=wgs)
  return out


def kernel_stamp_rows(kernel: str, handle=None):
  """Per-workgroup stamps of the last launch of `kernel` (a STAMP_KERNELS name): list of (begin_us, end_us, placement) for the stamped workgroups; placement =
  XCC_ID << 16 | SE / SH / CU byte (one value per physical CU)."""
  L = handle or lib()
  n = int(L.il_kernel_stamp_workgroups())
  buf = (C.c_uint64 * (4 * n))()
  check(L.il_kernel_stamp_rows(STAMP_KERNELS.index(kernel), buf))
  return [(int(buf[4 * w]) / 100.0, int(buf[4 * w + 1]) / 100.0, int(buf[4 * w + 2])) for w in range(n) if buf[4 * w] and buf[4 * w + 1]]


def kernel_stamp_gates(kernel: str, handle=None):
  """Overlapped launches (il_sac_update_gather_overlap): (first, last) time in us at which a workgroup of the last launch of `kernel` got past its wait for the other
  stream's launch, or None when no workgroup of that launch had such a wait (in-order launches)."""
  L = handle or lib()
  n = int(L.il_kernel_stamp_workgroups())
  buf = (C.c_uint64 * (4 * n))()
  check(L.il_kernel_stamp_rows(STAMP_KERNELS.index(kernel), buf))
  g = [int(buf[4 * w + 3]) / 100.0 for w in range(n) if buf[4 * w] and buf[4 * w + 1] and buf[4 * w + 3]]
  return (min(g), max(g)) if g else None


def check(rc: int):
  if rc != 0:
    raise RuntimeError(f'libil_hip error {rc}: {lib().il_last_error().decode()}')


def stream_ptr():
  import torch
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(t) -> bool:
  """The one definition of "a tensor the kernels may be handed": HIP device memory. Every guard of the host layer asks here (there is no CPU path to fall into);
  tests/host_emu, which runs the kernel sources on the host, substitutes it - nothing else may."""
  return bool(t.is_cuda)


def ptr(t):
  """Device pointer of a torch tensor (None -> NULL)."""
  return None if t is None else C.c_void_p(t.data_ptr())
