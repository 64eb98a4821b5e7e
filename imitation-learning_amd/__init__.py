"""imitation-learning_amd: MI355X-native (gfx950) off-policy update hot path of Kaixhin/imitation-learning.

Python mirrors of the reference's `memory.py`, `models.py`, `training.py` interfaces on top of libil_hip.so (hand-written
HIP kernels behind the C ABI in include/il_hip.h).  Import as `imitation_learning_amd` (the directory name carries a hyphen;
the importable alias at the repo root points its package path here).
"""
from . import _lib  # noqa: F401
from .acting import ActingWorker  # noqa: F401
from .memory import IndexStream, ReplayMemory, seed  # noqa: F401
from .models import (DropoutSoftActor, GAILDiscriminator, GMMILDiscriminator, PWILDiscriminator, REDDiscriminator, RewardRelabeller, SoftActor, TwinCritic, create_target_network,  # noqa: F401
                     make_gail_input, mix_expert_agent_transitions, update_target_network)
from .optim import Adam, AdamW  # noqa: F401
from .training import (BatchedPopulationPlan, PopulationPlan, UpdatePlan, adversarial_imitation_update, behavioural_cloning_update, sac_update,  # noqa: F401
                       target_estimation_update)

__version__ = '0.1.0'
