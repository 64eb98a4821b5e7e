"""CPU oracle for the SAC/GAIL/GMMIL/PWIL update hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy (float32) restatement of the arithmetic the
reference (Kaixhin/imitation-learning) performs on its hot path
(`training.py`, `models.py`, `memory.py`; every function cites the reference
file:line it follows).  It exists so that the HIP kernels can be checked on a
GPU box where `/root/reference` does not exist.

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import anything from here;
  * nothing under `imitation-learning_amd/` may import it -- the product path
    has no CPU fallback and fails loudly when the HIP library is missing.

Pinning: the reference has no tests / golden vectors of its own,
so the pin is the reference *code* executed on CPU torch in the build
container: `tests/golden/make_golden.py` imports `/root/reference` unmodified,
records inputs, injected noise and outputs into `tests/golden/*.npz`, and
`tests/test_oracle_golden.py` checks this restatement against those vectors.
Index draws are additionally pinned against numpy's own legacy RandomState
(the reference's actual third-party RNG, `memory.py:54`).
"""
