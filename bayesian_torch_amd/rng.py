"""Draw management for the fused kernels.

Two modes (``set_mode``):

``"philox"`` (default, the product path)
    eps and the Flipout signs are generated on chip from counter coordinates
    ``(seed, call, layer_id, tensor, global sample id, element)``; nothing weight-sized is
    written to or read from HBM for the draw.  ``layer.eps_*`` buffers are NOT updated
    (``layer.materialize_last_draw()`` regenerates them on demand from the same counters).

``"torch"`` (parity mode)
    the draw is made exactly where the reference makes it -- ``eps.normal_()`` /
    ``uniform_(-1, 1).sign()`` on the input's device with torch's default generator, in the
    reference's draw order (linear_variational.py:164-174, conv_flipout.py:385-402) -- and the
    kernels READ it.  ``layer.eps_*`` hold the last draw as in the reference.

The seed defaults to ``torch.initial_seed()``; ``call`` advances once per layer forward and
restarts whenever the seed changes, so ``torch.manual_seed(s)`` makes a run reproducible.
"""
import threading

import torch

_state = threading.local()
_layer_counter = [0]
_mode = ["philox"]


def set_mode(mode):
    if mode not in ("philox", "torch"):
        raise ValueError("rng mode must be 'philox' or 'torch'")
    _mode[0] = mode


def get_mode():
    return _mode[0]


def new_layer_id():
    _layer_counter[0] += 1
    return _layer_counter[0]


def assign_layer_ids(model, first=1):
    """Number the Bayesian layers of ``model`` by position (``model.modules()`` order): layer_id is the RNG coordinate
    that separates the layers' draw streams, so two copies of a model (``copy.deepcopy``, or the same architecture built
    twice in any order) draw identically for the same (seed, call).  ``dnn_to_bnn`` calls this; call it yourself for a
    hand-assembled model of ``bayesian_torch_amd.layers`` modules if construction-order independence matters."""
    n = first
    for m in model.modules():
        if hasattr(m, "_layer_id") and hasattr(m, "_kl_segments"):
            m._layer_id = n
            n += 1
    return n - first


def manual_seed(seed):
    """Pin the Philox seed explicitly (otherwise torch.initial_seed() is followed)."""
    _state.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state.pinned = True
    _state.call = 0


def _sync():
    if not getattr(_state, "pinned", False):
        s = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if getattr(_state, "seed", None) != s:
            _state.seed = s
            _state.call = 0


def seed():
    _sync()
    return _state.seed


def next_call():
    _sync()
    c = getattr(_state, "call", 0)
    _state.call = (c + 1) & 0xFFFFFFFF
    return c


def peek_call():
    _sync()
    return getattr(_state, "call", 0)


def set_call(c):
    _sync()
    _state.call = int(c) & 0xFFFFFFFF
