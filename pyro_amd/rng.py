"""Process-wide counter-based RNG state for the HIP kernels (Philox4x32-10).

The stream is OWNED by torch's default generator, as the reference's draws are (pyro/util.py:37-45 seeds
torch / random / numpy and nothing else; every ``rsample`` then goes through torch's generator): the Philox
seed is ``torch.initial_seed()`` and a call of ``torch.manual_seed`` -- by ``pyro.set_rng_seed`` or by the
user directly -- restarts the stream at block 0.  ``torch.manual_seed(s)`` alone therefore reproduces a
run; ``torch.get_rng_state`` / ``torch.set_rng_state`` / ``torch.random.fork_rng`` rewind it with the generator
(see _follow_state_calls; a ``manual_seed`` bound by ``from torch import manual_seed`` BEFORE this package was
imported bypasses the counter: re-seeding with the seed already set is then not noticed -- ``pyro.set_rng_seed``
is the route that never depends on that).  Every draw advances a host-side 64-bit block offset, so the sequence is reproducible and
independent of launch geometry; ranks of a multi-GPU job de-correlate by seeding with ``seed + rank``
exactly as the reference de-correlates chains (pyro/infer/mcmc/api.py:107).
"""
import random

import numpy as np
import torch

from . import kernels

_MASK = 0xFFFFFFFFFFFFFFFF
_STATE = {"seed": 0, "offset": 0, "torch_seed": None}
_SEEDINGS = [0]          # calls of torch.manual_seed / torch.seed so far


def _count_seedings():
    """``torch.manual_seed(s)`` with the seed ALREADY set leaves nothing to read off the generator (its
    initial seed is the same and its state only moves when somebody draws from it), yet it means "start
    over": the two entry points are wrapped to count their calls.  The wrappers change nothing else."""
    import functools

    def counting(fn):
        if getattr(fn, "_pyro_amd_counts", False):
            return fn

        @functools.wraps(fn)
        def seeded(*args, **kwargs):
            _SEEDINGS[0] += 1
            return fn(*args, **kwargs)
        seeded._pyro_amd_counts = True
        return seeded

    for name in ("manual_seed", "seed"):
        wrapped = counting(getattr(torch.random, name))
        setattr(torch.random, name, wrapped)
        setattr(torch, name, wrapped)


_count_seedings()

# torch.get_rng_state() / torch.set_rng_state() / torch.random.fork_rng() snapshot and restore the default
# generator; the reference's draws come out of that generator, so restoring it replays them
# (pyro/util.py:48-63 get_rng_state / set_rng_state are built on it).  The Philox position lives on the host
# beside the generator, not in its state bytes: the two module-level functions are wrapped (fork_rng calls them)
# so that a snapshot REMEMBERS the position and a restore of that snapshot rewinds it -- matched by the identity
# of the returned state tensor, else by its bytes (a state made by another process, or two snapshots with
# identical generator bytes taken at different Philox positions and then copied, cannot be told apart:
# pyro.get_rng_state / pyro.set_rng_state carry the position explicitly and are the exact route).
_SNAPSHOTS = {}          # id(state tensor) or bytes digest -> (weakref or None, Philox state)
_SNAPSHOT_LIMIT = 256


def _snapshot_key(t):
    import hashlib
    return hashlib.blake2b(t.numpy().tobytes(), digest_size=16).digest()


def _follow_state_calls():
    import functools
    import weakref
    if getattr(torch.random.get_rng_state, "_pyro_amd_follows", False):
        return
    get0, set0 = torch.random.get_rng_state, torch.random.set_rng_state

    @functools.wraps(get0)
    def get_rng_state(*args, **kwargs):
        st = get0(*args, **kwargs)
        try:
            _follow_torch()
            snap = dict(_STATE)
            if len(_SNAPSHOTS) >= _SNAPSHOT_LIMIT:
                for k in list(_SNAPSHOTS)[:_SNAPSHOT_LIMIT // 2]:
                    del _SNAPSHOTS[k]
            _SNAPSHOTS[id(st)] = (weakref.ref(st), snap)
            _SNAPSHOTS[_snapshot_key(st)] = (None, snap)
        except Exception:      # noqa: BLE001  (never in the way of torch's own function)
            pass
        return st

    @functools.wraps(set0)
    def set_rng_state(new_state, *args, **kwargs):
        out = set0(new_state, *args, **kwargs)
        try:
            ent = _SNAPSHOTS.get(id(new_state))
            if ent is None or ent[0] is None or ent[0]() is not new_state:
                ent = _SNAPSHOTS.get(_snapshot_key(new_state))
            if ent is not None:
                _STATE.update(ent[1])
                _STATE["torch_seed"] = (torch.initial_seed(), _SEEDINGS[0])
        except Exception:      # noqa: BLE001
            pass
        return out

    get_rng_state._pyro_amd_follows = True
    for name, fn in (("get_rng_state", get_rng_state), ("set_rng_state", set_rng_state)):
        setattr(torch.random, name, fn)
        setattr(torch, name, fn)


_follow_state_calls()


def _follow_torch():
    """The default generator was re-seeded since the last draw: the stream restarts under the new seed."""
    if torch.compiler.is_compiling():
        return          # (dynamo is recording: the wrapper that owns the draws synchronised before the call)
    ts = (torch.initial_seed(), _SEEDINGS[0])
    if ts != _STATE["torch_seed"]:
        _STATE["torch_seed"] = ts
        _STATE["seed"] = int(ts[0]) & _MASK
        _STATE["offset"] = 0


def current_seed():
    _follow_torch()
    return _STATE["seed"]
# While a hipGraph is being captured the block offset of a draw cannot be a launch constant (every
# replay would repeat the same numbers): draws are addressed relative to a device-resident base
# counter that the graph itself advances (pa_counter_add) -- see GraphCapture below.
_CAPTURE = {"active": None}


def set_rng_seed(seed):
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    _follow_torch()


def get_rng_state():
    return {"torch": torch.get_rng_state(), "random": random.getstate(),
            "numpy": np.random.get_state(), "philox": dict(_STATE)}


def set_rng_state(state):
    torch.set_rng_state(state["torch"])
    random.setstate(state["random"])
    np.random.set_state(state["numpy"])
    _STATE.update(state["philox"])
    _STATE["torch_seed"] = (torch.initial_seed(), _SEEDINGS[0])   # the restored generator owns the stream again


def reserve(n_elements, dtype):
    """Reserve Philox blocks for ``n_elements`` draws; returns (seed, offset, offset_dev) where
    ``offset_dev`` is None (eager) or the device base counter of the active graph capture."""
    per = 4 if dtype == torch.float32 else 2
    _follow_torch()
    off = _STATE["offset"]
    _STATE["offset"] = off + (int(n_elements) + per - 1) // per
    cap = _CAPTURE["active"]
    if cap is not None:
        return _STATE["seed"], off - cap.start, cap.base
    return _STATE["seed"], off, None


def reserve_blocks(n_blocks):
    """Reserve ``n_blocks`` whole Philox blocks (draws that key one block per element, such as the
    Gamma sampler's rejection loop); returns (seed, offset, offset_dev) like ``reserve``."""
    _follow_torch()
    off = _STATE["offset"]
    _STATE["offset"] = off + int(n_blocks)
    cap = _CAPTURE["active"]
    if cap is not None:
        return _STATE["seed"], off - cap.start, cap.base
    return _STATE["seed"], off, None


def normal(shape, dtype, device):
    """Standard normal draws from the Philox stream (HIP kernel, GPU only)."""
    n = 1
    for s in shape:
        n *= int(s)
    seed, off, off_dev = reserve(n, dtype)
    return kernels.philox_normal(tuple(shape), dtype, device, seed, off, off_dev)


_default_normal = normal


def uniform(shape, dtype, device):
    n = 1
    for s in shape:
        n *= int(s)
    seed, off, off_dev = reserve(n, dtype)
    return kernels.philox_uniform(tuple(shape), dtype, device, seed, off, off_dev)


class GraphCapture:
    """Makes the Philox stream replay-safe: inside ``with GraphCapture(device) as cap`` every draw
    reads its block offset as ``*base + relative offset``; ``cap.finish()`` (still inside the
    capture) appends the node that advances ``*base`` by the blocks one replay consumes.  A replay
    therefore draws exactly the numbers the same step would draw eagerly: call
    ``cap.before_replay()`` to (re)synchronise the base with the host-side offset and
    ``cap.after_replay()`` to advance the host mirror."""

    def __init__(self, device):
        # device words {Philox block base, publish sequence}: pa_publish_scalar's `counter`
        self.base = torch.zeros((2,), dtype=torch.int64, device=device)
        self.start = None
        self.used = None
        self._base_value = None      # what *base holds on the device, if known

    def __enter__(self):
        assert _CAPTURE["active"] is None, "nested RNG graph captures are not supported"
        _follow_torch()
        self.seed = _STATE["seed"]          # a launch constant of every draw recorded in here
        self.start = _STATE["offset"]
        _CAPTURE["active"] = self
        return self

    def finish(self, publish=None):
        """Append the node that advances ``*base`` by the blocks one replay consumes.  With
        ``publish`` = (device scalar, pinned float64[1], pinned int64[1]) the same node also hands
        that scalar (the step's loss) to the host (kernels.publish_scalar)."""
        self.used = _STATE["offset"] - self.start
        if publish is not None:
            kernels.publish_scalar(publish[0], publish[1], publish[2], self.base, self.used)
        else:
            kernels.counter_add(self.base, self.used)

    def finish_args(self, publish):
        """The arguments ``finish(publish)`` would launch with, for a caller that folds the node
        into its own last launch (the optimizer update: kernels.adam_step(publish=...))."""
        self.used = _STATE["offset"] - self.start
        return (publish[0], publish[1], publish[2], self.base, self.used)

    def __exit__(self, *exc):
        _CAPTURE["active"] = None
        _STATE["offset"] = self.start      # capturing executes nothing: no draws were consumed

    def stale(self):
        """The default generator was re-seeded after the capture: its draws carry the old seed."""
        _follow_torch()
        return getattr(self, "seed", _STATE["seed"]) != _STATE["seed"]

    def before_replay(self):
        if self._base_value != _STATE["offset"]:
            self.base[:1].fill_(_STATE["offset"])
            self._base_value = _STATE["offset"]

    def after_replay(self):
        _STATE["offset"] += self.used
        self._base_value += self.used
