"""Distributed optimizer wrapper: the MI355X-native counterpart of the reference's only
distributed hook, HorovodOptimizer (pyro/optim/horovod.py:12-55).

The reference all-reduces once PER PARAMETER tensor per step (six 4-byte all-reduces on its
example).  Here the gradients of all parameters are reduced with ONE collective per step over
the flat gradient buffer: ``torch.distributed.all_reduce`` on the ``nccl`` backend (= RCCL over
xGMI on ROCm; ``gloo`` in the CPU tests), followed by a division by the world size -- the
particle-sharded ELBO estimator is a mean over ranks (SURVEY 8e variant 1).  Parameters must be
identical on all ranks at the start: ``broadcast_parameters`` mirrors
hvd.broadcast_parameters(..., root_rank=0) (examples/svi_horovod.py:87-88).
"""
import torch
import torch.distributed as dist

from ..params import _PARAM_STORE


def _sorted(params):
    # deterministic order on every rank, as HorovodOptimizer does (horovod.py:52-55)
    return sorted(params, key=lambda p: _PARAM_STORE.param_name(p) or "")


class RcclOptimizer:
    def __init__(self, pyro_optim, group=None, average=True):
        self.optim = pyro_optim
        self.group = group
        self.average = average
        self._broadcast_done = set()
        self._avg_op = None            # None: untried, True / False: ReduceOp.AVG (un)available
        # test hook: run the collectives (and the multi-rank step structure of SVI) at world size 1
        # too, so that RCCL and its capture into a hipGraph can be exercised on a one-GPU box
        self.force_collective = False
        if hasattr(pyro_optim, "grad_hook"):
            pyro_optim.grad_hook = self._allreduce_flat

    @property
    def zeroes_grads(self):
        # the flat fused Adam zeroes the gradient buffer in its own launch
        return getattr(self.optim, "zeroes_grads", False)

    @property
    def fused_publish(self):
        return getattr(self.optim, "fused_publish", False)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def multi_rank(self):
        return self.world_size > 1 or (self.force_collective and dist.is_initialized())

    def _allreduce_flat(self, flat_grad):
        if not self.multi_rank:
            return
        if self.average and self._avg_op is not False:
            # RCCL averages inside the collective (one launch less on the step's critical path);
            # backends without ReduceOp.AVG (gloo) refuse synchronously, before anything is queued
            try:
                dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, group=self.group)
                self._avg_op = True
                return
            except (RuntimeError, ValueError, NotImplementedError):
                if self._avg_op is True:
                    raise
                self._avg_op = False
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat_grad.div_(self.world_size)

    def broadcast_parameters(self, params, root_rank=0):
        """Every rank starts from rank ``root_rank``'s values: ONE broadcast of the parameters packed
        into a flat buffer per dtype (the reference's hvd.broadcast_parameters is one collective per
        tensor, examples/svi_horovod.py:87-88)."""
        if self.world_size == 1:
            return
        by_dtype = {}
        for p in _sorted(params):
            by_dtype.setdefault((p.dtype, p.device), []).append(p)
        for ps in by_dtype.values():
            with torch.no_grad():
                flat = torch.cat([p.data.reshape(-1) for p in ps]) if len(ps) > 1 else \
                    ps[0].data.reshape(-1).clone()
                dist.broadcast(flat, src=root_rank, group=self.group)
                off = 0
                for p in ps:
                    n = p.numel()
                    p.data.copy_(flat[off:off + n].view_as(p.data))
                    off += n

    # ---- two-phase interface used by the hipGraph step (SVI(hip_graph=True)): the collective
    # stays an ordinary eager RCCL launch between two captured graphs
    # [loss + backward] -> reduce_gradients -> [optimizer update + gradient zeroing]
    def reduce_gradients(self, params):
        if not hasattr(self.optim, "grad_buffers"):
            raise RuntimeError("reduce_gradients needs the flat-buffer optimizer "
                               "(pyro_amd.optim.Adam / ClippedAdam)")
        for flat in self.optim.grad_buffers():     # one buffer unless parameters appeared late
            self._allreduce_flat(flat)

    def apply(self, params, *args, **kwargs):
        self.optim(_sorted(params), *args, skip_grad_hook=True, **kwargs)

    def __call__(self, params, *args, **kwargs):
        params = _sorted(params)
        fresh = [p for p in params if p not in self._broadcast_done]
        if fresh:
            # newly created parameters start identical on every rank
            self.broadcast_parameters(fresh)
            self._broadcast_done.update(fresh)
        if not hasattr(self.optim, "grad_hook") and self.world_size > 1:
            # generic per-parameter optimizer: pack -> one all-reduce -> unpack
            grads = [p.grad for p in params if p.grad is not None]
            if grads:
                flat = torch.cat([g.reshape(-1) for g in grads])
                self._allreduce_flat(flat)
                off = 0
                for g in grads:
                    n = g.numel()
                    g.copy_(flat[off:off + n].view_as(g))
                    off += n
        self.optim(params, *args, **kwargs)

    def get_state(self):
        return self.optim.get_state()

    def set_state(self, state):
        self.optim.set_state(state)
