"""Where does a capture with generated kernels fail? (developer probe)"""
import sys
import torch
sys.path.insert(0, ".")
from pyro_amd.ops import fuser

dev = torch.device("cuda:0")
x = torch.randn(64, 8, device=dev)
w = torch.randn(8, device=dev, requires_grad=True)
which = sys.argv[1]


def fwd():
    return ((x * w).sigmoid().log() * 0.5 + 1.0)


def fwd_sum():
    return ((x * w).sigmoid().log().sum(1) * 0.5).sum()


def step():
    w.grad = None
    loss = fwd_sum() + (w ** 2).sum()
    loss.backward()
    return loss.detach(), w.grad


fn = {"fwd": fwd, "fwd_sum": fwd_sum, "step": step}[which]
with fuser.Fuser():
    fn()
torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with fuser.Fuser():
        out = fn()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", which, flush=True)
