"""All conjugate Gaussian-chain sampler fixtures of the reference (tests/mcmc_cases.GAUSSIAN_CHAINS),
eager and graphed, outside the timed GPU suite:  PYTHONPATH=. python tools/run_gaussian_chains.py"""
import time

import torch

from tests import mcmc_cases as mc

if __name__ == "__main__":
    dev = torch.device("cuda")
    for case in sorted(mc.GAUSSIAN_CHAINS):
        for jit in (False, True):
            t0 = time.perf_counter()
            try:
                mc.run_gaussian_chain(dev, case, "nuts", jit_compile=jit)
                status = "ok"
            except AssertionError as e:
                status = "MISS {}".format(e)
            print("{:40s} {:8s} {:6.1f} s  {}".format(case, "graphed" if jit else "eager",
                                                     time.perf_counter() - t0, status), flush=True)
