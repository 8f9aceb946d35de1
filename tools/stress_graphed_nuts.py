"""Hazard hunt for NUTS(jit_compile=True) (developer tool): the reference's Gaussian-chain fixtures
with shortened runs, eager and graphed in ONE process, many rounds; every graphed run must equal the
eager run of the same fixture BIT FOR BIT (all samples of all chains) -- a far sharper detector
than the fixtures' statistical tolerances.   PYTHONPATH=. python tools/stress_graphed_nuts.py [rounds]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd.infer.mcmc import MCMC, NUTS
from tests import mcmc_cases as mc


def run(case, jit, dev, warmup=60, samples=40, C=8, dtype=torch.float32):
    dim, chain_len, num_obs = mc.GAUSSIAN_CHAINS[case][:3]
    data = torch.ones(num_obs, dim, dtype=dtype, device=dev)
    one = torch.ones((), dtype=dtype, device=dev)

    def model(data):
        loc = torch.zeros(dim, dtype=dtype, device=dev)
        with pyro.plate("dim", dim, dim=-1):
            for i in range(1, chain_len + 1):
                loc = pyro.sample("loc_{}".format(i), dist.Normal(loc, one))
            with pyro.plate("obs_plate", num_obs, dim=-2):
                pyro.sample("obs", dist.Normal(loc, one), obs=data)

    pyro.set_rng_seed(0)
    k = NUTS(model, jit_compile=jit)
    mcmc = MCMC(k, num_samples=samples, warmup_steps=warmup, num_chains=C)
    mcmc.run(data)
    s = mcmc.get_samples(group_by_chain=True)
    return torch.stack([s["loc_{}".format(i)] for i in range(1, chain_len + 1)]).clone(), k.num_leapfrog_steps


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda")
    cases = sorted(mc.GAUSSIAN_CHAINS)
    ref = {}
    bad = 0
    t0 = time.perf_counter()
    for r in range(rounds):
        order = cases if r % 2 == 0 else cases[::-1]
        for case in order:
            if case not in ref:
                ref[case] = run(case, False, dev)
            if r % 9 == 8:                      # interleave fresh eager runs: allocator churn
                again = run(case, False, dev)
                if not torch.equal(again[0], ref[case][0]):
                    bad += 1
                    print("round %d %s: EAGER run differs from the first eager run" % (r, case), flush=True)
            got = run(case, True, dev)
            same = torch.equal(got[0], ref[case][0]) and got[1] == ref[case][1]
            if not same:
                bad += 1
                d = (got[0] - ref[case][0]).abs()
                first = int((d.reshape(d.shape[0], d.shape[1], d.shape[2], -1).amax((0, 1, 3)) > 0).float().argmax())
                print("round %d %s: GRAPHED differs (max abs %.3e, first differing sample %d, leapfrogs %d vs %d)"
                      % (r, case, float(d.max()), first, got[1], ref[case][1]), flush=True)
        print("round %d done, %d mismatches so far, %.0f s" % (r, bad, time.perf_counter() - t0), flush=True)
    print("RESULT: %d rounds x %d fixtures, %d mismatches" % (rounds, len(cases), bad))
