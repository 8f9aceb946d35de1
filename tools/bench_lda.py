"""LDA factor kernel at the BASELINE config-4 size (developer tool): time per call from a hipGraph
of 10 calls, error against float64 at a reduced size."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pyro_amd import kernels as k
from tools.bench_glm_planes import graph_time

dev = torch.device("cuda:0")


def main():
    for (B, Wd, T, V) in [(100_000, 64, 8, 1024), (4096, 64, 8, 1024), (100_000, 64, 16, 1024)]:
        g = torch.Generator(device="cpu").manual_seed(0)
        words = torch.randint(0, V, (Wd, B), generator=g).to(dev)
        log_theta = torch.log_softmax(torch.randn((B, T), generator=g), -1).to(dev)
        log_phi = torch.log_softmax(torch.randn((T, V), generator=g), -1).to(dev)
        k.lda_set_index_mode(k.LDA_INDEX_OFF)
        try:
            us_a, _ = graph_time(lambda: k.lda_factor_fwd_bwd(words, log_theta, log_phi))
        except Exception as e:  # noqa: BLE001
            us_a = float("nan")
        k.lda_set_index_mode(k.LDA_INDEX_AUTO)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        index = k.lda_build_index(words, V)
        torch.cuda.synchronize()
        t_build = (time.perf_counter() - t0) * 1e3
        us, out = graph_time(lambda: k.lda_factor_fwd_bwd(words, log_theta, log_phi, index=index))
        print(f"   atomic route {us_a:8.1f} us; index build {t_build:.2f} ms ({index.numel() * 4 / 1e6:.1f} MB)")
        bytes_alg = Wd * B * 8.5
        # float64 through the same kernel family as the independent check of the f32 numbers
        o64 = k.lda_factor_fwd_bwd(words, log_theta.double(), log_phi.double(), index=index)
        err = [float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(out, o64)]
        print(f"lda B={B} Wd={Wd} T={T} V={V}: {us:8.1f} us  {bytes_alg/us/1e6:6.3f} TB/s(alg)  "
              f"max rel err vs f64: out {err[0]:.1e} g_theta {err[1]:.1e} g_phi {err[2]:.1e}")


if __name__ == "__main__":
    main()
