"""TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference algorithms on the two
hot paths (SVI ELBO gradient under pyro.plate; HMC/NUTS leapfrog) of pyro-ppl/pyro 1.9.1.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the *checker*.  Nothing under ``pyro_amd/`` imports it; the
product path has no CPU fallback.

Parity pinning: every module here is checked against golden vectors produced by the
unmodified reference (imported from /root/reference in the build container by
``tests/golden/make_golden.py``; the vectors are committed under ``tests/golden/``) in
``tests/test_oracle_vs_golden.py``.  Plain numpy (float64 unless told otherwise); the
floating-point formulas restate third-party torch.distributions arithmetic, cited per function.
"""
