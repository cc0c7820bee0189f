import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


FIELDS = {
    "bn256": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 7),
    "experiments": (3618502788666131213697322783095070105623107215331596699973092056135872020481, 3),
    # not a field of the reference: the BN254 scalar field (254 bits, 2-adicity 28, p mod 2^29 != 1) keeps
    # the "modulus is a run-time parameter" claim honest — both reference fields have S >= 32
    "bn254": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5),
}


@pytest.fixture(scope="session", params=["bn256", "experiments", "bn254"])
def field_name(request):
    return request.param


@pytest.fixture(scope="session")
def oracles():
    from oracle.oracle import Oracle
    return {k: Oracle(*v) for k, v in FIELDS.items()}


@pytest.fixture(scope="session")
def gpu_ctxs():
    """One hodor_ctx per field on cuda:0.  Fails loudly when the HIP extension is missing."""
    import hodor_amd
    ctxs = {k: hodor_amd.Context(v[0], v[1], device=0) for k, v in FIELDS.items()}
    yield ctxs
    for c in ctxs.values():
        c.close()
