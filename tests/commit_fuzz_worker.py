"""Worker of tests/test_gpu_plan_fuzz.py for the Merkle / FRI / batch-inversion schedule knobs (read once per
process): trees, FRI commits and batch inversions of a list of sizes must equal the CPU oracle's byte for byte."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import hodor_amd
    from oracle import pyref as P
    from oracle.oracle import Oracle
    logs = [int(x) for x in sys.argv[1].split(",")]
    O = Oracle(P.BN256.p, P.BN256.g)
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    for lg in logs:
        n = 1 << lg
        a = O.random_elements(n, 5000 + lg)
        assert np.array_equal(ctx.iop_create(a), O.iop_create(a)), ("tree", lg)
        if lg >= 2:   # the COSET2 format (combined leaf launch) under the same schedule knobs
            assert np.array_equal(ctx.iop_create_combined(a, hodor_amd.COSET2), O.iop_create_coset2(a)), ("coset2 tree", lg)
        inv = a.copy()
        O.poly_batch_inversion(inv)
        d = torch.from_numpy(a.view(np.int64)).cuda()
        ctx.poly_batch_inversion_dev(d, n)
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64), inv), ("batch inversion", lg)
        if 4 <= lg <= 13:
            for factor, out_deg in ((8, 1), (4, 2)):
                code = O.poly_lde(O.random_elements(n // factor, 5100 + lg), factor)
                want = O.fri_commit(code, factor, out_deg)["serialized"]
                proto = ctx.fri_commit(code, factor, out_deg)
                got = proto.serialized
                proto.free()
                assert got == want, ("fri commit", lg, factor, out_deg)
                want = O.fri_commit(code, factor, out_deg, combiner=1)["serialized"]
                proto = ctx.fri_commit(code, factor, out_deg, combiner=hodor_amd.COSET2)
                got = proto.serialized
                proto.free()
                assert got == want, ("coset2 fri commit", lg, factor, out_deg)
    print("COMMIT-FUZZ-OK")


if __name__ == "__main__":
    main()
