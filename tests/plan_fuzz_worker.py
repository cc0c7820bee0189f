"""Worker of tests/test_gpu_plan_fuzz.py: one process = one setting of the plan knobs (they are read once per
process).  Transforms a list of sizes on the GPU and compares every element with the CPU oracle."""
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

    def dev(a):
        return torch.from_numpy(a.view(np.int64)).cuda()

    def host(t):
        return t.cpu().numpy().view(np.uint64)

    for lg in logs:
        n = 1 << lg
        a = O.random_elements(n, 4000 + lg)
        d = dev(a)
        out = torch.empty_like(d)
        e = a.copy(); O.poly_fft(e)
        ctx.poly_fft_dev(d, out, lg); ctx.synchronize()
        assert np.array_equal(host(out), e), ("fft", lg)
        e = a.copy(); O.poly_ifft(e)
        ctx.poly_ifft_dev(d, out, lg); ctx.synchronize()
        assert np.array_equal(host(out), e), ("ifft", lg)
        e = a.copy(); O.poly_coset_fft(e)
        ctx.poly_coset_fft_dev(d, d, lg); ctx.synchronize()          # in place
        assert np.array_equal(host(d), e), ("coset_fft in place", lg)
        if lg >= 1 and lg <= 14:
            for f in (2, 8):
                big = torch.empty((n * f, 4), dtype=torch.int64, device="cuda")
                ctx.poly_lde_dev(dev(a), big, lg, f, coset=(f == 8)); ctx.synchronize()
                assert np.array_equal(host(big), O.poly_lde(a, f, coset=(f == 8))), ("lde", lg, f)
    # 4-step path under the same knobs (argv[2]: "log_n:world:log_chunks,..."): with a small radix cap the column
    # transforms (k_ntt_pass<1>, column mode) and the row transforms both take several passes — compact
    # intermediate widths, 2D twiddle on the first / last pass only, chunked split addressing — and every rank's
    # row block must equal the single-device transform checked above, the inverse must return the input
    if len(sys.argv) > 2 and sys.argv[2]:
        sys.path.insert(0, os.path.join(ROOT, "bench"))
        import sixstep_fullsize
        for spec in sys.argv[2].split(","):
            lg, world, log_chunks = (int(v) for v in spec.split(":"))
            assert lg in logs, "the sixstep sizes must be among the sizes compared with the oracle"
            assert sixstep_fullsize.run(ctx, lg, world, log_chunks, verbose=False, seed=4000 + lg), spec
    print("PLAN-FUZZ-OK", os.environ.get("HODOR_MAX_LOG_R"), os.environ.get("HODOR_TILE_LOG"), os.environ.get("HODOR_MIN_LOG_C"))


if __name__ == "__main__":
    main()
