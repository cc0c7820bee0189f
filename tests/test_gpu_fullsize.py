"""GPU parity at BASELINE.json's FULL sizes (-m gpu), every element, against known answers the CPU
oracle produced in the build container (tests/golden/fullsize_digests.json, generator
tests/golden/gen_fullsize.py):

  * inputs are regenerated on the device from (seed, index) by hodor_gen_elements_dev — the twin of
    oracle/hodor_oracle.c:o_gen_elements — and their digest must equal the oracle's input digest;
  * outputs are downloaded and compared through the BLAKE2s-256 digest of the whole buffer
    ("full compare <= 2^20, BLAKE2s-of-buffer compare above"), which is what the
    reference's own tests do element by element at 2^22 (src/fft/mod.rs:128-184) and 2^22 x 16
    (src/polynomials/mod.rs:1084-1130);
  * the Merkle root and the FRI prototype bytes are compared in full.

The digest is hashlib's BLAKE2s over host memory — no code of this repository sits between the
device buffer and the comparison.  Nothing here reads /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FULL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")))
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hodor_golden.json")))


def digest(t):
    a = t.cpu().numpy() if hasattr(t, "cpu") else np.ascontiguousarray(t)
    return hashlib.blake2s(memoryview(a).cast("B"), digest_size=32).hexdigest()


def dev_elements(ctx, n, seed):
    import torch
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, n, seed)
    return a


@pytest.mark.parametrize("log_n", [1, 7, 12, 17])
def test_generator_twin_matches_oracle(gpu_ctxs, oracles, field_name, log_n):
    """hodor_gen_elements_dev == o_gen_elements, any window of the stream, all three fields."""
    import torch
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    n = 1 << log_n
    for first, seed in ((0, 1), (12345678901, 0x484F444F52), ((1 << 40) + 3, 2 ** 64 - 1)):
        d = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.gen_elements_dev(d, first, n, seed)
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64), O.gen_elements(first, n, seed))


@pytest.mark.parametrize("log_n", sorted(int(k) for k in FULL["ntt"] if "coset_fft" not in FULL["ntt"][k] and int(k) <= 27))
def test_multi_gpu_bench_sizes_on_one_device_every_element(gpu_ctxs, log_n):
    """2^25 .. 2^27 (the totals of `bench.py --gpus 2 / 4 / 8`): the single-device transform of the generator
    stream equals the CPU oracle's digest, and its inverse returns the input (2^28 and 2^29 the same way through
    bench/big_digest.py: profiles/r02/big_digest.txt)."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["ntt"][str(log_n)]
    n = 1 << log_n
    a = dev_elements(ctx, n, e["seed"])
    ctx.synchronize()
    assert digest(a) == e["input"]
    b = torch.empty_like(a)
    ctx.poly_fft_dev(a, b, log_n)
    ctx.synchronize()
    assert digest(b) == e["fft"]
    ctx.poly_ifft_dev(b, b, log_n)
    ctx.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("log_n", sorted(int(k) for k in FULL["ntt"] if "coset_fft" in FULL["ntt"][k]))
def test_config1_ntt_every_element(gpu_ctxs, log_n):
    """BASELINE config[1] (2^24) and the CPU config's size (2^20): fft / coset_fft / ifft digests equal
    the CPU oracle's; ifft(fft(x)) == x bit for bit."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["ntt"][str(log_n)]
    n = 1 << log_n
    a = dev_elements(ctx, n, e["seed"])
    ctx.synchronize()
    assert digest(a) == e["input"]
    b = torch.empty_like(a)
    ctx.poly_fft_dev(a, b, log_n)
    ctx.synchronize()
    assert digest(b) == e["fft"]
    c = torch.empty_like(a)
    ctx.poly_ifft_dev(b, c, log_n)
    ctx.synchronize()
    assert torch.equal(a, c)
    ctx.poly_coset_fft_dev(a, b, log_n)
    ctx.synchronize()
    assert digest(b) == e["coset_fft"]
    ctx.poly_ifft_dev(a, b, log_n)
    ctx.synchronize()
    assert digest(b) == e["ifft"]
    ctx.poly_fft_dev(a, a, log_n)            # in place
    ctx.synchronize()
    assert digest(a) == e["fft"]


def test_the_reference_tests_at_their_own_sizes(gpu_ctxs):
    """test_parallel_radix4_fft (src/fft/mod.rs:128-184: 2^22 points) and test_various_ldes
    (src/polynomials/mod.rs:1084-1130: 2^22 coefficients x 16) over THEIR field, experiments::Fr: the device's transform
    and both of its LDE spellings — Polynomial::lde and the fft of the zero-padded vector, which is what filtering_lde's
    best_lde computes — equal, whole buffer, the results on which the oracle's restated parallel_fft /
    parallel_fft_radix_4 / parallel_DIT_fft resp. lde_using_multiple_cosets / best_lde / best_fft all agreed when
    tests/golden/gen_fullsize.py --ref-sizes wrote the digests."""
    import torch
    ctx, R = gpu_ctxs["experiments"], FULL["reference_tests"]
    e = R["parallel_radix4_fft"]
    log_n = e["log_n"]
    a = dev_elements(ctx, 1 << log_n, e["seed"])
    ctx.synchronize()
    assert digest(a) == e["input"]
    b = torch.empty_like(a)
    ctx.poly_fft_dev(a, b, log_n)
    ctx.synchronize()
    assert digest(b) == e["fft"]
    del a, b
    e = R["various_ldes"]
    log_n, f = e["log_n"], e["factor"]
    n = 1 << log_n
    a = dev_elements(ctx, n, e["seed"])
    ctx.synchronize()
    assert digest(a) == e["input"]
    out = torch.empty((n * f, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(a, out, log_n, f)
    ctx.synchronize()
    assert digest(out) == e["lde"]
    out.zero_()
    out[:n] = a
    ctx.poly_fft_dev(out, out, log_n + f.bit_length() - 1)      # "naive_lde": Polynomial::fft of the padded vector
    ctx.synchronize()
    assert digest(out) == e["lde"]


@pytest.mark.parametrize("log_n", sorted(int(k) for k in FULL["lde"]))
def test_config2_lde_and_commit_every_element(gpu_ctxs, log_n):
    """BASELINE config[2]: lde(8) of 2^22 coefficients and the IOP tree over it, whole buffers."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["lde"][str(log_n)]
    n, f = 1 << log_n, e["factor"]
    a = dev_elements(ctx, n, e["seed"])
    ctx.synchronize()
    assert digest(a) == e["input"]
    out = torch.empty((n * f, 4), dtype=torch.int64, device="cuda")
    nodes = torch.empty((n * f, 32), dtype=torch.uint8, device="cuda")
    ctx.poly_lde_dev(a, out, log_n, f)
    ctx.iop_create_dev(out, n * f, nodes)
    ctx.synchronize()
    assert digest(out) == e["lde"]
    assert bytes(nodes[1].cpu().numpy()).hex() == e["root"]
    assert digest(nodes) == e["nodes"]
    ctx.poly_lde_dev(a, out, log_n, f, coset=True)
    ctx.iop_create_dev(out, n * f, nodes)
    ctx.synchronize()
    assert digest(out) == e["coset_lde"]
    assert bytes(nodes[1].cpu().numpy()).hex() == e["coset_root"]
    del a, out, nodes
    torch.cuda.empty_cache()


@pytest.mark.parametrize("log_n", sorted(int(k) for k in FULL["fri"]))
def test_config3_fri_commit_bytes(gpu_ctxs, log_n):
    """BASELINE config[3]: FRI commit phase on the 2^26 codeword, "proof bytes identical to CPU"."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["fri"][str(log_n)]
    f = e["factor"]
    log_deg = log_n - (f.bit_length() - 1)
    a = dev_elements(ctx, 1 << log_deg, e["seed"])
    code = torch.empty((1 << log_n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(a, code, log_deg, f)
    ctx.synchronize()
    assert digest(code) == e["codeword"]
    proto = ctx.fri_commit_dev(code, 1 << log_n, f, e["out_deg_plus_one"])
    assert proto.num_steps == e["num_steps"]
    assert proto.serialized.hex() == e["serialized"]
    assert proto.final_root.hex() == e["final_root"]
    proto.free()
    del a, code
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- the GPU straight against the
# committed Python big-int + hashlib fixtures (no C oracle in between)
def _h2i(xs):
    return [int(x, 16) for x in xs]


def test_gpu_directly_against_golden_json(gpu_ctxs, field_name):
    from oracle import pyref as P
    from oracle.oracle import array_to_ints, ints_to_array
    F = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}[field_name]
    ctx, cases = gpu_ctxs[field_name], GOLD[field_name]["cases"]
    seen = set()
    for key, c in cases.items():
        if key.startswith("ntt_"):
            a = ints_to_array([F.to_mont(v) for v in _h2i(c["input"])])
            for name in ("fft", "ifft", "coset_fft", "icoset_fft"):
                b = a.copy()
                getattr(ctx, "poly_" + name)(b)
                assert [F.from_mont(v) for v in array_to_ints(b)] == _h2i(c[name]), (key, name)
            seen.add("ntt")
        elif key.startswith("lde_"):
            a = ints_to_array([F.to_mont(v) for v in _h2i(c["input"])])
            assert [F.from_mont(v) for v in array_to_ints(ctx.poly_lde(a, c["factor"]))] == _h2i(c["lde"])
            assert [F.from_mont(v) for v in array_to_ints(ctx.poly_lde(a, c["factor"], coset=True))] == \
                _h2i(c["coset_lde"])
            seen.add("lde")
        elif key.startswith("merkle_"):
            leafs = ints_to_array(_h2i(c["leafs_mont"]))
            nodes = ctx.iop_create(leafs)
            assert [bytes(x).hex() for x in nodes[1:]] == c["nodes"][1:]
            seen.add("merkle")
        elif key.startswith("fri_"):
            coeffs = ints_to_array([F.to_mont(v) for v in _h2i(c["coeffs"])])
            lde = ctx.poly_lde(coeffs, c["lde_factor"])
            proto = ctx.fri_commit(lde, c["lde_factor"], c["out_deg_plus_one"])
            assert proto.serialized.hex() == c["serialized"]
            proto.free()
            seen.add("fri")
    assert seen == {"ntt", "lde", "merkle", "fri"}
