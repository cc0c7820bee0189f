"""The host-only proof verifier (hodor_fri_verify_proof*, src/fri/verifier.rs:131-289 restated in csrc/abi_host.hip) parses
bytes a prover hands it: it must never crash, hang or allocate without bound on malformed input, and the strict variant
must accept nothing but the proof it was given.  Seeded mutations of valid proofs (both tree formats): byte flips,
truncations, extensions, every 8-byte-aligned word replaced by hostile counts.  A crash here kills pytest — that is the
assertion; `bench/asan_suite.sh` runs the same file against the ASAN + UBSan build.  No device needed."""
import random

import pytest

import hodor_amd
from oracle import pyref as P

F = P.BN256
LOG_DEG, FACTOR, INDEX = 3, 4, 7
HOSTILE = (0, 1, 2, 3, 31, 32, 33, 1 << 16, 1 << 31, (1 << 32) - 1, 1 << 32, 1 << 40, 1 << 62, (1 << 63) - 1, 1 << 63,
           (1 << 64) - 1, (1 << 64) // 32, (1 << 64) // 32 + 1, (1 << 64) // 64 + 1)


@pytest.fixture(scope="module")
def proofs():
    coeffs = [pow(3, 50 + i, F.p) for i in range(1 << LOG_DEG)]
    lde = P.poly_lde(F, coeffs, FACTOR)
    out = {}
    for combiner in (P.TRIVIAL, P.COSET2):
        proto = P.fri_commit(F, lde, FACTOR, 1, combiner=combiner)
        proof = P.fri_produce_proof(F, proto, lde, INDEX, FACTOR, 1, combiner=combiner)
        out[combiner] = P.fri_proof_to_bytes(proof)
    return out, F.to_mont(lde[INDEX]), len(lde)


def _mutations(raw, rng):
    for _ in range(300):                                    # single-byte damage
        i = rng.randrange(len(raw))
        yield raw[:i] + bytes([raw[i] ^ (1 << rng.randrange(8))]) + raw[i + 1:]
    for cut in sorted({0, 1, 7, 8, 9, 40, len(raw) // 2, len(raw) - 9, len(raw) - 8, len(raw) - 1} |
                      {rng.randrange(len(raw)) for _ in range(40)}):
        yield raw[:cut]                                     # truncation
    for extra in (1, 8, 32, 1000):
        yield raw + bytes(rng.randrange(256) for _ in range(extra))
    for off in range(0, len(raw) - 7, 8):                   # every aligned word as a hostile count / index / length
        for v in HOSTILE if off < 64 or off > len(raw) - 64 else HOSTILE[::4]:
            yield raw[:off] + int(v).to_bytes(8, "little") + raw[off + 8:]
    yield b""
    yield bytes(len(raw))
    yield bytes([255]) * len(raw)


@pytest.mark.parametrize("combiner", [P.TRIVIAL, P.COSET2])
def test_verifier_survives_malformed_proofs_and_the_strict_one_accepts_only_the_original(proofs, combiner):
    raws, expected, n = proofs
    raw = raws[combiner]
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    assert ctx.fri_verify_proof_combined(raw, combiner, INDEX, expected) is True
    assert ctx.fri_verify_proof_strict(raw, n, FACTOR, 1, INDEX, expected, combiner=combiner) is True
    rng = random.Random(1234 + combiner)
    tried = accepted_lenient = 0
    for bad in _mutations(raw, rng):
        if bad == raw:
            continue
        tried += 1
        for strict in (True, False):
            try:
                ok = (ctx.fri_verify_proof_strict(bad, n, FACTOR, 1, INDEX, expected, combiner=combiner) if strict
                      else ctx.fri_verify_proof_combined(bad, combiner, INDEX, expected))
            except hodor_amd.HodorError:
                ok = False                                  # the reference's Err(..) cases
            if strict:
                assert ok is False, "the strict verifier accepted a damaged proof (%d bytes)" % len(bad)
            elif ok:
                # The reference's verifier never reads proof.output_coeffs_at_degree_plus_one and uses the other two
                # trailing words only as Domain::new_for_size(initial_degree_plus_one * lde_factor), which rounds up to a
                # power of two (src/fri/verifier.rs:24, :146; src/domains/mod.rs:21-44): the restated walk accepts a proof
                # whose three trailing words were changed without changing that domain — and nothing else.  (The strict
                # variant binds all three to the caller's expectation.)
                accepted_lenient += 1
                assert len(bad) == len(raw) and bad[:-24] == raw[:-24]
                initial, _, factor = (int.from_bytes(bad[len(bad) - 24 + 8 * k:len(bad) - 16 + 8 * k], "little") for k in range(3))
                size = 1
                while size < initial * factor:
                    size <<= 1
                assert size == n
    assert tried > 500 and accepted_lenient > 0
    ctx.close()
