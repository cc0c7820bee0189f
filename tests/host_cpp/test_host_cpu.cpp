// test_host_cpu.cpp — the HOST-ONLY half of the C++ mirror (hodor_amd/csrc/host/hodor.hpp) executed without a device:
// Field on a device-less context, Domain, the coset combiners, FRIProof::from_bytes / to_bytes, NaiveFriIop::verify_proof
// (_strict), TrivialBlake2sIOP::verify_query, Transcript — against a fixture the Python restatement (oracle/pyref.py) wrote,
// and the refusal of every device-side constructor.  The device half is tests/host_cpp/test_host.cpp (-m gpu).
//   usage: test_host_cpu fixture.bin
// fixture: u64 proof_len | proof | u64 index | 32 B expected value | u64 domain | u64 lde_factor | 32 B transcript challenge
//          bytes | 32 B transcript challenge element | 32 B a leaf value | u64 leaf index | u64 path_len | path | 32 B root
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../../hodor_amd/csrc/host/hodor.hpp"

using namespace hodor;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static const uint64_t MODULUS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> fx((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t o = 0;
    auto u64 = [&]() { uint64_t v; memcpy(&v, fx.data() + o, 8); o += 8; return v; };
    auto bytes = [&](size_t n) { std::vector<uint8_t> b(fx.begin() + o, fx.begin() + o + n); o += n; return b; };
    auto fr = [&]() { Fr v; memcpy(v.l, fx.data() + o, 32); o += 32; return v; };

    Field F(MODULUS, 7, -1);                         // no device: host helpers only
    CHECK(F.S() == 32);
    CHECK(F.mul(F.one(), F.one()) == F.one());
    CHECK(F.mul(F.inverse(F.from_u64(12345)), F.from_u64(12345)) == F.one());
    {   // Domain::new_for_size (src/domains/mod.rs:21-44): rounding up, generator of the right order, Err beyond 2^S
        Domain d = Domain::new_for_size(F, 1000);
        CHECK(d.size == 1024 && d.power_of_two == 10);
        CHECK(F.pow(d.generator, 1024) == F.one() && !(F.pow(d.generator, 512) == F.one()));
        bool threw = false;
        try { Domain::new_for_size(F, (1ull << 32) + 1); } catch (const SynthesisError &e) { threw = e.code == HODOR_ERR_SIZE; }
        CHECK(threw);
        CHECK((Domain::coset_for_natural_index_and_size(5, 16) == std::vector<size_t>{5, 13}));
        CHECK((Domain::coset_for_natural_index_and_size(13, 16) == std::vector<size_t>{5, 13}));
        CHECK((Domain::index_and_size_for_next_domain(13, 16) == std::pair<size_t, size_t>{5, 8}));
    }
    {   // Worker (src/fft/multicore.rs:16-107): chunk sizes, log_num_cpus, and a scope that runs every chunk exactly once
        Worker w(6);
        CHECK(w.log_num_cpus() == 2 && Worker(1).log_num_cpus() == 0 && Worker(256).log_num_cpus() == 8);
        CHECK(w.get_chunk_size(3) == 1 && w.get_chunk_size(6) == 1 && w.get_chunk_size(100) == 16);
        std::vector<Fr> v(100, F.zero());
        MutSlice all(v.data(), v.size(), nullptr);                 // (no handle behind it: nothing to write back)
        w.scope(v.size(), [&](Worker::Scope &scope, size_t chunk) {
            size_t i = 0;
            for (auto c : all.chunks_mut(chunk)) {
                scope.spawn([&F, c, i, chunk] { size_t k = i * chunk; for (Fr &e : c) e = F.from_u64(++k); });
                i++;
            }
        });
        bool ok = true;
        for (size_t k = 0; k < v.size(); k++) ok = ok && v[k] == F.from_u64(k + 1);
        CHECK(ok);
        CHECK(all.chunks_mut(16).size() == 7 && all.chunks_mut(16).back().size() == 4);
    }
    for (size_t n : {4u, 16u, 1024u})                 // the combiners' index maps are inverse to each other
        for (size_t i = 0; i < n; i++) {
            CHECK(TrivialCombiner::tree_index_into_natural_index(TrivialCombiner::natural_index_into_tree_index(i, n), n) == i);
            CHECK(Coset2Combiner::tree_index_into_natural_index(Coset2Combiner::natural_index_into_tree_index(i, n), n) == i);
        }

    // the proof the Python restatement produced: parse, re-encode, verify, tamper
    const std::vector<uint8_t> raw = bytes(u64());
    const size_t index = u64();
    const Fr expected = fr();
    const size_t domain = u64(), lde_factor = u64();
    FRIProof proof = FRIProof::from_bytes(raw);
    CHECK(proof.to_bytes() == raw);
    CHECK(proof.lde_factor == lde_factor && proof.initial_degree_plus_one * lde_factor == domain);
    CHECK(NaiveFriIop::verify_proof(F, proof, index, expected));
    CHECK(NaiveFriIop::verify_proof_strict(F, proof, domain, lde_factor, 1, index, expected));
    CHECK(!NaiveFriIop::verify_proof_strict(F, proof, domain, lde_factor * 2, 1, index, expected));
    {
        Fr wrong = expected;
        wrong.l[0] ^= 1;
        CHECK(!NaiveFriIop::verify_proof(F, proof, index, wrong));
        FRIProof bad = FRIProof::from_bytes(raw);
        bad.queries[2].path_[0][5] ^= 0x40;
        CHECK(!NaiveFriIop::verify_proof(F, bad, index, expected));
        bool threw = false;
        try { FRIProof::from_bytes(std::vector<uint8_t>(raw.begin(), raw.begin() + raw.size() / 2)); }
        catch (const SynthesisError &) { threw = true; }
        CHECK(threw);
    }

    // Transcript (src/transcript/mod.rs:26-80): commit 32 bytes and one field element, two challenges
    const Hash32 want_bytes = bytes(32);
    const Fr want_elem = fr();
    {
        Transcript t(F);
        std::vector<uint8_t> msg(32);
        for (int i = 0; i < 32; i++) msg[i] = (uint8_t)i;
        t.commit_bytes(msg);
        t.commit_field_element(F.from_u64(12345));
        CHECK(t.get_challenge_bytes() == want_bytes);
        CHECK(t.get_challenge() == want_elem);
    }

    // one Merkle opening of the restated tree: IOP::verify (src/iop/blake2s_trivial_iop.rs:236-249)
    {
        TrivialBlake2sIopQuery q;
        q.value_ = fr();
        q.index = u64();
        const size_t plen = u64();
        for (size_t k = 0; k < plen; k++) q.path_.push_back(bytes(32));
        const Hash32 root = bytes(32);
        CHECK(TrivialBlake2sIOP::verify_query(F, q, root));
        q.value_.l[1] ^= 2;
        CHECK(!TrivialBlake2sIOP::verify_query(F, q, root));
    }
    CHECK(o == fx.size());

    // every device-side constructor refuses on this field: there is no CPU path behind the mirror
    {
        bool threw = false;
        try { auto p = from_coeffs(F, std::vector<Fr>(8, F.one())); (void)p; } catch (const SynthesisError &e) { threw = e.code == HODOR_ERR_DEVICE; }
        CHECK(threw);
        threw = false;
        try { auto p = Polynomial<Values>::new_for_size(F, 8); (void)p; } catch (const SynthesisError &e) { threw = e.code == HODOR_ERR_DEVICE; }
        CHECK(threw);
    }
    if (failures) { printf("%d check(s) failed\n", failures); return 1; }
    printf("host-only checks passed\n");
    return 0;
}
