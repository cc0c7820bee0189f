// gen_fixtures.rs — OUR file (not part of matter-labs/hodor): an in-crate test that makes the Rust
// reference itself produce the known answers of tests/golden/fullsize_digests.json.  run.sh copies it
// into a scratch copy of the crate's src/ (it needs the pub(crate) items bn256::Fr and
// fft::multicore::Worker).  NEVER COMPILED in the build image (no rustc) — expect to fix small
// compile errors on first use; the algorithmic content (stream, call sequence, byte formats) is what
// the C oracle and the GPU path are checked against.
use crate::bn256::Fr;
use crate::fft::multicore::Worker;
use crate::fri::{FriIop, FriProofPrototype, NaiveFriIop};
use crate::iop::blake2s_trivial_iop::{Blake2sIopTree, TrivialBlake2sIOP};
use crate::iop::IopTree;
use crate::polynomials::{Coefficients, Polynomial, Values};
use ff::{PrimeField, PrimeFieldRepr};

const SEED_NTT: u64 = 0x484F444F52;

// SplitMix64 output number m of the stream seeded `seed` (oracle/hodor_oracle.c:splitmix64_out)
fn sm64(seed: u64, m: u64) -> u64 {
    let mut z = seed.wrapping_add((m.wrapping_add(1)).wrapping_mul(0x9E3779B97F4A7C15));
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}

// element i: first of 16 candidates (4 outputs each, top limb masked to NUM_BITS - 192 bits) below p,
// as a canonical residue -> Fr::from_repr (o_gen_elements)
fn gen_elements(first: u64, count: usize, seed: u64) -> Vec<Fr> {
    let mask = (1u64 << (Fr::NUM_BITS - 192)) - 1;
    (0..count as u64).map(|r| {
        let i = first + r;
        for t in 0..16u64 {
            let mut repr = <Fr as PrimeField>::Repr::default();
            for k in 0..4u64 { repr.as_mut()[k as usize] = sm64(seed, 4 * (16 * i + t) + k); }
            repr.as_mut()[3] &= mask;
            if let Ok(x) = Fr::from_repr(repr) { return x; }
        }
        let mut repr = <Fr as PrimeField>::Repr::default();
        for k in 0..3u64 { repr.as_mut()[k as usize] = sm64(seed, 4 * (16 * i + 15) + k); }
        Fr::from_repr(repr).unwrap()
    }).collect()
}

// BLAKE2s-256 (no key, no personalisation) of the raw little-endian Montgomery limbs = hashlib.blake2s
fn digest(values: &[Fr]) -> String {
    let mut st = blake2s_simd::Params::new().hash_length(32).to_state();
    let mut buf = [0u8; 32];
    for v in values {
        v.into_raw_repr().write_le(&mut buf[..]).unwrap();
        st.update(&buf);
    }
    hex::encode(st.finalize().as_bytes())
}

fn raw_bytes(v: &Fr) -> [u8; 32] { let mut b = [0u8; 32]; v.into_raw_repr().write_le(&mut b[..]).unwrap(); b }

#[test]
#[ignore]
fn gen_fixtures() {
    let worker = Worker::new();
    let mut out = String::from("{\n \"ntt\": {\n");
    let sizes = [20u32, 22, 24];
    for (si, log_n) in sizes.iter().enumerate() {
        let n = 1usize << log_n;
        let a = gen_elements(0, n, SEED_NTT);
        let fwd: Polynomial<Fr, Values> = Polynomial::<Fr, Coefficients>::from_coeffs(a.clone()).unwrap().fft(&worker);
        let cos = Polynomial::<Fr, Coefficients>::from_coeffs(a.clone()).unwrap().coset_fft(&worker);
        let inv = Polynomial::<Fr, Values>::from_values(a.clone()).unwrap().ifft(&worker);
        let back = Polynomial::<Fr, Values>::from_values(fwd.as_ref().to_vec()).unwrap().ifft(&worker);
        assert!(back.as_ref() == &a[..]);
        out += &format!("  \"{}\": {{\"input\": \"{}\", \"fft\": \"{}\", \"coset_fft\": \"{}\", \"ifft\": \"{}\"}}{}\n",
                        log_n, digest(&a), digest(fwd.as_ref()), digest(cos.as_ref()), digest(inv.as_ref()),
                        if si + 1 < sizes.len() { "," } else { "" });
    }
    out += " },\n \"lde\": {\n";
    let sizes = [18u32, 22];
    for (si, log_n) in sizes.iter().enumerate() {
        let a = gen_elements(0, 1usize << log_n, SEED_NTT + 1);
        let lde = Polynomial::<Fr, Coefficients>::from_coeffs(a.clone()).unwrap().lde(&worker, 8).unwrap();
        let tree = Blake2sIopTree::<Fr>::create(lde.as_ref());
        let clde = Polynomial::<Fr, Coefficients>::from_coeffs(a.clone()).unwrap().coset_lde(&worker, 8).unwrap();
        let ctree = Blake2sIopTree::<Fr>::create(clde.as_ref());
        out += &format!("  \"{}\": {{\"input\": \"{}\", \"lde\": \"{}\", \"root\": \"{}\", \"coset_lde\": \"{}\", \"coset_root\": \"{}\"}}{}\n",
                        log_n, digest(&a), digest(lde.as_ref()), hex::encode(tree.get_root().as_ref()),
                        digest(clde.as_ref()), hex::encode(ctree.get_root().as_ref()),
                        if si + 1 < sizes.len() { "," } else { "" });
    }
    out += " },\n \"fri\": {\n";
    let sizes = [17u32, 23];
    for (si, log_deg) in sizes.iter().enumerate() {
        let a = gen_elements(0, 1usize << log_deg, SEED_NTT + 2);
        let code = Polynomial::<Fr, Coefficients>::from_coeffs(a).unwrap().lde(&worker, 8).unwrap();
        let proto = NaiveFriIop::<Fr, TrivialBlake2sIOP<Fr>>::proof_from_lde(&code, 8, 1, &worker).unwrap();
        // canonical prototype encoding (oracle/hodor_oracle.h: o_fri_serialize):
        // u64le num_steps | roots[num_steps + 1] | challenges[num_steps] (raw LE) | final_root | u64le n_final | final coeffs
        let roots = proto.get_roots();
        let num_steps = proto.challenges.len() as u64;
        let mut ser: Vec<u8> = num_steps.to_le_bytes().to_vec();
        for r in roots.iter() { ser.extend_from_slice(r.as_ref()); }
        for c in proto.challenges.iter() { ser.extend_from_slice(&raw_bytes(c)); }
        ser.extend_from_slice(proto.get_final_root().as_ref());
        let fin = proto.get_final_coefficients();
        ser.extend_from_slice(&(fin.len() as u64).to_le_bytes());
        for c in fin.iter() { ser.extend_from_slice(&raw_bytes(c)); }
        out += &format!("  \"{}\": {{\"codeword\": \"{}\", \"serialized\": \"{}\", \"final_root\": \"{}\"}}{}\n",
                        log_deg + 3, digest(code.as_ref()), hex::encode(&ser), hex::encode(proto.get_final_root().as_ref()),
                        if si + 1 < sizes.len() { "," } else { "" });
    }
    out += " }\n}\n";
    let path = std::env::var("HODOR_FIXTURES_OUT").unwrap_or_else(|_| "fullsize_digests_rust.json".to_string());
    std::fs::write(&path, out).unwrap();
    println!("wrote {}", path);
}
