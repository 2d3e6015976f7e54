//! refgen — asks the REFERENCE (dusk-poseidon 0.42.0-rc.0 and the crates it delegates to) for everything the repository's oracle
//! could only recollect, on inputs that need no RNG, and writes it as JSON (tests/golden/reference_fixtures.json):
//!
//!   tags        dusk-safe's tag-input bytes and `BlsScalar::hash_to_scalar` of them (src/hades/permutation/scalar.rs:29-31) for the
//!               io-patterns of src/hash.rs:62-85: Merkle4, Merkle2, Other 3/5/15/42 -> 1, (3,3), (5,2), (4,7), (42,5), chunked [3,39]
//!   digests     `Hash::finalize` (src/hash.rs:128-155) for those shapes on three input families
//!   truncated   `Hash::finalize_truncated` (src/hash.rs:164-183): the JubJubScalar's canonical bytes
//!   encryption  `dusk_poseidon::encrypt` (src/encryption.rs:62-95) for lengths 2 / 21 / 42, the tag dusk_safe::encrypt derives, and
//!               one `decrypt` failure (tests/encryption.rs:60-115)
//!
//! Input families (both sides can derive them): "seq" x_j = j + 1; "inv7pow" x_j = 7^-(j+1) (full-width values); "kat" = the ten
//! inputs of the crate's known-answer test (src/hades.rs:94-131), read from tests/golden/hades_kat.json.
//! A scalar is written as the 4 little-endian u64 limbs of its Montgomery form (`BlsScalar.0`: the memory the C ABI takes).
//!
//! GPU-free and library-free on purpose: bindings/rust/run_parity.sh needs an MI355X AND cargo on one machine; this needs cargo.
//! NOT COMPILED in the repository's build image (no Rust toolchain): uses only `Hash`, `Domain`, `encrypt`, `decrypt`,
//! `dusk_safe::{Call, Safe, Sponge, Encryption, encrypt}` the way the reference's own sources and tests do.
use std::cell::RefCell;
use std::fmt::Write as _;
use std::rc::Rc;

use dusk_bls12_381::BlsScalar;
use dusk_bytes::Serializable;
use dusk_jubjub::{JubJubAffine, JubJubScalar, GENERATOR_EXTENDED};
use dusk_poseidon::{decrypt, encrypt, Domain, Hash};
use dusk_safe::{Call, Safe, Sponge};

const KAT_JSON: &str = include_str!("../../../../tests/golden/hades_kat.json");

// ---- the tag, captured from the real crates: a Safe whose tag() is ScalarPermutation::tag's body and whose permute() counts
// (the reference's own `Test` pattern, src/hades.rs:63-92, with the tag kept instead of zeroed) ----
#[derive(Default, Clone)]
struct ProbeLog {
    tag_input: Vec<u8>,
    tag: Option<BlsScalar>,
    permutations: usize,
}

#[derive(Clone)]
struct TagProbe(Rc<RefCell<ProbeLog>>);

impl Safe<BlsScalar, 5> for TagProbe {
    fn permute(&mut self, _state: &mut [BlsScalar; 5]) {
        self.0.borrow_mut().permutations += 1;
    }
    fn tag(&mut self, input: &[u8]) -> BlsScalar {
        let t = BlsScalar::hash_to_scalar(input);
        let mut log = self.0.borrow_mut();
        log.tag_input = input.to_vec();
        log.tag = Some(t);
        t
    }
    fn add(&mut self, right: &BlsScalar, left: &BlsScalar) -> BlsScalar {
        right + left
    }
}

impl dusk_safe::Encryption<BlsScalar, 5> for TagProbe {
    fn subtract(&mut self, minuend: &BlsScalar, subtrahend: &BlsScalar) -> BlsScalar {
        minuend - subtrahend
    }
    fn is_equal(&mut self, lhs: &BlsScalar, rhs: &BlsScalar) -> bool {
        lhs == rhs
    }
}

fn hash_tag(domain: Domain, absorb_lens: &[usize], output_len: usize) -> ProbeLog {
    let mut iopattern: Vec<Call> = absorb_lens.iter().map(|l| Call::Absorb(*l)).collect();
    iopattern.push(Call::Squeeze(output_len));
    let log = Rc::new(RefCell::new(ProbeLog::default()));
    let _sponge = Sponge::start(TagProbe(log.clone()), iopattern, u64::from(domain)).expect("a valid io-pattern");
    let out = log.borrow().clone();
    out
}

fn encryption_tag(message_len: usize) -> ProbeLog {
    let log = Rc::new(RefCell::new(ProbeLog::default()));
    let message = vec![BlsScalar::zero(); message_len];
    let _ = dusk_safe::encrypt(TagProbe(log.clone()), Domain::Encryption, &message, &[BlsScalar::zero(), BlsScalar::zero()], &BlsScalar::zero())
        .expect("dusk_safe::encrypt on the probe");
    let out = log.borrow().clone();
    out
}

// ---- inputs ----
fn seq(n: usize) -> Vec<BlsScalar> {
    (0..n).map(|j| BlsScalar::from(j as u64 + 1)).collect()
}

fn inv7pow(n: usize) -> Vec<BlsScalar> {
    let g = BlsScalar::from(7u64).invert().unwrap();
    let mut x = BlsScalar::one();
    (0..n)
        .map(|_| {
            x *= g;
            x
        })
        .collect()
}

fn kat_inputs() -> Vec<BlsScalar> {
    // the 64-hex-digit strings of "inputs_le_hex": little-endian canonical bytes (src/hades.rs:94-131 parses them with from_hex_str)
    let start = KAT_JSON.find("\"inputs_le_hex\"").expect("inputs_le_hex in hades_kat.json");
    let end = start + KAT_JSON[start..].find(']').expect("end of inputs_le_hex");
    let mut out = Vec::new();
    for piece in KAT_JSON[start..end].split('"') {
        if piece.len() == 64 && piece.bytes().all(|b| b.is_ascii_hexdigit()) {
            let mut bytes = [0u8; 32];
            for k in 0..32 {
                bytes[k] = u8::from_str_radix(&piece[2 * k..2 * k + 2], 16).unwrap();
            }
            out.push(BlsScalar::from_bytes(&bytes).ok().expect("a canonical scalar"));
        }
    }
    assert_eq!(out.len(), 10, "the ten KAT inputs");
    out
}

// ---- JSON by hand (no serde: the dependency list stays the dusk crates) ----
fn limbs(x: &BlsScalar) -> String {
    format!("[{}, {}, {}, {}]", x.0[0], x.0[1], x.0[2], x.0[3])
}

fn limbs_list(xs: &[BlsScalar]) -> String {
    format!("[{}]", xs.iter().map(limbs).collect::<Vec<_>>().join(", "))
}

fn bytes_list(b: &[u8]) -> String {
    format!("[{}]", b.iter().map(|v| v.to_string()).collect::<Vec<_>>().join(", "))
}

fn usize_list(b: &[usize]) -> String {
    format!("[{}]", b.iter().map(|v| v.to_string()).collect::<Vec<_>>().join(", "))
}

fn hex(b: &[u8]) -> String {
    let mut s = String::new();
    for v in b {
        write!(s, "{:02x}", v).unwrap();
    }
    s
}

fn domain_name(d: Domain) -> &'static str {
    match d {
        Domain::Merkle4 => "merkle4",
        Domain::Merkle2 => "merkle2",
        Domain::Encryption => "encryption",
        Domain::Other => "other",
    }
}

fn shape_name(d: Domain, lens: &[usize], out: usize) -> String {
    format!("{}_{}_{}", domain_name(d), lens.iter().map(|l| l.to_string()).collect::<Vec<_>>().join("+"), out)
}

fn main() {
    let default_out = concat!(env!("CARGO_MANIFEST_DIR"), "/../../../tests/golden/reference_fixtures.json");
    let out_path = std::env::args().nth(1).unwrap_or_else(|| default_out.to_string());

    // the io-patterns of src/hash.rs:62-85 the repository's tests use (tests/hash.rs:101-116, 188-203, 277-292; BASELINE configs[3])
    let shapes: Vec<(Domain, Vec<usize>, usize)> = vec![
        (Domain::Merkle4, vec![4], 1), (Domain::Merkle2, vec![2], 1), (Domain::Other, vec![3], 1), (Domain::Other, vec![5], 1),
        (Domain::Other, vec![15], 1), (Domain::Other, vec![42], 1), (Domain::Other, vec![3], 3), (Domain::Other, vec![5], 2),
        (Domain::Other, vec![4], 7), (Domain::Other, vec![42], 5), (Domain::Other, vec![3, 39], 1),
    ];
    let kat = kat_inputs();
    let mut tags = Vec::new();
    let mut digests = Vec::new();
    let mut truncated = Vec::new();
    for (domain, lens, out_len) in shapes.iter() {
        let total: usize = lens.iter().sum();
        let name = shape_name(*domain, lens, *out_len);
        let log = hash_tag(*domain, lens, *out_len);
        tags.push(format!(
            "{{\"name\": \"{}\", \"domain\": \"{}\", \"absorb_lens\": {}, \"output_len\": {}, \"tag_input\": {}, \"tag_limbs\": {}}}",
            name, domain_name(*domain), usize_list(lens), out_len, bytes_list(&log.tag_input), limbs(&log.tag.expect("tag"))
        ));
        let mut families: Vec<(&str, Vec<BlsScalar>)> = vec![("seq", seq(total)), ("inv7pow", inv7pow(total))];
        if total <= kat.len() {
            families.push(("kat", kat[..total].to_vec()));
        }
        for (family, input) in families.iter() {
            let mut h = Hash::new(*domain);
            h.output_len(*out_len);
            let mut off = 0;
            for l in lens.iter() {
                h.update(&input[off..off + l]);
                off += l;
            }
            let d = h.finalize();
            assert_eq!(d.len(), *out_len);
            digests.push(format!(
                "{{\"name\": \"{}\", \"domain\": \"{}\", \"absorb_lens\": {}, \"output_len\": {}, \"input\": \"{}\", \"input_limbs\": {}, \"output_limbs\": {}}}",
                name, domain_name(*domain), usize_list(lens), out_len, family, limbs_list(input), limbs_list(&d)
            ));
            // finalize_truncated: (bls & MASK).reduce().0 handed to JubJubScalar::from_raw — written as the JubJubScalar's canonical
            // little-endian bytes, which are the raw limbs from_raw received (the value is below 2^250 < r)
            let t: Vec<JubJubScalar> = h.finalize_truncated();
            let t_hex: Vec<String> = t.iter().map(|s| format!("\"{}\"", hex(&s.to_bytes()))).collect();
            truncated.push(format!(
                "{{\"name\": \"{}\", \"domain\": \"{}\", \"absorb_lens\": {}, \"output_len\": {}, \"input\": \"{}\", \"output_le_hex\": [{}]}}",
                name, domain_name(*domain), usize_list(lens), out_len, family, t_hex.join(", ")
            ));
        }
    }

    // encryption: a fixed secret point (the reference's own pattern, tests/encryption.rs:16-28, with a fixed scalar), a fixed nonce
    let secret: JubJubAffine = (GENERATOR_EXTENDED * &JubJubScalar::from(12345u64)).into();
    let nonce = BlsScalar::from(0x6e6f6e6365u64);
    let mut encryption = Vec::new();
    for len in [2usize, 21, 42] {
        let message = inv7pow(len);
        let cipher = encrypt(&message, &secret, &nonce).expect("encrypt");
        assert_eq!(cipher.len(), len + 1);
        assert_eq!(decrypt(&cipher, &secret, &nonce).expect("decrypt"), message);
        let wrong_nonce_fails = decrypt(&cipher, &secret, &(nonce + BlsScalar::one())).is_err();
        let log = encryption_tag(len);
        encryption.push(format!(
            "{{\"len\": {}, \"secret_limbs\": {}, \"nonce_limbs\": {}, \"message\": \"inv7pow\", \"message_limbs\": {}, \"cipher_limbs\": {}, \"tag_input\": {}, \"tag_limbs\": {}, \"permutations\": {}, \"wrong_nonce_fails\": {}}}",
            len, limbs_list(&[secret.get_u(), secret.get_v()]), limbs(&nonce), limbs_list(&message), limbs_list(&cipher),
            bytes_list(&log.tag_input), limbs(&log.tag.expect("tag")), log.permutations, wrong_nonce_fails
        ));
    }

    let mut s = String::new();
    s.push_str("{\n");
    s.push_str(" \"_about\": \"REFERENCE outputs, written by bindings/rust/refgen (cargo run --release) from dusk-poseidon 0.42.0-rc.0, dusk-safe 0.3, dusk-bls12_381 0.14, dusk-jubjub 0.15; consumed by tests/test_reference_fixtures.py\",\n");
    s.push_str(" \"schema\": 1,\n");
    s.push_str(" \"source\": \"reference\",\n");
    s.push_str(" \"scalar_encoding\": \"4 little-endian u64 limbs of the Montgomery form (BlsScalar.0)\",\n");
    for (key, rows) in [("tags", &tags), ("digests", &digests), ("truncated", &truncated), ("encryption", &encryption)] {
        write!(s, " \"{}\": [\n  {}\n ]{}\n", key, rows.join(",\n  "), if key == "encryption" { "" } else { "," }).unwrap();
    }
    s.push_str("}\n");
    std::fs::write(&out_path, s).expect("writing the fixture file");
    println!("wrote {} ({} tags, {} digests, {} truncated, {} encryption cases)", out_path, tags.len(), digests.len(), truncated.len(), encryption.len());
}
