//! Run on a machine with an MI355X AND the dusk crates: this is the test that turns the tag from
//! "parity unpinned" into pinned.  Shapes follow the reference's
//! tests/hash.rs (3/5/15 inputs; (3,3), (5,2), (4,7) outputs) plus the Merkle domains and config 4.
use dusk_bls12_381::BlsScalar;
use dusk_poseidon::{Domain, Hash};
use dusk_poseidon_hip::HashBatch;
use ff::Field;
use rand::rngs::StdRng;
use rand::SeedableRng;

#[test]
fn gpu_matches_reference() {
    let mut rng = StdRng::seed_from_u64(0xbeef);
    for (domain, n_in, n_out) in [
        (Domain::Merkle4, 4, 1), (Domain::Merkle2, 2, 1), (Domain::Other, 3, 1), (Domain::Other, 5, 1),
        (Domain::Other, 15, 1), (Domain::Other, 3, 3), (Domain::Other, 5, 2), (Domain::Other, 4, 7), (Domain::Other, 42, 5),
    ] {
        let hb = HashBatch::with_output_len(domain, n_in, n_out).unwrap();
        let input: Vec<BlsScalar> = (0..n_in * 1000).map(|_| BlsScalar::random(&mut rng)).collect();
        let got = hb.digest(&input);
        for i in 0..1000 {
            let mut h = Hash::new(domain);
            h.output_len(n_out);
            h.update(&input[i * n_in..(i + 1) * n_in]);
            assert_eq!(h.finalize(), &got[i * n_out..(i + 1) * n_out]);
        }
    }
}
