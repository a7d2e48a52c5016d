//! bx-upstream-vectors: dumps known-answer vectors from risc0-zkp 3.0.3 / risc0-core 3.0.0 in the formats
//! tests/golden/upstream/README.md specifies, one vector per **choice** row of oracle/README.md.
//!
//!     cargo run --release -- ../../tests/golden/upstream
//!
//! NEVER COMPILED where it was written (no Rust toolchain, no network).  The paths below are risc0-zkp's public items as the
//! author recalls them; if rustc disagrees, fix the `use` lines — the JSON this program must write is fixed by the README and by
//! tools/upstream_vectors/twin.py, which writes the same files from this repository's oracle (the loader is tested against those).
//!
//! All integers are CANONICAL field elements (`as_u32()`), never Montgomery words.
use std::{fs, path::PathBuf};

use rand::{rngs::StdRng, Rng as _, SeedableRng};
use risc0_core::field::{
    baby_bear::{BabyBear, BabyBearElem as Elem, BabyBearExtElem as ExtElem},
    Elem as _, ExtElem as _,
};
use risc0_zkp::{
    core::{
        digest::Digest,
        hash::{
            poseidon2::{
                consts::{M_INT_DIAG_HZN, ROUND_CONSTANTS},
                poseidon2_mix, Poseidon2HashSuite, CELLS,
            },
            HashSuite,
        },
        ntt::{evaluate_ntt, expand, interpolate_ntt},
    },
    hal::{cpu::CpuHal, Buffer, Hal},
    prove::{merkle::MerkleTreeProver, write_iop::WriteIOP},
};
use serde_json::{json, Value};
use sha2::{Digest as _, Sha256};

fn c(e: &Elem) -> u32 { e.as_u32() }
fn cs(v: &[Elem]) -> Vec<u32> { v.iter().map(c).collect() }
fn ext(e: &ExtElem) -> Vec<u32> { e.subelems().iter().map(c).collect() }
fn dg(d: &Digest) -> Vec<u32> {
    // a Poseidon2 digest holds 8 field elements as MONTGOMERY words (Digest::from(elems.map(as_u32_montgomery))): decode them
    d.as_words().iter().map(|w| Elem::new_raw(*w).as_u32()).collect()
}
fn rnd_elems(rng: &mut StdRng, n: usize) -> Vec<Elem> { (0..n).map(|_| Elem::new(rng.gen_range(0..Elem::P))).collect() }
fn rnd_ext(rng: &mut StdRng) -> ExtElem { ExtElem::from_subelems(rnd_elems(rng, 4)) }

fn main() {
    let out_dir = PathBuf::from(std::env::args().nth(1).expect("usage: bx-upstream-vectors <tests/golden/upstream>"));
    fs::create_dir_all(&out_dir).unwrap();
    let mut rng = StdRng::seed_from_u64(0xB0D1E55);
    let suite: HashSuite<BabyBear> = Poseidon2HashSuite::new_suite();
    let hal = CpuHal::<BabyBear>::new(suite.clone());
    let mut files: Vec<(&str, Value)> = vec![];

    // ---- poseidon2_consts.json: 213 round constants (4x24 external, 21 internal, 4x24 external), 24 internal diagonal entries
    files.push(("poseidon2_consts.json", json!({"round_constants": cs(&ROUND_CONSTANTS[..]), "internal_diag": cs(&M_INT_DIAG_HZN[..])})));

    // ---- poseidon2_vectors.json: permutation, sponge (full blocks, a 5-element PADDED block, 16, 17, 40, empty), pair hash, Poseidon2Rng
    let mut perms = vec![];
    for k in 0..3 {
        let mut cells: [Elem; CELLS] = core::array::from_fn(|i| if k == 0 { Elem::new(i as u32) } else { Elem::new(rng.gen_range(0..Elem::P)) });
        let inp = cs(&cells);
        poseidon2_mix(&mut cells);
        perms.push(json!({"in": inp, "out": cs(&cells)}));
    }
    let mut slices = vec![];
    for n in [0usize, 1, 5, 15, 16, 17, 32, 40] {
        let v = rnd_elems(&mut rng, n);
        slices.push(json!({"in": cs(&v), "digest": dg(&suite.hashfn.hash_elem_slice(&v))}));
    }
    let mut pairs = vec![];
    for _ in 0..3 {
        let (a, b) = (*suite.hashfn.hash_elem_slice(&rnd_elems(&mut rng, 8)), *suite.hashfn.hash_elem_slice(&rnd_elems(&mut rng, 9)));
        pairs.push(json!({"a": dg(&a), "b": dg(&b), "out": dg(&suite.hashfn.hash_pair(&a, &b))}));
    }
    // Poseidon2Rng: mix(digest) / random_elem / random_ext_elem / random_bits, interleaved the way the prover uses them
    let mut r = suite.rng.new_rng();
    let mut script = vec![];
    for round in 0..4 {
        let d = *suite.hashfn.hash_elem_slice(&rnd_elems(&mut rng, 8 + round));
        r.mix(&d);
        script.push(json!({"op": "mix", "digest": dg(&d)}));
        for _ in 0..(3 + 5 * round) {
            script.push(json!({"op": "random_elem", "value": c(&r.random_elem())}));
        }
        script.push(json!({"op": "random_ext_elem", "value": ext(&r.random_ext_elem())}));
        for bits in [1usize, 7, 12, 22, 26, 31] {
            script.push(json!({"op": "random_bits", "bits": bits, "value": r.random_bits(bits)}));
        }
    }
    files.push(("poseidon2_vectors.json", json!({"permutation": perms, "hash_elem_slice": slices, "hash_pair": pairs, "rng": script})));

    // ---- ntt_vectors.json: interpolate_ntt (natural evaluations -> bit-reversed coefficients), expand + evaluate_ntt (-> 4n evaluations)
    let mut ntt = vec![];
    for log_n in [1usize, 2, 3, 5, 8, 12] {
        let n = 1 << log_n;
        let evals = rnd_elems(&mut rng, n);
        let mut io = evals.clone();
        interpolate_ntt::<Elem, Elem>(&mut io);
        let mut big = vec![Elem::ZERO; 4 * n];
        expand(&mut big, &io, 2);
        evaluate_ntt::<Elem, Elem>(&mut big, 2);
        ntt.push(json!({"size": n, "evals_natural": cs(&evals), "interpolate_out": cs(&io), "expand_bits": 2, "evaluate_out": cs(&big)}));
    }
    files.push(("ntt_vectors.json", json!({"cases": ntt})));

    // ---- zk_shift_vectors.json (Hal::zk_shift on one polynomial in interpolate_ntt's output order)
    let mut zk = vec![];
    for log_n in [1usize, 4, 9] {
        let v = rnd_elems(&mut rng, 1 << log_n);
        let buf = hal.copy_from_elem("zk", &v);
        hal.zk_shift(&buf, 1);
        let mut out = vec![];
        buf.view(|s| out = cs(s));
        zk.push(json!({"size": v.len(), "in": cs(&v), "out": out}));
    }
    files.push(("zk_shift_vectors.json", json!({"cases": zk})));

    // ---- fri_fold_vectors.json: in = 4 planes of 16*count base elements (SoA ext), out = 4 planes of count
    let mut fri = vec![];
    for count in [1usize, 4, 64] {
        let inp = rnd_elems(&mut rng, 64 * count);
        let mix = rnd_ext(&mut rng);
        let (bi, bo) = (hal.copy_from_elem("in", &inp), hal.alloc_elem("out", 4 * count));
        hal.fri_fold(&bo, &bi, &mix);
        let mut out = vec![];
        bo.view(|s| out = cs(s));
        fri.push(json!({"count": count, "in_soa": cs(&inp), "mix": ext(&mix), "out_soa": out}));
    }
    files.push(("fri_fold_vectors.json", json!({"cases": fri})));

    // ---- mix_poly_coeffs_vectors.json: input_size polynomials of `count` coefficients mixed into the ext accumulators `combos` names
    let mut mixes = vec![];
    for (input_size, count, n_combos) in [(3usize, 8usize, 2usize), (7, 16, 3)] {
        let inp = rnd_elems(&mut rng, count * input_size);
        let combos: Vec<u32> = (0..input_size).map(|i| (i % n_combos) as u32).collect();
        let init: Vec<ExtElem> = (0..n_combos * count).map(|_| rnd_ext(&mut rng)).collect();
        let (mix_start, mix) = (rnd_ext(&mut rng), rnd_ext(&mut rng));
        let out = hal.copy_from_extelem("out", &init);
        hal.mix_poly_coeffs(&out, &mix_start, &mix, &hal.copy_from_elem("in", &inp), &hal.copy_from_u32("combos", &combos), input_size, count);
        let mut res: Vec<u32> = vec![];
        out.view(|s| res = s.iter().flat_map(ext).collect());
        mixes.push(json!({"count": count, "input_size": input_size, "combos": combos, "mix_start": ext(&mix_start), "mix": ext(&mix),
                          "in": cs(&inp), "init_ext_aos": init.iter().flat_map(ext).collect::<Vec<u32>>(), "out_ext_aos": res}));
    }
    files.push(("mix_poly_coeffs_vectors.json", json!({"cases": mixes})));

    // ---- merkle_vectors.json: MerkleTreeProver over a rows x cols matrix with 50 queries: the words `commit` writes (the top layer),
    //      the root it mixes into the transcript, and the words `prove(idx)` writes for two openings
    let mut merkle = vec![];
    for (rows, cols) in [(64usize, 3usize), (256, 16), (1024, 20)] {
        let m = rnd_elems(&mut rng, rows * cols); // column-major
        let tree = MerkleTreeProver::new(&hal, &hal.copy_from_elem("m", &m), rows, cols, 50);
        let mut iop = WriteIOP::new(suite.rng.as_ref());
        tree.commit(&mut iop);
        let top: Vec<u32> = iop.proof.clone();
        let mut opens = vec![];
        for idx in [0usize, rows - 1, rows / 3] {
            let before = iop.proof.len();
            let _ = tree.prove(&mut iop, idx);
            opens.push(json!({"idx": idx, "words_montgomery": iop.proof[before..].to_vec()}));
        }
        merkle.push(json!({"rows": rows, "cols": cols, "queries": 50, "matrix": cs(&m), "root": dg(tree.root()),
                           "commit_words_montgomery": top, "openings": opens}));
    }
    files.push(("merkle_vectors.json", json!({"cases": merkle})));

    // ---- write the files and a MANIFEST with their SHA-256 (merge it into tests/golden/MANIFEST.json under "upstream")
    let mut manifest = serde_json::Map::new();
    for (name, value) in &files {
        let text = serde_json::to_string(value).unwrap();
        fs::write(out_dir.join(name), &text).unwrap();
        manifest.insert(name.to_string(), json!(hex::encode(Sha256::digest(text.as_bytes()))));
    }
    fs::write(out_dir.join("MANIFEST.upstream.json"),
              serde_json::to_string_pretty(&json!({"generator": "tools/upstream_vectors", "risc0-zkp": "3.0.3", "risc0-core": "3.0.0", "sha256": manifest})).unwrap()).unwrap();
    println!("wrote {} files to {}", files.len() + 1, out_dir.display());
}
