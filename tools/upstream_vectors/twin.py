"""The Python twin of tools/upstream_vectors/src/main.rs: writes the SAME files, in the same JSON formats, from this repository's
CPU oracle instead of risc0-zkp.

    python tools/upstream_vectors/twin.py <out_dir>

Two uses: (1) it is the executable specification of the formats (tests/test_upstream_vectors.py is run against its output on every
host, so the loader is known to work before anyone has a Rust toolchain); (2) `diff -r` of its output directory against the Rust
program's — both seeded differently, so compare by running the LOADER on the Rust files, not the bytes — is the one-command parity
flip: every convention oracle/README.md marks "choice"/"recalled" has a vector here.
Test infrastructure: imports oracle/.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_lib as ol  # noqa: E402

P = ol.P


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    L = ol.lib()
    rng = np.random.default_rng(0xB0D1E55)
    canon = lambda n: rng.integers(0, P, n, dtype=np.uint32)  # noqa: E731
    files = {}

    rc, dgn = np.zeros(213, np.uint32), np.zeros(24, np.uint32)
    L.bxo_poseidon2_get_params(rc, dgn)
    files["poseidon2_consts.json"] = {"round_constants": rc.tolist(), "internal_diag": dgn.tolist()}

    def digest_of(canon_words):
        d = np.zeros(8, np.uint32)
        x = ol.encode(canon_words) if len(canon_words) else np.zeros(1, np.uint32)
        L.bxo_hash_elem_slice(d, x, len(canon_words), 1)
        return d  # Montgomery words

    perms = []
    for k in range(3):
        cells = np.arange(24, dtype=np.uint32) if k == 0 else canon(24)
        m = ol.encode(cells)
        L.bxo_poseidon2_mix(m)
        perms.append({"in": cells.tolist(), "out": ol.decode(m).tolist()})
    slices = []
    for n in (0, 1, 5, 15, 16, 17, 32, 40):
        v = canon(n)
        slices.append({"in": v.tolist(), "digest": ol.decode(digest_of(v)).tolist()})
    pairs = []
    for _ in range(3):
        a, b = digest_of(canon(8)), digest_of(canon(9))
        out = np.zeros(8, np.uint32)
        L.bxo_hash_pair(out, a, b)
        pairs.append({"a": ol.decode(a).tolist(), "b": ol.decode(b).tolist(), "out": ol.decode(out).tolist()})
    state = np.zeros(25, np.uint32)
    script = []
    for rnd in range(4):
        d = digest_of(canon(8 + rnd))
        state, _ = ol.transcript_step(state, d, 0)
        script.append({"op": "mix", "digest": ol.decode(d).tolist()})
        for _ in range(3 + 5 * rnd):
            state, e = ol.transcript_step(state, np.zeros(0, np.uint32), 1)
            script.append({"op": "random_elem", "value": int(ol.decode(e)[0])})
        state, e = ol.transcript_step(state, np.zeros(0, np.uint32), 4)
        script.append({"op": "random_ext_elem", "value": ol.decode(e).tolist()})
        for bits in (1, 7, 12, 22, 26, 31):
            script.append({"op": "random_bits", "bits": bits, "value": int(L.bxo_rng_random_bits(state, bits))})
    files["poseidon2_vectors.json"] = {"permutation": perms, "hash_elem_slice": slices, "hash_pair": pairs, "rng": script}

    ntt = []
    for log_n in (1, 2, 3, 5, 8, 12):
        n = 1 << log_n
        ev = canon(n)
        io = ol.encode(ev)
        L.bxo_batch_interpolate_ntt(io, 1, n)
        big = np.zeros(4 * n, np.uint32)
        L.bxo_batch_expand_into_evaluate_ntt(big, io, 1, n, 2)
        ntt.append({"size": n, "evals_natural": ev.tolist(), "interpolate_out": ol.decode(io).tolist(), "expand_bits": 2, "evaluate_out": ol.decode(big).tolist()})
    files["ntt_vectors.json"] = {"cases": ntt}

    zk = []
    for log_n in (1, 4, 9):
        v = canon(1 << log_n)
        io = ol.encode(v)
        L.bxo_zk_shift(io, 1, v.size)
        zk.append({"size": int(v.size), "in": v.tolist(), "out": ol.decode(io).tolist()})
    files["zk_shift_vectors.json"] = {"cases": zk}

    fri = []
    for count in (1, 4, 64):
        inp, mix = canon(64 * count), canon(4)
        out = np.zeros(4 * count, np.uint32)
        L.bxo_fri_fold(out, ol.encode(inp), ol.encode(mix), count)
        fri.append({"count": count, "in_soa": inp.tolist(), "mix": mix.tolist(), "out_soa": ol.decode(out).tolist()})
    files["fri_fold_vectors.json"] = {"cases": fri}

    mixes = []
    # Hal::mix_poly_coeffs(output, mix_start, mix, input, combos, input_size, count): input_size polynomials of `count` coefficients
    for input_size, count, n_combos in ((3, 8, 2), (7, 16, 3)):
        inp = canon(count * input_size)
        combos = np.array([i % n_combos for i in range(input_size)], np.uint32)
        init = canon(4 * n_combos * count)
        ms, mx = canon(4), canon(4)
        out = ol.encode(init)
        L.bxo_mix_poly_coeffs(out, ol.encode(ms), ol.encode(mx), ol.encode(inp), combos, input_size, count)
        mixes.append({"count": count, "input_size": input_size, "combos": combos.tolist(), "mix_start": ms.tolist(), "mix": mx.tolist(),
                      "in": inp.tolist(), "init_ext_aos": init.tolist(), "out_ext_aos": ol.decode(out).tolist()})
    files["mix_poly_coeffs_vectors.json"] = {"cases": mixes}

    merkle = []
    for rows, cols in ((64, 3), (256, 16), (1024, 20)):
        m = canon(rows * cols)
        nodes, top = merkle_nodes(L, ol.encode(m), rows, cols, 50)
        opens = []
        for idx in (0, rows - 1, rows // 3):
            opens.append({"idx": idx, "words_montgomery": merkle_open(nodes, ol.encode(m), rows, cols, top, idx)})
        merkle.append({"rows": rows, "cols": cols, "queries": 50, "matrix": m.tolist(), "root": ol.decode(nodes[8:16]).tolist(),
                       "commit_words_montgomery": nodes[8 * top:16 * top].tolist(), "openings": opens})
    files["merkle_vectors.json"] = {"cases": merkle}

    manifest = {}
    for name, value in files.items():
        text = json.dumps(value, separators=(",", ":"))
        open(os.path.join(out_dir, name), "w").write(text)
        manifest[name] = hashlib.sha256(text.encode()).hexdigest()
    json.dump({"generator": "tools/upstream_vectors/twin.py (this repository's oracle, NOT risc0)", "sha256": manifest},
              open(os.path.join(out_dir, "MANIFEST.upstream.json"), "w"), indent=1)
    return sorted(files)


def merkle_nodes(L, matrix_mont, rows, cols, queries):
    """[EXT] MerkleTreeParams::new / MerkleTreeProver::new: nodes[2*rows] digests (Montgomery words), leaf r at rows + r; the top layer
    is the largest power of two <= queries below the leaf layer."""
    nodes = np.zeros(16 * rows, np.uint32)
    leaves = np.zeros(8 * rows, np.uint32)
    L.bxo_hash_rows(leaves, matrix_mont, rows, cols)
    nodes[8 * rows:] = leaves
    size = rows
    while size > 1:
        L.bxo_hash_fold(nodes, size, size // 2)
        size //= 2
    layers, top_layer = rows.bit_length() - 1, 0
    for i in range(1, layers):
        if (1 << i) > queries:
            break
        top_layer = i
    return nodes, 1 << top_layer


def merkle_open(nodes, matrix_mont, rows, cols, top, idx):
    """[EXT] MerkleTreeProver::prove: the row's column values, then the sibling digests from the leaf layer up to (excluding) the top layer"""
    words = [int(matrix_mont[c * rows + idx]) for c in range(cols)]
    node = idx + rows
    while node >= 2 * top:
        words += nodes[8 * (node ^ 1):8 * (node ^ 1) + 8].tolist()
        node >>= 1
    return words


if __name__ == "__main__":
    print("wrote", main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/bx_upstream_twin"))
