"""Generates the golden fixtures in this directory from oracle/np_oracle.py (definition-level big-int math).

Run from the repo root:  python tests/golden/make_golden.py
No reference code is imported (the reference has no Python and vendors none of the prover arithmetic,
SURVEY.md §0/§8c).  The only externally-published vectors are
  * the BabyBear constants (P, Montgomery constants, roots of unity) — re-derived here and compared to the
    table recorded in SURVEY.md Appendix A.1;
  * the Poseidon2 BabyBear t=24 known-answer test (permutation of 0..23) published with the Poseidon2
    reference instance and used by upstream's own `poseidon2_test_vectors` unit test [EXT].
Everything else is self-generated from seeded inputs.
All vectors are stored as CANONICAL integers (not Montgomery).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import np_oracle as npo  # noqa: E402

P = npo.P


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    # ---- field constants (verbatim) ----
    survey_fwd = [1, 2013265920, 284861408, 1801542727, 567209306, 740045640, 918899846, 1881002012, 1453957774,
                  65325759, 1538055801, 515192888, 483885487, 157393079, 1695124103, 2005211659, 1540072241, 88064245,
                  1542985445, 1269900459, 1461624142, 825701067, 682402162, 1311873874, 1164520853, 352275361, 18769, 137]
    survey_rev = [1, 2013265920, 1728404513, 1592366214, 196396260, 1253260071, 72041623, 1091445674, 145223211,
                  1446820157, 1030796471, 2010749425, 1827366325, 1239938613, 246299276, 596347512, 1893145354,
                  246074437, 1525739923, 1194341128, 1463599021, 704606912, 95395244, 15672543, 647517488, 584175179,
                  137728885, 749463956]
    fwd = [npo.rou(k) for k in range(28)]
    rev = [pow(x, -1, P) for x in fwd]
    assert fwd == survey_fwd and rev == survey_rev
    assert pow(P, -1, 2**32) == 0x88000001 and (2**64) % P == 1172168163 and (2**32) % P == 268435454
    dump("babybear_consts.json", {
        "P": P, "M": 0x88000001, "R2": 1172168163, "R": 268435454, "generator": 31, "ext_beta": 11,
        "rou_fwd": fwd, "rou_rev": rev})

    # ---- Poseidon2 KAT (published vector; asserted against the derived-constant implementation) ----
    kat_out = [0x2ED3E23D, 0x12921FB0, 0x0E659E79, 0x61D81DC9, 0x32BAE33B, 0x62486AE3, 0x1E681B60, 0x24B91325,
               0x2A2EF5B9, 0x50E8593E, 0x5BC818EC, 0x10691997, 0x35A14520, 0x2BA6A3C5, 0x279D47EC, 0x55014E81,
               0x5953A67F, 0x2F403111, 0x6B8828FF, 0x1801301F, 0x2749207A, 0x3DC9CF21, 0x3C985BA2, 0x57A99864]
    got = npo.poseidon2_permute(list(range(24)))
    assert got == kat_out, "derived constants do not reproduce the published KAT"
    rng = random.Random(0xB0D1E55)
    extra = []
    for _ in range(4):
        st = [rng.randrange(P) for _ in range(24)]
        extra.append({"in": st, "out": npo.poseidon2_permute(st)})
    dump("poseidon2_kat.json", {
        "source": "Poseidon2 reference instance, BabyBear t=24 (external rounds 8, internal 21, x^7); permutation of 0..23",
        "kat_in": list(range(24)), "kat_out": kat_out,
        "round_constants_first8": npo.RC[:8], "round_constants_count": len(npo.RC),
        "round_constants": npo.RC, "internal_diag": npo.DIAG_HZN,
        "seeded": extra})

    # ---- sponge / pair / merkle ----
    sponge = []
    for n in (0, 1, 15, 16, 17, 32, 40):
        el = [rng.randrange(P) for _ in range(n)]
        sponge.append({"in": el, "digest": npo.hash_elems(el)})
    a = [rng.randrange(P) for _ in range(8)]
    b = [rng.randrange(P) for _ in range(8)]
    rows, cols = 16, 20
    mat = [[rng.randrange(P) for _ in range(rows)] for _ in range(cols)]  # column-major: mat[c][r]
    leaves = [npo.hash_elems([mat[c][r] for c in range(cols)]) for r in range(rows)]
    dump("poseidon2_sponge.json", {
        "sponge": sponge, "pair": {"a": a, "b": b, "out": npo.hash_pair(a, b)},
        "merkle": {"rows": rows, "cols": cols, "matrix_colmajor": mat, "leaves": leaves, "root": npo.merkle_root(leaves)}})

    # ---- NTT family ----
    ntt = []
    for bits in (1, 2, 4, 6, 8):
        n = 1 << bits
        ev = [rng.randrange(P) for _ in range(n)]
        co = npo.interpolate(ev)
        sh = npo.zk_shift(co)
        lde = npo.expand_evaluate(sh, 2)
        plain = npo.expand_evaluate(co, 0)
        assert plain == ev
        ntt.append({"bits": bits, "evals": ev, "coeffs_bitrev": co, "shifted_bitrev": sh, "lde4": lde})
    dump("ntt_vectors.json", {"cases": ntt})

    # ---- FRI fold / fp4 ----
    m = 16
    f = [[rng.randrange(P) for _ in range(4)] for _ in range(16 * m)]
    mix = [rng.randrange(P) for _ in range(4)]
    x = [rng.randrange(P) for _ in range(4)]
    y = [rng.randrange(P) for _ in range(4)]
    dump("fri_vectors.json", {
        "fp4": {"x": x, "y": y, "xy": npo.f4_mul(x, y), "x_inv": npo.f4_inv(x)},
        "fold": {"coeffs_natural": f, "mix": mix, "out_natural": npo.fri_fold(f, mix)}})
    control_ids()
    print("golden fixtures written to", HERE)


def _splitmix64(x):
    m = (1 << 64) - 1
    z = (x + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def control_ids():
    """Control IDs of the synthetic circuit (include/bx_prover.h, "code"; include/bx_circuit.h) from the DEFINITION: the code columns
    as Montgomery words, the interpolating polynomial of each (O(n^2) sums), its values on the coset 3<w_4N> (Horner per point), the
    row sponge and the tree — big-int Python only (oracle/np_oracle.py).  A fifth implementation next to the HIP prover, the C oracle,
    the library's host path and the generated table; small shapes only (a case costs minutes here)."""
    code_seed = 0x434F4E54524F4C21  # "CONTROL!"
    r_mont = (1 << 32) % P

    def word(col, row):  # the Montgomery WORD of a code cell; the field element it stands for is word * R^-1
        v = _splitmix64(code_seed ^ ((col << 32) | row)) >> 33
        return v - P if v >= P else v

    cases = []
    for po2, wc in ((9, 1), (9, 3), (10, 2)):
        n = 1 << po2
        act = n - min(1994, n // 4)
        rows = [[] for _ in range(4 * n)]
        for c in range(wc):
            if c == 0:
                col_words = [r_mont if r == 0 else 0 for r in range(n)]
            elif c == 1:
                col_words = [r_mont if r == act - 1 else 0 for r in range(n)]
            else:
                col_words = [word(c, r) for r in range(n)]
            evals = [npo.from_mont(w) for w in col_words]          # canonical field elements
            ev4 = npo.expand_evaluate(npo.zk_shift(npo.interpolate(evals)), 2)
            for r in range(4 * n):
                rows[r].append(ev4[r])
        root = npo.merkle_root([npo.hash_elems(row) for row in rows])
        cases.append({"po2": po2, "w_code": wc, "control_id": root})  # canonical digest elements
        print("control id", po2, wc, root, flush=True)
    dump("control_ids.json", {"what": "Merkle root of the committed code group of the synthetic circuit, canonical elements", "cases": cases})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "control_ids":
        control_ids()
    else:
        main()
