"""Independent big-int restatement of the hot-path math, from the *definitions* (not the butterflies).

TEST INFRASTRUCTURE ONLY.  Used to (a) cross-check the C oracle (oracle/bx_oracle.c) and (b) generate the
small golden fixtures under tests/golden/ (tests/golden/make_golden.py).  Everything here works on canonical
integers mod P with Python ints; conversion to/from the Montgomery words that cross the HAL boundary happens
at the edges only.  The algorithms are stated mathematically so that an error in the butterfly code of the C
oracle or of the HIP kernels cannot be mirrored here:

  * interpolate  = inverse DFT by the O(n^2) definition, output index bit-reversed
  * evaluate/LDE = Horner evaluation of the polynomial at every point of <w_{n*2^e}> (the coset shift is zk_shift)
  * fri_fold     = the polynomial identity  g(y) = sum_k mix^k f_k(y),  f(x) = sum_k x^k f_k(x^16)
  * Poseidon2    = matrix form (explicit 24x24 M_E, M_I), constants from the Grain LFSR

Upstream items restated (not vendored in /root/reference; pinned by its Cargo.lock:9155,9012):
risc0-zkp 3.0.3 core/ntt.rs, core/hash/poseidon2/{mod,consts}.rs, hal/cpu.rs; risc0-core 3.0.0 field/baby_bear.rs.
"""
P = 15 * 2**27 + 1
R = 2**32
RINV = pow(R, -1, P)
BETA = 11  # Fp4 = Fp[X]/(X^4 + 11)


def to_mont(x):
    return (x * R) % P


def from_mont(x):
    return (x * RINV) % P


def rou(k):
    """primitive 2^k-th root of unity: 137^(2^(27-k))  (ROU_FWD[k])"""
    return pow(137, 2 ** (27 - k), P)


def bitrev(i, bits):
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def log2(n):
    k = n.bit_length() - 1
    assert 1 << k == n
    return k


# ---------------------------------------------------------------- polynomials over Fp
def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def interpolate(evals):
    """natural-order evaluations on <w_n>  ->  coefficients, stored at bit-reversed positions"""
    n = len(evals)
    bits = log2(n)
    w = rou(bits)
    winv = pow(w, -1, P)
    ninv = pow(n, -1, P)
    out = [0] * n
    for j in range(n):
        acc = 0
        for k in range(n):
            acc += evals[k] * pow(winv, (j * k) % n, P)
        out[bitrev(j, bits)] = (acc % P) * ninv % P
    return out


def expand_evaluate(coeffs_bitrev, expand_bits):
    """bit-reversed coefficients of a degree<n poly -> natural-order evaluations on <w_{n*2^e}>"""
    n = len(coeffs_bitrev)
    bits = log2(n)
    coeffs = [coeffs_bitrev[bitrev(j, bits)] for j in range(n)]
    m = n << expand_bits
    w = rou(bits + expand_bits)
    return [poly_eval(coeffs, pow(w, k, P)) for k in range(m)]


def zk_shift(coeffs_bitrev):
    n = len(coeffs_bitrev)
    bits = log2(n)
    return [(c * pow(3, bitrev(i, bits), P)) % P for i, c in enumerate(coeffs_bitrev)]


# ---------------------------------------------------------------- Fp4
def f4_mul(a, b):
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += a[i] * b[j]
    for k in range(6, 3, -1):
        r[k - 4] -= BETA * r[k]
    return [x % P for x in r[:4]]


def f4_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f4_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = f4_mul(r, a)
        a = f4_mul(a, a)
        e >>= 1
    return r


def f4_inv(a):
    return f4_pow(a, P**4 - 2)


def f4_poly_eval(coeffs_f4, x):
    acc = [0, 0, 0, 0]
    for c in reversed(coeffs_f4):
        acc = f4_add(f4_mul(acc, x), c)
    return acc


# ---------------------------------------------------------------- FRI fold (by the polynomial identity)
def fri_fold(coeffs_f4_natural, mix):
    """f (natural-order ext coefficients, len 16*m) -> g with g_j = sum_k mix^k f_{16 j + k}."""
    m = len(coeffs_f4_natural) // 16
    out = []
    for j in range(m):
        acc = [0, 0, 0, 0]
        cur = [1, 0, 0, 0]
        for k in range(16):
            acc = f4_add(acc, f4_mul(cur, coeffs_f4_natural[16 * j + k]))
            cur = f4_mul(cur, mix)
        out.append(acc)
    return out


# ---------------------------------------------------------------- Poseidon2 (matrix form)
CELLS, RATE, RF_HALF, RP = 24, 16, 4, 21
M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
DIAG_HZN = [
    0x409133F0, 0x1667A8A1, 0x06A6C7B6, 0x6F53160E, 0x273B11D1, 0x03176C5D, 0x72F9BBF9, 0x73CEBA91,
    0x5CDEF81D, 0x01393285, 0x46DAEE06, 0x065D7BA6, 0x52D72D6F, 0x05DD05E0, 0x3BAB4B63, 0x6ADA3842,
    0x2FC5FBEC, 0x770D61B0, 0x5715AAE9, 0x03EF0E90, 0x75B6C770, 0x242ADF5F, 0x00D0CA4C, 0x36C0E388,
]


def grain_round_constants(field=1, sbox=0, n=31, t=24, rf=8, rp=21, p=P):
    """Poseidon reference constant generator (Grain LFSR in self-shrinking mode); Poseidon2 count t*R_F + R_P."""
    def tobits(v, w):
        return [int(c) for c in bin(v)[2:].zfill(w)]

    bits = tobits(field, 2) + tobits(sbox, 4) + tobits(n, 12) + tobits(t, 12) + tobits(rf, 10) + tobits(rp, 10) + [1] * 30

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb

    for _ in range(160):
        step()

    def nextbit():
        while True:
            if step():
                return step()
            step()

    out = []
    while len(out) < t * rf + rp:
        v = 0
        for _ in range(n):
            v = (v << 1) | nextbit()
        if v < p:
            out.append(v)
    return out


RC = grain_round_constants()
_ME = [[(2 if i // 4 == j // 4 else 1) * M4[i % 4][j % 4] for j in range(CELLS)] for i in range(CELLS)]
_MI = [[1 + (DIAG_HZN[i] if i == j else 0) for j in range(CELLS)] for i in range(CELLS)]


def _matvec(m, s):
    return [sum(m[i][j] * s[j] for j in range(CELLS)) % P for i in range(CELLS)]


def poseidon2_permute(state):
    s = _matvec(_ME, [x % P for x in state])
    k = 0
    for _ in range(RF_HALF):
        s = _matvec(_ME, [pow((s[i] + RC[k + i]) % P, 7, P) for i in range(CELLS)])
        k += CELLS
    for _ in range(RP):
        s[0] = pow((s[0] + RC[k]) % P, 7, P)
        k += 1
        s = _matvec(_MI, s)
    for _ in range(RF_HALF):
        s = _matvec(_ME, [pow((s[i] + RC[k + i]) % P, 7, P) for i in range(CELLS)])
        k += CELLS
    return s


def hash_elems(elems):
    """overwrite-mode sponge; returns the 8 canonical output elements"""
    s = [0] * CELLS
    n = len(elems)
    i = 0
    while n - i >= RATE:
        s[:RATE] = elems[i : i + RATE]
        s = poseidon2_permute(s)
        i += RATE
    if i < n or n == 0:
        blk = list(elems[i:]) + [0] * (RATE - (n - i))
        s[:RATE] = blk
        s = poseidon2_permute(s)
    return s[:8]


def hash_pair(a, b):
    return poseidon2_permute(list(a) + list(b) + [0] * 8)[:8]


def merkle_root(leaf_digests):
    layer = list(leaf_digests)
    while len(layer) > 1:
        layer = [hash_pair(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    return layer[0]
