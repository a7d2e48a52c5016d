// host_arith_check.cpp — CPU check of the exact arithmetic source the gfx950 Poseidon2/NTT kernels are built from
// (boundless_amd/csrc/fp.hpp, poseidon2_arith.hpp), compiled with -DBX_CHECK_BOUNDS so every documented bound and every
// 64-bit accumulation is asserted while extreme and random operands are pushed through.  Exact results come from
// (unsigned / signed) __int128 arithmetic mod P.  Built and run by tests/test_host_arith_cpu.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fp.hpp"
#include "poseidon2_arith.hpp"
#include "circuit_dev.hpp"
#include "lazy_ext.hpp"
#include "poseidon2_params.hpp"
#include "transcript.hpp"

using namespace bx;
typedef unsigned __int128 u128;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static const uint64_t RINV = 943718400ull;  // 2^-32 mod P
static uint32_t redc_exact(u128 v) { return (uint32_t)((v % P) * RINV % P); }
#define REQUIRE(cond)                                                     \
    do {                                                                  \
        if (!(cond)) {                                                    \
            fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__);    \
            return 1;                                                     \
        }                                                                 \
    } while (0)

// the circuit's cell value sum_t prod_f pool[idx(t,f)] through the signed three-level form (every sredc operand is asserted
// against SREDC_MAX by BX_CHECK_BOUNDS) against the plain canonical loop, pools drawn from the extreme words and at random
template <int TT, int GG>
static int check_cons_sum() {
    const uint32_t edge[] = {0, 1, P - 1, P / 2, P / 2 + 1, P / 2 - 1, MONT_ONE, P - MONT_ONE};
    uint32_t pool[Circuit::POOL];
    for (int iter = 0; iter < 60000; ++iter) {
        for (unsigned i = 0; i < Circuit::POOL; ++i) {
            const uint64_t r = rnd64();
            // a third of the pools all-extreme (worst magnitudes: +-P/2 everywhere), a third mixed, a third random
            pool[i] = iter % 3 == 0 ? edge[3 + r % 3] : iter % 3 == 1 ? ((r & 1) ? edge[(r >> 8) % 8] : (uint32_t)((r >> 8) % P)) : (uint32_t)((r >> 8) % P);
        }
        const uint32_t want = cons_sum<0, 0>(pool, TT, GG), got = cons_sum<TT, GG>(pool, TT, GG);
        if (got != want || got >= P) {
            fprintf(stderr, "cons_sum<%d,%d> mismatch: got %u want %u\n", TT, GG, got, want);
            return 1;
        }
    }
    return 0;
}

// NTT butterflies of ntt_r16.hpp.  (1) the fused-reduction butterfly takes ANY u32 operands (its own BX_CHECK_BOUNDS assertion
// watches the 64-bit accumulators) and returns the right residues; chains of them stay correct with no reduction in between;
// reduce_any() canonicalises any word.  (2) the lazy canonical forward butterfly as the kernels use it across a step boundary: the
// "a" operand is reduced where it is consumed, the "b" operand and both outputs are in [0, 2P) and never wrap.
static int check_ntt_butterflies() {
    const uint32_t edge[] = {0u, 1u, 2u, P - 1u, P, P + 1u, 2u * P - 1u, 2u * P, 2u * P + 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu, MONT_ONE, P - MONT_ONE};
    const uint32_t wedge[] = {0u, 1u, 2u, MONT_ONE, P - MONT_ONE, P / 2, P / 2 + 1, P - 2u, P - 1u};
    const int ne = (int)(sizeof edge / sizeof edge[0]), nw = (int)(sizeof wedge / sizeof wedge[0]);
    for (int iter = 0; iter < 400000; ++iter) {
        const uint64_t r = rnd64(), r2 = rnd64();
        uint32_t a = iter < ne * ne * nw ? edge[iter % ne] : (iter & 1) ? edge[r % ne] : (uint32_t)r;
        uint32_t b = iter < ne * ne * nw ? edge[(iter / ne) % ne] : (iter & 2) ? edge[(r >> 8) % ne] : (uint32_t)(r >> 32);
        const uint32_t w = iter < ne * ne * nw ? wedge[(iter / (ne * ne)) % nw] : (iter & 4) ? wedge[r2 % nw] : (uint32_t)((r2 >> 8) % P);
        const uint32_t wn = w ? P - w : 0u;  // P - 0 = P is not a table entry: the stage tables hold w in [1, P)
        if (w == 0) continue;
        const uint32_t wu = redc_exact((u128)(a % P) * MONT_ONE + (u128)(b % P) * w);
        const uint32_t wd = redc_exact((u128)(a % P) * MONT_ONE + (u128)(b % P) * wn);
        uint32_t u = a, d = b;
        bfly_fused(u, d, w, wn);
        REQUIRE(u % P == wu && d % P == wd);
        REQUIRE(reduce_any(u) == wu && reduce_any(d) == wd && reduce_any(a) == a % P);
        // four more stages on the unreduced outputs (any u32 in, any u32 out), against the canonical chain
        uint32_t cu = wu, cd = wd;
        for (int k = 0; k < 4; ++k) {
            const uint32_t wk = (uint32_t)(rnd64() % (P - 1)) + 1u;
            bfly_fused(u, d, wk, P - wk);
            const uint32_t t = fp_mul(cd, wk), nu = fp_add(cu, t), nd = fp_sub(cu, t);
            cu = nu;
            cd = nd;
            REQUIRE(u % P == cu && d % P == cd);
        }
        // lazy canonical butterfly across a step boundary: a, b in [0, 2P) as the previous step's last stage left them
        const uint32_t la = a % (2u * P), lb = b % (2u * P);
        const uint32_t ar = fp_reduce(la);                 // IN_LAZY: the operand that is added is reduced where it is consumed
        const uint32_t t = fp_mul(lb, w);                  // the multiplied operand is taken as it is
        const uint64_t su = (uint64_t)ar + t, sd = (uint64_t)ar + P - t;
        REQUIRE(t < P && su < 2ull * P && sd < 2ull * P && sd > 0);  // OUT_LAZY outputs fit the word and stay below 2P
        REQUIRE((uint32_t)su % P == fp_add(la % P, fp_mul(lb % P, w)) && (uint32_t)sd % P == fp_sub(la % P, fp_mul(lb % P, w)));
    }
    return 0;
}

// LazyExtAcc (mix_poly_coeffs, batch_evaluate_any, eval_check's mixing): sum_k w_k * x_k against f4_scale + f4_add, for term
// counts around every fold boundary, worst-case magnitudes (weights +-P/2, x = P - 1) and random operands
static int check_lazy_ext_acc() {
    const int counts[] = {0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 100, 336, 1000};
    for (int mode = 0; mode < 3; ++mode)
        for (int n : counts) {
            LazyExtAcc acc;
            acc.reset();
            Fp4 want = f4_zero();
            for (int k = 0; k < n; ++k) {
                Fp4 w;
                uint32_t x;
                for (int c = 0; c < 4; ++c)
                    w.c[c] = mode == 0 ? (uint32_t)(rnd64() % P) : ((rnd64() & 1) ? P / 2 : P / 2 + 1);  // centred: +P/2 or -P/2
                x = mode == 0 ? (uint32_t)(rnd64() % P) : (mode == 1 ? P - 1 : (uint32_t)(rnd64() % 3) * (P / 2));
                const i32 wc[4] = {fp_centre_w(w.c[0]), fp_centre_w(w.c[1]), fp_centre_w(w.c[2]), fp_centre_w(w.c[3])};
                acc.add(wc, x);
                want = f4_add(want, f4_scale(w, x));
            }
            const Fp4 got = acc.finish();
            for (int c = 0; c < 4; ++c)
                if (got.c[c] != want.c[c] || got.c[c] >= P) {
                    fprintf(stderr, "LazyExtAcc mismatch: mode %d, %d terms, component %d\n", mode, n, c);
                    return 1;
                }
        }
    return 0;
}

int main() {
    if (check_lazy_ext_acc()) return 1;
    if (check_ntt_butterflies()) return 1;
    if (check_cons_sum<64, 4>() || check_cons_sum<48, 3>() || check_cons_sum<16, 3>() || check_cons_sum<32, 3>() || check_cons_sum<8, 2>() || check_cons_sum<5, 1>() ||
        check_cons_sum<7, 4>() || check_cons_sum<64, 5>() || check_cons_sum<1, 1>() || check_cons_sum<25, 2>())
        return 1;
    REQUIRE((u128)RINV * ((u128)1 << 32) % P == 1);
    // ---- canonical field ops at the edges and at random ----
    const uint32_t edge[] = {0, 1, 2, P - 1, P - 2, (P - 1) / 2, MONT_ONE, R2, R3, 0x7fffffffu % P};
    for (uint32_t a : edge)
        for (uint32_t b : edge) {
            REQUIRE(fp_mul(a, b) == redc_exact((u128)a * b));
            REQUIRE(fp_add(a, b) == (uint32_t)(((uint64_t)a + b) % P));
            REQUIRE(fp_sub(a, b) == (uint32_t)(((uint64_t)a + P - b) % P));
        }
    for (int i = 0; i < 2000000; ++i) {
        uint32_t a = (uint32_t)(rnd64() % P), b = (uint32_t)(rnd64() % P);
        REQUIRE(fp_mul(a, b) == redc_exact((u128)a * b));
    }
    // ---- lazy multiply-add across its whole contract: a*b + c < 2^64 - (2^32-1)P ----
    for (int i = 0; i < 2000000; ++i) {
        uint32_t a = (uint32_t)rnd64(), b = (uint32_t)rnd64(), c = (uint32_t)rnd64();
        u128 v = (u128)a * b + c;
        if (v + (u128)0xffffffffu * P >= ((u128)1 << 64)) continue;
        uint32_t r = fp_mad_lazy(a, b, c);
        REQUIRE(r % P == redc_exact(v));
        REQUIRE((u128)r <= v / ((u128)1 << 32) + P);
    }
    // ---- signed Montgomery reduction over its whole contract ----
    typedef __int128 s128;
    const auto smod = [](s128 v) { s128 r = v % (s128)P; return (uint32_t)(r < 0 ? r + P : r); };  // exact residue in [0, P)
    const auto res = [&](i32 v) { return smod((s128)v); };
    {
        const i64 ts[] = {0, 1, -1, (i64)P, -(i64)P, SREDC_MAX, -SREDC_MAX, SREDC_MAX - 1, ((i64)1 << 32), -((i64)1 << 32), 0x7fffffffll,
                          -(i64)0x80000000ll, (i64)0xffffffffll};
        for (i64 tv : ts) {
            i32 r = sredc(tv);
            REQUIRE(res(r) == (uint32_t)((u128)smod((s128)tv) * RINV % P));
            // t/2^32 - P/2 - 1 <= r <= t/2^32 + P/2 + 1
            s128 lo = (s128)tv - ((s128)(P / 2 + 1) << 32), hi = (s128)tv + ((s128)(P / 2 + 1) << 32);
            REQUIRE(((s128)r << 32) >= lo - ((s128)1 << 32) && ((s128)r << 32) <= hi + ((s128)1 << 32));
        }
        for (int i = 0; i < 2000000; ++i) {
            i64 tv = (i64)(rnd64() % (uint64_t)(2 * SREDC_MAX + 1)) - SREDC_MAX;
            i32 r = sredc(tv);
            REQUIRE(res(r) == (uint32_t)((u128)smod((s128)tv) * RINV % P));
            REQUIRE((r < 0 ? -(i64)r : (i64)r) <= ub(tv < 0 ? -tv : tv));
        }
    }
    // ---- S-box on the whole admissible input range incl. both edges ----
    {
        const i64 sb_in[] = {0, 1, -1, (i64)P - 1, -((i64)P - 1), B_EXT, -B_EXT, B_INT, -B_INT, B_INT - 1, (i64)(P / 2), -(i64)(P / 2)};
        const auto want7 = [&](i32 x) {  // x^7 * 2^(-6*32)
            u128 xr = res(x), e = xr;
            for (int k = 0; k < 6; ++k) e = (u128)redc_exact(e * xr);
            return (uint32_t)e;
        };
        for (i64 x : sb_in) {
            if (x > B_INT || x < -B_INT) continue;
            i32 r = sbox7s((i32)x);
            REQUIRE(res(r) == want7((i32)x) && (r < 0 ? -(i64)r : (i64)r) < (i64)((double)P * 0.885));
        }
        for (int i = 0; i < 500000; ++i) {
            i32 x = (i32)((i64)(rnd64() % (uint64_t)(2 * B_INT + 1)) - B_INT);
            REQUIRE(res(sbox7s(x)) == want7(x));
        }
    }
    // ---- the returns to 32 bits over |y| <= B_Y incl. the edges ----
    {
        const i64 ys[] = {0, 1, -1, (i64)P, -(i64)P, 0xffffffffll, -(i64)0xffffffffll, 0x100000000ll, -0x100000000ll, B_Y, -B_Y, B_Y - 1,
                          0x7fffffffll, 0x80000000ll, -0x80000000ll, 0x17fffffffll, -0x180000001ll};
        const auto red_mid = [](i64 y, uint32_t a) { return red64ks<K1_MID, K2_MID>(y, a); };
        const auto red_end = [](i64 y, uint32_t a) { return red64ks<K1_END, K2_END>(y, a); };
        const auto mag = [](i32 v) { return v < 0 ? -(i64)v : (i64)v; };
        for (i64 y : ys)
            for (uint32_t add : edge) {
                const uint32_t a = add % P;
                const uint32_t yr = smod((s128)y);
                i32 r = redc64s(y, a);
                REQUIRE(res(r) == (uint32_t)((u128)((yr + (uint64_t)a) % P) * RINV % P) && mag(r) <= B_EXT);
                i32 m = red_mid(y, a), e = red_end(y, a);
                REQUIRE(res(m) == (uint32_t)(((u128)yr * K1_MID + a) % P * RINV % P) && mag(m) <= B_MIDOUT);
                REQUIRE(res(e) == (uint32_t)(((u128)yr * K1_END + a) % P * RINV % P) && mag(e) < (i64)P);
                REQUIRE(canon(e) == res(e));
            }
        for (int i = 0; i < 500000; ++i) {
            i64 y = (i64)(rnd64() % (uint64_t)(2 * B_Y + 1)) - B_Y;
            uint32_t a = (uint32_t)(rnd64() % P);
            const uint32_t yr = smod((s128)y);
            REQUIRE(res(redc64s(y, a)) == (uint32_t)((u128)((yr + (uint64_t)a) % P) * RINV % P));
            REQUIRE(res(red_mid(y, a)) == (uint32_t)(((u128)yr * K1_MID + a) % P * RINV % P));
            REQUIRE(canon(red_end(y, 0)) == (uint32_t)((u128)yr * K1_END % P * RINV % P));
        }
    }
    // the representation constants are what the exponent bookkeeping says (R = 2^32 mod P)
    {
        const auto rp = [](int64_t e) { return cx_rpow(e); };
        REQUIRE(cx_mul(rp(1), rp(-1)) == 1 && rp(1) == MONT_ONE && rp(2) == R2 && rp(3) == R3 && rp(-1) == RINV);
        int64_t e = 1;                      // canonical Montgomery input
        e -= 1;                             // initial layer, bare REDC
        for (int r = 0; r < 3; ++r) e = 7 * e - 6 - 1;  // S-box, layer, bare REDC
        REQUIRE(e == -399);
        REQUIRE(cx_mul(rp(7 * e - 6), K1_MID) == rp(2));  // (y * K1_MID) * R^-1 is back at R^1
        e = 1;                              // after the internal rounds
        e = 7 * e - 6 - 1;                  // round 4
        for (int r = 0; r < 2; ++r) e = 7 * e - 6 - 1;  // rounds 5, 6
        REQUIRE(e == -56 && cx_mul(rp(7 * e - 6), K1_END) == rp(2));
        REQUIRE(p2_rc_scale(0) == rp(1) && p2_rc_scale(24) == rp(-6) && p2_rc_scale(48) == rp(-55) && p2_rc_scale(72) == rp(-398));
        REQUIRE(p2_rc_scale(96) == R2 && p2_rc_scale(117) == R2 && p2_rc_scale(141) == rp(1) && p2_rc_scale(165) == rp(-6) &&
                p2_rc_scale(189) == rp(-55));
    }
    // ---- the external layer on extreme int32 cells ----
    {
        i32 s[24];
        i64 y[24];
        const int M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
        for (int t = 0; t < 20003; ++t) {
            for (int i = 0; i < 24; ++i)
                s[i] = t == 0 ? 0x7fffffff : t == 1 ? (i32)0x80000000 : t == 2 ? ((i & 1) ? 0x7fffffff : (i32)0x80000000) : (i32)(uint32_t)rnd64();
            m_ext64s(s, y);
            for (int i = 0; i < 24; ++i) {
                s128 want = 0;
                for (int j = 0; j < 24; ++j) want += (s128)(M4[i & 3][j & 3] * (i / 4 == j / 4 ? 2 : 1)) * s[j];
                REQUIRE((s128)y[i] == want && (y[i] < 0 ? -y[i] : y[i]) <= B_Y);
            }
        }
    }
    // ---- internal-round sum_r at its edges ----
    for (i64 sum : {(i64)0, (i64)P, -(i64)P, (i64)0xffffffffll, (i64)0x7fffffffll, (i64)0x80000000ll, -(i64)0x80000000ll, -(i64)0x80000001ll,
                    24 * B_INT, -24 * B_INT, ((i64)1 << 37) - 1, -(((i64)1 << 37) - 1)})
        REQUIRE(res(internal_sum_rs(sum)) == (uint32_t)((u128)smod((s128)sum) * (((u128)1 << 32) % P) % P));
    for (int i = 0; i < 500000; ++i) {
        i64 sum = (i64)(rnd64() >> 26) - ((i64)1 << 37);
        if (sum <= -((i64)1 << 37)) continue;
        REQUIRE(res(internal_sum_rs(sum)) == (uint32_t)((u128)smod((s128)sum) * (((u128)1 << 32) % P) % P));
    }
    // ---- the whole permutation, device order and arithmetic, vs the plain canonical implementation ----
    {
        HostPoseidon2 ref;
        ref.load(POSEIDON2_RC, POSEIDON2_DIAG);
        uint32_t prm[240];
        memset(prm, 0, sizeof prm);
        for (int i = 0; i < 213; ++i) prm[i] = (uint32_t)((uint64_t)POSEIDON2_RC[i] * p2_rc_scale(i) % P);
        for (int i = 0; i < 24; ++i) prm[216 + i] = fp_encode(POSEIDON2_DIAG[i]);
        const uint32_t pool[] = {0, 1, 2, P - 1, P - 2, (P - 1) / 2, MONT_ONE, R2};
        for (int t = 0; t < 6000; ++t) {
            uint32_t a[24], b[24];
            for (int i = 0; i < 24; ++i) {
                uint64_t r = rnd64();
                a[i] = t < 2000 ? pool[r % 8] : (t == 2000 ? P - 1 : (uint32_t)(r % P));
            }
            uint32_t v[24];
            memcpy(b, a, sizeof a);
            memcpy(v, a, sizeof a);
            ref.mix_scalar(a);
            poseidon2_mix_bounded<216>(b, prm);
            REQUIRE(memcmp(a, b, sizeof a) == 0);
            ref.mix(v);  // the AVX2 form where the CPU has it (the transcript's and the verifier's), else the scalar one again
            REQUIRE(memcmp(a, v, sizeof a) == 0);
        }
        // published KAT through the bounded form
        uint32_t k[24];
        for (int i = 0; i < 24; ++i) k[i] = fp_encode((uint32_t)i);
        poseidon2_mix_bounded<216>(k, prm);
        REQUIRE(fp_decode(k[0]) == 0x2ed3e23du && fp_decode(k[1]) == 0x12921fb0u && fp_decode(k[23]) == 0x57a99864u);
    }
    {   // the host permutation the prover's transcript and the verifier run (vector form where available): published KAT
        HostPoseidon2 h;
        h.load(POSEIDON2_RC, POSEIDON2_DIAG);
        uint32_t k[24];
        for (int i = 0; i < 24; ++i) k[i] = fp_encode((uint32_t)i);
        h.mix(k);
        REQUIRE(fp_decode(k[0]) == 0x2ed3e23du && fp_decode(k[1]) == 0x12921fb0u && fp_decode(k[23]) == 0x57a99864u);
        printf("host permutation: %s form\n", h.vec ? "AVX2" : "scalar");
    }
    printf("host_arith_check ok\n");
    return 0;
}
