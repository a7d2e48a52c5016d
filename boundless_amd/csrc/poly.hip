// poly.hip — FRI fold, DEEP mixing/evaluation/division and element-wise helpers for gfx950.
//
// Restates risc0_zkp::hal::Hal::{fri_fold, mix_poly_coeffs, batch_evaluate_any, eltwise_*, gather_sample} and
// risc0_zkp::core::poly::poly_divide (risc0-zkp 3.0.3, reference Cargo.lock:9155), reached from
// bento/crates/workflow/src/tasks/prove.rs:41-49 through Prover::finalize / fri_prove.
// All of these stream each input word once; they are HBM-bound except batch_evaluate_any and
// poly_divide, whose Fp4 products make them VALU-bound.
#define BX_PLAIN_MAD 1  // the signed multiply-adds of lazy_ext.hpp are left to the compiler here (no loop-carried cell state)
#include <algorithm>

#include "ctx.hpp"
#include "lazy_ext.hpp"

namespace bx {

struct W4 {  // a centred Fp4 weight
    i32 c[4];
};
__device__ __forceinline__ W4 ldw4(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return W4{{(i32)v.x, (i32)v.y, (i32)v.z, (i32)v.w}};
}
__device__ __forceinline__ Fp4 ld4(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Fp4{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st4(uint32_t* p, const Fp4& a) {
    *reinterpret_cast<uint4*>(p) = make_uint4(a.c[0], a.c[1], a.c[2], a.c[3]);
}

// ---- fri_fold: out[k*count + idx] = sum_i mix^i * in[(k*16 + rev4(i))*count + idx] ----
__global__ __launch_bounds__(256) void fri_fold_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, Fp4 mix,
                                                       const uint32_t* __restrict__ mix_dev, size_t count) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    if (mix_dev) mix = Fp4{{mix_dev[0], mix_dev[1], mix_dev[2], mix_dev[3]}};  // a challenge drawn on the device (bx_transcript_step)
    // Horner from the highest power keeps one running product: tot = (((f15*mix + f14)*mix + ...)*mix + f0
    Fp4 tot = f4_zero();
#pragma unroll
    for (int i = BX_FRI_FOLD - 1; i >= 0; --i) {
        size_t r = (size_t)bit_reverse((uint32_t)i, 4) * count + idx;
        Fp4 f{{in[r], in[16 * count + r], in[32 * count + r], in[48 * count + r]}};
        tot = f4_add(f4_mul(tot, mix), f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k * count + idx] = tot.c[k];
}

// ---- mix_poly_coeffs ----
// prep (one workgroup): order[] = polynomial indices grouped by combo, starts[c] = first slot of combo c,
// pows[i] = mix_start * mix^i.
__global__ void mix_prep_kernel(const uint32_t* __restrict__ combos, uint32_t input_size, uint32_t n_combos,
                                uint32_t* __restrict__ order, uint32_t* __restrict__ starts, uint32_t* __restrict__ pows,
                                Fp4 mix_start, Fp4 mix) {
    for (uint32_t i = threadIdx.x; i < input_size; i += blockDim.x) {  // stored centred: the weights of a LazyExtAcc
        const Fp4 w = f4_mul(mix_start, f4_pow(mix, i));
        st4(pows + 4 * (size_t)i, Fp4{{(uint32_t)fp_centre_w(w.c[0]), (uint32_t)fp_centre_w(w.c[1]), (uint32_t)fp_centre_w(w.c[2]),
                                       (uint32_t)fp_centre_w(w.c[3])}});
    }
    // order[] = a counting sort of the columns by combo.  Up to 1024 columns every thread ranks its own columns against an LDS copy
    // of the combo ids (the single-thread loop this replaces walked n_combos x input_size dependent global loads: 23 us per call,
    // four calls per proof, on an otherwise idle GPU)
    __shared__ uint32_t ids[1024];
    if (input_size <= 1024) {
        for (uint32_t i = threadIdx.x; i < input_size; i += blockDim.x) ids[i] = combos[i];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < input_size; i += blockDim.x) {
            const uint32_t c = ids[i];
            uint32_t pos = 0;
            for (uint32_t j = 0; j < input_size; ++j) pos += (ids[j] < c) || (ids[j] == c && j < i);
            order[pos] = i;
        }
        for (uint32_t c = threadIdx.x; c <= n_combos; c += blockDim.x) {
            uint32_t pos = 0;
            for (uint32_t j = 0; j < input_size; ++j) pos += ids[j] < c;
            starts[c] = pos;
        }
    } else if (threadIdx.x == 0) {
        uint32_t pos = 0;
        for (uint32_t c = 0; c < n_combos; ++c) {
            starts[c] = pos;
            for (uint32_t i = 0; i < input_size; ++i)
                if (combos[i] == c) order[pos++] = i;
        }
        starts[n_combos] = pos;
    }
}
__global__ __launch_bounds__(256) void mix_poly_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                       const uint32_t* __restrict__ order, const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ pows, size_t count) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    uint32_t combo = blockIdx.y;
    uint32_t lo = starts[combo], hi = starts[combo + 1];
    if (lo == hi) return;
    uint32_t* o = out + 4 * ((size_t)combo * count + idx);
    LazyExtAcc acc;
    acc.reset();
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t i = order[k];
        const W4 w = ldw4(pows + 4 * (size_t)i);  // wave-uniform, centred by mix_prep_kernel
        acc.add(w.c, in[(size_t)i * count + idx]);
    }
    st4(o, f4_add(ld4(o), acc.finish()));
}
// Four consecutive coefficients per thread (one 16-byte load per polynomial, two polynomials in flight): the one-word
// form above keeps a single 4-byte load per lane in flight and is bound by memory latency (3.0 TB/s on 256 polynomials).
__global__ __launch_bounds__(256) void mix_poly_x4_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                          const uint32_t* __restrict__ order, const uint32_t* __restrict__ starts,
                                                          const uint32_t* __restrict__ pows, size_t count) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= count) return;
    const uint32_t combo = blockIdx.y;
    const uint32_t lo = starts[combo], hi = starts[combo + 1];
    if (lo == hi) return;
    LazyExtAcc acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q].reset();
    uint32_t k = lo;
    for (; k + 2 <= hi; k += 2) {
        const uint32_t i0 = order[k], i1 = order[k + 1];
        const uint4 v0 = *reinterpret_cast<const uint4*>(in + (size_t)i0 * count + idx);
        const uint4 v1 = *reinterpret_cast<const uint4*>(in + (size_t)i1 * count + idx);
        const W4 w0 = ldw4(pows + 4 * (size_t)i0), w1 = ldw4(pows + 4 * (size_t)i1);  // wave-uniform
        acc[0].add(w0.c, v0.x); acc[1].add(w0.c, v0.y); acc[2].add(w0.c, v0.z); acc[3].add(w0.c, v0.w);
        acc[0].add(w1.c, v1.x); acc[1].add(w1.c, v1.y); acc[2].add(w1.c, v1.z); acc[3].add(w1.c, v1.w);
    }
    if (k < hi) {
        const uint32_t i0 = order[k];
        const uint4 v0 = *reinterpret_cast<const uint4*>(in + (size_t)i0 * count + idx);
        const W4 w0 = ldw4(pows + 4 * (size_t)i0);
        acc[0].add(w0.c, v0.x); acc[1].add(w0.c, v0.y); acc[2].add(w0.c, v0.z); acc[3].add(w0.c, v0.w);
    }
    uint32_t* o = out + 4 * ((size_t)combo * count + idx);
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + 4 * q, f4_add(ld4(o + 4 * q), acc[q].finish()));
}
// ---- batch_evaluate_any ----
// grid = (segments, evals).  A workgroup evaluates SEG = 256*K consecutive coefficients of polynomial which[e] at x:
//   x^seg_base * sum_{t<256} x^t * sum_{i<K} c[base + t + 256 i] * (x^256)^i.
// The 256 + K powers are built once per workgroup by doubling in LDS (pw[d + i] = pw[d] * pw[i]), so the per-coefficient
// cost is one Fp4-by-Fp product (4 Montgomery products) plus one LDS broadcast read.
// Bit-reversed storage (brev_log = n > 0, poly_size = 2^n >= one segment): position j = seg * 2^15 + ii * 256 + t holds the
// coefficient of x^bitrev_n(j) = x^(rev(seg)) * (x^(2^(n-15)))^rev7(ii) * (x^(2^(n-8)))^rev8(t), so the same two power tables
// serve with bases x^(2^(n-8)) and x^(2^(n-15)) and bit-reversed table indices.
constexpr int EV_K = 128, EV_T = 256;
__global__ __launch_bounds__(EV_T) void eval_partial_kernel(const uint32_t* __restrict__ coeffs, size_t poly_size,
                                                            const uint32_t* __restrict__ which, const uint32_t* __restrict__ xs,
                                                            uint32_t* __restrict__ partials, uint32_t segs, int brev_log) {
    __shared__ uint32_t xpow[EV_T * 4];  // x^t (reused for the final reduction)
    __shared__ uint32_t ypow[EV_K * 4];  // (x^256)^i
    const uint32_t e = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const Fp4 x0 = ld4(xs + 4 * (size_t)e);
    Fp4 x = x0;  // base of the 256-entry table: x, or x^(2^(n-8)) for bit-reversed storage
    Fp4 xk = x0;  // x^(2^(n-15)), base of the 128-entry table for bit-reversed storage
    if (brev_log) {
        for (int i = 0; i < brev_log - 15; ++i) xk = f4_mul(xk, xk);
        x = xk;
        for (int i = 0; i < 7; ++i) x = f4_mul(x, x);
    }
    const size_t seg_elems = (size_t)EV_T * EV_K;
    const size_t base = (size_t)seg * seg_elems;
    const uint32_t* c = coeffs + (size_t)which[e] * poly_size + base;
    const size_t remaining = poly_size - base;
    if (tid == 0) {
        st4(xpow, f4_one());
        st4(xpow + 4, x);
    }
    __syncthreads();
    for (uint32_t d = 2; d < EV_T; d <<= 1) {  // pw[d .. 2d) = pw[d-1] * x * pw[0 .. d)  ==  pw[d] * pw[i]
        if (tid < d) {
            Fp4 top = f4_mul(ld4(xpow + 4 * (d - 1)), x);  // x^d
            st4(xpow + 4 * (d + tid), f4_mul(top, ld4(xpow + 4 * tid)));
        }
        __syncthreads();
    }
    const Fp4 X = brev_log ? xk : f4_mul(ld4(xpow + 4 * (EV_T - 1)), x);  // x^256 (natural order)
    if (tid == 0) {
        st4(ypow, f4_one());
        st4(ypow + 4, X);
    }
    __syncthreads();
    for (uint32_t d = 2; d < EV_K; d <<= 1) {
        if (tid < d) {
            Fp4 top = f4_mul(ld4(ypow + 4 * (d - 1)), X);
            st4(ypow + 4 * (d + tid), f4_mul(top, ld4(ypow + 4 * tid)));
        }
        __syncthreads();
    }
    // the (x^256)^i table becomes the centred weights of a LazyExtAcc: 10 instead of 32 instructions per coefficient
    if (tid < EV_K) {
        const Fp4 w = ld4(ypow + 4 * tid);
        st4(ypow + 4 * tid, Fp4{{(uint32_t)fp_centre_w(w.c[0]), (uint32_t)fp_centre_w(w.c[1]), (uint32_t)fp_centre_w(w.c[2]),
                                 (uint32_t)fp_centre_w(w.c[3])}});
    }
    __syncthreads();
    LazyExtAcc lz;
    lz.reset();
    if (brev_log) {
#pragma unroll 16
        for (int i = 0; i < EV_K; ++i) lz.add(ldw4(ypow + 4 * (__brev((uint32_t)i) >> 25)).c, c[(size_t)tid + (size_t)EV_T * i]);
    } else if (remaining >= seg_elems) {
#pragma unroll 16
        for (int i = 0; i < EV_K; ++i) lz.add(ldw4(ypow + 4 * i).c, c[(size_t)tid + (size_t)EV_T * i]);
    } else {
        for (int i = 0; i < EV_K; ++i) {
            size_t j = (size_t)tid + (size_t)EV_T * i;
            if (j < remaining) lz.add(ldw4(ypow + 4 * i).c, c[j]);
        }
    }
    Fp4 acc = f4_mul(lz.finish(), ld4(xpow + 4 * (brev_log ? (__brev(tid) >> 24) : tid)));
    __syncthreads();
    st4(xpow + 4 * tid, acc);
    __syncthreads();
    for (int s = EV_T / 2; s > 0; s >>= 1) {
        if ((int)tid < s) st4(xpow + 4 * tid, f4_add(ld4(xpow + 4 * tid), ld4(xpow + 4 * (tid + s))));
        __syncthreads();
    }
    if (tid == 0) {
        // bit-reversed storage: the segment's exponent is rev(seg) over the brev_log - 15 segment bits (none at 2^15)
        const uint64_t ex = brev_log ? (brev_log > 15 ? (uint64_t)(__brev(seg) >> (32 - (brev_log - 15))) : 0ull) : (uint64_t)base;
        st4(partials + 4 * ((size_t)e * segs + seg), f4_mul(ld4(xpow), f4_pow(x0, ex)));
    }
}
// The same with 16-byte loads and per-evaluation tables, for whole segments (poly_size a multiple of 2^15, 16-byte aligned).
// The dword kernel above rebuilds its power tables in every workgroup — ~20 dependent ext products and a dozen barriers before the
// first coefficient is read, once per (evaluation, segment) — and was latency-bound at 2 TB/s.  Here eval_tables_kernel builds the
// tables once per evaluation point; a workgroup of eval_partial_x4_kernel then only streams: a thread owns the four consecutive
// positions 4t .. 4t+3 of each of the segment's 32 runs of 1024 (32 dwordx4 loads, 8 in flight), its own weight comes from one
// 16-byte load, the 32 run weights are wave-uniform scalar loads, and the 256 partial sums meet through one cross-lane reduction
// and one barrier.
//   position j = seg * 2^15 + i * 1024 + 4 t + k   (i < 32, t < 256, k < 4)
//   natural order:   exponent j             = seg 2^15 + 1024 i + 4 t + k          A = x,            B = x^4,          C = x^1024
//   bit-reversed:    exponent bitrev_n(j)   = rev(seg) + 2^(n-15) rev5(i) + 2^(n-10) rev8(t) + 2^(n-2) rev2(k)
//                                                                                 C = x^(2^(n-15)), B = C^32,         A = B^256
// value = S[seg] * sum_t B^e(t) * sum_k A^e(k) * sum_i C^e(i) c[j]; the innermost sums run unreduced (LazyExtAcc).
// Table of one evaluation (words): [0,1024) tw[t] = B^e(t) | [1024,1152) iw[i] = C^e(i), centred | [1152,1168) A^e(k), k < 4 |
// [1168, 1168 + 4 segs) S[seg] = x^(seg 2^15) or x^rev(seg).  Already indexed by position: the main kernel never bit-reverses.
constexpr uint32_t EVT_TW = 0, EVT_IW = 1024, EVT_KW = 1152, EVT_SW = 1168;
__global__ __launch_bounds__(EV_T) void eval_tables_kernel(const uint32_t* __restrict__ xs, uint32_t* __restrict__ tables, uint32_t segs,
                                                           uint32_t stride, int brev_log_all, const uint32_t* __restrict__ flags, int brev_log_flagged) {
    __shared__ uint32_t pw[512 * 4];
    const uint32_t e = blockIdx.x, tid = threadIdx.x;
    const int brev_log = flags ? ((flags[e] & 1u) ? brev_log_flagged : 0) : brev_log_all;  // per evaluation when several buffers share a launch
    uint32_t* T = tables + (size_t)e * stride;
    const Fp4 x0 = ld4(xs + 4 * (size_t)e);
    Fp4 A, B, Cc, G;  // G: base of the segment powers
    if (brev_log) {
        Cc = x0;
        for (int i = 0; i < brev_log - 15; ++i) Cc = f4_mul(Cc, Cc);
        B = Cc;
        for (int i = 0; i < 5; ++i) B = f4_mul(B, B);
        A = B;
        for (int i = 0; i < 8; ++i) A = f4_mul(A, A);
        G = x0;
    } else {
        A = x0;
        B = f4_mul(f4_mul(x0, x0), f4_mul(x0, x0));
        Cc = B;
        for (int i = 0; i < 8; ++i) Cc = f4_mul(Cc, Cc);
        G = Cc;
        for (int i = 0; i < 5; ++i) G = f4_mul(G, G);  // x^(2^15)
    }
    // powers of `base` by doubling in LDS: pw[d + i] = base^d * pw[i]
    const auto powers = [&](const Fp4& base, uint32_t count) {
        __syncthreads();
        if (tid == 0) {
            st4(pw, f4_one());
            st4(pw + 4, base);
        }
        __syncthreads();
        for (uint32_t d = 2; d < count; d <<= 1) {
            const Fp4 top = f4_mul(ld4(pw + 4 * (d - 1)), base);
            for (uint32_t i = tid; i < d; i += EV_T) st4(pw + 4 * (d + i), f4_mul(top, ld4(pw + 4 * i)));
            __syncthreads();
        }
    };
    powers(B, EV_T);
    st4(T + EVT_TW + 4 * tid, ld4(pw + 4 * (brev_log ? (__brev(tid) >> 24) : tid)));
    powers(Cc, 32);
    if (tid < 32) {
        const Fp4 w = ld4(pw + 4 * (brev_log ? (__brev(tid) >> 27) : tid));
        st4(T + EVT_IW + 4 * tid, Fp4{{(uint32_t)fp_centre_w(w.c[0]), (uint32_t)fp_centre_w(w.c[1]), (uint32_t)fp_centre_w(w.c[2]),
                                       (uint32_t)fp_centre_w(w.c[3])}});
    }
    if (tid == 0) {
        const Fp4 A2 = f4_mul(A, A), A3 = f4_mul(A2, A);
        st4(T + EVT_KW, f4_one());
        st4(T + EVT_KW + 4, brev_log ? A2 : A);  // k = 1, 2 carry exponents 1, 2 in natural order and rev2(k) = 2, 1 bit-reversed
        st4(T + EVT_KW + 8, brev_log ? A : A2);
        st4(T + EVT_KW + 12, A3);
    }
    uint32_t cnt = 2;
    while (cnt < segs) cnt <<= 1;  // segs <= 512 (2^24 coefficients)
    powers(G, cnt);
    const int sbits = brev_log ? brev_log - 15 : 0;
    for (uint32_t sgm = tid; sgm < segs; sgm += EV_T)
        st4(T + EVT_SW + 4 * sgm, ld4(pw + 4 * (sbits ? (__brev(sgm) >> (32 - sbits)) : (brev_log ? 0u : sgm))));
}
__device__ __forceinline__ Fp4 f4_shfl_down(const Fp4& v, int d) {
    return Fp4{{(uint32_t)__shfl_down((int)v.c[0], d), (uint32_t)__shfl_down((int)v.c[1], d), (uint32_t)__shfl_down((int)v.c[2], d),
                (uint32_t)__shfl_down((int)v.c[3], d)}};
}
__global__ __launch_bounds__(EV_T) void eval_partial_x4_kernel(const uint32_t* __restrict__ coeffs, size_t poly_size,
                                                               const uint32_t* __restrict__ which, const uint32_t* __restrict__ tables,
                                                               uint32_t stride, uint32_t* __restrict__ partials, uint32_t segs,
                                                               const unsigned long long* __restrict__ poly_ptrs) {
    __shared__ uint32_t red[4 * 4];
    const uint32_t e = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const uint32_t* T = tables + (size_t)e * stride;
    const uint32_t* poly = poly_ptrs ? reinterpret_cast<const uint32_t*>(poly_ptrs[e]) : coeffs + (size_t)which[e] * poly_size;
    const uint4* c4 = reinterpret_cast<const uint4*>(poly + (size_t)seg * ((size_t)EV_T * EV_K)) + tid;
    LazyExtAcc lz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lz[k].reset();
    // software pipeline, fully unrolled (the accumulators' fold counters become compile-time): the next batch of EV_B loads is
    // issued before the current one is consumed, so 8..16 x 16 bytes per lane are in flight at any time.  Left to itself the
    // compiler emitted load / wait / use, one 16-byte load in flight per lane.
    constexpr int EV_B = 8;
    uint4 cur[EV_B], nxt[EV_B];
#pragma unroll
    for (int u = 0; u < EV_B; ++u) cur[u] = c4[(size_t)u * EV_T];
#pragma unroll
    for (int ib = 0; ib < 32; ib += EV_B) {
        if (ib + EV_B < 32) {
#pragma unroll
            for (int u = 0; u < EV_B; ++u) nxt[u] = c4[(size_t)(ib + EV_B + u) * EV_T];
        }
#pragma unroll
        for (int u = 0; u < EV_B; ++u) {
            const W4 w = ldw4(T + EVT_IW + 4 * (ib + u));  // wave-uniform
            lz[0].add_centred(w.c, fp_centre_w(cur[u].x)); lz[1].add_centred(w.c, fp_centre_w(cur[u].y));
            lz[2].add_centred(w.c, fp_centre_w(cur[u].z)); lz[3].add_centred(w.c, fp_centre_w(cur[u].w));
        }
#pragma unroll
        for (int u = 0; u < EV_B; ++u) cur[u] = nxt[u];
    }
    Fp4 inner = f4_add(f4_add(lz[0].finish(), f4_mul(ld4(T + EVT_KW + 4), lz[1].finish())),
                       f4_add(f4_mul(ld4(T + EVT_KW + 8), lz[2].finish()), f4_mul(ld4(T + EVT_KW + 12), lz[3].finish())));
    Fp4 acc = f4_mul(inner, ld4(T + EVT_TW + 4 * tid));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc = f4_add(acc, f4_shfl_down(acc, d));
    if ((tid & 63u) == 0) st4(red + 4 * (tid >> 6), acc);
    __syncthreads();
    if (tid == 0) {
        const Fp4 tot = f4_add(f4_add(ld4(red), ld4(red + 4)), f4_add(ld4(red + 8), ld4(red + 12)));
        st4(partials + 4 * ((size_t)e * segs + seg), f4_mul(tot, ld4(T + EVT_SW + 4 * seg)));
    }
}
__global__ void eval_final_kernel(const uint32_t* __restrict__ partials, uint32_t segs, uint32_t* __restrict__ out,
                                  uint32_t evals) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= evals) return;
    Fp4 acc = f4_zero();
    for (uint32_t s = 0; s < segs; ++s) acc = f4_add(acc, ld4(partials + 4 * ((size_t)e * segs + s)));
    st4(out + 4 * (size_t)e, acc);
}

// ---- element-wise ----
__global__ void eltwise_add_kernel(uint32_t* __restrict__ out, const uint32_t* a, const uint32_t* b, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fp_add(a[i], b[i]);
}
__global__ void eltwise_mul_factor_kernel(uint32_t* __restrict__ io, uint32_t factor, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) io[i] = fp_mul(io[i], factor);
}
__global__ void eltwise_zeroize_kernel(uint32_t* __restrict__ io, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (io[i] == 0xffffffffu) io[i] = 0u;
}
__global__ __launch_bounds__(256) void eltwise_sum_ext_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                              size_t count, size_t to_add) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    Fp4 tot = f4_zero();
    for (size_t j = 0; j < to_add; ++j) tot = f4_add(tot, ld4(in + 4 * (j * count + idx)));
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k * count + idx] = tot.c[k];
}
__global__ void gather_sample_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t idx, size_t size,
                                     size_t stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < size) dst[i] = src[idx + i * stride];
}
// the queued form (bx_ctx::gq): one workgroup per descriptor {dst, src + idx, size, stride}
__global__ __launch_bounds__(64) void gather_batch_kernel(const uint4* __restrict__ descs) {
    const uint4 a = descs[2 * blockIdx.x], b = descs[2 * blockIdx.x + 1];
    uint32_t* __restrict__ dst = (uint32_t*)(((unsigned long long)a.y << 32) | a.x);
    const uint32_t* __restrict__ src = (const uint32_t*)(((unsigned long long)a.w << 32) | a.z);
    const uint32_t size = b.x, stride = b.y;
    for (uint32_t i = threadIdx.x; i < size; i += 64u) dst[i] = src[(size_t)i * stride];
}

// The chunk walks below are chains of dependent Fp4 products; their loads are independent, so they are issued eight at a
// time (SCAN_B elements = 128 bytes per lane in flight) instead of one per product.
constexpr int SCAN_B = 8;

// ---- poly_divide: q_{i-1} = p_i + z q_i (top down), in place; remainder = p_0 + z q_0 ----
// Three phases over chunks of DIV_L coefficients: (1) each chunk's carry-out assuming zero carry-in,
// (2) sequential composition of the chunk maps carry -> local + z^L * carry (one workgroup), (3) replay.
// The phases nest: the chunk values are themselves an array to be divided by (x - z^L) — "the carry entering chunk ch" is
// that division's quotient coefficient — so arrays longer than DIV_DIRECT recurse with chunks of DIV_L and the one-workgroup
// kernel only ever sees <= DIV_DIRECT entries (2^20 -> 2^15 -> 2^10: five launches, 123 -> 40 us per 2^20 coefficients).
constexpr int DIV_L = 32;
constexpr size_t DIV_DIRECT = 2048;
__global__ void div_local_kernel(const uint32_t* __restrict__ poly, size_t size, Fp4 z, uint32_t* __restrict__ local,
                                 size_t chunks) {
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= chunks) return;
    size_t lo = ch * DIV_L, hi = lo + DIV_L < size ? lo + DIV_L : size;
    Fp4 cur = f4_zero();
    for (size_t top = hi; top > lo;) {
        const size_t nb = top - lo < (size_t)SCAN_B ? top - lo : (size_t)SCAN_B;
        Fp4 v[SCAN_B];
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) v[k] = ld4(poly + 4 * (top - 1 - k));
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) cur = f4_add(f4_mul(z, cur), v[k]);
        top -= nb;
    }
    st4(local + 4 * ch, cur);
}
// carry_in[ch] = value of `cur` entering chunk ch from above.  One workgroup: thread t owns a contiguous run of chunks
// (thread 0 the highest), reduces it to the affine map carry -> a*carry + b, the maps are composed across threads with a
// log-step (Hillis-Steele) scan in LDS, and each thread replays its run with the carry that enters it.
__global__ void div_scan_kernel(uint32_t* __restrict__ local_then_carry, size_t chunks, Fp4 zL, uint32_t* __restrict__ rem) {
    extern __shared__ uint32_t sh[];  // per thread: a (4 words) | b (4 words)
    const uint32_t nt = blockDim.x, tid = threadIdx.x;
    size_t per = (chunks + nt - 1) / nt;
    size_t hi = chunks > (size_t)tid * per ? chunks - (size_t)tid * per : 0;
    size_t lo = hi > per ? hi - per : 0;
    Fp4 a = f4_one(), b = f4_zero();
    for (size_t ch = hi; ch-- > lo;) {
        b = f4_add(f4_mul(zL, b), ld4(local_then_carry + 4 * ch));
        a = f4_mul(a, zL);
    }
    // inclusive scan of F_t = f_t o f_(t-1) o ... o f_0 with (a2,b2) o (a1,b1) = (a2*a1, a2*b1 + b2)
    for (uint32_t d = 1; d < nt; d <<= 1) {
        st4(sh + 8 * tid, a);
        st4(sh + 8 * tid + 4, b);
        __syncthreads();
        if (tid >= d) {
            Fp4 pa = ld4(sh + 8 * (tid - d)), pb = ld4(sh + 8 * (tid - d) + 4);
            b = f4_add(f4_mul(a, pb), b);
            a = f4_mul(a, pa);
        }
        __syncthreads();
    }
    st4(sh + 8 * tid + 4, b);
    __syncthreads();
    if (tid == nt - 1) st4(rem, b);  // composition of every run applied to carry 0 = the remainder
    Fp4 carry = tid == 0 ? f4_zero() : ld4(sh + 8 * (tid - 1) + 4);
    for (size_t ch = hi; ch-- > lo;) {
        Fp4 l = ld4(local_then_carry + 4 * ch);
        st4(local_then_carry + 4 * ch, carry);
        carry = f4_add(f4_mul(zL, carry), l);
    }
}
__global__ void div_apply_kernel(uint32_t* __restrict__ poly, size_t size, Fp4 z, const uint32_t* __restrict__ carry_in,
                                 size_t chunks) {
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= chunks) return;
    size_t lo = ch * DIV_L, hi = lo + DIV_L < size ? lo + DIV_L : size;
    Fp4 cur = ld4(carry_in + 4 * ch);
    for (size_t top = hi; top > lo;) {
        const size_t nb = top - lo < (size_t)SCAN_B ? top - lo : (size_t)SCAN_B;
        Fp4 v[SCAN_B];
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) v[k] = ld4(poly + 4 * (top - 1 - k));
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) {
                st4(poly + 4 * (top - 1 - k), cur);
                cur = f4_add(f4_mul(z, cur), v[k]);
            }
        top -= nb;
    }
}

// ---- prefix_products: inclusive running product of ext elements, same three-phase shape as poly_divide ----
constexpr int PP_L = 64;
constexpr size_t PP_DIRECT = 2048;
// blockIdx.y = sequence of a batch (each with its own n elements of io and `chunks` aggregates)
__global__ void pp_local_kernel(const uint32_t* __restrict__ io, size_t n, size_t seq_stride, uint32_t* __restrict__ agg, size_t chunks,
                                size_t agg_stride) {
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= chunks) return;
    io += 4 * seq_stride * blockIdx.y;
    agg += 4 * agg_stride * blockIdx.y;
    size_t lo = ch * PP_L, hi = lo + PP_L < n ? lo + PP_L : n;
    Fp4 p = f4_one();
    for (size_t b = lo; b < hi; b += SCAN_B) {
        const size_t nb = hi - b < (size_t)SCAN_B ? hi - b : (size_t)SCAN_B;
        Fp4 v[SCAN_B];
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) v[k] = ld4(io + 4 * (b + k));
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) p = f4_mul(p, v[k]);
    }
    st4(agg + 4 * ch, p);
}
// agg[ch] <- product of all chunks before ch (exclusive scan), one workgroup, log-step scan across threads
__global__ void pp_scan_kernel(uint32_t* __restrict__ agg, size_t chunks, size_t agg_stride) {
    extern __shared__ uint32_t sh[];
    agg += 4 * agg_stride * blockIdx.x;
    const uint32_t nt = blockDim.x, tid = threadIdx.x;
    size_t per = (chunks + nt - 1) / nt;
    size_t lo = (size_t)tid * per < chunks ? (size_t)tid * per : chunks;
    size_t hi = lo + per < chunks ? lo + per : chunks;
    Fp4 mine = f4_one();
    for (size_t ch = lo; ch < hi; ++ch) mine = f4_mul(mine, ld4(agg + 4 * ch));
    Fp4 incl = mine;
    for (uint32_t d = 1; d < nt; d <<= 1) {
        st4(sh + 4 * tid, incl);
        __syncthreads();
        if (tid >= d) incl = f4_mul(incl, ld4(sh + 4 * (tid - d)));
        __syncthreads();
    }
    st4(sh + 4 * tid, incl);
    __syncthreads();
    Fp4 carry = tid == 0 ? f4_one() : ld4(sh + 4 * (tid - 1));
    for (size_t ch = lo; ch < hi; ++ch) {
        Fp4 a = ld4(agg + 4 * ch);
        st4(agg + 4 * ch, carry);
        carry = f4_mul(carry, a);
    }
}
// exclusive form for the inner levels: io[i] <- carry * prod_{lo <= k < i} io[k]
__global__ void pp_apply_excl_kernel(uint32_t* __restrict__ io, size_t n, size_t seq_stride, const uint32_t* __restrict__ carry_in,
                                     size_t chunks, size_t carry_stride) {
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= chunks) return;
    io += 4 * seq_stride * blockIdx.y;
    carry_in += 4 * carry_stride * blockIdx.y;
    size_t lo = ch * PP_L, hi = lo + PP_L < n ? lo + PP_L : n;
    Fp4 p = ld4(carry_in + 4 * ch);
    for (size_t b = lo; b < hi; b += SCAN_B) {
        const size_t nb = hi - b < (size_t)SCAN_B ? hi - b : (size_t)SCAN_B;
        Fp4 v[SCAN_B];
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) v[k] = ld4(io + 4 * (b + k));
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) {
                st4(io + 4 * (b + k), p);
                p = f4_mul(p, v[k]);
            }
    }
}
__global__ void pp_apply_kernel(uint32_t* __restrict__ io, size_t n, const uint32_t* __restrict__ carry_in, size_t chunks) {
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= chunks) return;
    io += 4 * n * blockIdx.y;
    carry_in += 4 * chunks * blockIdx.y;
    size_t lo = ch * PP_L, hi = lo + PP_L < n ? lo + PP_L : n;
    Fp4 p = ld4(carry_in + 4 * ch);
    for (size_t b = lo; b < hi; b += SCAN_B) {
        const size_t nb = hi - b < (size_t)SCAN_B ? hi - b : (size_t)SCAN_B;
        Fp4 v[SCAN_B];
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) v[k] = ld4(io + 4 * (b + k));
#pragma unroll
        for (int k = 0; k < SCAN_B; ++k)
            if ((size_t)k < nb) {
                p = f4_mul(p, v[k]);
                st4(io + 4 * (b + k), p);
            }
    }
}
// entries [index[0], index[last]) are written.  The per-cycle grouping of upstream's scatter only orders writes that hit
// the same offset, which its circuits never produce, so one pass over the range is equivalent.  Nothing is read back by
// the host: a bad offset or index range raises a word of the ctx's deferred error flags (reported by the next bx_d2h / bx_sync).
__global__ void scatter_kernel(uint32_t* __restrict__ into, const uint32_t* __restrict__ index, size_t index_len,
                               const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ values, size_t entries, size_t into_len,
                               uint32_t* __restrict__ flag) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = index[0], hi = index[index_len - 1];  // wave-uniform
    if (lo > hi || hi > entries) {
        if (e == 0) *(volatile uint32_t*)(flag + FLAG_SLOT_SCATTER_INDEX) = 1u;
        return;
    }
    if (e < lo || e >= hi) return;
    const uint32_t o = offsets[e];
    if (o < into_len) {
        into[o] = values[e];
    } else {
        *(volatile uint32_t*)(flag + FLAG_SLOT_SCATTER_RANGE) = 1u;
    }
}

const char* ensure_scratch(bx_ctx* c, size_t words) {
    if (c->scratch_words >= words) return nullptr;
    if (c->d_scratch) {
        BX_HIP(c, stream_wait(c));
        BX_HIP(c, hipFree(c->d_scratch));
        c->d_scratch = nullptr;
        c->scratch_words = 0;
    }
    size_t want = words < (1u << 20) ? (1u << 20) : words;
    BX_HIP(c, hipMalloc(&c->d_scratch, want * 4));
    c->scratch_words = want;
    return nullptr;
}

}  // namespace bx

using namespace bx;

static inline Fp4 host4(const uint32_t* w) { return Fp4{{w[0], w[1], w[2], w[3]}}; }
static inline unsigned grid1d(size_t n, unsigned bs = 256, size_t cap = 65536) {
    size_t b = (n + bs - 1) / bs;
    return (unsigned)(b > cap ? cap : (b ? b : 1));
}

extern "C" const char* bx_fri_fold(bx_ctx* c, bx_buf out, bx_buf in, const uint32_t mix[4]) try {
    if (!c) return "bx_fri_fold: null ctx";
    BX_REQUIRE(c, out.len % 4 == 0 && in.len == out.len * BX_FRI_FOLD, "fri_fold: input.len must be 16 * output.len");
    BX_ENTER(c);
    size_t count = out.len / 4;
    OpScope op(c, "fri_fold", 4.0 * (double)(in.len + out.len));
    if (!count) return nullptr;
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)in.dptr, host4(mix), (const uint32_t*)nullptr, count);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_fri_fold")
extern "C" const char* bx_fri_fold_dev(bx_ctx* c, bx_buf out, bx_buf in, bx_buf mix_ext) try {
    if (!c) return "bx_fri_fold_dev: null ctx";
    BX_REQUIRE(c, out.len % 4 == 0 && in.len == out.len * BX_FRI_FOLD, "fri_fold_dev: input.len must be 16 * output.len");
    BX_REQUIRE(c, mix_ext.dptr != nullptr && mix_ext.len >= 4, "fri_fold_dev: mix is one ext element in device memory");
    BX_ENTER(c);
    size_t count = out.len / 4;
    OpScope op(c, "fri_fold", 4.0 * (double)(in.len + out.len));
    if (!count) return nullptr;
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)in.dptr, f4_zero(), (const uint32_t*)mix_ext.dptr, count);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_fri_fold_dev")

extern "C" const char* bx_mix_poly_coeffs(bx_ctx* c, bx_buf out, const uint32_t mix_start[4], const uint32_t mix[4], bx_buf in,
                                          bx_buf combos, size_t input_size, size_t count) try {
    if (!c) return "bx_mix_poly_coeffs: null ctx";
    BX_REQUIRE(c, mul_le(input_size, count, in.len) && combos.len >= input_size, "mix_poly_coeffs: input/combos too small");
    BX_REQUIRE(c, count > 0 && mul_le(4, count, out.len) && out.len % (4 * count) == 0, "mix_poly_coeffs: out.len not a multiple of 4*count");
    BX_ENTER(c);
    if (!input_size) return nullptr;
    size_t n_combos = out.len / (4 * count);
    OpScope op(c, "mix_poly_coeffs", 4.0 * (double)(input_size * count) + 32.0 * (double)out.len / 4.0);
    // scratch: order[input_size] | starts[n_combos+1] | pows[4*input_size] (16B aligned first)
    size_t pows_off = 0, order_off = 4 * input_size, starts_off = order_off + input_size;
    BX_TRY(ensure_scratch(c, starts_off + n_combos + 8));
    uint32_t* s = c->d_scratch;
    hipLaunchKernelGGL(mix_prep_kernel, dim3(1), dim3(256), 0, c->stream, (const uint32_t*)combos.dptr, (uint32_t)input_size,
                       (uint32_t)n_combos, s + order_off, s + starts_off, s + pows_off, host4(mix_start), host4(mix));
    BX_LAUNCH_CHECK(c);
    if (count % 4 == 0 && ((uintptr_t)in.dptr & 15u) == 0)
        hipLaunchKernelGGL(mix_poly_x4_kernel, dim3((unsigned)((count / 4 + 255) / 256), (unsigned)n_combos), dim3(256), 0, c->stream,
                           (uint32_t*)out.dptr, (const uint32_t*)in.dptr, s + order_off, s + starts_off, s + pows_off, count);
    else
        hipLaunchKernelGGL(mix_poly_kernel, dim3((unsigned)((count + 255) / 256), (unsigned)n_combos), dim3(256), 0, c->stream,
                           (uint32_t*)out.dptr, (const uint32_t*)in.dptr, s + order_off, s + starts_off, s + pows_off, count);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_mix_poly_coeffs")

static const char* evaluate_any_impl(bx_ctx* c, bx_buf coeffs, size_t poly_count, bx_buf which, bx_buf xs, bx_buf out, bool bitrev,
                                     const char* name) {
    BX_REQUIRE(c, poly_count > 0 && coeffs.len % poly_count == 0, "batch_evaluate_any: coeffs.len not a multiple of poly_count");
    size_t evals = which.len;
    BX_REQUIRE(c, xs.len == 4 * evals && out.len == 4 * evals, "batch_evaluate_any: xs/out must hold one ext elem per eval");
    BX_ENTER(c);
    size_t poly_size = coeffs.len / poly_count;
    size_t seg_elems = (size_t)EV_T * EV_K;
    int brev_log = 0;
    if (bitrev) {
        BX_REQUIRE(c, is_pow2(poly_size) && poly_size >= seg_elems, "batch_evaluate_any_bitrev: polynomial size must be a power of two >= 2^15");
        brev_log = ilog2(poly_size);
    }
    OpScope op(c, name, 4.0 * (double)(poly_size * evals));
    if (!evals) return nullptr;
    size_t segs = (poly_size + seg_elems - 1) / seg_elems;
    BX_REQUIRE(c, evals <= 65535, "batch_evaluate_any: more than 65535 evaluations in one call");
    BX_TRY(ensure_scratch(c, 4 * evals * segs));
    if (c->eval_x4 && poly_size % seg_elems == 0 && ((uintptr_t)coeffs.dptr & 15u) == 0 && segs <= 512) {
        const uint32_t stride = EVT_SW + 4 * (uint32_t)segs;  // a multiple of four words: every table entry stays 16-byte aligned
        const size_t part_words = (4 * evals * segs + 3) & ~(size_t)3;
        BX_TRY(ensure_scratch(c, part_words + (size_t)stride * evals));
        uint32_t* tables = c->d_scratch + part_words;
        hipLaunchKernelGGL(eval_tables_kernel, dim3((unsigned)evals), dim3(EV_T), 0, c->stream, (const uint32_t*)xs.dptr, tables, (uint32_t)segs,
                           stride, brev_log, (const uint32_t*)nullptr, 0);
        BX_LAUNCH_CHECK(c);
        hipLaunchKernelGGL(eval_partial_x4_kernel, dim3((unsigned)segs, (unsigned)evals), dim3(EV_T), 0, c->stream,
                           (const uint32_t*)coeffs.dptr, poly_size, (const uint32_t*)which.dptr, (const uint32_t*)tables, stride, c->d_scratch,
                           (uint32_t)segs, (const unsigned long long*)nullptr);
    } else {
        hipLaunchKernelGGL(eval_partial_kernel, dim3((unsigned)segs, (unsigned)evals), dim3(EV_T), 0, c->stream,
                           (const uint32_t*)coeffs.dptr, poly_size, (const uint32_t*)which.dptr, (const uint32_t*)xs.dptr,
                           c->d_scratch, (uint32_t)segs, brev_log);
    }
    BX_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(eval_final_kernel, dim3((unsigned)((evals + 63) / 64)), dim3(64), 0, c->stream, c->d_scratch,
                       (uint32_t)segs, (uint32_t*)out.dptr, (uint32_t)evals);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
extern "C" const char* bx_batch_evaluate_any(bx_ctx* c, bx_buf coeffs, size_t poly_count, bx_buf which, bx_buf xs, bx_buf out) try {
    if (!c) return "bx_batch_evaluate_any: null ctx";
    return evaluate_any_impl(c, coeffs, poly_count, which, xs, out, false, "batch_evaluate_any");
} BX_ABI_CATCH(c, "bx_batch_evaluate_any")
extern "C" const char* bx_batch_evaluate_any_bitrev(bx_ctx* c, bx_buf coeffs, size_t poly_count, bx_buf which, bx_buf xs, bx_buf out) try {
    if (!c) return "bx_batch_evaluate_any_bitrev: null ctx";
    return evaluate_any_impl(c, coeffs, poly_count, which, xs, out, true, "batch_evaluate_any");
} BX_ABI_CATCH(c, "bx_batch_evaluate_any_bitrev")

// Extension: the tap evaluations of several coefficient buffers in ONE launch set (the DEEP step evaluates columns of four
// groups): evaluation i reads the poly_size coefficients at device address poly_ptrs[i] (bit-reversed storage when flags[i] & 1).
extern "C" const char* bx_batch_evaluate_ptrs(bx_ctx* c, bx_buf poly_ptrs, bx_buf flags, size_t poly_size, bx_buf xs, bx_buf out) try {
    if (!c) return "bx_batch_evaluate_ptrs: null ctx";
    const size_t evals = flags.len, seg_elems = (size_t)EV_T * EV_K;
    BX_REQUIRE(c, poly_ptrs.len == 2 * evals && xs.len == 4 * evals && out.len == 4 * evals, "batch_evaluate_ptrs: one pointer (two words), one flag, one point and one result per evaluation");
    BX_REQUIRE(c, is_pow2(poly_size) && poly_size >= seg_elems && poly_size / seg_elems <= 512, "batch_evaluate_ptrs: polynomial size must be a power of two in [2^15, 2^24]");
    BX_REQUIRE(c, evals <= 65535 && ((uintptr_t)poly_ptrs.dptr & 7u) == 0, "batch_evaluate_ptrs: at most 65535 evaluations, pointers 8-byte aligned");
    BX_ENTER(c);
    OpScope op(c, "batch_evaluate_any", 4.0 * (double)(poly_size * evals));
    if (!evals) return nullptr;
    const size_t segs = poly_size / seg_elems;
    const uint32_t stride = EVT_SW + 4 * (uint32_t)segs;
    const size_t part_words = (4 * evals * segs + 3) & ~(size_t)3;
    BX_TRY(ensure_scratch(c, part_words + (size_t)stride * evals));
    uint32_t* tables = c->d_scratch + part_words;
    hipLaunchKernelGGL(eval_tables_kernel, dim3((unsigned)evals), dim3(EV_T), 0, c->stream, (const uint32_t*)xs.dptr, tables, (uint32_t)segs, stride, 0,
                       (const uint32_t*)flags.dptr, ilog2(poly_size));
    BX_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(eval_partial_x4_kernel, dim3((unsigned)segs, (unsigned)evals), dim3(EV_T), 0, c->stream, (const uint32_t*)nullptr, poly_size,
                       (const uint32_t*)nullptr, (const uint32_t*)tables, stride, c->d_scratch, (uint32_t)segs, (const unsigned long long*)poly_ptrs.dptr);
    BX_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(eval_final_kernel, dim3((unsigned)((evals + 63) / 64)), dim3(64), 0, c->stream, c->d_scratch, (uint32_t)segs, (uint32_t*)out.dptr,
                       (uint32_t)evals);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_batch_evaluate_ptrs")

extern "C" const char* bx_eltwise_add_elem(bx_ctx* c, bx_buf out, bx_buf a, bx_buf b) try {
    if (!c) return "bx_eltwise_add_elem: null ctx";
    BX_REQUIRE(c, out.len == a.len && a.len == b.len, "eltwise_add_elem: length mismatch");
    BX_ENTER(c);
    OpScope op(c, "eltwise_add_elem", 12.0 * (double)out.len);
    hipLaunchKernelGGL(eltwise_add_kernel, dim3(grid1d(out.len)), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)a.dptr, (const uint32_t*)b.dptr, out.len);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_add_elem")
extern "C" const char* bx_eltwise_mul_factor(bx_ctx* c, bx_buf io, uint32_t factor_mont) try {
    if (!c) return "bx_eltwise_mul_factor: null ctx";
    BX_REQUIRE(c, factor_mont < P, "eltwise_mul_factor: factor is not a canonical Montgomery word");
    BX_ENTER(c);
    if (!io.len) return nullptr;
    OpScope op(c, "eltwise_mul_factor", 8.0 * (double)io.len);
    hipLaunchKernelGGL(eltwise_mul_factor_kernel, dim3(grid1d(io.len)), dim3(256), 0, c->stream, (uint32_t*)io.dptr, factor_mont, io.len);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_mul_factor")
extern "C" const char* bx_eltwise_copy_elem(bx_ctx* c, bx_buf out, bx_buf in) try {
    if (!c) return "bx_eltwise_copy_elem: null ctx";
    BX_REQUIRE(c, out.len == in.len, "eltwise_copy_elem: length mismatch");
    BX_ENTER(c);
    OpScope op(c, "eltwise_copy_elem", 8.0 * (double)out.len);
    BX_HIP(c, hipMemcpyAsync(out.dptr, in.dptr, out.len * 4, hipMemcpyDeviceToDevice, c->stream));
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_copy_elem")
extern "C" const char* bx_eltwise_zeroize_elem(bx_ctx* c, bx_buf io) try {
    if (!c) return "bx_eltwise_zeroize_elem: null ctx";
    BX_ENTER(c);
    OpScope op(c, "eltwise_zeroize_elem", 8.0 * (double)io.len);
    hipLaunchKernelGGL(eltwise_zeroize_kernel, dim3(grid1d(io.len)), dim3(256), 0, c->stream, (uint32_t*)io.dptr, io.len);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_zeroize_elem")
extern "C" const char* bx_eltwise_sum_extelem(bx_ctx* c, bx_buf out, bx_buf in) try {
    if (!c) return "bx_eltwise_sum_extelem: null ctx";
    BX_REQUIRE(c, out.len % 4 == 0 && out.len > 0 && in.len % out.len == 0, "eltwise_sum_extelem: in.len not a multiple of out.len");
    BX_ENTER(c);
    size_t count = out.len / 4, to_add = in.len / out.len;
    OpScope op(c, "eltwise_sum_extelem", 4.0 * (double)(in.len + out.len));
    hipLaunchKernelGGL(eltwise_sum_ext_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream,
                       (uint32_t*)out.dptr, (const uint32_t*)in.dptr, count, to_add);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_eltwise_sum_extelem")
extern "C" const char* bx_gather_sample(bx_ctx* c, bx_buf dst, bx_buf src, size_t idx, size_t size, size_t stride) try {
    if (!c) return "bx_gather_sample: null ctx";
    BX_REQUIRE(c, dst.len >= size, "gather_sample: dst too small");
    BX_REQUIRE(c, size == 0 || (idx < src.len && mul_le(size - 1, stride, src.len - 1 - idx)), "gather_sample: source index out of range");
    if (!size) return nullptr;
    // Small gathers are queued and launched together by the next call that touches the ctx (ctx.hpp, "Deferred ... gather_sample"):
    // the openings of a proof are ~5 000 of them.  Stream order as the caller sees it is kept: anything that could observe the
    // difference flushes the queue first, and so does a gather that touches memory a queued one writes.
    // (not on an adopted stream, bx_set_stream: its owner enqueues work of its own there without going through this library)
    if (c->gather_defer && c->stream == c->own_stream && !c->profile && trace_level() == 0 && size <= 4096 && stride <= 0xffffffffu) {
        const uintptr_t d0 = (uintptr_t)dst.dptr, d1 = d0 + 4 * size;
        const uintptr_t s0 = (uintptr_t)((const uint32_t*)src.dptr + idx), s1 = s0 + 4 * ((size - 1) * stride + 1);
        if (c->gq_n && ((s0 < c->gq_dst_hi && c->gq_dst_lo < s1) || (d0 < c->gq_dst_hi && c->gq_dst_lo < d1) || (d0 < c->gq_src_hi && c->gq_src_lo < d1)))
            BX_TRY(gather_flush(c));  // reads or overwrites what a queued gather writes, or overwrites what one reads
        if (c->gq_n == bx_ctx::GQ_MAX) BX_TRY(gather_flush(c));
        if (!c->gq_n) c->gq_dst_lo = c->gq_src_lo = ~(uintptr_t)0, c->gq_dst_hi = c->gq_src_hi = 0;
        const uint32_t w[8] = {(uint32_t)d0, (uint32_t)((unsigned long long)d0 >> 32), (uint32_t)s0, (uint32_t)((unsigned long long)s0 >> 32),
                               (uint32_t)size, (uint32_t)stride, 0u, 0u};
        c->gq.insert(c->gq.end(), w, w + 8);
        c->gq_n += 1;
        c->gq_dst_lo = std::min(c->gq_dst_lo, d0), c->gq_dst_hi = std::max(c->gq_dst_hi, d1);
        c->gq_src_lo = std::min(c->gq_src_lo, s0), c->gq_src_hi = std::max(c->gq_src_hi, s1);
        return nullptr;
    }
    BX_ENTER(c);
    hipLaunchKernelGGL(gather_sample_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)dst.dptr,
                       (const uint32_t*)src.dptr, idx, size, stride);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_gather_sample")

namespace bx {
// Launch the queued gathers: the descriptors go up through the pinned ring (h2d_staged: no wait) into d_gq, one workgroup each.
// The queue is emptied FIRST: everything called from here (h2d_staged, and stream_wait when the ring wraps) checks it too.
const char* gather_flush(bx_ctx* c) {
    const size_t n = c->gq_n;
    if (!n) return nullptr;
    std::vector<uint32_t> q;
    q.swap(c->gq);
    c->gq_n = 0;
    BX_HIP(c, hipSetDevice(c->device));
    if (!c->d_gq) BX_HIP(c, hipMalloc((void**)&c->d_gq, bx_ctx::GQ_MAX * 32));
    BX_TRY(h2d_staged(c, bx_buf{c->d_gq, 8 * bx_ctx::GQ_MAX}, q.data(), 8 * n));
    hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)n), dim3(64), 0, c->stream, (const uint4*)c->d_gq);
    BX_LAUNCH_CHECK(c);
    q.clear();
    c->gq.swap(q);  // keep the capacity
    return nullptr;
}
}  // namespace bx

// in-place division of the AoS ext array `arr` (n entries) by (x - z); `scratch` has room for every level's chunk values
static const char* divide_rec(bx_ctx* c, uint32_t* arr, size_t n, Fp4 z, uint32_t* scratch, uint32_t* rem) {
    if (n <= DIV_DIRECT) {
        // one workgroup: thread t owns a run of entries; the "chunk" multiplier of the kernel is z itself here
        unsigned nt = n >= 1024 ? 1024 : 64;
        hipLaunchKernelGGL(div_scan_kernel, dim3(1), dim3(nt), nt * 32, c->stream, arr, n, z, rem);
        BX_LAUNCH_CHECK(c);
        return nullptr;
    }
    const size_t chunks = (n + DIV_L - 1) / DIV_L;
    hipLaunchKernelGGL(div_local_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t*)arr, n, z, scratch,
                       chunks);
    BX_LAUNCH_CHECK(c);
    BX_TRY(divide_rec(c, scratch, chunks, f4_pow(z, DIV_L), scratch + 4 * chunks, rem));  // chunk values -> carries entering the chunks
    hipLaunchKernelGGL(div_apply_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, c->stream, arr, n, z, (const uint32_t*)scratch,
                       chunks);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
static size_t scan_scratch_words(size_t n, size_t L, size_t direct, size_t count) {
    size_t words = 8;
    while (n > direct) {
        n = (n + L - 1) / L;
        words += 4 * n * count;
    }
    return words;
}

extern "C" const char* bx_poly_divide(bx_ctx* c, bx_buf poly, const uint32_t z[4], bx_buf rem_out) try {
    if (!c) return "bx_poly_divide: null ctx";
    BX_REQUIRE(c, poly.len % 4 == 0 && rem_out.len >= 4, "poly_divide: poly must be AoS ext, remainder buffer >= 4 words");
    BX_ENTER(c);
    size_t size = poly.len / 4;
    if (!size) return nullptr;
    OpScope op(c, "poly_divide", 8.0 * (double)poly.len);
    if (c->scan_lookback && ((uintptr_t)poly.dptr & 15u) == 0 && ((uintptr_t)rem_out.dptr & 15u) == 0)
        return poly_divide_lookback(c, (uint32_t*)poly.dptr, size, 1, z, (uint32_t*)rem_out.dptr, nullptr);
    BX_TRY(ensure_scratch(c, scan_scratch_words(size, DIV_L, DIV_DIRECT, 1)));
    return divide_rec(c, (uint32_t*)poly.dptr, size, host4(z), c->d_scratch, (uint32_t*)rem_out.dptr);
} BX_ABI_CATCH(c, "bx_poly_divide")

// exclusive running products, in place, of `count` sequences of n entries (sequence k at arr + 4 * k * stride)
static const char* excl_scan_rec(bx_ctx* c, uint32_t* arr, size_t n, size_t stride, size_t count, uint32_t* scratch) {
    if (n <= PP_DIRECT) {
        unsigned nt = n >= 1024 ? 1024 : 64;
        hipLaunchKernelGGL(pp_scan_kernel, dim3((unsigned)count), dim3(nt), nt * 16, c->stream, arr, n, stride);
        BX_LAUNCH_CHECK(c);
        return nullptr;
    }
    const size_t chunks = (n + PP_L - 1) / PP_L;
    hipLaunchKernelGGL(pp_local_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)count), dim3(256), 0, c->stream, (const uint32_t*)arr, n,
                       stride, scratch, chunks, chunks);
    BX_LAUNCH_CHECK(c);
    BX_TRY(excl_scan_rec(c, scratch, chunks, chunks, count, scratch + 4 * chunks * count));
    hipLaunchKernelGGL(pp_apply_excl_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)count), dim3(256), 0, c->stream, arr, n, stride,
                       (const uint32_t*)scratch, chunks, chunks);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

extern "C" const char* bx_batch_prefix_products(bx_ctx* c, bx_buf io, size_t count) try {
    if (!c) return "bx_batch_prefix_products: null ctx";
    BX_REQUIRE(c, io.len % 4 == 0, "prefix_products: buffer must hold AoS ext elements");
    BX_REQUIRE(c, count >= 1 && (io.len / 4) % count == 0, "prefix_products: the buffer does not split into `count` equal sequences");
    BX_REQUIRE(c, count <= 65535, "prefix_products: too many sequences");
    BX_ENTER(c);
    size_t n = io.len / 4 / count;
    if (n < 2) return nullptr;
    OpScope op(c, "prefix_products", 8.0 * (double)io.len);
    if (c->scan_lookback && ((uintptr_t)io.dptr & 15u) == 0) return prefix_products_lookback(c, (uint32_t*)io.dptr, n, count);
    // chunk products -> exclusive scan of them (recursively, PP_L per level) -> inclusive replay of every chunk with its carry
    size_t chunks = (n + PP_L - 1) / PP_L;
    BX_TRY(ensure_scratch(c, 4 * chunks * count + scan_scratch_words(chunks, PP_L, PP_DIRECT, count)));
    hipLaunchKernelGGL(pp_local_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)count), dim3(256), 0, c->stream,
                       (const uint32_t*)io.dptr, n, n, c->d_scratch, chunks, chunks);
    BX_LAUNCH_CHECK(c);
    BX_TRY(excl_scan_rec(c, c->d_scratch, chunks, chunks, count, c->d_scratch + 4 * chunks * count));
    hipLaunchKernelGGL(pp_apply_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)count), dim3(256), 0, c->stream, (uint32_t*)io.dptr, n,
                       c->d_scratch, chunks);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_batch_prefix_products")
extern "C" const char* bx_prefix_products(bx_ctx* c, bx_buf io) try {
    if (!c) return "bx_prefix_products: null ctx";
    return bx_batch_prefix_products(c, io, 1);
} BX_ABI_CATCH(c, "bx_prefix_products")

extern "C" const char* bx_scatter(bx_ctx* c, bx_buf into, bx_buf index, bx_buf offsets, bx_buf values) try {
    if (!c) return "bx_scatter: null ctx";
    BX_REQUIRE(c, offsets.len == values.len, "scatter: offsets and values must have the same length");
    BX_ENTER(c);
    if (index.len < 2 || offsets.len == 0) return nullptr;
    // asynchronous like every other entry point: range errors are raised on the device and reported by the next
    // blocking call on this ctx (bx_d2h / bx_sync)
    OpScope op(c, "scatter", 12.0 * (double)offsets.len);
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((offsets.len + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)into.dptr,
                       (const uint32_t*)index.dptr, index.len, (const uint32_t*)offsets.dptr, (const uint32_t*)values.dptr, offsets.len,
                       into.len, c->h_flag);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_scatter")
