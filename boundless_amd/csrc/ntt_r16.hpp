// ntt_r16.hpp — register-resident radix-16 NTT passes for gfx950 (included by ntt.hip).
//
// The measured bound of this path is VALU issue: a radix-2 butterfly is ~11 integer instructions and
// everything else (LDS traffic, index arithmetic, barriers) is overhead on the same issue port.  The v1 kernels did one
// stage per LDS round trip (2 reads + 2 writes + twiddle read + index math per butterfly).  Here every thread keeps
// 16 elements in VGPRs and runs up to four consecutive stages on them ("step"); LDS is touched only to regroup
// elements between steps, the first step is fed straight from global memory and the last one stores straight back:
// for a 2^12 sub-transform that is 2 LDS writes + 2 LDS reads per element instead of 24 + 24.
//
// Tile = 2^lrows rows x 2^lt adjacent columns, `lr` <= lrows radix-2 stages along the row index:
//   pass A (contiguous blocks)  lt = 0, rows are consecutive words, several 2^lr blocks per tile when lr < lrows;
//   pass B (across blocks)      rows are 2^row_shift words apart, T = 2^lt adjacent positions side by side.
// Step (s0, K) runs stages s0+1 .. s0+K: row = hi * 2^(s0+K) + mid * 2^s0 + lo; a "unit" (hi, lo, t) owns the 2^K
// elements mid = 0..2^K-1; a thread owns 16 / 2^K units, enumerated so that consecutive lanes take consecutive
// (t, lo) — global accesses coalesce and LDS accesses (layout i + (i >> 4)) are conflict-free or 2-way.
// Stage-s twiddles w_{2^s}^e come straight from the global stage table at [2^(s-1) + e]: every entry is used by exactly
// one unit per tile, so staging the table in LDS would cost more LDS traffic than the data itself; consecutive lanes read
// consecutive entries (or broadcast), and the table (<= 32 KiB) lives in L1/L2.
#pragma once
#include "ctx.hpp"

namespace bx {

struct R16Args {
    uint32_t* out;
    const uint32_t* in;
    const uint32_t* tw;     // stage table (forward or inverse roots)
    const uint32_t* twist;  // pass A only: per-element factor (nullptr = none)
    const uint32_t* post = nullptr;  // pass A inverse only: per-position factor applied on the final store
    uint32_t scale;         // pass A inverse without twist: 1/M
    int lr, lrows, lt;
    int expand;     // pass A forward: load shift (out[i] = in[i >> expand])
    int row_shift;  // log2 of the global distance between consecutive rows (0 for pass A)
    uint32_t tile_stride;  // words between consecutive tiles (pass A: tile size, pass B: T)
    size_t in_col_stride, out_col_stride;
    uint32_t tiles;
    uint32_t cols;  // number of columns (polynomials) in the launch
    uint32_t cpw;   // columns per workgroup (multi-column pass A)
};

__device__ __forceinline__ uint32_t lds_phys(uint32_t i) { return i + (i >> 4); }

// raw buffer descriptor over `bytes` bytes from a wave-uniform pointer (gfx950: DATA_FORMAT word 0x00020000, stride 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t col_rsrc(const uint32_t* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p), 0, bytes, 0x00020000);
}

// Butterflies.  Canonical in, canonical out by default; two conditional subtractions per butterfly can be dropped where the
// value is consumed by a twiddle product next (a lazy Montgomery product accepts operands < 2P: 2P * P < 2.42 P^2):
//   DIF (inverse):  b' = (a - b) * w   — the difference feeds the product directly: a - b + P in (0, 2P), no v_min;
//   DIT (forward):  an output that is a "b" (multiplied) operand of the next stage of the same register-resident step may
//                   stay in [0, 2P): a + t without the subtraction, a - t + P without the v_min.  LA / LB say which.
template <bool INV, bool LA = false, bool LB = false>
__device__ __forceinline__ void bfly_tw(uint32_t& a, uint32_t& b, uint32_t w) {
    if (!INV) {
        const uint32_t t = fp_mul(b, w);  // b < 2P allowed
        const uint32_t d = a - t;
        b = LB ? d + P : umin(d, d + P);
        const uint32_t u = a + t;
        a = LA ? u : umin(u, u - P);
    } else {
        const uint32_t u = fp_add(a, b);
        b = fp_mul(a - b + P, w);
        a = u;
    }
}
template <bool LA = false, bool LB = false>
__device__ __forceinline__ void bfly_one(uint32_t& a, uint32_t& b) {
    const uint32_t u = a + b, d = a - b;
    a = LA ? u : umin(u, u - P);
    b = LB ? d + P : umin(d, d + P);
}
// DIT stage with half = 2^(k-1) inside a K-stage step: may the outputs of butterfly (j, j + half) stay lazy?  Both outputs
// are "b" operands of stage k+1 iff bit 2*half of j is set.  With a trivial (w = 1) first twiddle group (S0ZERO) the b
// operand of a next-stage butterfly with jj' = 0 is added, not multiplied, and must be canonical: that is output j of a
// butterfly that is itself in the trivial group (jj = 0); output j + half never lands there.
#define BX_DIT_LAZY_B(last, j, half) (!(last) && (((j) & (2 * (half))) != 0))

// the K stages of one step on the thread's 16 registers
template <int K, bool INV, bool S0ZERO, int SKIP>
__device__ __forceinline__ void step_compute(uint32_t (&x)[16], const uint32_t* __restrict__ ltw, int s0, int lt, uint32_t tid,
                                             uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        const uint32_t rl = (tid + nt * w) >> lt;
        const uint32_t lo = S0ZERO ? 0u : (rl & ((1u << s0) - 1u));
        uint32_t* xu = &x[w * E];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int k = INV ? K - kk : kk + 1;  // DIT ascends, DIF descends
            if (k <= SKIP) continue;
            const int half = 1 << (k - 1);
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                if (S0ZERO && jj == 0) {
#pragma unroll
                    for (int j = 0; j < E; j += 2 * half) {
                        if (!INV && BX_DIT_LAZY_B(kk == K - 1, j, half)) bfly_one<false, true>(xu[j], xu[j + half]);
                        else bfly_one(xu[j], xu[j + half]);
                    }
                } else {
                    const uint32_t wv = ltw[(1u << (s0 + k - 1)) + lo + ((uint32_t)jj << s0)];
#pragma unroll
                    for (int j = jj; j < E; j += 2 * half) {
                        if (!INV && BX_DIT_LAZY_B(kk == K - 1, j, half)) bfly_tw<INV, true, true>(xu[j], xu[j + half], wv);
                        else bfly_tw<INV>(xu[j], xu[j + half], wv);
                    }
                }
            }
        }
    }
}

// Twiddle prefetch: the (2^K - 1) stage twiddles of each unit, fetched into registers ahead of the LDS regrouping that
// precedes the step, so the L2 round trip overlaps the barrier / LDS traffic instead of stalling the butterflies.
// Slot (w, half - 1 + jj) of tw[] holds the twiddle of stage k (half = 2^(k-1)), butterfly jj.
template <int K, bool S0ZERO, int SKIP>
__device__ __forceinline__ void tw_load(uint32_t (&tw)[16], const uint32_t* __restrict__ ltw, int s0, int lt, uint32_t tid, uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        const uint32_t rl = (tid + nt * w) >> lt;
        const uint32_t lo = S0ZERO ? 0u : (rl & ((1u << s0) - 1u));
#pragma unroll
        for (int k = 1; k <= K; ++k) {
            if (k <= SKIP) continue;
            const int half = 1 << (k - 1);
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                if (S0ZERO && jj == 0) continue;
                tw[w * E + half - 1 + jj] = ltw[(1u << (s0 + k - 1)) + lo + ((uint32_t)jj << s0)];
            }
        }
    }
}
// IN_LAZY (forward only): the registers arrive in [0, 2P) (the previous step left its last stage uncorrected); the "a" operands of
// the first stage are reduced here, where they are consumed — the "b" operands feed a product and need nothing.  OUT_LAZY
// (forward only): the last stage leaves both outputs in [0, 2P) for a consumer that reduces on use (the next step, or the twist
// product of pass A).  Together: 8 conditional subtractions per step boundary instead of 16.
template <int K, bool INV, bool S0ZERO, int SKIP, bool IN_LAZY = false, bool OUT_LAZY = false>
__device__ __forceinline__ void step_compute_tw(uint32_t (&x)[16], const uint32_t (&tw)[16]) {
    static_assert(!(INV && (IN_LAZY || OUT_LAZY)), "lazy step boundaries are implemented for the forward (DIT) butterflies only");
    static_assert(!(IN_LAZY && S0ZERO), "a step with trivial first twiddles adds its b operands: they must arrive canonical");
    constexpr int U = 16 >> K, E = 1 << K;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        uint32_t* xu = &x[w * E];
        if (IN_LAZY) {
#pragma unroll
            for (int j = 0; j < E; j += 2) xu[j] = fp_reduce(xu[j]);
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int k = INV ? K - kk : kk + 1;
            if (k <= SKIP) continue;
            const int half = 1 << (k - 1);
            const bool last = kk == K - 1;
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                if (S0ZERO && jj == 0) {
#pragma unroll
                    for (int j = 0; j < E; j += 2 * half) {
                        if (!INV && OUT_LAZY && last) bfly_one<true, true>(xu[j], xu[j + half]);
                        else if (!INV && BX_DIT_LAZY_B(last, j, half)) bfly_one<false, true>(xu[j], xu[j + half]);
                        else bfly_one(xu[j], xu[j + half]);
                    }
                } else {
                    const uint32_t wv = tw[w * E + half - 1 + jj];
#pragma unroll
                    for (int j = jj; j < E; j += 2 * half) {
                        if (!INV && ((OUT_LAZY && last) || BX_DIT_LAZY_B(last, j, half))) bfly_tw<INV, true, true>(xu[j], xu[j + half], wv);
                        else bfly_tw<INV>(xu[j], xu[j + half], wv);
                    }
                }
            }
        }
    }
}

// row/col of register (w, mid) for step (s0, K)
template <int K>
__device__ __forceinline__ void unit_coords(int w, int s0, int lt, uint32_t tid, uint32_t nt, uint32_t& base_row, uint32_t& t) {
    const uint32_t q = tid + nt * w;
    t = q & ((1u << lt) - 1u);
    const uint32_t rl = q >> lt;
    const uint32_t lo = rl & ((1u << s0) - 1u);
    const uint32_t hi = rl >> s0;
    base_row = (hi << (s0 + K)) | lo;
}

// LDS index of register (w, mid): element i = ib + mid * 2^(s0+lt) with ib = (base_row << lt) | t.  When the element
// stride is a multiple of 16 the padding term splits, phys(i) = phys(ib) + mid * (stride + stride/16): one add per access.
template <int K>
__device__ __forceinline__ void lds_put(const uint32_t (&x)[16], uint32_t* __restrict__ s, int s0, int lt, uint32_t tid, uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
    const int sh = s0 + lt;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        uint32_t base_row, t;
        unit_coords<K>(w, s0, lt, tid, nt, base_row, t);
        const uint32_t ib = (base_row << lt) | t;
        if (sh >= 4) {
            const uint32_t pb = lds_phys(ib), step = (1u << sh) + (1u << (sh - 4));
#pragma unroll
            for (int mid = 0; mid < E; ++mid) s[pb + (uint32_t)mid * step] = x[w * E + mid];
        } else {
#pragma unroll
            for (int mid = 0; mid < E; ++mid) s[lds_phys(ib + ((uint32_t)mid << sh))] = x[w * E + mid];
        }
    }
}
template <int K>
__device__ __forceinline__ void lds_get(uint32_t (&x)[16], const uint32_t* __restrict__ s, int s0, int lt, uint32_t tid, uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
    const int sh = s0 + lt;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        uint32_t base_row, t;
        unit_coords<K>(w, s0, lt, tid, nt, base_row, t);
        const uint32_t ib = (base_row << lt) | t;
        if (sh >= 4) {
            const uint32_t pb = lds_phys(ib), step = (1u << sh) + (1u << (sh - 4));
#pragma unroll
            for (int mid = 0; mid < E; ++mid) x[w * E + mid] = s[pb + (uint32_t)mid * step];
        } else {
#pragma unroll
            for (int mid = 0; mid < E; ++mid) x[w * E + mid] = s[lds_phys(ib + ((uint32_t)mid << sh))];
        }
    }
}

// global <-> registers for step (s0, K).  PASS_A: rows are consecutive words (lt == 0) and the twist/scale/expand apply.
template <int K, bool INV, bool PASS_A>
__device__ __forceinline__ void glb_get(uint32_t (&x)[16], const R16Args& a, const uint32_t* __restrict__ src, size_t tile_off, int s0,
                                        uint32_t tid, uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
    if (PASS_A && !INV && K == 4 && s0 == 0 && (a.expand == 0 || a.expand == 2)) {
        // 16 consecutive words per thread: wide loads (one dwordx4 per thread when expanding by 4)
        const size_t g = tile_off + (size_t)tid * 16;
        if (a.expand == 2) {
            uint4 v = *reinterpret_cast<const uint4*>(src + (g >> 2));
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = vv[i >> 2];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = reinterpret_cast<const uint4*>(src + g)[i];
                x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
            }
        }
        return;
    }
    const int rsh = a.row_shift + s0;
    if (!PASS_A) {
        // pass B: buffer loads — ONE per-lane byte offset per unit in voffset, the row of register `mid` in the scalar offset, so
        // a load needs no vector address arithmetic (the 64-bit per-element form cost ~5 multiply-class VALU instructions per
        // element: 13 % of the pass's issue cycles).  The descriptor covers the rest of the column from the tile's first word.
        const __amdgpu_buffer_rsrc_t rs = col_rsrc(src + tile_off, (uint32_t)((a.in_col_stride - tile_off) * 4));
#pragma unroll
        for (int w = 0; w < U; ++w) {
            uint32_t base_row, t;
            unit_coords<K>(w, s0, a.lt, tid, nt, base_row, t);
            const uint32_t vo = ((base_row << a.row_shift) + t) << 2;
#pragma unroll
            for (int mid = 0; mid < E; ++mid)
                x[w * E + mid] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (uint32_t)mid << (rsh + 2), 0);
        }
        return;
    }
    // 32-bit offsets from wave-uniform bases keep one VGPR per address
    const uint32_t* sp = src + (tile_off >> a.expand);
#pragma unroll
    for (int w = 0; w < U; ++w) {
        uint32_t base_row, t;
        unit_coords<K>(w, s0, a.lt, tid, nt, base_row, t);
        const uint32_t ob = (base_row << a.row_shift) + t;
#pragma unroll
        for (int mid = 0; mid < E; ++mid) {
            const uint32_t off = ob + ((uint32_t)mid << rsh);
            x[w * E + mid] = sp[off >> a.expand];
        }
    }
    if (PASS_A && INV) {
        if (a.twist) {
            const uint32_t* tp = a.twist + tile_off;
#pragma unroll
            for (int w = 0; w < U; ++w) {
                uint32_t base_row, t;
                unit_coords<K>(w, s0, a.lt, tid, nt, base_row, t);
                const uint32_t ob = (base_row << a.row_shift) + t;
#pragma unroll
                for (int mid = 0; mid < E; ++mid) x[w * E + mid] = fp_mul(x[w * E + mid], tp[ob + ((uint32_t)mid << rsh)]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = fp_mul(x[i], a.scale);
        }
    }
}
template <int K, bool INV, bool PASS_A>
__device__ __forceinline__ void glb_put(const uint32_t (&x)[16], const R16Args& a, uint32_t* __restrict__ dst, size_t tile_off, int s0,
                                        uint32_t tid, uint32_t nt) {
    constexpr int U = 16 >> K, E = 1 << K;
    if (PASS_A && INV && K == 4 && s0 == 0) {
        uint4* o = reinterpret_cast<uint4*>(dst + tile_off + (size_t)tid * 16);
        if (a.post) {  // fused zk_shift: the factor of position p of the column
            const uint4* f = reinterpret_cast<const uint4*>(a.post + tile_off + (size_t)tid * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 g = f[i];
                o[i] = make_uint4(fp_mul(x[4 * i], g.x), fp_mul(x[4 * i + 1], g.y), fp_mul(x[4 * i + 2], g.z), fp_mul(x[4 * i + 3], g.w));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = make_uint4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
        return;
    }
    uint32_t* dp = dst + tile_off;
    const int rsh = a.row_shift + s0;
    if (!PASS_A) {  // buffer stores: per-lane offset per unit + scalar row offset (see glb_get)
        const __amdgpu_buffer_rsrc_t rs = col_rsrc(dp, (uint32_t)((a.out_col_stride - tile_off) * 4));
#pragma unroll
        for (int w = 0; w < U; ++w) {
            uint32_t base_row, t;
            unit_coords<K>(w, s0, a.lt, tid, nt, base_row, t);
            const uint32_t vo = ((base_row << a.row_shift) + t) << 2;
#pragma unroll
            for (int mid = 0; mid < E; ++mid) __builtin_amdgcn_raw_buffer_store_b32(x[w * E + mid], rs, vo, (uint32_t)mid << (rsh + 2), 0);
        }
        return;
    }
    const bool tw = PASS_A && !INV && a.twist != nullptr;
    const uint32_t* tp = tw ? a.twist + tile_off : nullptr;
#pragma unroll
    for (int w = 0; w < U; ++w) {
        uint32_t base_row, t;
        unit_coords<K>(w, s0, a.lt, tid, nt, base_row, t);
        const uint32_t ob = (base_row << a.row_shift) + t;
        if (tw) {
            uint32_t f[E];
#pragma unroll
            for (int mid = 0; mid < E; ++mid) f[mid] = tp[ob + ((uint32_t)mid << rsh)];
#pragma unroll
            for (int mid = 0; mid < E; ++mid) dp[ob + ((uint32_t)mid << rsh)] = fp_mul(x[w * E + mid], f[mid]);
        } else {
#pragma unroll
            for (int mid = 0; mid < E; ++mid) dp[ob + ((uint32_t)mid << rsh)] = x[w * E + mid];
        }
    }
}

// Forward (DIT) order: step 0 (s0 = 0, from global, trivial twiddles, optional skipped stages), then the higher steps.
// `twn` receives the next step's twiddles (prefetched before the regrouping).
template <int KN>
__device__ __forceinline__ void prefetch_next(uint32_t (&twn)[16], const uint32_t* ltw, int s0n, int lt, uint32_t tid, uint32_t nt) {
    tw_load<KN, false, 0>(twn, ltw, s0n, lt, tid, nt);
}
__device__ __forceinline__ void prefetch_dispatch(uint32_t (&twn)[16], const uint32_t* ltw, int lr, int s0n, int lt, uint32_t tid,
                                                  uint32_t nt) {
    const int kn = lr - s0n < 4 ? lr - s0n : 4;
    switch (kn) {
        case 4: prefetch_next<4>(twn, ltw, s0n, lt, tid, nt); break;
        case 3: prefetch_next<3>(twn, ltw, s0n, lt, tid, nt); break;
        case 2: prefetch_next<2>(twn, ltw, s0n, lt, tid, nt); break;
        default: prefetch_next<1>(twn, ltw, s0n, lt, tid, nt); break;
    }
}
template <int K, bool PASS_A, int SKIP>
__device__ __forceinline__ void fwd_first(uint32_t (&x)[16], uint32_t (&twn)[16], const R16Args& a, uint32_t* s, const uint32_t* ltw,
                                          const uint32_t* src, uint32_t* dst, size_t tile_off, bool only, uint32_t tid, uint32_t nt) {
    uint32_t tw0[16];
    tw_load<K, true, SKIP>(tw0, ltw, 0, a.lt, tid, nt);
    glb_get<K, false, PASS_A>(x, a, src, tile_off, 0, tid, nt);
    if (!only) prefetch_dispatch(twn, ltw, a.lr, 4, a.lt, tid, nt);
    if (only) {
        step_compute_tw<K, false, true, SKIP>(x, tw0);
        glb_put<K, false, PASS_A>(x, a, dst, tile_off, 0, tid, nt);
    } else {
        step_compute_tw<K, false, true, SKIP, false, true>(x, tw0);  // lazy out: the next step reduces what it adds
        lds_put<K>(x, s, 0, a.lt, tid, nt);
        __syncthreads();
    }
}
// A step's regrouping writes the very LDS words its own thread read (same K, same s0): only the barrier AFTER the writes is
// needed — the next step reads other threads' words.
template <int K, bool PASS_A>
__device__ __forceinline__ void fwd_next(uint32_t (&x)[16], uint32_t (&twc)[16], const R16Args& a, uint32_t* s, const uint32_t* ltw,
                                         uint32_t* dst, size_t tile_off, int s0, bool last, uint32_t tid, uint32_t nt) {
    lds_get<K>(x, s, s0, a.lt, tid, nt);
    uint32_t twn[16];
    if (!last) prefetch_dispatch(twn, ltw, a.lr, s0 + 4, a.lt, tid, nt);
    if (last) {
        // pass A's twist product takes operands in [0, 2P): leave the last stage lazy there
        if (PASS_A && a.twist != nullptr) step_compute_tw<K, false, false, 0, true, true>(x, twc);
        else step_compute_tw<K, false, false, 0, true, false>(x, twc);
        glb_put<K, false, PASS_A>(x, a, dst, tile_off, s0, tid, nt);
    } else {
        step_compute_tw<K, false, false, 0, true, true>(x, twc);
        lds_put<K>(x, s, s0, a.lt, tid, nt);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) twc[i] = twn[i];
    }
}
// Inverse (DIF) order: the top step comes from global (twist / scale applied on load), step 0 goes back to global.
template <int K, bool PASS_A>
__device__ __forceinline__ void inv_top(uint32_t (&x)[16], uint32_t (&twn)[16], const R16Args& a, uint32_t* s, const uint32_t* ltw,
                                        const uint32_t* src, size_t tile_off, int s0, uint32_t tid, uint32_t nt) {
    uint32_t tw0[16];
    tw_load<K, false, 0>(tw0, ltw, s0, a.lt, tid, nt);
    glb_get<K, true, PASS_A>(x, a, src, tile_off, s0, tid, nt);
    if (s0 >= 8) {
        tw_load<4, false, 0>(twn, ltw, s0 - 4, a.lt, tid, nt);
    } else {
        tw_load<4, true, 0>(twn, ltw, 0, a.lt, tid, nt);
    }
    step_compute_tw<K, true, false, 0>(x, tw0);
    lds_put<K>(x, s, s0, a.lt, tid, nt);
    __syncthreads();
}

// One workgroup per (tile, column).  blockDim.x = tile elements / 16.  LR/LT > 0 / >= 0 bake the hot geometries in at
// compile time (stage loops unroll, shifts and LDS strides become immediates); LR = 0 is the generic runtime version.
template <bool INV, bool PASS_A, int SKIP, int LR, int LT, int MAXT = 512>
__global__ __launch_bounds__(MAXT) void ntt_r16_kernel(R16Args a) {
    extern __shared__ uint32_t lds[];
    if (LR > 0) {
        a.lr = LR;
        a.lt = LT;
        if (!PASS_A) a.lrows = LR;
    }
    uint32_t* s = lds;
    const uint32_t* __restrict__ ltw = a.tw;  // global stage table (see the header comment)
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    // Consecutive workgroups walk consecutive tiles of one column, so the chip streams contiguous memory (a
    // column-fastest order makes every workgroup in flight hit addresses 2^m words apart and hot-spots HBM channels:
    // measured 13 % slower).  Pass B additionally keeps tiles that share a 128-byte line on one XCD (block b is observed
    // on XCD b % 8; speed only).
    const uint32_t tl = (uint32_t)__builtin_ctz(a.tiles);  // tiles is a power of two
    const uint32_t bt = blockIdx.x & (a.tiles - 1u), col = blockIdx.x >> tl;
    const uint32_t tile = (!PASS_A && (a.tiles % 8u == 0u)) ? (bt % 8u) * (a.tiles / 8u) + bt / 8u : bt;
    const size_t tile_off = (size_t)tile * a.tile_stride;
    const uint32_t* src = a.in + (size_t)col * a.in_col_stride;
    uint32_t* dst = a.out + (size_t)col * a.out_col_stride;

    uint32_t x[16], twc[16];
    const int ns = (a.lr + 3) >> 2;
    if (!INV) {
        const int K0 = a.lr < 4 ? a.lr : 4;
        switch (K0) {
            case 4: fwd_first<4, PASS_A, SKIP>(x, twc, a, s, ltw, src, dst, tile_off, ns == 1, tid, nt); break;
            case 3: fwd_first<3, PASS_A, SKIP>(x, twc, a, s, ltw, src, dst, tile_off, true, tid, nt); break;
            case 2: fwd_first<2, PASS_A, SKIP>(x, twc, a, s, ltw, src, dst, tile_off, true, tid, nt); break;
            default: fwd_first<1, PASS_A, SKIP>(x, twc, a, s, ltw, src, dst, tile_off, true, tid, nt); break;
        }
#pragma unroll
        for (int si = 1; si < ns; ++si) {
            const int s0 = 4 * si;
            const int K = a.lr - s0 < 4 ? a.lr - s0 : 4;
            const bool last = si == ns - 1;
            switch (K) {
                case 4: fwd_next<4, PASS_A>(x, twc, a, s, ltw, dst, tile_off, s0, last, tid, nt); break;
                case 3: fwd_next<3, PASS_A>(x, twc, a, s, ltw, dst, tile_off, s0, true, tid, nt); break;
                case 2: fwd_next<2, PASS_A>(x, twc, a, s, ltw, dst, tile_off, s0, true, tid, nt); break;
                default: fwd_next<1, PASS_A>(x, twc, a, s, ltw, dst, tile_off, s0, true, tid, nt); break;
            }
        }
    } else {
        if (ns == 1) {
            switch (a.lr) {
                case 4: glb_get<4, true, PASS_A>(x, a, src, tile_off, 0, tid, nt); step_compute<4, true, true, 0>(x, ltw, 0, a.lt, tid, nt); glb_put<4, true, PASS_A>(x, a, dst, tile_off, 0, tid, nt); break;
                case 3: glb_get<3, true, PASS_A>(x, a, src, tile_off, 0, tid, nt); step_compute<3, true, true, 0>(x, ltw, 0, a.lt, tid, nt); glb_put<3, true, PASS_A>(x, a, dst, tile_off, 0, tid, nt); break;
                case 2: glb_get<2, true, PASS_A>(x, a, src, tile_off, 0, tid, nt); step_compute<2, true, true, 0>(x, ltw, 0, a.lt, tid, nt); glb_put<2, true, PASS_A>(x, a, dst, tile_off, 0, tid, nt); break;
                default: glb_get<1, true, PASS_A>(x, a, src, tile_off, 0, tid, nt); step_compute<1, true, true, 0>(x, ltw, 0, a.lt, tid, nt); glb_put<1, true, PASS_A>(x, a, dst, tile_off, 0, tid, nt); break;
            }
            return;
        }
        const int s_top = 4 * (ns - 1);
        switch (a.lr - s_top) {
            case 4: inv_top<4, PASS_A>(x, twc, a, s, ltw, src, tile_off, s_top, tid, nt); break;
            case 3: inv_top<3, PASS_A>(x, twc, a, s, ltw, src, tile_off, s_top, tid, nt); break;
            case 2: inv_top<2, PASS_A>(x, twc, a, s, ltw, src, tile_off, s_top, tid, nt); break;
            default: inv_top<1, PASS_A>(x, twc, a, s, ltw, src, tile_off, s_top, tid, nt); break;
        }
#pragma unroll
        for (int si = ns - 2; si >= 1; --si) {
            lds_get<4>(x, s, 4 * si, a.lt, tid, nt);
            uint32_t twn[16];
            if (si >= 2) {
                tw_load<4, false, 0>(twn, ltw, 4 * (si - 1), a.lt, tid, nt);
            } else {
                tw_load<4, true, 0>(twn, ltw, 0, a.lt, tid, nt);
            }
            step_compute_tw<4, true, false, 0>(x, twc);
            lds_put<4>(x, s, 4 * si, a.lt, tid, nt);  // the words this thread just read: no barrier before the writes
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) twc[i] = twn[i];
        }
        lds_get<4>(x, s, 0, a.lt, tid, nt);
        step_compute_tw<4, true, true, 0>(x, twc);
        glb_put<4, true, PASS_A>(x, a, dst, tile_off, 0, tid, nt);
    }
}

// Pass A of the forward transform for the hot geometry (2^12-element tiles, three radix-16 steps), several columns per
// workgroup: the twist slice of the tile and the twiddles of all three steps depend only on the tile, so they are loaded
// once and reused for `a.cpw` consecutive columns.  Cuts the twist re-reads (one 2^m-word table per column otherwise:
// measured 0.9 GB fetched per launch for 0.2 GB of input) and ~60 address/load instructions per column.
// 158 VGPRs = 3 waves per SIMD; forcing 4 (128 VGPRs, 34 spill instructions) or 5 measured 3 % / 18 % slower.
template <int SKIP>
__global__ __launch_bounds__(256, 1) void ntt_passA_fwd12_multi_kernel(R16Args a) {
    extern __shared__ uint32_t lds[];
    uint32_t* s = lds;
    const uint32_t* __restrict__ ltw = a.tw;
    const uint32_t tid = threadIdx.x, nt = 256u;
    a.lr = 12;
    a.lrows = 12;
    a.lt = 0;
    const uint32_t tile = blockIdx.x & (a.tiles - 1u), colg = blockIdx.x >> (uint32_t)__builtin_ctz(a.tiles);  // power of two
    const size_t tile_off = (size_t)tile << 12;
    uint32_t tw0[16], tw1[16], tw2[16], f[16];
    tw_load<4, true, SKIP>(tw0, ltw, 0, 0, tid, nt);
    tw_load<4, false, 0>(tw1, ltw, 4, 0, tid, nt);
    tw_load<4, false, 0>(tw2, ltw, 8, 0, tid, nt);
    const bool has_twist = a.twist != nullptr;
    if (has_twist) {
        const uint32_t* tp = a.twist + tile_off;  // last step: thread owns rows tid + 256 * mid
#pragma unroll
        for (int mid = 0; mid < 16; ++mid) f[mid] = tp[tid + 256u * mid];
    }
    const uint32_t c0 = colg * a.cpw;
    // Two LDS tiles, alternating by column: column c's first regrouping then cannot overtake another wave's last reads of column
    // c - 1 (those are in the other tile), and a regrouping that rewrites the words its own thread read needs no barrier before
    // it — two barriers per column instead of four.  Step boundaries are lazy (see step_compute_tw); the twist product takes
    // the last stage's outputs unreduced.
    for (uint32_t c = 0; c < a.cpw && c0 + c < a.cols; ++c) {
        const uint32_t* src = a.in + (size_t)(c0 + c) * a.in_col_stride;
        uint32_t* dst = a.out + (size_t)(c0 + c) * a.out_col_stride + tile_off;
        uint32_t* sc = s + (c & 1u) * (4096u + 256u);
        uint32_t x[16];
        // buffer loads / stores: the per-lane byte offset is loop-invariant, the column and the row of register `mid` are scalar
        if (a.expand == 2) {
            const __amdgpu_buffer_rsrc_t rs = col_rsrc(src + (tile_off >> 2), 4096u);
            const __attribute__((ext_vector_type(4))) uint32_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = v[i >> 2];
        } else {
            const __amdgpu_buffer_rsrc_t rs = col_rsrc(src + tile_off, 16384u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __attribute__((ext_vector_type(4))) uint32_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 64u, 16u * i, 0);
                x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
            }
        }
        step_compute_tw<4, false, true, SKIP, false, true>(x, tw0);
        lds_put<4>(x, sc, 0, 0, tid, nt);
        __syncthreads();
        lds_get<4>(x, sc, 4, 0, tid, nt);
        step_compute_tw<4, false, false, 0, true, true>(x, tw1);
        lds_put<4>(x, sc, 4, 0, tid, nt);
        __syncthreads();
        lds_get<4>(x, sc, 8, 0, tid, nt);
        const __amdgpu_buffer_rsrc_t rd = col_rsrc(dst, 16384u);
        if (has_twist) {
            step_compute_tw<4, false, false, 0, true, true>(x, tw2);
#pragma unroll
            for (int mid = 0; mid < 16; ++mid) __builtin_amdgcn_raw_buffer_store_b32(fp_mul(x[mid], f[mid]), rd, tid * 4u, 1024u * mid, 0);
        } else {
            step_compute_tw<4, false, false, 0, true, false>(x, tw2);
#pragma unroll
            for (int mid = 0; mid < 16; ++mid) __builtin_amdgcn_raw_buffer_store_b32(x[mid], rd, tid * 4u, 1024u * mid, 0);
        }
    }
}

}  // namespace bx
