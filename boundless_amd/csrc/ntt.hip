// ntt.hip — batched radix-2 NTT / LDE over BabyBear for gfx950 (column-major, one polynomial per column).
//
// Entry points restate risc0_zkp::hal::Hal::{batch_interpolate_ntt, batch_evaluate_ntt,
// batch_expand_into_evaluate_ntt, batch_bit_reverse, zk_shift} (risc0-zkp 3.0.3, reference Cargo.lock:9155),
// reached from bento/crates/workflow/src/tasks/prove.rs:41-49.  The upstream CPU code is a recursive
// radix-2 DIF/DIT; the arithmetic is exact, so any factorisation of the same DFT gives identical words.
//
// MI355X design: a size-2^m transform is factored "four-step" as 2^m = 2^m_hi * 2^m_lo.
//   pass A : 2^m_lo contiguous blocks of 2^m_hi elements, transformed with perfectly coalesced loads/stores; the
//            inter-pass twist w_M^(j_lo*k2) (and the 1/M scale for the inverse) comes from a precomputed table that is
//            streamed in the same order as the data.
//   pass B : 2^m_lo-point transforms across the blocks (stride 2^m_hi), T adjacent positions per workgroup so every
//            row access is a T*4-byte contiguous segment.
// Forward (evaluate) = A then B, bit-reversed coefficients -> natural evaluations; the zero-padding "expand" is
// folded into pass A's load (out[i] = in[i >> bits]; the first `bits` stages are skipped exactly like upstream).
// Inverse (interpolate) = B then A.  Each pass moves every element through HBM once: 2 reads + 2 writes per element
// instead of m.  Sizes with m <= m_hi need pass A only.
// Two kernel families implement the passes:
//   ntt_r16.hpp  (default) register-resident radix-16 steps, LDS only for regrouping — the fast path for tiles >= 2^10;
//   this file    ntt_block_kernel / ntt_strided_kernel: one radix-2 stage per LDS round trip — the general fallback for
//                small or unaligned shapes and unusual expand_bits (tunable ntt_fast = 0 forces it; both are tested).
#include "ctx.hpp"
#include "ntt_r16.hpp"

namespace bx {

// ---------------------------------------------------------------------------------------------------------
// in-LDS radix-2 stages over `rows` = 2^lr points laid out with a unit of `T` = 2^lt adjacent lanes:
// element (r, t) lives at s[r*T + t].  ltw is the stage table (ltw[half + e] = w_{2*half}^e).
// ---------------------------------------------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ void lds_stages(uint32_t* __restrict__ s, const uint32_t* __restrict__ ltw, int lr, int lt,
                                           int first_stage, int tid, int nthreads) {
    const uint32_t bfly = 1u << (lr - 1 + lt);
    const uint32_t tmask = (1u << lt) - 1u;
    if (!INV) {
        for (int stage = first_stage; stage <= lr; ++stage) {
            const uint32_t half = 1u << (stage - 1);
            for (uint32_t x = tid; x < bfly; x += nthreads) {
                uint32_t t = x & tmask, q = x >> lt;
                uint32_t lo = q & (half - 1u);
                uint32_t i = (((q - lo) << 1) | lo);
                uint32_t ia = (i << lt) | t, ib = ((i + half) << lt) | t;
                uint32_t a = s[ia];
                uint32_t b = fp_mul(s[ib], ltw[half + lo]);
                s[ia] = fp_add(a, b);
                s[ib] = fp_sub(a, b);
            }
            __syncthreads();
        }
    } else {
        for (int stage = lr; stage >= 1; --stage) {
            const uint32_t half = 1u << (stage - 1);
            for (uint32_t x = tid; x < bfly; x += nthreads) {
                uint32_t t = x & tmask, q = x >> lt;
                uint32_t lo = q & (half - 1u);
                uint32_t i = (((q - lo) << 1) | lo);
                uint32_t ia = (i << lt) | t, ib = ((i + half) << lt) | t;
                uint32_t a = s[ia], b = s[ib];
                s[ia] = fp_add(a, b);
                s[ib] = fp_mul(fp_sub(a, b), ltw[half + lo]);
            }
            __syncthreads();
        }
    }
}

// pass A: one workgroup per (block, column).  LDS = data[R] + stage table[R].
template <bool INV>
__global__ void ntt_block_kernel(uint32_t* out, const uint32_t* in,  // may alias (in-place): no __restrict__
                                 const uint32_t* __restrict__ tw, const uint32_t* __restrict__ twist, uint32_t scale,
                                 int lr, int expand_bits, int first_stage, size_t in_col_stride, size_t out_col_stride) {
    extern __shared__ uint32_t lds[];
    const uint32_t R = 1u << lr;
    uint32_t* s = lds;
    uint32_t* ltw = lds + R;
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t block = blockIdx.x, col = blockIdx.y;
    const uint32_t* src = in + col * in_col_stride + ((block << lr) >> expand_bits);
    uint32_t* dst = out + col * out_col_stride + (block << lr);
    const uint32_t* twb = twist ? twist + (block << lr) : nullptr;

    for (uint32_t i = tid; i < R; i += nt) ltw[i] = tw[i];
    if (!INV) {
        for (uint32_t i = tid; i < R; i += nt) s[i] = src[i >> expand_bits];
        __syncthreads();
        lds_stages<false>(s, ltw, lr, 0, first_stage, tid, nt);
        if (twb) {
            for (uint32_t i = tid; i < R; i += nt) dst[i] = fp_mul(s[i], twb[i]);
        } else {
            for (uint32_t i = tid; i < R; i += nt) dst[i] = s[i];
        }
    } else {
        if (twb) {
            for (uint32_t i = tid; i < R; i += nt) s[i] = fp_mul(src[i], twb[i]);
        } else {
            for (uint32_t i = tid; i < R; i += nt) s[i] = fp_mul(src[i], scale);
        }
        __syncthreads();
        lds_stages<true>(s, ltw, lr, 0, 1, tid, nt);
        for (uint32_t i = tid; i < R; i += nt) dst[i] = s[i];
    }
}

// pass B: one workgroup per (tile of T adjacent positions, column); rows are 2^m_hi apart.  In place.
// `tile_swz`: blockIdx.x -> tile remap so that tiles sharing a 128-byte line run on the same XCD (block b is
// observed on XCD b % 8; a performance hint only).
template <bool INV>
__global__ void ntt_strided_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ tw, int lr, int lt, int m_hi,
                                   size_t col_stride, uint32_t tiles) {
    extern __shared__ uint32_t lds[];
    const uint32_t rows = 1u << lr, T = 1u << lt;
    uint32_t* s = lds;
    uint32_t* ltw = lds + (rows << lt);
    const int tid = threadIdx.x, nt = blockDim.x;
    uint32_t b = blockIdx.x;
    uint32_t tile = (tiles % 8u == 0u) ? (b % 8u) * (tiles / 8u) + b / 8u : b;
    uint32_t* base = io + (size_t)blockIdx.y * col_stride + ((size_t)tile << lt);
    const uint32_t total = rows << lt, tmask = T - 1u;

    for (uint32_t i = tid; i < rows; i += nt) ltw[i] = tw[i];
    for (uint32_t e = tid; e < total; e += nt) s[e] = base[((size_t)(e >> lt) << m_hi) + (e & tmask)];
    __syncthreads();
    lds_stages<INV>(s, ltw, lr, lt, 1, tid, nt);
    for (uint32_t e = tid; e < total; e += nt) base[((size_t)(e >> lt) << m_hi) + (e & tmask)] = s[e];
}

// twist[b*2^m_hi + i] = w_M^(+-rev_{m_lo}(b) * i) (* 1/M for the inverse)
__global__ void twist_build_kernel(uint32_t* __restrict__ out, uint32_t root, uint32_t scale, int m, int m_hi) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ((size_t)1 << m)) return;
    uint32_t b = (uint32_t)(p >> m_hi), i = (uint32_t)(p & (((size_t)1 << m_hi) - 1));
    uint64_t e = ((uint64_t)bit_reverse(b, m - m_hi) * i) & (((uint64_t)1 << m) - 1);
    out[p] = fp_mul(fp_pow(root, e), scale);
}

__global__ void zk_tab_kernel(uint32_t* __restrict__ lo, uint32_t* __restrict__ hi, int n, int lo_bits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (1u << lo_bits)) lo[i] = fp_pow(MONT_THREE, (uint64_t)bit_reverse(i, lo_bits) << (n - lo_bits));
    if (i < (1u << (n - lo_bits))) hi[i] = fp_pow(MONT_THREE, bit_reverse(i, n - lo_bits));
}
__global__ void zk_shift_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                int n, int lo_bits, size_t total) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint32_t pos = (uint32_t)(i & (((size_t)1 << n) - 1));
        uint32_t f = fp_mul(lo[pos & ((1u << lo_bits) - 1u)], hi[pos >> lo_bits]);
        io[i] = fp_mul(io[i], f);
    }
}
__global__ void bit_reverse_kernel(uint32_t* __restrict__ io, int n, size_t total) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        uint32_t i = (uint32_t)(g & (((size_t)1 << n) - 1));
        uint32_t r = bit_reverse(i, n);
        if (i < r) {
            size_t base = g - i;
            uint32_t a = io[base + i], b = io[base + r];
            io[base + i] = b;
            io[base + r] = a;
        }
    }
}

// Tiled bit reversal for n >= 12: index i = (a:6 | b:n-12 | c:6) maps to (rev c | rev b | rev a), so the 64x64 tile
// {all a, all c} with middle bits b lands, transposed, on the tile with middle bits rev(b).  A workgroup loads the pair
// (b, rev b), swaps the two tiles through LDS and stores them: each word moves HBM->HBM once.  Every lane moves 16 bytes per
// access (a wave instruction covers four 256-byte rows) and all eight loads of a thread are issued before the first LDS write:
// with one word per lane the launch had too few bytes in flight to cover the HBM latency (28 % of the roofline).
__global__ __launch_bounds__(256) void bit_reverse_tiled_kernel(uint32_t* __restrict__ io, int n) {
    __shared__ uint32_t A[64][65], B[64][65];
    const int nb = n - 12;
    const uint32_t b = blockIdx.x, rb = bit_reverse(b, nb);
    if (rb < b) return;
    uint32_t* col = io + ((size_t)blockIdx.y << n);
    const uint32_t l16 = threadIdx.x & 15u, r = threadIdx.x >> 4;  // lane: words 4 l16 .. 4 l16 + 3 of row r + 16 p
    const int hs = n - 6;
    const bool two = rb != b;
    uint4 va[4], vb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const size_t row = (size_t)(r + 16u * p) << hs;
        va[p] = *reinterpret_cast<const uint4*>(col + row + (b << 6) + 4u * l16);
        if (two) vb[p] = *reinterpret_cast<const uint4*>(col + row + (rb << 6) + 4u * l16);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t* ar = &A[r + 16u * p][4u * l16];
        ar[0] = va[p].x, ar[1] = va[p].y, ar[2] = va[p].z, ar[3] = va[p].w;
        if (two) {
            uint32_t* br = &B[r + 16u * p][4u * l16];
            br[0] = vb[p].x, br[1] = vb[p].y, br[2] = vb[p].z, br[3] = vb[p].w;
        }
    }
    __syncthreads();
    // output word c = 4 l16 + i of row a comes from A[rev6(c)][rev6(a)], and rev6(4 l16 + i) = 16 rev2(i) + rev4(l16)
    const uint32_t s0 = bit_reverse(l16, 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t a = r + 16u * p, ra = bit_reverse(a, 6);
        const size_t row = (size_t)a << hs;
        *reinterpret_cast<uint4*>(col + row + (rb << 6) + 4u * l16) = make_uint4(A[s0][ra], A[32u + s0][ra], A[16u + s0][ra], A[48u + s0][ra]);
        if (two) *reinterpret_cast<uint4*>(col + row + (b << 6) + 4u * l16) = make_uint4(B[s0][ra], B[32u + s0][ra], B[16u + s0][ra], B[48u + s0][ra]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
const char* ntt_init_tables(bx_ctx* c) {
    const size_t n = (size_t)1 << TW_LOG;
    std::vector<uint32_t> f(n, 0), r(n, 0);
    uint32_t rou[28], rev[28];
    rou[27] = fp_encode(137u);
    for (int k = 26; k >= 0; --k) rou[k] = fp_mul(rou[k + 1], rou[k + 1]);
    for (int k = 0; k < 28; ++k) rev[k] = fp_inv(rou[k]);
    for (int s = 1; s <= TW_LOG; ++s) {
        size_t half = (size_t)1 << (s - 1);
        uint32_t cf = MONT_ONE, cr = MONT_ONE;
        for (size_t e = 0; e < half; ++e) {
            f[half + e] = cf;
            r[half + e] = cr;
            cf = fp_mul(cf, rou[s]);
            cr = fp_mul(cr, rev[s]);
        }
    }
    BX_HIP(c, hipMalloc(&c->d_tw_fwd, n * 4));
    BX_HIP(c, hipMalloc(&c->d_tw_inv, n * 4));
    BX_HIP(c, hipMemcpy(c->d_tw_fwd, f.data(), n * 4, hipMemcpyHostToDevice));
    BX_HIP(c, hipMemcpy(c->d_tw_inv, r.data(), n * 4, hipMemcpyHostToDevice));
    return nullptr;
}
void ntt_free_tables(bx_ctx* c) {
    if (c->d_tw_fwd) (void)hipFree(c->d_tw_fwd);
    if (c->d_tw_inv) (void)hipFree(c->d_tw_inv);
    for (auto& kv : c->twist) (void)hipFree(kv.second);
    for (auto& kv : c->zk) {
        (void)hipFree(kv.second.lo);
        (void)hipFree(kv.second.hi);
    }
    for (auto& kv : c->zk_full) (void)hipFree(kv.second);
    c->twist.clear();
    c->zk.clear();
    c->zk_full.clear();
}

static const char* get_twist(bx_ctx* c, int m, int m_hi, bool inverse, uint32_t** out) {
    TwistKey key{m, m_hi, inverse ? 1 : 0};
    auto it = c->twist.find(key);
    if (it != c->twist.end()) {
        *out = it->second;
        return nullptr;
    }
    uint32_t* d = nullptr;
    size_t M = (size_t)1 << m;
    BX_HIP(c, hipMalloc(&d, M * 4));
    uint32_t root = fp_pow(fp_encode(137u), (uint64_t)1 << (27 - m));  // w_M
    uint32_t scale = MONT_ONE;
    if (inverse) {
        root = fp_inv(root);
        scale = fp_inv(fp_encode((uint32_t)M));
    }
    hipLaunchKernelGGL(twist_build_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->stream, d, root, scale, m,
                       m_hi);
    BX_LAUNCH_CHECK(c);
    c->twist[key] = d;
    *out = d;
    return nullptr;
}

struct Split {
    int m_hi, m_lo, lt;
};
static Split choose_split(bx_ctx* c, int m) {
    Split sp;
    int blk = (int)c->ntt_block_log;
    if (blk > TW_LOG) blk = TW_LOG;
    // sizes beyond 2^24 (segments of po2 23 / 24): a taller contiguous pass keeps the strided pass at <= 2^12 rows where it can,
    // i.e. its LDS tile (<= 2^14 elements) at least four positions wide (measured: po2 23 proof 0.44 -> 0.38 s)
    if (m - blk > 12 && blk < TW_LOG) blk = m - 12 < TW_LOG ? m - 12 : TW_LOG;
    if (m <= blk) {
        sp.m_hi = m;
        sp.m_lo = 0;
        sp.lt = 0;
        return sp;
    }
    sp.m_hi = blk;
    sp.m_lo = m - blk;
    if (sp.m_lo > TW_LOG) {  // > 2^26: not reachable for BabyBear segment sizes, guarded by the callers
        sp.m_lo = TW_LOG;
        sp.m_hi = m - TW_LOG;
    }
    int tile = (int)c->ntt_tile_log;
    if (tile < sp.m_lo) tile = sp.m_lo;
    sp.lt = tile - sp.m_lo;
    if (sp.lt > sp.m_hi) sp.lt = sp.m_hi;
    return sp;
}

template <typename K>
static const char* allow_lds(bx_ctx* c, K kernel, size_t bytes) {
    if (bytes > 64 * 1024) BX_HIP(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return nullptr;
}

// forward transform of `count` columns: in (size M >> expand_bits per column) -> out (size M per column).
// ---- register-radix-16 fast path (ntt_r16.hpp) -------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <bool INV, bool PASS_A, int SKIP, int LR, int LT, int MAXT = 512>
static const char* launch_r16_geom(bx_ctx* c, R16Args a, size_t count) {
    uint32_t tile_elems = 1u << (a.lrows + a.lt);
    size_t lds = ((size_t)tile_elems + (tile_elems >> 4)) * 4;
    a.cols = (uint32_t)count;
    BX_REQUIRE(c, (size_t)a.tiles * count < ((size_t)1 << 31), "ntt: too many workgroups in one launch");
    BX_REQUIRE(c, tile_elems / 16 <= (uint32_t)MAXT, "ntt: tile larger than the kernel's launch bound");
    BX_TRY(allow_lds(c, (ntt_r16_kernel<INV, PASS_A, SKIP, LR, LT, MAXT>), lds));
    hipLaunchKernelGGL((ntt_r16_kernel<INV, PASS_A, SKIP, LR, LT, MAXT>), dim3(a.tiles * (unsigned)count), dim3(tile_elems / 16), lds,
                       c->stream, a);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
template <int SKIP>
static const char* launch_passA_multi(bx_ctx* c, R16Args a, size_t count, uint32_t cpw) {
    a.cols = (uint32_t)count;
    a.cpw = cpw;
    size_t lds = ((size_t)4096 + 256) * 4 * 2;  // two tiles, alternating by column
    unsigned groups = (unsigned)((count + cpw - 1) / cpw);
    hipLaunchKernelGGL((ntt_passA_fwd12_multi_kernel<SKIP>), dim3(a.tiles * groups), dim3(256), lds, c->stream, a);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
// hot geometries (BASELINE sizes 2^20 / 2^22, default tunables) get a compile-time specialisation
template <bool INV, bool PASS_A, int SKIP>
static const char* launch_r16(bx_ctx* c, const R16Args& a, size_t count) {
    if (PASS_A && !INV && a.lr == 12 && a.lrows == 12 && a.lt == 0 && c->ntt_cols_per_wg > 1 && count > 1 &&
        (a.expand == 0 || a.expand == 2)) {
        if (SKIP == 2) return launch_passA_multi<2>(c, a, count, (uint32_t)c->ntt_cols_per_wg);
        if (SKIP == 0) return launch_passA_multi<0>(c, a, count, (uint32_t)c->ntt_cols_per_wg);
    }
    if (PASS_A && a.lr == 12 && a.lt == 0) return launch_r16_geom<INV, PASS_A, SKIP, 12, 0>(c, a, count);
    if (!PASS_A && a.lr == 10 && a.lt == 3) return launch_r16_geom<INV, PASS_A, SKIP, 10, 3>(c, a, count);
    if (!PASS_A && a.lr == 10 && a.lt == 4) return launch_r16_geom<INV, PASS_A, SKIP, 10, 4, 1024>(c, a, count);  // 64-byte rows
    // po2 21 .. 24 segments (compose.yml:67 runs 21): strided passes of 2^11 .. 2^13 rows on the 2^14-element tile, 2^13 contiguous pass
    if (!PASS_A && a.lr == 11 && a.lt == 3) return launch_r16_geom<INV, PASS_A, SKIP, 11, 3, 1024>(c, a, count);
    if (!PASS_A && a.lr == 12 && a.lt == 2) return launch_r16_geom<INV, PASS_A, SKIP, 12, 2, 1024>(c, a, count);
    if (!PASS_A && a.lr == 13 && a.lt == 1) return launch_r16_geom<INV, PASS_A, SKIP, 13, 1, 1024>(c, a, count);
    if (PASS_A && a.lr == 13 && a.lt == 0 && a.lrows == 13) return launch_r16_geom<INV, PASS_A, SKIP, 13, 0>(c, a, count);
    if (!PASS_A && a.lrows + a.lt > 13) return launch_r16_geom<INV, PASS_A, SKIP, 0, 0, 1024>(c, a, count);
    if (!PASS_A && a.lr == 8 && a.lt == 5) return launch_r16_geom<INV, PASS_A, SKIP, 8, 5>(c, a, count);
    return launch_r16_geom<INV, PASS_A, SKIP, 0, 0>(c, a, count);
}

// pass A over `count` columns of size 2^m made of 2^(m - m_hi) blocks; returns false in *ok if the shape is not covered
static const char* fast_pass_a(bx_ctx* c, bool inv, uint32_t* out, const uint32_t* in, const uint32_t* twist, uint32_t scale,
                               size_t count, int m, int m_hi, int expand, int skip, bool* ok, const uint32_t* post = nullptr,
                               bool* post_applied = nullptr) {
    *ok = false;
    int lrows = (int)c->ntt_tile_a_log;
    if (lrows < m_hi) lrows = m_hi;
    if (lrows > m) lrows = m;
    if (lrows < 10 || m_hi < 1 || m_hi > TW_LOG) return nullptr;  // tile must fill at least one wave (16 elements/lane)
    if (!aligned16(out) || !aligned16(in) || (((size_t)1 << m) >> expand) % 4 != 0) return nullptr;
    if (!(skip == 0 || (skip == 2 && !inv))) return nullptr;
    R16Args a;
    a.out = out; a.in = in; a.tw = inv ? c->d_tw_inv : c->d_tw_fwd; a.twist = twist; a.scale = scale;
    // the final store of the inverse pass A applies `post` only in its 16-words-per-thread form (the last step is K = 4 at s0 = 0 whenever m_hi >= 4)
    const bool can_post = inv && post && m_hi >= 4;
    a.post = can_post ? post : nullptr;
    if (post_applied) *post_applied = can_post;
    a.lr = m_hi; a.lrows = lrows; a.lt = 0; a.expand = expand; a.row_shift = 0;
    a.tile_stride = 1u << lrows;
    a.in_col_stride = ((size_t)1 << m) >> expand; a.out_col_stride = (size_t)1 << m;
    a.tiles = 1u << (m - lrows);
    *ok = true;
    if (inv) return launch_r16<true, true, 0>(c, a, count);
    if (skip == 2) return launch_r16<false, true, 2>(c, a, count);
    return launch_r16<false, true, 0>(c, a, count);
}
// tile width (log2 positions) of pass B for a 2^m transform split at m_hi; -1 when the fast path does not cover the shape
static int pass_b_lt(bx_ctx* c, int m, int m_hi) {
    int m_lo = m - m_hi;
    int tile = (int)c->ntt_tile_b_log;
    // keep every row access at least 64 bytes wide: tall passes (>= 2^10 rows) take the 2^14-element tile (1024 threads);
    // measured on the 2^22 LDE: -6 % time and 1.5x fewer HBM write bytes than 32-byte rows
    // ... and the tallest ones (2^11 .. 2^13 rows) the same 2^14-element tile, as wide as it still allows (po2 24 proof 1.22 -> 0.86 s)
    if (c->ntt_tile_b_wide && m_lo + 4 > tile) tile = m_lo + 4 <= 14 ? m_lo + 4 : 14;
    if (tile < m_lo) tile = m_lo;
    int lt = tile - m_lo;
    if (lt > m_hi) lt = m_hi;
    if (m_lo + lt < 10 || m_lo < 1 || m_lo > TW_LOG || m_lo + lt > 14) return -1;
    return lt;
}
static const char* fast_pass_b(bx_ctx* c, bool inv, uint32_t* io, size_t count, int m, int m_hi, bool* ok) {
    *ok = false;
    int m_lo = m - m_hi;
    const int lt = pass_b_lt(c, m, m_hi);
    if (lt < 0) return nullptr;
    // pass B addresses a column through a buffer descriptor with 32-bit byte offsets and a 32-bit num_records (ntt_r16.hpp glb_get /
    // glb_put): a column must stay below 4 GiB, or loads past the wrap would return 0 and stores be dropped silently
    BX_REQUIRE(c, (((size_t)1 << m) * 4) >> 32 == 0, "ntt: a column of 2^30 words or more does not fit pass B's 32-bit buffer offsets");
    R16Args a;
    a.out = io; a.in = io; a.tw = inv ? c->d_tw_inv : c->d_tw_fwd; a.twist = nullptr; a.scale = MONT_ONE;
    a.lr = m_lo; a.lrows = m_lo; a.lt = lt; a.expand = 0; a.row_shift = m_hi;
    a.tile_stride = 1u << lt;
    a.in_col_stride = a.out_col_stride = (size_t)1 << m;
    a.tiles = 1u << (m_hi - lt);
    *ok = true;
    if (inv) return launch_r16<true, false, 0>(c, a, count);
    return launch_r16<false, false, 0>(c, a, count);
}

// `expand_bits` = load shift (out[i] = in[i >> bits]); `skip_bits` = leading stages skipped (== expand_bits for the
// expanding form, and for the in-place Hal::batch_evaluate_ntt(io, count, expand_bits) with no load shift).
static const char* forward(bx_ctx* c, uint32_t* out, const uint32_t* in, size_t count, int m, int expand_bits,
                           int skip_bits) {
    if (m == 0 || count == 0) {
        if (out != in) BX_HIP(c, hipMemcpyAsync(out, in, count * 4, hipMemcpyDeviceToDevice, c->stream));
        return nullptr;
    }
    Split sp = choose_split(c, m);
    BX_REQUIRE(c, sp.m_hi <= TW_LOG && sp.m_lo <= TW_LOG, "ntt: size too large");
    BX_REQUIRE(c, expand_bits <= sp.m_hi && skip_bits <= sp.m_hi, "ntt: expand_bits larger than the block pass");
    size_t M = (size_t)1 << m;
    uint32_t* twist = nullptr;
    if (sp.m_lo) BX_TRY(get_twist(c, m, sp.m_hi, false, &twist));
    bool done_a = false, done_b = false;
    // Column groups (SURVEY hard part 5): run pass A and pass B back to back on `ntt_group_cols` columns at a time so that
    // pass B finds what pass A just wrote in the 256 MB Infinity Cache instead of HBM (0 = one pass A and one pass B over all
    // columns).  Measured on the 2^22 LDE, see DESIGN.md §4 "NTT traffic".
    if (c->ntt_fast && c->ntt_group_cols > 0 && sp.m_lo && (size_t)c->ntt_group_cols < count) {
        const size_t g = (size_t)c->ntt_group_cols;
        bool all = true;
        for (size_t c0 = 0; c0 < count && all; c0 += g) {
            const size_t n = count - c0 < g ? count - c0 : g;
            bool oa = false, ob = false;
            BX_TRY(fast_pass_a(c, false, out + c0 * M, in + c0 * (M >> expand_bits), twist, MONT_ONE, n, m, sp.m_hi, expand_bits, skip_bits, &oa));
            if (oa) BX_TRY(fast_pass_b(c, false, out + c0 * M, n, m, sp.m_hi, &ob));
            all = oa && ob;
            if (!all && c0 != 0) return set_msg(c, "ntt: column-group path became unavailable mid-way");
        }
        if (all) return nullptr;
    }
    if (c->ntt_fast) BX_TRY(fast_pass_a(c, false, out, in, twist, MONT_ONE, count, m, sp.m_hi, expand_bits, skip_bits, &done_a));
    if (!done_a) {
        unsigned R = 1u << sp.m_hi;
        unsigned threads = R / 2 < 64 ? 64 : (R / 2 > 256 ? 256 : R / 2);
        size_t lds = (size_t)R * 8;
        BX_TRY(allow_lds(c, ntt_block_kernel<false>, lds));
        hipLaunchKernelGGL(ntt_block_kernel<false>, dim3(1u << sp.m_lo, (unsigned)count), dim3(threads), lds, c->stream, out,
                           in, c->d_tw_fwd, twist, MONT_ONE, sp.m_hi, expand_bits, skip_bits + 1, M >> expand_bits, M);
        BX_LAUNCH_CHECK(c);
    }
    if (sp.m_lo && c->ntt_fast) BX_TRY(fast_pass_b(c, false, out, count, m, sp.m_hi, &done_b));
    if (sp.m_lo && !done_b) {
        unsigned rows = 1u << sp.m_lo;
        unsigned total = rows << sp.lt;
        unsigned threads = total / 2 > 1024 ? 1024 : (total / 2 < 64 ? 64 : total / 2);
        if (threads > 512 && total <= 8192) threads = 512;
        size_t lds = (size_t)total * 4 + (size_t)rows * 4;
        BX_TRY(allow_lds(c, ntt_strided_kernel<false>, lds));
        unsigned tiles = 1u << (sp.m_hi - sp.lt);
        hipLaunchKernelGGL(ntt_strided_kernel<false>, dim3(tiles, (unsigned)count), dim3(threads), lds, c->stream, out,
                           c->d_tw_fwd, sp.m_lo, sp.lt, sp.m_hi, M, tiles);
        BX_LAUNCH_CHECK(c);
    }
    return nullptr;
}

static const char* inverse(bx_ctx* c, uint32_t* io, size_t count, int m, const uint32_t* post = nullptr, bool* post_applied = nullptr) {
    if (post_applied) *post_applied = false;
    if (m == 0 || count == 0) return nullptr;
    Split sp = choose_split(c, m);
    BX_REQUIRE(c, sp.m_hi <= TW_LOG && sp.m_lo <= TW_LOG, "ntt: size too large");
    size_t M = (size_t)1 << m;
    uint32_t* twist = nullptr;
    bool done_a = false, done_b = false;
    if (sp.m_lo) BX_TRY(get_twist(c, m, sp.m_hi, true, &twist));
    if (sp.m_lo && c->ntt_fast) BX_TRY(fast_pass_b(c, true, io, count, m, sp.m_hi, &done_b));
    if (sp.m_lo && !done_b) {
        unsigned rows = 1u << sp.m_lo;
        unsigned total = rows << sp.lt;
        unsigned threads = total / 2 > 1024 ? 1024 : (total / 2 < 64 ? 64 : total / 2);
        if (threads > 512 && total <= 8192) threads = 512;
        size_t lds = (size_t)total * 4 + (size_t)rows * 4;
        BX_TRY(allow_lds(c, ntt_strided_kernel<true>, lds));
        unsigned tiles = 1u << (sp.m_hi - sp.lt);
        hipLaunchKernelGGL(ntt_strided_kernel<true>, dim3(tiles, (unsigned)count), dim3(threads), lds, c->stream, io,
                           c->d_tw_inv, sp.m_lo, sp.lt, sp.m_hi, M, tiles);
        BX_LAUNCH_CHECK(c);
    }
    uint32_t scale = fp_inv(fp_encode((uint32_t)M));
    if (c->ntt_fast) BX_TRY(fast_pass_a(c, true, io, io, twist, scale, count, m, sp.m_hi, 0, 0, &done_a, post, post_applied));
    if (!done_a && post_applied) *post_applied = false;
    if (!done_a) {
        unsigned R = 1u << sp.m_hi;
        unsigned threads = R / 2 < 64 ? 64 : (R / 2 > 256 ? 256 : R / 2);
        size_t lds = (size_t)R * 8;
        BX_TRY(allow_lds(c, ntt_block_kernel<true>, lds));
        hipLaunchKernelGGL(ntt_block_kernel<true>, dim3(1u << sp.m_lo, (unsigned)count), dim3(threads), lds, c->stream, io,
                           io, c->d_tw_inv, twist, scale, sp.m_hi, 0, 1, M, M);
        BX_LAUNCH_CHECK(c);
    }
    return nullptr;
}

static const char* get_zk(bx_ctx* c, int n, ZkTab* out) {
    auto it = c->zk.find(n);
    if (it != c->zk.end()) {
        *out = it->second;
        return nullptr;
    }
    ZkTab t;
    t.lo_bits = (n + 1) / 2;
    size_t nlo = (size_t)1 << t.lo_bits, nhi = (size_t)1 << (n - t.lo_bits);
    BX_HIP(c, hipMalloc(&t.lo, nlo * 4));
    BX_HIP(c, hipMalloc(&t.hi, nhi * 4));
    hipLaunchKernelGGL(zk_tab_kernel, dim3((unsigned)((nlo + 255) / 256)), dim3(256), 0, c->stream, t.lo, t.hi, n, t.lo_bits);
    BX_LAUNCH_CHECK(c);
    c->zk[n] = t;
    *out = t;
    return nullptr;
}

__global__ void zk_full_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, int n,
                               int lo_bits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << n)) out[i] = fp_mul(lo[i & ((1u << lo_bits) - 1u)], hi[i >> lo_bits]);
}
// 3^bitrev_n(i) for every position of a size-2^n column (4 MB at n = 20, read through L2 by the fused inverse pass)
static const char* get_zk_full(bx_ctx* c, int n, const uint32_t** out) {
    auto it = c->zk_full.find(n);
    if (it != c->zk_full.end()) {
        *out = it->second;
        return nullptr;
    }
    ZkTab t;
    BX_TRY(get_zk(c, n, &t));
    uint32_t* full = nullptr;
    BX_HIP(c, hipMalloc(&full, ((size_t)1 << n) * 4));
    hipLaunchKernelGGL(zk_full_kernel, dim3((unsigned)((((size_t)1 << n) + 255) / 256)), dim3(256), 0, c->stream, full, t.lo, t.hi, n,
                       t.lo_bits);
    BX_LAUNCH_CHECK(c);
    c->zk_full[n] = full;
    *out = full;
    return nullptr;
}

}  // namespace bx

using namespace bx;

static const char* zk_shift_impl(bx_ctx* c, bx_buf io, size_t count);

// Extension: batch_interpolate_ntt followed by zk_shift in one call.  When the register-radix path covers the shape the
// shift rides on the final store of the inverse transform (one extra product per element instead of a read-modify-write
// pass over the coefficients); otherwise it is the two calls.
extern "C" const char* bx_batch_interpolate_zk(bx_ctx* c, bx_buf io, size_t count) try {
    if (!c) return "bx_batch_interpolate_zk: null ctx";
    BX_REQUIRE(c, count > 0 && io.len % count == 0 && is_pow2(io.len / count), "batch_interpolate_zk: io.len/count must be a power of two");
    BX_ENTER(c);
    const int m = ilog2(io.len / count);
    bool fused = false;
    {
        OpScope op(c, "batch_interpolate_ntt", 8.0 * (double)io.len);
        const uint32_t* post = nullptr;
        if (m >= 12 && c->ntt_fast) BX_TRY(get_zk_full(c, m, &post));
        BX_TRY(inverse(c, (uint32_t*)io.dptr, count, m, post, &fused));
    }
    if (fused) return nullptr;
    return zk_shift_impl(c, io, count);
} BX_ABI_CATCH(c, "bx_batch_interpolate_zk")

extern "C" const char* bx_batch_interpolate_ntt(bx_ctx* c, bx_buf io, size_t count) try {
    if (!c) return "bx_batch_interpolate_ntt: null ctx";
    BX_REQUIRE(c, count > 0 && io.len % count == 0 && is_pow2(io.len / count), "batch_interpolate_ntt: io.len/count must be a power of two");
    BX_ENTER(c);
    OpScope op(c, "batch_interpolate_ntt", 8.0 * (double)io.len);
    return inverse(c, (uint32_t*)io.dptr, count, ilog2(io.len / count));
} BX_ABI_CATCH(c, "bx_batch_interpolate_ntt")

extern "C" const char* bx_batch_evaluate_ntt(bx_ctx* c, bx_buf io, size_t count, size_t expand_bits) try {
    if (!c) return "bx_batch_evaluate_ntt: null ctx";
    BX_REQUIRE(c, count > 0 && io.len % count == 0 && is_pow2(io.len / count), "batch_evaluate_ntt: io.len/count must be a power of two");
    int m = ilog2(io.len / count);
    BX_REQUIRE(c, expand_bits <= (size_t)m, "batch_evaluate_ntt: expand_bits > log2(size)");
    BX_ENTER(c);
    OpScope op(c, "batch_evaluate_ntt", 8.0 * (double)io.len);
    return forward(c, (uint32_t*)io.dptr, (const uint32_t*)io.dptr, count, m, 0, (int)expand_bits);
} BX_ABI_CATCH(c, "bx_batch_evaluate_ntt")

extern "C" const char* bx_batch_expand_into_evaluate_ntt(bx_ctx* c, bx_buf out, bx_buf in, size_t count, size_t expand_bits) try {
    if (!c) return "bx_batch_expand_into_evaluate_ntt: null ctx";
    BX_REQUIRE(c, count > 0 && in.len % count == 0 && is_pow2(in.len / count), "batch_expand_into_evaluate_ntt: in.len/count must be a power of two");
    BX_REQUIRE(c, expand_bits < 32 && out.len >> expand_bits == in.len && out.len == (in.len << expand_bits), "batch_expand_into_evaluate_ntt: out.len != in.len << expand_bits");
    BX_REQUIRE(c, out.dptr != in.dptr || expand_bits == 0, "batch_expand_into_evaluate_ntt: in-place expansion");
    BX_ENTER(c);
    OpScope op(c, "batch_expand_into_evaluate_ntt", 4.0 * (double)in.len + 4.0 * (double)out.len);
    int m = ilog2(out.len / count);
    return forward(c, (uint32_t*)out.dptr, (const uint32_t*)in.dptr, count, m, (int)expand_bits, (int)expand_bits);
} BX_ABI_CATCH(c, "bx_batch_expand_into_evaluate_ntt")

extern "C" const char* bx_batch_bit_reverse(bx_ctx* c, bx_buf io, size_t count) try {
    if (!c) return "bx_batch_bit_reverse: null ctx";
    BX_REQUIRE(c, count > 0 && io.len % count == 0 && is_pow2(io.len / count), "batch_bit_reverse: io.len/count must be a power of two");
    BX_ENTER(c);
    OpScope op(c, "batch_bit_reverse", 8.0 * (double)io.len);
    int n = ilog2(io.len / count);
    if (n == 0) return nullptr;
    if (n >= 12 && n <= 30 && ((uintptr_t)io.dptr & 15u) == 0) {  // the tiled kernel moves 16 bytes per lane
        hipLaunchKernelGGL(bit_reverse_tiled_kernel, dim3(1u << (n - 12), (unsigned)count), dim3(256), 0, c->stream,
                           (uint32_t*)io.dptr, n);
        BX_LAUNCH_CHECK(c);
        return nullptr;
    }
    size_t blocks = (io.len + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bit_reverse_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (uint32_t*)io.dptr, n, io.len);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_batch_bit_reverse")

// Bit reversal of AoS extension-field arrays (16-byte elements): one thread swaps element i with element rev(i), i < rev(i).
__global__ void bit_reverse_ext_kernel(uint4* __restrict__ io, int n, size_t total) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(g & (((size_t)1 << n) - 1));
        const uint32_t r = bit_reverse(i, n);
        if (i < r) {
            uint4* base = io + (g - i);
            const uint4 a = base[i], b = base[r];
            base[i] = b;
            base[r] = a;
        }
    }
}
extern "C" const char* bx_batch_bit_reverse_ext(bx_ctx* c, bx_buf io_ext, size_t count) try {
    if (!c) return "bx_batch_bit_reverse_ext: null ctx";
    BX_REQUIRE(c, count > 0 && count <= io_ext.len / 4 && io_ext.len % (4 * count) == 0 && is_pow2(io_ext.len / (4 * count)),
               "batch_bit_reverse_ext: io.len/(4*count) must be a power of two");
    BX_REQUIRE(c, ((uintptr_t)io_ext.dptr & 15) == 0, "batch_bit_reverse_ext: buffer must be 16-byte aligned");
    BX_ENTER(c);
    OpScope op(c, "batch_bit_reverse", 8.0 * (double)io_ext.len);
    const size_t elems = io_ext.len / 4;
    const int n = ilog2(elems / count);
    if (n == 0) return nullptr;
    size_t blocks = (elems + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bit_reverse_ext_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (uint4*)io_ext.dptr, n, elems);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_batch_bit_reverse_ext")

extern "C" const char* bx_zk_shift(bx_ctx* c, bx_buf io, size_t count) try {
    if (!c) return "bx_zk_shift: null ctx";
    return zk_shift_impl(c, io, count);
} BX_ABI_CATCH(c, "bx_zk_shift")
static const char* zk_shift_impl(bx_ctx* c, bx_buf io, size_t count) {
    BX_REQUIRE(c, count > 0 && io.len % count == 0 && is_pow2(io.len / count), "zk_shift: io.len/count must be a power of two");
    BX_ENTER(c);
    OpScope op(c, "zk_shift", 8.0 * (double)io.len);
    int n = ilog2(io.len / count);
    ZkTab t;
    BX_TRY(get_zk(c, n, &t));
    size_t blocks = (io.len + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(zk_shift_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (uint32_t*)io.dptr, t.lo, t.hi, n,
                       t.lo_bits, io.len);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
