// scan.hip — single-pass scans over AoS ext arrays for gfx950: the DEEP quotient (poly_divide) and prefix_products.
//
// Restates risc0_zkp::core::poly::poly_divide (run once per tap point of every DEEP combination polynomial by
// Prover::finalize) and risc0_zkp::hal::Hal::prefix_products (risc0-zkp 3.0.3, reference Cargo.lock:9155), reached from
// bento/crates/workflow/src/tasks/prove.rs:41-49.
//
// Both are first-order recurrences over 16-byte elements whose algorithmic traffic is one read and one write of the array.
// Round 2 ran them as reduce / scan-the-chunk-values (recursively, down to one workgroup) / replay: five launches per call, every
// element read twice with 512-byte strides between lanes, 5-17 % of the HBM roofline.  Here each is ONE launch: a workgroup owns
// a tile of 2048 elements (loaded coalesced, transposed through LDS so that a lane owns 8 consecutive elements), scans it in
// registers and wave shuffles, publishes its aggregate, and obtains the carry entering the tile by DECOUPLED LOOK-BACK: wave 0
// inspects the 64 preceding tiles at once, takes the nearest published inclusive value and the aggregates after it, and moves
// on 64 tiles at a time until it finds one.  Tiles are handed out by a ticket counter, so a tile's predecessors are always
// running or finished and the spin terminates.  Each array is read once and written once.
//
// Publication protocol.  An ext value is four words < 2^31, so bit 31 of every word is free: a slot is written with four
// relaxed agent-scope atomic stores of (word | 2^31) and read with four relaxed agent-scope atomic loads; the value is taken
// only when all four words carry the bit.  Every word validates itself, so no ordering between the four is needed and a torn
// read is simply retried.  Slots must start at zero: the state lives in two alternating buffers, and every launch clears the
// extent the OTHER buffer was last used with (calls on a ctx are stream-ordered), so no memset launch is ever needed.
//
// poly_divide specifics: the recurrence runs from the top coefficient down, cur <- z cur + p_i with out_i = cur before the
// update; an element's map is affine with a known slope (z), so a span of L elements is (z^L, B) and only B is scanned or
// published — the slopes z^8, z^16 .. z^2048, z^(2048*64) come from the host as kernel arguments.
#define BX_PLAIN_MAD 1  // the signed multiply-adds of lazy_ext.hpp are left to the compiler here
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "lazy_ext.hpp"

namespace bx {

constexpr int SC_T = 256, SC_I = 8, SC_TILE = SC_T * SC_I;
constexpr uint32_t SC_VALID = 0x80000000u;
constexpr uint32_t SC_HDR = 4;  // words before a sequence's slots: [0] ticket

__device__ __forceinline__ Fp4 sc_ld4(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Fp4{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void sc_publish(uint32_t* slot, const Fp4& v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(slot + k, v.c[k] | SC_VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool sc_try_read(const uint32_t* slot, Fp4& v) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __hip_atomic_load(slot + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!((w[0] & w[1] & w[2] & w[3]) & SC_VALID)) return false;
#pragma unroll
    for (int k = 0; k < 4; ++k) v.c[k] = w[k] & ~SC_VALID;
    return true;
}
__device__ __forceinline__ Fp4 sc_shfl_up(const Fp4& v, int d) {
    return Fp4{{(uint32_t)__shfl_up((int)v.c[0], d), (uint32_t)__shfl_up((int)v.c[1], d), (uint32_t)__shfl_up((int)v.c[2], d),
                (uint32_t)__shfl_up((int)v.c[3], d)}};
}
__device__ __forceinline__ Fp4 sc_shfl_xor(const Fp4& v, int d) {
    return Fp4{{(uint32_t)__shfl_xor((int)v.c[0], d), (uint32_t)__shfl_xor((int)v.c[1], d), (uint32_t)__shfl_xor((int)v.c[2], d),
                (uint32_t)__shfl_xor((int)v.c[3], d)}};
}
// every thread of the grid clears its share of the other state buffer; then the workgroup draws its tile number
__device__ __forceinline__ uint32_t sc_begin(uint32_t* __restrict__ seq_state, uint32_t* __restrict__ clear, size_t clear_words, uint32_t* sh_tile) {
    const size_t nthreads = (size_t)gridDim.x * gridDim.y * blockDim.x;
    for (size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < clear_words; i += nthreads) clear[i] = 0u;
    if (threadIdx.x == 0) *sh_tile = atomicAdd(seq_state, 1u);
    __syncthreads();
    return *sh_tile;
}

struct DivSeq {
    Fp4 z;      // the point
    Fp4 zp[6];  // z^(8 * 2^k), k < 6: slopes of spans of 1, 2, .. 32 lanes
    Fp4 z512;   // slope of a wave (64 lanes x 8)
    Fp4 zL;     // slope of a tile
    Fp4 zLp[6]; // zL^(2^k), k < 6: a lane's look-back weight zL^lane is the product over the set bits of its number
    Fp4 zL64;   // slope of 64 tiles
    uint32_t poly, pad[3];  // which polynomial of the buffer this sequence divides
};
// base^lane from the table of base^(2^k): at most six lazy products instead of a square-and-multiply ladder of generic ones
__device__ __forceinline__ Fp4 sc_lane_pow(const Fp4 (&tab)[6], uint32_t lane) {
    Fp4 r = f4_one();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const Fp4 t = f4_mul_lz(r, tab[k]);
        if (lane & (1u << k)) r = t;
    }
    return r;
}
struct DivArgs {
    DivSeq s[8];
};

// LDS layout of a tile: lane t's 8 elements at [9 t, 9 t + 8) (in 16-byte units): the pad makes the blocked accesses conflict-free
__device__ __forceinline__ uint32_t sc_slot(uint32_t u) { return (u >> 3) * 9u + (u & 7u); }

__global__ __launch_bounds__(SC_T) void div_lookback_kernel(uint32_t* __restrict__ polys, size_t size, DivArgs args, uint32_t* __restrict__ state,
                                                            uint32_t seq_stride, uint32_t tiles, uint32_t* __restrict__ clear, size_t clear_words,
                                                            uint32_t* __restrict__ rems) {
    __shared__ uint4 sh[SC_T * 9];
    __shared__ uint32_t wtot[4 * 4], carry_sh[4], agg_sh[4], sh_tile;
    const uint32_t q = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t* st = state + (size_t)q * seq_stride;
    const uint32_t tile = sc_begin(st, clear, clear_words, &sh_tile);
    const DivSeq& S = args.s[q];
    uint4* poly = reinterpret_cast<uint4*>(polys) + (size_t)S.poly * size;
    const size_t hi = size - (size_t)tile * SC_TILE;                    // positions [hi - valid, hi), top first: u = hi - 1 - pos
    const uint32_t valid = hi < (size_t)SC_TILE ? (uint32_t)hi : (uint32_t)SC_TILE;  // elements of this tile that exist
    // coalesced load (descending addresses), transposed through LDS
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint32_t u = (uint32_t)k * SC_T + tid;
        sh[sc_slot(u)] = u < valid ? poly[hi - 1 - u] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    Fp4 v[SC_I];
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint4 w = sh[9 * tid + k];
        v[k] = Fp4{{w.x, w.y, w.z, w.w}};
    }
    const uint32_t mine = valid > SC_I * tid ? (valid - SC_I * tid < (uint32_t)SC_I ? valid - SC_I * tid : (uint32_t)SC_I) : 0u;
    const C4 zc = f4_centre(S.z);
    Fp4 cur = f4_zero();
#pragma unroll
    for (int k = 0; k < SC_I; ++k)
        if ((uint32_t)k < mine) cur = f4_add(f4_mul_cc(zc, f4_centre(cur)), v[k]);
    // inclusive scan over the lanes of a wave: I_t = z^8 I_(t-1) + B_t, by doubling with the known slopes
    Fp4 I = cur;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const Fp4 prev = sc_shfl_up(I, 1 << k);
        if (lane >= (1u << k)) I = f4_add(I, f4_mul_lz(S.zp[k], prev));
    }
    if (lane == 63) *reinterpret_cast<uint4*>(wtot + 4 * wv) = make_uint4(I.c[0], I.c[1], I.c[2], I.c[3]);
    __syncthreads();
    Fp4 cw = f4_zero();  // carry entering this wave from the waves above it (tile carry still zero)
    for (uint32_t w = 0; w < wv; ++w) cw = f4_add(f4_mul_lz(S.z512, cw), sc_ld4(wtot + 4 * w));
    const Fp4 zlane = sc_lane_pow(S.zp, lane);  // z^(8 lane): what a carry entering the wave is multiplied by on its way to this lane
    Fp4 excl = sc_shfl_up(I, 1);
    if (lane == 0) excl = f4_zero();
    excl = f4_add(excl, f4_mul_lz(zlane, cw));
    if (tid == SC_T - 1) {
        const Fp4 agg = f4_add(I, f4_mul(S.z512, cw));  // the whole tile with zero carry-in
        *reinterpret_cast<uint4*>(agg_sh) = make_uint4(agg.c[0], agg.c[1], agg.c[2], agg.c[3]);
        if (tile > 0) sc_publish(st + SC_HDR + 8 * (size_t)tile, agg);
    }
    __syncthreads();
    if (wv == 0) {
        // look back: lane l inspects tile (base - l); tiles above the first one count as "inclusive, zero"
        Fp4 carry = f4_zero(), wbase = f4_one();
        const Fp4 lpow = sc_lane_pow(S.zLp, lane);
        int base = (int)tile - 1;
        while (tile > 0) {
            const int id = base - (int)lane;
            Fp4 val = f4_zero();
            bool is_incl = true;
            if (id >= 0) {
                const uint32_t* slot = st + SC_HDR + 8 * (size_t)id;
                for (;;) {
                    if (sc_try_read(slot + 4, val)) { is_incl = true; break; }
                    if (sc_try_read(slot, val)) { is_incl = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const unsigned long long m = __ballot(is_incl);
            const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : 64u;  // nearest tile whose inclusive value is known
            Fp4 part = lane <= first ? f4_mul_lz(lpow, val) : f4_zero();
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part = f4_add(part, sc_shfl_xor(part, d));
            carry = f4_add(carry, f4_mul(wbase, part));
            if (first < 64u) break;
            wbase = f4_mul(wbase, S.zL64);
            base -= 64;
        }
        if (lane == 0) {
            const Fp4 incl = f4_add(f4_mul(S.zL, carry), sc_ld4(agg_sh));
            sc_publish(st + SC_HDR + 8 * (size_t)tile + 4, incl);
            *reinterpret_cast<uint4*>(carry_sh) = make_uint4(carry.c[0], carry.c[1], carry.c[2], carry.c[3]);
        }
    }
    __syncthreads();
    // replay: the carry entering this lane, then its elements
    Fp4 zt = zlane;
    for (uint32_t w = 0; w < wv; ++w) zt = f4_mul_lz(zt, S.z512);
    cur = f4_add(excl, f4_mul_lz(zt, sc_ld4(carry_sh)));
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        if ((uint32_t)k < mine) {
            sh[9 * tid + k] = make_uint4(cur.c[0], cur.c[1], cur.c[2], cur.c[3]);
            cur = f4_add(f4_mul_cc(zc, f4_centre(cur)), v[k]);
        }
    }
    // the lane that owns coefficient 0 leaves the division with the remainder in `cur`
    if (tile == tiles - 1 && mine > 0 && SC_I * tid + mine == valid) *reinterpret_cast<uint4*>(rems + 4 * (size_t)q) = make_uint4(cur.c[0], cur.c[1], cur.c[2], cur.c[3]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint32_t u = (uint32_t)k * SC_T + tid;
        if (u < valid) poly[hi - 1 - u] = sh[sc_slot(u)];
    }
}

// prefix_products: io[i] <- io[0] * .. * io[i], `count` sequences of n elements back to back; forward, multiplicative
__global__ __launch_bounds__(SC_T) void pp_lookback_kernel(uint32_t* __restrict__ io, size_t n, uint32_t* __restrict__ state, uint32_t seq_stride,
                                                           uint32_t* __restrict__ clear, size_t clear_words) {
    __shared__ uint4 sh[SC_T * 9];
    __shared__ uint32_t wtot[4 * 4], carry_sh[4], agg_sh[4], sh_tile;
    const uint32_t q = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t* st = state + (size_t)q * seq_stride;
    const uint32_t tile = sc_begin(st, clear, clear_words, &sh_tile);
    uint4* seq = reinterpret_cast<uint4*>(io) + (size_t)q * n;
    const size_t lo = (size_t)tile * SC_TILE;
    const uint32_t valid = n - lo < (size_t)SC_TILE ? (uint32_t)(n - lo) : (uint32_t)SC_TILE;
    const uint4 one4 = make_uint4(MONT_ONE, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint32_t u = (uint32_t)k * SC_T + tid;
        sh[sc_slot(u)] = u < valid ? seq[lo + u] : one4;  // elements past the end are ones: they change no product
    }
    __syncthreads();
    Fp4 pre[SC_I];  // running products of this lane's elements
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint4 w = sh[9 * tid + k];
        const Fp4 x = Fp4{{w.x, w.y, w.z, w.w}};
        pre[k] = k ? f4_mul_lz(pre[k - 1], x) : x;
    }
    Fp4 I = pre[SC_I - 1];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const Fp4 prev = sc_shfl_up(I, 1 << k);
        if (lane >= (1u << k)) I = f4_mul_lz(I, prev);
    }
    if (lane == 63) *reinterpret_cast<uint4*>(wtot + 4 * wv) = make_uint4(I.c[0], I.c[1], I.c[2], I.c[3]);
    __syncthreads();
    Fp4 cw = f4_one();
    for (uint32_t w = 0; w < wv; ++w) cw = f4_mul_lz(cw, sc_ld4(wtot + 4 * w));
    Fp4 excl = sc_shfl_up(I, 1);
    if (lane == 0) excl = f4_one();
    excl = f4_mul_lz(excl, cw);
    if (tid == SC_T - 1) {
        const Fp4 agg = f4_mul(I, cw);
        *reinterpret_cast<uint4*>(agg_sh) = make_uint4(agg.c[0], agg.c[1], agg.c[2], agg.c[3]);
        if (tile > 0) sc_publish(st + SC_HDR + 8 * (size_t)tile, agg);
    }
    __syncthreads();
    if (wv == 0) {
        Fp4 carry = f4_one();
        int base = (int)tile - 1;
        while (tile > 0) {
            const int id = base - (int)lane;
            Fp4 val = f4_one();
            bool is_incl = true;
            if (id >= 0) {
                const uint32_t* slot = st + SC_HDR + 8 * (size_t)id;
                for (;;) {
                    if (sc_try_read(slot + 4, val)) { is_incl = true; break; }
                    if (sc_try_read(slot, val)) { is_incl = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const unsigned long long m = __ballot(is_incl);
            const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : 64u;
            Fp4 part = lane <= first ? val : f4_one();
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part = f4_mul_lz(part, sc_shfl_xor(part, d));
            carry = f4_mul_lz(carry, part);
            if (first < 64u) break;
            base -= 64;
        }
        if (lane == 0) {
            sc_publish(st + SC_HDR + 8 * (size_t)tile + 4, f4_mul(carry, sc_ld4(agg_sh)));
            *reinterpret_cast<uint4*>(carry_sh) = make_uint4(carry.c[0], carry.c[1], carry.c[2], carry.c[3]);
        }
    }
    __syncthreads();
    const C4 e = f4_centre(f4_mul_lz(excl, sc_ld4(carry_sh)));
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const Fp4 o = f4_mul_cc(e, f4_centre(pre[k]));
        sh[9 * tid + k] = make_uint4(o.c[0], o.c[1], o.c[2], o.c[3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SC_I; ++k) {
        const uint32_t u = (uint32_t)k * SC_T + tid;
        if (u < valid) seq[lo + u] = sh[sc_slot(u)];
    }
}

// two alternating state buffers; returns the one to use and what to clear in the other
const char* scan_state(bx_ctx* c, size_t words, uint32_t** use, uint32_t** clear, size_t* clear_words) {
    if (c->scan_cap < words) {
        BX_HIP(c, stream_wait(c));
        for (int b = 0; b < 2; ++b) {
            if (c->d_scan[b]) BX_HIP(c, hipFree(c->d_scan[b]));
            c->d_scan[b] = nullptr;
        }
        size_t cap = words < ((size_t)1 << 16) ? ((size_t)1 << 16) : words;
        for (int b = 0; b < 2; ++b) {
            BX_HIP(c, hipMalloc(&c->d_scan[b], cap * 4));
            BX_HIP(c, hipMemsetAsync(c->d_scan[b], 0, cap * 4, c->stream));
            c->scan_used[b] = 0;
        }
        c->scan_cap = cap;
    }
    const int b = c->scan_next;
    c->scan_next ^= 1;
    *use = c->d_scan[b];
    *clear = c->d_scan[b ^ 1];
    *clear_words = c->scan_used[b ^ 1];
    c->scan_used[b ^ 1] = 0;  // this launch clears it
    c->scan_used[b] = words;
    return nullptr;
}

static DivSeq div_seq(const uint32_t z[4]) {
    DivSeq s;
    s.z = Fp4{{z[0], z[1], z[2], z[3]}};
    Fp4 p = f4_pow(s.z, SC_I);
    for (int k = 0; k < 6; ++k) {
        s.zp[k] = p;
        p = f4_mul(p, p);
    }
    s.z512 = p;                               // z^(8 * 64)
    s.zL = f4_mul(f4_mul(p, p), f4_mul(p, p));  // z^2048
    p = s.zL;
    for (int k = 0; k < 6; ++k) {
        s.zLp[k] = p;
        p = f4_mul(p, p);
    }
    s.zL64 = p;
    return s;
}

// `count` polynomials of `size` AoS ext coefficients back to back, polynomial q divided in place by (x - zs[q]); rems[4q..] = remainder
const char* poly_divide_lookback(bx_ctx* c, uint32_t* polys, size_t size, size_t count, const uint32_t* zs, uint32_t* rems, const uint32_t* which) {
    const size_t tiles = (size + SC_TILE - 1) / SC_TILE;
    const uint32_t seq_stride = SC_HDR + 8 * (uint32_t)tiles;
    for (size_t q0 = 0; q0 < count; q0 += 8) {
        const size_t nq = count - q0 < 8 ? count - q0 : 8;
        DivArgs args;
        for (size_t q = 0; q < nq; ++q) {
            args.s[q] = div_seq(zs + 4 * (q0 + q));
            args.s[q].poly = which ? which[q0 + q] : (uint32_t)(q0 + q);
        }
        uint32_t *use, *clear;
        size_t clear_words;
        BX_TRY(scan_state(c, (size_t)seq_stride * nq, &use, &clear, &clear_words));
        hipLaunchKernelGGL(div_lookback_kernel, dim3((unsigned)tiles, (unsigned)nq), dim3(SC_T), 0, c->stream, polys, size, args, use,
                           seq_stride, (uint32_t)tiles, clear, clear_words, rems + 4 * q0);
        BX_LAUNCH_CHECK(c);
    }
    return nullptr;
}
const char* prefix_products_lookback(bx_ctx* c, uint32_t* io, size_t n, size_t count) {
    const size_t tiles = (n + SC_TILE - 1) / SC_TILE;
    const uint32_t seq_stride = SC_HDR + 8 * (uint32_t)tiles;
    uint32_t *use, *clear;
    size_t clear_words;
    BX_TRY(scan_state(c, (size_t)seq_stride * count, &use, &clear, &clear_words));
    hipLaunchKernelGGL(pp_lookback_kernel, dim3((unsigned)tiles, (unsigned)count), dim3(SC_T), 0, c->stream, io, n, use, seq_stride, clear, clear_words);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

}  // namespace bx

using namespace bx;

extern "C" const char* bx_poly_divide_batch(bx_ctx* c, bx_buf polys, size_t count, const uint32_t* zs, bx_buf rems_out) try {
    if (!c) return "bx_poly_divide_batch: null ctx";
    BX_REQUIRE(c, count >= 1 && count <= 65535 && polys.len % (4 * count) == 0, "poly_divide_batch: the buffer does not split into `count` AoS ext polynomials");
    BX_REQUIRE(c, zs != nullptr && rems_out.len >= 4 * count, "poly_divide_batch: one point and one remainder slot per polynomial");
    BX_REQUIRE(c, ((uintptr_t)polys.dptr & 15u) == 0 && ((uintptr_t)rems_out.dptr & 15u) == 0, "poly_divide_batch: buffers must be 16-byte aligned");
    BX_ENTER(c);
    const size_t size = polys.len / 4 / count;
    if (!size) return nullptr;
    OpScope op(c, "poly_divide", 8.0 * (double)polys.len);
    return poly_divide_lookback(c, (uint32_t*)polys.dptr, size, count, zs, (uint32_t*)rems_out.dptr, nullptr);
} BX_ABI_CATCH(c, "bx_poly_divide_batch")
extern "C" const char* bx_poly_divide_batch_indexed(bx_ctx* c, bx_buf polys, size_t n_polys, size_t count, const uint32_t* which, const uint32_t* zs,
                                                    bx_buf rems_out) try {
    if (!c) return "bx_poly_divide_batch_indexed: null ctx";
    BX_REQUIRE(c, n_polys >= 1 && n_polys <= polys.len / 4 && polys.len % (4 * n_polys) == 0, "poly_divide_batch_indexed: the buffer does not split into n_polys AoS ext polynomials");
    BX_REQUIRE(c, count <= 65535 && which != nullptr && zs != nullptr && rems_out.len >= 4 * count, "poly_divide_batch_indexed: one index, one point and one remainder slot per division");
    BX_REQUIRE(c, ((uintptr_t)polys.dptr & 15u) == 0 && ((uintptr_t)rems_out.dptr & 15u) == 0, "poly_divide_batch_indexed: buffers must be 16-byte aligned");
    for (size_t q = 0; q < count; ++q) BX_REQUIRE(c, which[q] < n_polys, "poly_divide_batch_indexed: polynomial index out of range");
    {
        std::vector<uint32_t> seen(which, which + count);
        std::sort(seen.begin(), seen.end());
        BX_REQUIRE(c, std::adjacent_find(seen.begin(), seen.end()) == seen.end(), "poly_divide_batch_indexed: a polynomial may be divided once per call");
    }
    BX_ENTER(c);
    const size_t size = polys.len / 4 / n_polys;
    if (!size || !count) return nullptr;
    OpScope op(c, "poly_divide", 32.0 * (double)size * (double)count);
    return poly_divide_lookback(c, (uint32_t*)polys.dptr, size, count, zs, (uint32_t*)rems_out.dptr, which);
} BX_ABI_CATCH(c, "bx_poly_divide_batch_indexed")
