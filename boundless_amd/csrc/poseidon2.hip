// poseidon2.hip — Poseidon2 (BabyBear, t=24, rate 16, 4+21+4 rounds, x^7) Merkle hashing for gfx950.
//
// Restates risc0_zkp::hal::Hal::{hash_rows, hash_fold} with the `poseidon2` hash suite and
// risc0_zkp::core::hash::poseidon2::{poseidon2_mix, unpadded_hash} (risc0-zkp 3.0.3, reference Cargo.lock:9155),
// reached from bento/crates/workflow/src/tasks/prove.rs:41-49 via MerkleTreeProver::new.
//
// MI355X design: the permutation is pure 32-bit integer VALU work (~1.4k Montgomery products per
// 64 absorbed bytes) — no MFMA shape exists for it and it is nowhere near HBM-bound, so the kernel is organised
// around registers, not memory: one matrix row per lane, the 24-word state in VGPRs for the whole sponge,
// round constants fetched through wave-uniform (scalar) loads, loads of the column-major matrix coalesced
// across the 64 lanes of a wave (lane r reads matrix[c*rows + r]).
#include "ctx.hpp"
#include "poseidon2_arith.hpp"
#include "../../include/bx_image.h"

namespace bx {

constexpr int CELLS = 24, RATE = 16, RF_HALF = 4, RP = 21;
constexpr int DIAG_OFF = 216;  // diagonal starts 16-byte aligned in the device parameter table

// ---------------------------------------------------------------------------------------------------------------
// Instruction budget.  The kernel is VALU-issue-bound and on gfx950 a 32-bit multiply-class op (v_mad_[iu]64_[iu]32,
// v_mul_lo/hi_u32, v_min_u32, v_lshl_add_u64) costs ~1.9 ns per wave and SIMD and a plain add/sub/and/mov ~1.05 ns
// (profiles/r01_microbench2_instr_cost.jsonl).  The permutation is organised to minimise that weighted instruction count,
// not multiplications (every bound below is asserted on the host by tests/host_arith_check.cpp, which compiles
// poseidon2_arith.hpp for the CPU and compares the whole permutation with the plain canonical implementation):
//   * cells are SIGNED 32-bit integers, only bounded in magnitude: the signed Montgomery reduction maps |t| <= 1.2 P^2 to
//     |r| < |t|/2^32 + P/2, so products of cells stay below 0.95 P by themselves and the permutation contains no
//     conditional subtraction at all except for the 24 words that leave it;
//   * linear layers run UNREDUCED in 64 bits (v_mad_i64_i32 multiplies by the small matrix entries and accumulates in one
//     instruction; external-layer outputs < 2^38, internal-layer sum < 2^37);
//   * the return to 32 bits is one bare REDC per cell with the next round constant in its accumulator:
//     x = REDC(y + rc') [REDC = * 2^-32 mod P].  The factor 2^-32 is not compensated: the cells of a round share a known
//     representation factor R^e that the (homogeneous) linear layers carry along and the S-box maps to R^(7e-6); the round
//     constants are stored pre-scaled, and only the two layers that must hand back Montgomery form (before the internal
//     rounds, and at the end) multiply by a correction constant (poseidon2_arith.hpp: representation tracking).
// Results are congruent mod P at every step and the words that leave the permutation are canonical, so the output is
// bit-identical to the reduce-everywhere form (6.6 k VALU instructions per permutation by PMC, down from ~14.5 k).
//
// Magnitude bounds (rho = P / 2^32 = 0.46875; sredc(t) in [t/2^32 - P/2, t/2^32 + P/2); int32 holds 1.0667 P):
//   redc64s output                     |x| <= 57 + P/2                           (external rounds' S-box input)
//   sbox7s(|x| <= 0.945 P)             |x2| < 0.919 P, |x3| < 0.907 P, |x4| < 0.896 P, |x7| < 0.881 P; products <= 0.893 P^2 < 1.2 P^2
//   m_ext64s(any int32 cells)          rows sum to <= 112, |y| <= 112 * 2^31 < 2^38
//   red64ks<MID>/<END> output          |x| < K1 + 26 + P/2  = 0.626 P / 0.555 P    (K1 = R^2801, R^400 mod P)
//   internal rounds                    cell i: sredc(d_i s_i + sum_r [+ rc]) with d_i < P: B -> rho B + P/2 + 2 has its fixed
//                                      point at 0.9412 P, 0.945 P is invariant; |sum| < 22.7 P < 2^37; |sum_r| < 0.791 P thanks to
//                                      the balanced low word of sum.
// ---------------------------------------------------------------------------------------------------------------
// Device parameter table (round constants scaled by p2_rc_scale(i) so that they can ride in a REDC accumulator of the
// round's representation): [0,96) external rounds 0-3 | [96,117) internal rounds | [117,213) external rounds 4-7 |
// [216,240) internal diagonal (plain Montgomery form, used as a multiplier; 16-byte aligned).
// Input: cells < P (canonical).  Output: canonical.
__device__ __forceinline__ void poseidon2_mix(uint32_t* io, const uint32_t* __restrict__ prm) {
    i32 s[CELLS];
    i64 y[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; ++i) s[i] = (i32)io[i];
    // initial external layer; round-0 constants ride in the reduction
    m_ext64s(s, y);
    redc64s_all(y, prm, s);
    // external rounds 0..3: S-box, layer, reduction with the NEXT round's constants (after round 3 only cell 0 has one:
    // the first internal round's)
#pragma unroll 1
    for (int r = 0; r < RF_HALF; ++r) {
        sbox7s_n<CELLS / 2>(s);  // two halves: 12 independent chains are enough ILP and halve the live 64-bit temporaries
        sbox7s_n<CELLS / 2>(s + CELLS / 2);
        m_ext64s(s, y);
        if (r < RF_HALF - 1) {
            redc64s_all(y, prm + (r + 1) * CELLS, s);
        } else {  // back to the Montgomery representation for the internal rounds
            red64ks_all<K1_MID, K2_MID, true>(y, prm[96], s);
        }
    }
    // internal rounds: cells[i] = sum + diag[i]*cells[i].  sum is accumulated in 64 bits, turned into
    // sum_r = sum * 2^32 mod P by one reduction, and rides in each cell's REDC accumulator (cell 0: together with the next
    // constant).
    // The 24 diagonal words are wave-uniform.  Loaded as scalars the compiler re-issues the s_load (+ s_waitcnt stall) in every
    // internal round; held in VGPRs they cost 24 registers.  So: six 16-byte vector loads through an offset the compiler cannot prove uniform (an opaque zero), issued once,
    // then v_readfirstlane into SGPRs — values the compiler can neither rematerialise from memory nor widen, and that the
    // multiply-adds read as their one scalar operand.
    uint32_t diag[CELLS];
    {
        uint32_t zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        const uint4* dp = reinterpret_cast<const uint4*>(prm + DIAG_OFF + zero);
#pragma unroll
        for (int i = 0; i < CELLS / 4; ++i) {
            const uint4 v = dp[i];
            diag[4 * i] = __builtin_amdgcn_readfirstlane(v.x);
            diag[4 * i + 1] = __builtin_amdgcn_readfirstlane(v.y);
            diag[4 * i + 2] = __builtin_amdgcn_readfirstlane(v.z);
            diag[4 * i + 3] = __builtin_amdgcn_readfirstlane(v.w);
        }
    }
#pragma unroll 1
    for (int r = 0; r < RP - 1; ++r) {
        internal_round<false>(s, diag, prm + 97 + r);
    }
    internal_round<true>(s, diag, prm + 117);  // external round 4's constants ride in the last internal round
    // external rounds 4..7
#pragma unroll 1
    for (int r = 0; r < RF_HALF; ++r) {
        sbox7s_n<CELLS / 2>(s);  // two halves: 12 independent chains are enough ILP and halve the live 64-bit temporaries
        sbox7s_n<CELLS / 2>(s + CELLS / 2);
        m_ext64s(s, y);
        if (r < RF_HALF - 1) {
            redc64s_all(y, prm + 117 + (r + 1) * CELLS, s);
        } else {
            red64ks_all<K1_END, K2_END, false>(y, 0u, s);
#pragma unroll
            for (int i = 0; i < CELLS; ++i) io[i] = canon(s[i]);  // canonical Montgomery words leave the permutation
        }
    }
}

// Occupancy: the stage-wise order fixes the register pressure (the statements are volatile), so the stages are 12 cells wide
// (3 four-cell groups in the linear layer): 108-120 VGPRs, four waves per SIMD without spills.  Five (96 VGPRs) spills.
// hash_rows: lane = row.  matrix is column-major rows x cols.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_rows_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ matrix,
                                                        const uint32_t* __restrict__ prm, uint32_t rows, uint32_t cols) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint32_t s[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; ++i) s[i] = 0u;
    const uint32_t* p = matrix + r;
    uint32_t c = 0;
    for (; c + RATE <= cols; c += RATE) {
#pragma unroll
        for (int i = 0; i < RATE; ++i) s[i] = p[(size_t)(c + i) * rows];
        poseidon2_mix(s, prm);
    }
    if (c < cols || cols == 0) {
#pragma unroll
        for (int i = 0; i < RATE; ++i) s[i] = (c + i < cols) ? p[(size_t)(c + i) * rows] : 0u;
        poseidon2_mix(s, prm);
    }
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)r * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// hash_fold: lane = output node; io[out+i] = H(io[in+2i] || io[in+2i+1]).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_fold_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ prm,
                                                        uint32_t input_size, uint32_t output_size) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= output_size) return;
    const uint4* src = reinterpret_cast<const uint4*>(io + ((size_t)input_size + 2 * (size_t)i) * 8);
    uint32_t s[CELLS];
    uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
    s[4] = v1.x; s[5] = v1.y; s[6] = v1.z; s[7] = v1.w;
    s[8] = v2.x; s[9] = v2.y; s[10] = v2.z; s[11] = v2.w;
    s[12] = v3.x; s[13] = v3.y; s[14] = v3.z; s[15] = v3.w;
#pragma unroll
    for (int k = 16; k < CELLS; ++k) s[k] = 0u;
    poseidon2_mix(s, prm);
    uint4* o = reinterpret_cast<uint4*>(io + ((size_t)output_size + i) * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// hash_fold with an indirection: out[j] = H(in[sel[2j]] || in[sel[2j+1]]).  The levels of a SPARSE Merkle tree (the zkVM
// memory image, bx_image.h: 2^22 leaves of which a few hundred exist) are folded with it: the host lists, per level, which
// two digests of the level below (or that level's all-zero digest) feed each surviving parent.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_fold_indexed_kernel(uint32_t* out, const uint32_t* in,  // out may be a range of the pool `in` names (no sel entry inside it): not __restrict__
                                                                const uint32_t* __restrict__ sel, const uint32_t* __restrict__ prm,
                                                                uint32_t count, uint32_t n_in) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    uint2 pr = reinterpret_cast<const uint2*>(sel)[j];
    pr.x = pr.x < n_in ? pr.x : n_in - 1;  // sel comes from the caller: an entry beyond the pool must not become an out-of-bounds read
    pr.y = pr.y < n_in ? pr.y : n_in - 1;
    const uint4* a = reinterpret_cast<const uint4*>(in + (size_t)pr.x * 8);
    const uint4* b = reinterpret_cast<const uint4*>(in + (size_t)pr.y * 8);
    uint32_t s[CELLS];
    uint4 v0 = a[0], v1 = a[1], v2 = b[0], v3 = b[1];
    s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
    s[4] = v1.x; s[5] = v1.y; s[6] = v1.z; s[7] = v1.w;
    s[8] = v2.x; s[9] = v2.y; s[10] = v2.z; s[11] = v2.w;
    s[12] = v3.x; s[13] = v3.y; s[14] = v3.z; s[15] = v3.w;
#pragma unroll
    for (int k = 16; k < CELLS; ++k) s[k] = 0u;
    poseidon2_mix(s, prm);
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)j * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// Several levels of a LARGE layer in one launch, every lane busy: lane j folds the 2^L consecutive input digests
// [2^L j, 2^L (j+1)) depth first — at most L digests are alive while the next pair is hashed — and writes every intermediate
// node to its place in the tree.  2^L - 1 permutations per lane, so the lane efficiency is that of the per-layer kernel, but a
// 2^22-leaf tree needs two launches down to the 2^17-node layer where the latency-bound kernel takes over instead of five, and
// the intermediate layers are never read back.
__device__ __forceinline__ void fold_pair(uint32_t* out8, const uint32_t* a8, const uint32_t* b8, const uint32_t* __restrict__ prm) {
    uint32_t s[CELLS];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = a8[k], s[8 + k] = b8[k];
#pragma unroll
    for (int k = 16; k < CELLS; ++k) s[k] = 0u;
    poseidon2_mix(s, prm);
#pragma unroll
    for (int k = 0; k < 8; ++k) out8[k] = s[k];
}
__device__ __forceinline__ void ld_digest(uint32_t* d, const uint32_t* p) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void st_digest(uint32_t* p, const uint32_t* d) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(d[0], d[1], d[2], d[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(d[4], d[5], d[6], d[7]);
}
// node `idx` of the layer `lvl` levels above the input layer (lvl = 1: parents of inputs), computed depth first
template <int LVL>
__device__ __forceinline__ void fold_subtree(uint32_t* out8, uint32_t* __restrict__ io, const uint32_t* __restrict__ prm, size_t input_size, size_t idx) {
    uint32_t a[8], b[8];
    if constexpr (LVL == 1) {
        ld_digest(a, io + (input_size + 2 * idx) * 8);
        ld_digest(b, io + (input_size + 2 * idx + 1) * 8);
    } else {
        fold_subtree<LVL - 1>(a, io, prm, input_size, 2 * idx);
        fold_subtree<LVL - 1>(b, io, prm, input_size, 2 * idx + 1);
    }
    fold_pair(out8, a, b, prm);
    st_digest(io + ((input_size >> LVL) + idx) * 8, out8);
}
template <int L>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_fold_deep_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ prm,
                                                                                               uint32_t input_size) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (input_size >> L)) return;
    uint32_t top[8];
    fold_subtree<L>(top, io, prm, input_size, j);
}

// Small layers in one launch: a workgroup owns `per_wg` (<= 512) consecutive input digests and folds them `levels`
// levels deep through LDS, writing every intermediate layer.  Used once a layer no longer fills the chip, where each
// separate launch would cost a full single-wave permutation latency.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_fold_multi_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ prm,
                                                              uint32_t input_size, uint32_t per_wg, int levels) {
    __shared__ uint32_t sh[256 * 8];
    const uint32_t tid = threadIdx.x;
    uint32_t width = per_wg >> 1;  // outputs of this workgroup at the current level
    uint32_t out_size = input_size >> 1;
    uint32_t s[CELLS];
    if (tid < width) {
        const uint32_t i = blockIdx.x * width + tid;
        const uint4* src = reinterpret_cast<const uint4*>(io + ((size_t)input_size + 2 * (size_t)i) * 8);
        uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
        s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
        s[4] = v1.x; s[5] = v1.y; s[6] = v1.z; s[7] = v1.w;
        s[8] = v2.x; s[9] = v2.y; s[10] = v2.z; s[11] = v2.w;
        s[12] = v3.x; s[13] = v3.y; s[14] = v3.z; s[15] = v3.w;
#pragma unroll
        for (int k = 16; k < CELLS; ++k) s[k] = 0u;
        poseidon2_mix(s, prm);
        uint4* o = reinterpret_cast<uint4*>(io + ((size_t)out_size + i) * 8);
        o[0] = make_uint4(s[0], s[1], s[2], s[3]);
        o[1] = make_uint4(s[4], s[5], s[6], s[7]);
    }
    for (int lvl = 1; lvl < levels; ++lvl) {
        if (tid < width) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sh[tid * 8 + k] = s[k];
        }
        __syncthreads();
        width >>= 1;
        out_size >>= 1;
        if (tid < width) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s[k] = sh[tid * 16 + k];
#pragma unroll
            for (int k = 16; k < CELLS; ++k) s[k] = 0u;
            poseidon2_mix(s, prm);
            uint4* o = reinterpret_cast<uint4*>(io + ((size_t)out_size + blockIdx.x * width + tid) * 8);
            o[0] = make_uint4(s[0], s[1], s[2], s[3]);
            o[1] = make_uint4(s[4], s[5], s[6], s[7]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same permutation spread over a QUAD of lanes, for the layers that no longer fill the chip.  There the cost of a level
// is the latency of one permutation on one lane (6.6 k dependent-ish instructions, ~12 us), not throughput; four lanes per
// permutation cut the per-lane instruction stream to ~2.3 k.  Lane q of a quad holds cells 4k + q (slot k = 0..5):
//   * an external layer's M4 block k is the slot-k values of the four lanes: each lane computes its own row of M4 from the
//     three other lanes' values (v_mov_b32_dpp quad_perm rotations) with per-lane coefficients; the column sums T_j are lane-local;
//   * the S-boxes and the returns to 32 bits are per cell and reuse the stage-wise helpers at width 6;
//   * an internal round sums six cells per lane and all-reduces the 64-bit partial sums over the quad with two DPP steps;
//     every lane runs the S-box of its slot 0 and only lane 0 (cell 0) keeps it.
// Per-lane round constants and diagonal entries come from a copy of the parameter table in LDS.  The arithmetic per cell is
// exactly poseidon2_mix's (same representation tracking, same bounds), so the words that leave are the same words.
constexpr int QS = CELLS / 4;  // slots per lane
__device__ __forceinline__ i32 quad_rot1(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0x39, 0xf, 0xf, false); }  // from lane (q+1)&3
__device__ __forceinline__ i32 quad_rot2(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); }  // (q+2)&3
__device__ __forceinline__ i32 quad_rot3(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0x93, 0xf, 0xf, false); }  // (q+3)&3
__device__ __forceinline__ i32 quad_xor1(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ i64 add_u32v(i64 c, uint32_t k, int alt) {  // c + k, k a per-lane unsigned word
#if defined(__HIP_DEVICE_COMPILE__)
    i64 r;
    BX_MAD_ASM("v_mad_u64_u32", "%1, 1, %2", "v"(k), "v"(c));
    return r;
#else
    (void)alt;
    return c + (i64)k;
#endif
}
struct QuadCoef {
    i32 c0, c1, c3;  // this lane's row of M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]], rotated to start at its own column (c2 = 1)
};
__device__ __forceinline__ void m_ext_quad(const i32* s, i64* y, const QuadCoef& m) {
    i64 t = 0;
#pragma unroll
    for (int k = 0; k < QS; ++k) {
        const i32 x1 = quad_rot1(s[k]), x2 = quad_rot2(s[k]), x3 = quad_rot3(s[k]);
        i64 w = smul(s[k], m.c0, k);
        w = smad(x1, m.c1, w, k + 1);
        w = smadc<1>(x2, w, k + 2);
        w = smad(x3, m.c3, w, k + 3);
        y[k] = w;
        t += w;
    }
#pragma unroll
    for (int k = 0; k < QS; ++k) y[k] += t;
}
__device__ __forceinline__ void redc_quad(const i64* y, const uint32_t* __restrict__ rc /* LDS, this lane's entries 4k apart */, i32* s) {
    i64 acc[QS];
#pragma unroll
    for (int k = 0; k < QS; ++k) acc[k] = add_u32v(y[k], rc[4 * k], k);
    sredc_n<QS>(acc, s);
}
template <uint32_t K1, uint32_t K2>
__device__ __forceinline__ void redk_quad(const i64* y, uint32_t add0 /* this lane's addend for slot 0 (0 = none) */, i32* s) {
    i64 acc[QS];
#pragma unroll
    for (int k = 0; k < QS; ++k) acc[k] = umul_k((uint32_t)y[k], K1, k);
    acc[0] = add_u32v(acc[0], add0, 0);
#pragma unroll
    for (int k = 0; k < QS; ++k) acc[k] = smad_k((i32)(y[k] >> 32), K2, acc[k], k + 1);
    sredc_n<QS>(acc, s);
}
template <bool RC_ALL>
__device__ __forceinline__ void internal_round_quad(i32* s, const i32* diag, uint32_t rc0 /* lane 0: the round's constant, else 0 */,
                                                    const uint32_t* __restrict__ rc_all, uint32_t keep_mask) {
    const i32 sb = sbox7s(s[0]);
    s[0] = (i32)(((uint32_t)sb & keep_mask) | ((uint32_t)s[0] & ~keep_mask));  // only cell 0 has an S-box in these rounds
    i64 pa = smulc<1>(s[0], 0);
#pragma unroll
    for (int k = 1; k < QS; ++k) pa = smadc<1>(s[k], pa, k);
    {  // all-reduce over the quad
        i64 o = (i64)(((uint64_t)(uint32_t)quad_xor1((i32)(pa >> 32)) << 32) | (uint32_t)quad_xor1((i32)(uint32_t)pa));
        pa += o;
        o = (i64)(((uint64_t)(uint32_t)quad_rot2((i32)(pa >> 32)) << 32) | (uint32_t)quad_rot2((i32)(uint32_t)pa));
        pa += o;
    }
    const i64 c = smulc<1>(internal_sum_rs(pa), 1);
    i64 tt[QS];
#pragma unroll
    for (int k = 0; k < QS; ++k) {
        if (RC_ALL) tt[k] = smad(s[k], diag[k], add_u32v(c, rc_all[4 * k], 2 * k), 2 * k + 1);
        else if (k == 0) tt[k] = smad(s[k], diag[k], add_u32v(c, rc0, 0), 1);
        else tt[k] = smad(s[k], diag[k], c, k + 1);
    }
    sredc_n<QS>(tt, s);
}
// in: this lane's cells (canonical); out: this lane's cells (canonical).  shp = the parameter table in LDS.
__device__ __forceinline__ void poseidon2_mix_quad(uint32_t* io, const uint32_t* __restrict__ shp, uint32_t q) {
    i32 s[QS];
    i64 y[QS];
    const bool odd = q & 1;
    const QuadCoef m{odd ? 6 : 5, odd ? 1 : 7, odd ? 4 : 3};
    const uint32_t keep = q == 0 ? 0xffffffffu : 0u;
#pragma unroll
    for (int k = 0; k < QS; ++k) s[k] = (i32)io[k];
    m_ext_quad(s, y, m);
    redc_quad(y, shp + q, s);
#pragma unroll 1
    for (int r = 0; r < RF_HALF; ++r) {
        sbox7s_n<QS>(s);
        m_ext_quad(s, y, m);
        if (r < RF_HALF - 1) redc_quad(y, shp + (r + 1) * CELLS + q, s);
        else redk_quad<K1_MID, K2_MID>(y, shp[96] & keep, s);
    }
    i32 diag[QS];
#pragma unroll
    for (int k = 0; k < QS; ++k) diag[k] = (i32)shp[DIAG_OFF + 4 * k + q];
#pragma unroll 1
    for (int r = 0; r < RP - 1; ++r) internal_round_quad<false>(s, diag, shp[97 + r] & keep, nullptr, keep);
    internal_round_quad<true>(s, diag, 0u, shp + 117 + q, keep);
#pragma unroll 1
    for (int r = 0; r < RF_HALF; ++r) {
        sbox7s_n<QS>(s);
        m_ext_quad(s, y, m);
        if (r < RF_HALF - 1) redc_quad(y, shp + 117 + (r + 1) * CELLS + q, s);
        else redk_quad<K1_END, K2_END>(y, 0u, s);
    }
#pragma unroll
    for (int k = 0; k < QS; ++k) io[k] = canon(s[k]);
}

// Small Merkle layers, four lanes per node: a workgroup owns `per_wg` (<= 512) consecutive input digests and folds them
// `levels` levels deep through LDS, writing every intermediate layer (as hash_fold_multi_kernel does with one lane per node).
__global__ __launch_bounds__(1024) void hash_fold_quad_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ prm, uint32_t input_size,
                                                              uint32_t per_wg, int levels) {
    __shared__ uint32_t sh[256 * 8];
    __shared__ uint32_t shp[DIAG_OFF + 24];
    const uint32_t tid = threadIdx.x, q = tid & 3, node = tid >> 2;
    for (uint32_t i = tid; i < DIAG_OFF + 24; i += blockDim.x) shp[i] = prm[i];
    uint32_t width = per_wg >> 1;  // nodes this workgroup produces at the current level
    uint32_t out_size = input_size >> 1;
    for (int lvl = 0; lvl < levels; ++lvl) {
        const bool active = node < width;
        uint32_t s[QS];
        if (active) {
            if (lvl == 0) {
                const uint32_t* src = io + ((size_t)input_size + 2 * ((size_t)blockIdx.x * width + node)) * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = src[4 * k + q];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = sh[node * 16 + 4 * k + q];
            }
            s[4] = 0u;
            s[5] = 0u;
        }
        __syncthreads();  // the level's inputs are in registers (and, at level 0, the parameter table is in LDS)
        if (active) {
            poseidon2_mix_quad(s, shp, q);
            uint32_t* o = io + ((size_t)out_size + (size_t)blockIdx.x * width + node) * 8;
            o[q] = s[0];
            o[4 + q] = s[1];
            sh[node * 8 + q] = s[0];
            sh[node * 8 + 4 + q] = s[1];
        }
        __syncthreads();
        width >>= 1;
        out_size >>= 1;
    }
}

// One step of the Fiat-Shamir transcript on the device (transcript.hpp's Transcript::commit / random_elem, i.e. risc0_zkp's
// Poseidon2Rng): four lanes hold the 24 cells as poseidon2_mix_quad wants them (lane q: cells 4k + q).  A step is 1-2 sequential
// permutations (8.7 us each on a quad) against ~150 us for the host round trip it replaces; the long sponges of the transcript
// (coeff_u, the final coefficients) stay on the host, where a permutation takes ~1 us.
__global__ __launch_bounds__(64) void transcript_step_kernel(uint32_t* __restrict__ state, const uint32_t* __restrict__ digests, uint32_t n_commit,
                                                            uint32_t* __restrict__ out, uint32_t n_elems, const uint32_t* __restrict__ prm) {
    __shared__ uint32_t shp[DIAG_OFF + 24];
    const uint32_t tid = threadIdx.x, q = tid & 3;
    for (uint32_t i = tid; i < DIAG_OFF + 24; i += blockDim.x) shp[i] = prm[i];
    __syncthreads();
    if (tid >= 4) return;
    uint32_t s[QS];
#pragma unroll
    for (int k = 0; k < QS; ++k) s[k] = state[4 * k + q];
    uint32_t pool = state[24];
    for (uint32_t i = 0; i < n_commit; ++i) {
        if (pool != 0) {  // elements were handed out since the last permutation: the pool is discarded first
            poseidon2_mix_quad(s, shp, q);
            pool = 0;
        }
        s[0] = fp_add(s[0], digests[8 * i + q]);
        s[1] = fp_add(s[1], digests[8 * i + 4 + q]);
        poseidon2_mix_quad(s, shp, q);
    }
    for (uint32_t e = 0; e < n_elems; ++e) {
        if (pool == 16) {
            poseidon2_mix_quad(s, shp, q);
            pool = 0;
        }
        const uint32_t slot = pool >> 2;
        const uint32_t v = slot == 0 ? s[0] : slot == 1 ? s[1] : slot == 2 ? s[2] : s[3];
        if ((pool & 3u) == q) out[e] = v;
        ++pool;
    }
#pragma unroll
    for (int k = 0; k < QS; ++k) state[4 * k + q] = s[k];
    if (q == 0) state[24] = pool;
}

const char* poseidon2_upload_params(bx_ctx* c) {
    uint32_t h[DIAG_OFF + 24] = {0};
    // round constants ride in REDC accumulators, pre-scaled to the representation of the round they are added in
    for (int i = 0; i < 213; ++i) h[i] = (uint32_t)((uint64_t)(c->h_rc[i] % P) * p2_rc_scale(i) % P);
    for (int i = 0; i < 24; ++i) h[DIAG_OFF + i] = fp_encode(c->h_diag[i]);
    if (!c->d_p2) BX_HIP(c, hipMalloc(&c->d_p2, sizeof h));
    BX_HIP(c, hipMemcpyAsync(c->d_p2, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    BX_HIP(c, stream_wait(c));
    return nullptr;
}

static const char* launch_hash_rows(bx_ctx* c, uint32_t* out, const uint32_t* matrix, size_t rows, size_t cols) {
    if (rows == 0) return nullptr;
    unsigned bs = (unsigned)c->hash_rows_block;
    hipLaunchKernelGGL(hash_rows_kernel, dim3((unsigned)((rows + bs - 1) / bs)), dim3(bs), 0, c->stream, out, matrix, c->d_p2,
                       (uint32_t)rows, (uint32_t)cols);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
static const char* launch_hash_fold(bx_ctx* c, uint32_t* io, size_t input_size, size_t output_size) {
    if (output_size == 0) return nullptr;
    hipLaunchKernelGGL(hash_fold_kernel, dim3((unsigned)((output_size + 255) / 256)), dim3(256), 0, c->stream, io, c->d_p2,
                       (uint32_t)input_size, (uint32_t)output_size);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

}  // namespace bx

using namespace bx;

extern "C" const char* bx_poseidon2_set_params(bx_ctx* c, const uint32_t* rc213, const uint32_t* diag24) try {
    if (!c) return "bx_poseidon2_set_params: null ctx";
    BX_REQUIRE(c, rc213 && diag24, "poseidon2_set_params: null table");
    // a prover snapshots the table for its host transcript at create time and bx_verify_segment uses the compiled-in one:
    // changing the device table under a live prover would make its trees and its transcript disagree
    BX_REQUIRE(c, c->live_provers == 0, "poseidon2_set_params: destroy the provers of this ctx first (their transcripts hold the old table)");
    BX_ENTER(c);
    for (int i = 0; i < 213; ++i) c->h_rc[i] = rc213[i] % P;
    for (int i = 0; i < 24; ++i) c->h_diag[i] = diag24[i] % P;
    return poseidon2_upload_params(c);
} BX_ABI_CATCH(c, "bx_poseidon2_set_params")
extern "C" const char* bx_poseidon2_get_params(bx_ctx* c, uint32_t* rc213, uint32_t* diag24) try {
    if (!c) return "bx_poseidon2_get_params: null ctx";
    for (int i = 0; i < 213; ++i) rc213[i] = c->h_rc[i];
    for (int i = 0; i < 24; ++i) diag24[i] = c->h_diag[i];
    return nullptr;
} BX_ABI_CATCH(c, "bx_poseidon2_get_params")

extern "C" const char* bx_hash_rows(bx_ctx* c, bx_buf out, bx_buf matrix) try {
    if (!c) return "bx_hash_rows: null ctx";
    BX_REQUIRE(c, out.len % 8 == 0, "hash_rows: digest buffer length not a multiple of 8 words");
    size_t rows = out.len / 8;
    BX_REQUIRE(c, rows > 0 && matrix.len % rows == 0, "hash_rows: matrix.len not a multiple of rows");
    BX_REQUIRE(c, rows <= 0xffffffffu, "hash_rows: too many rows");
    size_t cols = matrix.len / rows;
    BX_ENTER(c);
    OpScope op(c, "hash_rows", 4.0 * (double)matrix.len + 32.0 * (double)rows);
    return launch_hash_rows(c, (uint32_t*)out.dptr, (const uint32_t*)matrix.dptr, rows, cols);
} BX_ABI_CATCH(c, "bx_hash_rows")

extern "C" const char* bx_hash_fold(bx_ctx* c, bx_buf io, size_t input_size, size_t output_size) try {
    if (!c) return "bx_hash_fold: null ctx";
    BX_REQUIRE(c, output_size <= io.len / 32 && input_size == 2 * output_size, "hash_fold: input_size must be 2*output_size, and the buffer hold 2*input_size digests");
    BX_ENTER(c);
    OpScope op(c, "hash_fold", 96.0 * (double)output_size);
    return launch_hash_fold(c, (uint32_t*)io.dptr, input_size, output_size);
} BX_ABI_CATCH(c, "bx_hash_fold")

extern "C" const char* bx_hash_fold_indexed(bx_ctx* c, bx_buf out, bx_buf in, bx_buf sel, size_t count) try {
    if (!c) return "bx_hash_fold_indexed: null ctx";
    BX_REQUIRE(c, count <= out.len / 8 && count <= sel.len / 2, "hash_fold_indexed: out or sel too small");
    BX_REQUIRE(c, count <= 0xffffffffu && in.len / 8 <= 0xffffffffu, "hash_fold_indexed: too many digests");
    if (count == 0) return nullptr;
    BX_REQUIRE(c, out.dptr && in.dptr && sel.dptr && in.len >= 8, "hash_fold_indexed: null or empty buffer");
    BX_ENTER(c);
    OpScope op(c, "hash_fold_indexed", 104.0 * (double)count);
    hipLaunchKernelGGL(hash_fold_indexed_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)out.dptr,
                       (const uint32_t*)in.dptr, (const uint32_t*)sel.dptr, c->d_p2, (uint32_t)count, (uint32_t)(in.len / 8));
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_hash_fold_indexed")

// every layer above the leaves nodes[rows .. 2 rows), down to the root nodes[1]
static const char* merkle_fold_layers(bx_ctx* c, uint32_t* n, size_t rows) {
    // Large layers: full-utilisation launches (lane = output node, or a lane folds 2^L inputs depth first).  Once a layer no
    // longer fills the chip (<= fold_fuse_below inputs) the remaining levels are latency-bound, so a workgroup folds 512 inputs
    // nine levels deep through LDS in one launch.
    size_t size = rows;
    const size_t fuse_below = (size_t)c->fold_fuse_below;
    while (size > 1) {
        if (size <= fuse_below) {
            const size_t cap = c->fold_quad ? (size_t)c->fold_quad_wg : 512;
            size_t per_wg = size < cap ? size : cap;
            int levels = ilog2(per_wg);
            if (c->fold_quad)
                hipLaunchKernelGGL(hash_fold_quad_kernel, dim3((unsigned)(size / per_wg)), dim3((unsigned)(2 * per_wg)), 0, c->stream, n,
                                   c->d_p2, (uint32_t)size, (uint32_t)per_wg, levels);
            else
                hipLaunchKernelGGL(hash_fold_multi_kernel, dim3((unsigned)(size / per_wg)), dim3(256), 0, c->stream, n, c->d_p2,
                                   (uint32_t)size, (uint32_t)per_wg, levels);
            BX_LAUNCH_CHECK(c);
            size >>= levels;
        } else if (c->fold_deep >= 3 && (size >> 3) >= (size_t)c->fold_deep_min_lanes && (size >> 3) >= fuse_below) {
            // three levels per launch while that still leaves a lane per SIMD slot of the chip and lands above the fused kernel's range
            hipLaunchKernelGGL(hash_fold_deep_kernel<3>, dim3((unsigned)(((size >> 3) + 255) / 256)), dim3(256), 0, c->stream, n, c->d_p2, (uint32_t)size);
            BX_LAUNCH_CHECK(c);
            size >>= 3;
        } else if (c->fold_deep >= 2 && (size >> 2) >= (size_t)c->fold_deep_min_lanes && (size >> 2) >= fuse_below) {
            hipLaunchKernelGGL(hash_fold_deep_kernel<2>, dim3((unsigned)(((size >> 2) + 255) / 256)), dim3(256), 0, c->stream, n, c->d_p2, (uint32_t)size);
            BX_LAUNCH_CHECK(c);
            size >>= 2;
        } else {
            BX_TRY(launch_hash_fold(c, n, size, size / 2));
            size >>= 1;
        }
    }
    return nullptr;
}

extern "C" const char* bx_merkle_build(bx_ctx* c, bx_buf nodes, bx_buf matrix, size_t rows) try {
    if (!c) return "bx_merkle_build: null ctx";
    BX_REQUIRE(c, is_pow2(rows) && rows <= nodes.len / 16 && nodes.len == 16 * rows, "merkle_build: nodes must hold 2*rows digests, rows a power of two");
    BX_REQUIRE(c, matrix.len % rows == 0, "merkle_build: matrix.len not a multiple of rows");
    BX_ENTER(c);
    uint32_t* n = (uint32_t*)nodes.dptr;
    {
        OpScope op(c, "hash_rows", 4.0 * (double)matrix.len + 32.0 * (double)rows);
        BX_TRY(launch_hash_rows(c, n + 8 * rows, (const uint32_t*)matrix.dptr, rows, matrix.len / rows));
    }
    OpScope op(c, "hash_fold", 96.0 * (double)(rows - 1));
    return merkle_fold_layers(c, n, rows);
} BX_ABI_CATCH(c, "bx_merkle_build")
// Extension: the fold half of bx_merkle_build alone — the leaves are already in nodes[rows .. 2 rows).
extern "C" const char* bx_merkle_fold(bx_ctx* c, bx_buf nodes, size_t rows) try {
    if (!c) return "bx_merkle_fold: null ctx";
    BX_REQUIRE(c, is_pow2(rows) && rows <= nodes.len / 16 && nodes.len == 16 * rows, "merkle_fold: nodes must hold 2*rows digests, rows a power of two");
    BX_ENTER(c);
    OpScope op(c, "hash_fold", 96.0 * (double)(rows - 1));
    return merkle_fold_layers(c, (uint32_t*)nodes.dptr, rows);
} BX_ABI_CATCH(c, "bx_merkle_fold")

extern "C" const char* bx_transcript_step(bx_ctx* c, bx_buf state, bx_buf digests, size_t n_commit, bx_buf out_ext, size_t n_ext) try {
    if (!c) return "bx_transcript_step: null ctx";
    BX_REQUIRE(c, state.dptr != nullptr && state.len >= 25, "transcript_step: the state is 24 cells and the pool counter");
    BX_REQUIRE(c, n_commit <= 64 && n_ext <= 64, "transcript_step: at most 64 commits and 64 challenges per step");
    BX_REQUIRE(c, digests.len >= 8 * n_commit && out_ext.len >= 4 * n_ext, "transcript_step: digests / out too small");
    BX_ENTER(c);
    if (!n_commit && !n_ext) return nullptr;
    OpScope op(c, "transcript_step", 4.0 * (double)(50 + 8 * n_commit + 4 * n_ext));
    hipLaunchKernelGGL(transcript_step_kernel, dim3(1), dim3(64), 0, c->stream, (uint32_t*)state.dptr, (const uint32_t*)digests.dptr, (uint32_t)n_commit,
                       (uint32_t*)out_ext.dptr, (uint32_t)(4 * n_ext), c->d_p2);
    BX_LAUNCH_CHECK(c);
    return nullptr;
} BX_ABI_CATCH(c, "bx_transcript_step")
