// circuit.hip — device side of the synthetic circuit that stands in for risc0-circuit-rv32im under
// `ProverServer::prove_segment` (bento/crates/workflow/src/tasks/prove.rs:41-49): witness generation, the accumulate
// step and eval_check.  include/bx_prover.h ("The synthetic circuit") is the normative text; circuit.hpp holds the shape
// rules shared with the host verifier.  Upstream's counterparts are the machine-generated `witgen`/`step_exec`, `accum`
// and `eval_check` kernels of risc0-circuit-rv32im-sys 4.0.1 (reference Cargo.lock:8996), which are not vendored.
//
// All three stages are VALU-bound (no HBM or MFMA roofline applies): a derived cell / a constraint costs T*(G-1)
// Montgomery products, the loads around it are one coalesced dword per lane and column.
// The signed multiply-adds are left to the compiler in this translation unit (poseidon2.hip pins them as single asm
// statements because its loop-carried cells get widened): here the terms of a cell share their inner products (7 pool entries
// give at most 28 distinct pairs), and only un-pinned code lets hipcc find them: 446 instead of 605 instructions per cell
// for (T, G) = (48, 3), and none of the s_nops it places between adjacent asm statements.
#define BX_PLAIN_MAD 1
#include "circuit.hpp"
#include "ctx.hpp"
#include "circuit_dev.hpp"
#include "lazy_ext.hpp"
#include "../../include/bx_circuit.h"

namespace bx {

// ---- witness: code group (selectors + public control words) ----
// A function of the shape alone (SYNTH_CODE_SEED is a constant): its committed root is the circuit's control ID.
__global__ void witness_code_kernel(uint32_t* __restrict__ code, Circuit cc) {
    const uint32_t n = 1u << cc.po2;
    const size_t total = (size_t)n * cc.wc, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        code[i] = synth_code_cell(cc, (uint32_t)(i >> cc.po2), (uint32_t)(i & (n - 1)));
}
// ---- witness: free data columns (the permuted copies 4p+3, p < pairs, are placed by Hal::scatter afterwards) ----
// Rows >= active_rows() are the ZK noise rows: drawn from the noise seed, in the permuted copies too.
__global__ void witness_free_kernel(uint32_t* __restrict__ data, Circuit cc, uint64_t gseed, uint64_t nseed) {
    const uint32_t n = 1u << cc.po2, act = cc.active_rows();
    const size_t total = (size_t)n * cc.F, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t c = (uint32_t)(i >> cc.po2), r = (uint32_t)(i & (n - 1));
        if (r < act && (c & 3u) == 3u && (c >> 2) < cc.pairs) continue;
        data[i] = synth_word(r < act ? gseed : nseed, c, r);
    }
}
// offsets of pair p's scatter: entry r of column 4p+2 goes to column 4p+3, row perm_p(r)   (built once per prover)
__global__ void perm_offsets_kernel(uint32_t* __restrict__ offsets, uint32_t* __restrict__ index, Circuit cc) {
    const uint32_t n = 1u << cc.po2, act = cc.active_rows();
    const size_t total = (size_t)n * cc.pairs, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t p = (uint32_t)(i >> cc.po2), r = (uint32_t)(i & (n - 1));
        offsets[i] = r < act ? (4 * p + 3) * n + cc.perm_row(p, r) : 0u;
    }
    // one entry per active cycle, none for the noise cycles
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) index[i] = (uint32_t)(i < act ? i : act);
}

// ---- witness: derived data columns, one thread per row, columns in order (column F+j reads F+j-1 .. F+j-4) ----
template <int TT, int GG>
__global__ __launch_bounds__(256) void witness_derive_kernel(uint32_t* __restrict__ data, const uint32_t* __restrict__ code, Circuit cc) {
    const uint32_t n = 1u << cc.po2;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t rb1 = (r + n - 1) & (n - 1), rb2 = (r + n - 2) & (n - 1);
    // rings: the eight previous derived columns (csel(0..7) before the first), free columns j, j+1, j+2, code csel(j..j+3)
    const auto code_at = [&](unsigned i) -> uint32_t {
        const int col = cc.csel_col(i);
        return col < 0 ? MONT_ONE : code[(size_t)col * n + r];
    };
    uint32_t ring[8], u[3], k[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) ring[q] = code_at((unsigned)q);
#pragma unroll
    for (int q = 0; q < 3; ++q) u[q] = data[(size_t)((uint32_t)q % cc.F) * n + r];
#pragma unroll
    for (int q = 0; q < 4; ++q) k[q] = code_at((unsigned)q);
    for (uint32_t j = 0; j < cc.J; ++j) {
        uint32_t pool[Circuit::POOL];
        pool[0] = u[0];
        const int sb = Circuit::slot1_back(j);
        pool[1] = sb == 0 ? u[0] : data[(size_t)j * n + (sb == 1 ? rb1 : rb2)];
        pool[2] = u[1]; pool[3] = u[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) pool[4 + q] = ring[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) pool[12 + q] = k[q];
        const uint32_t d = cons_sum<TT, GG>(pool, cc.T, cc.G);
        data[(size_t)(cc.F + j) * n + r] = d;
#pragma unroll
        for (int q = 7; q > 0; --q) ring[q] = ring[q - 1];
        ring[0] = d;
        u[0] = u[1]; u[1] = u[2]; u[2] = data[(size_t)((j + 3) % cc.F) * n + r];
        k[0] = k[1]; k[1] = k[2]; k[2] = k[3]; k[3] = code_at(j + 4);
    }
}

// ---- accumulate: the columns the accumulators run over are saved before the data group is interpolated in place ----
__global__ void accum_gather_kernel(uint32_t* __restrict__ srcvals, const uint32_t* __restrict__ data, Circuit cc) {
    const uint32_t n = 1u << cc.po2;
    const size_t total = (size_t)n * cc.E, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t e = (uint32_t)(i >> cc.po2), r = (uint32_t)(i & (n - 1));
        srcvals[i] = data[(size_t)cc.acc_src(e) * n + r];
    }
}
// run[e][r] = beta_e + x (AoS ext), the input of Hal::prefix_products
__global__ void accum_build_kernel(uint32_t* __restrict__ run, const uint32_t* __restrict__ srcvals, const uint32_t* __restrict__ betas, Circuit cc) {
    const uint32_t n = 1u << cc.po2;
    const size_t total = (size_t)n * cc.E, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t e = (uint32_t)(i >> cc.po2);
        const uint4 b = *reinterpret_cast<const uint4*>(betas + 4 * e);  // wave-uniform
        *reinterpret_cast<uint4*>(run + 4 * i) = make_uint4(fp_add(b.x, srcvals[i]), b.y, b.z, b.w);
    }
}
// accumulator e, component k -> accum column 4e+k; columns >= 4E are noise
__global__ void accum_store_kernel(uint32_t* __restrict__ accum, const uint32_t* __restrict__ run, Circuit cc, uint64_t gseed) {
    const uint32_t n = 1u << cc.po2;
    const size_t total = (size_t)n * cc.E, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t e = (uint32_t)(i >> cc.po2), r = (uint32_t)(i & (n - 1));
        const uint4 v = *reinterpret_cast<const uint4*>(run + 4 * i);
        uint32_t* o = accum + (size_t)(4 * e) * n + r;
        o[0] = v.x; o[n] = v.y; o[2 * (size_t)n] = v.z; o[3 * (size_t)n] = v.w;
    }
    const size_t noise = (size_t)n * (cc.wa - 4 * cc.E);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < noise; i += stride) {
        const uint32_t c = 4 * cc.E + (uint32_t)(i >> cc.po2), r = (uint32_t)(i & (n - 1));
        accum[(size_t)c * n + r] = synth_word(gseed, c, r);
    }
}

// ---- eval_check: check(x) = sum_i poly_mix^i C_i(x) / ((3x)^N - 1) on the 4N domain, one thread per domain point ----
struct ZInv {
    uint32_t v[4];  // 1 / (3^N w_4^m - 1), m = row mod 4
    uint32_t g[2];  // the statement's public words (Circuit::globals())
};
template <int TT, int GG>
__global__ __launch_bounds__(256) void eval_check_kernel(uint32_t* __restrict__ check, const uint32_t* __restrict__ ecode,
                                                         const uint32_t* __restrict__ edata, const uint32_t* __restrict__ eacc, Circuit cc,
                                                         const uint32_t* __restrict__ mixpows, const uint32_t* __restrict__ mixpows_c,
                                                         const uint32_t* __restrict__ betas, ZInv zinv) {
    const uint32_t dom = 4u << cc.po2;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dom) return;
    const uint32_t ib = (i + dom - 4u) & (dom - 1u);   // one row back: x * w_N^-1 = w_4N^(row - 4)
    const uint32_t ib2 = (i + dom - 8u) & (dom - 1u);  // two rows back
    LazyExtAcc mixacc;  // sum_j poly_mix^j * C_j over the derived-column constraints (ext weight x base value)
    mixacc.reset();
    const auto code_at = [&](unsigned q) -> uint32_t {
        const int col = cc.csel_col(q);
        return col < 0 ? MONT_ONE : ecode[(size_t)col * dom + i];
    };
    uint32_t ring[8], u[3], k[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) ring[q] = code_at((unsigned)q);
#pragma unroll
    for (int q = 0; q < 3; ++q) u[q] = edata[(size_t)((uint32_t)q % cc.F) * dom + i];
#pragma unroll
    for (int q = 0; q < 4; ++q) k[q] = code_at((unsigned)q);
    for (uint32_t j = 0; j < cc.J; ++j) {
        uint32_t pool[Circuit::POOL];
        pool[0] = u[0];
        const int sb = Circuit::slot1_back(j);
        pool[1] = sb == 0 ? u[0] : edata[(size_t)j * dom + (sb == 1 ? ib : ib2)];
        pool[2] = u[1]; pool[3] = u[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) pool[4 + q] = ring[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) pool[12 + q] = k[q];
        const uint32_t d = edata[(size_t)(cc.F + j) * dom + i];
        const uint32_t cons = fp_sub(d, cons_sum<TT, GG>(pool, cc.T, cc.G));
        const uint4 m = *reinterpret_cast<const uint4*>(mixpows_c + 4 * (size_t)j);  // wave-uniform, centred
        const i32 w[4] = {(i32)m.x, (i32)m.y, (i32)m.z, (i32)m.w};
        mixacc.add(w, cons);
#pragma unroll
        for (int q = 7; q > 0; --q) ring[q] = ring[q - 1];
        ring[0] = d;
        u[0] = u[1]; u[1] = u[2]; u[2] = edata[(size_t)((j + 3) % cc.F) * dom + i];
        k[0] = k[1]; k[1] = k[2]; k[2] = k[3]; k[3] = code_at(j + 4);
    }
    Fp4 tot = mixacc.finish();
    const uint32_t first = ecode[i];
    for (uint32_t e = 0; e < cc.E; ++e) {
        Fp4 a, ab;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.c[k] = eacc[(size_t)(4 * e + k) * dom + i];
            ab.c[k] = eacc[(size_t)(4 * e + k) * dom + ib];
        }
        Fp4 inner = f4_scale(ab, fp_sub(MONT_ONE, first));
        inner.c[0] = fp_add(inner.c[0], first);
        const uint4 b = *reinterpret_cast<const uint4*>(betas + 4 * (size_t)e);
        Fp4 fac{{fp_add(b.x, edata[(size_t)cc.acc_src(e) * dom + i]), b.y, b.z, b.w}};
        const uint4 m = *reinterpret_cast<const uint4*>(mixpows + 4 * (size_t)(cc.J + e));
        tot = f4_add(tot, f4_mul(Fp4{{m.x, m.y, m.z, m.w}}, f4_sub(a, f4_mul(inner, fac))));
    }
    if (cc.pairs) {
        const uint32_t last = ecode[(size_t)dom + i];
        for (uint32_t p = 0; p < cc.pairs; ++p) {
            Fp4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                d.c[k] = fp_sub(eacc[(size_t)(4 * (2 * p + 1) + k) * dom + i], eacc[(size_t)(4 * (2 * p) + k) * dom + i]);
            const uint4 m = *reinterpret_cast<const uint4*>(mixpows + 4 * (size_t)(cc.J + cc.E + p));
            tot = f4_add(tot, f4_mul(Fp4{{m.x, m.y, m.z, m.w}}, f4_scale(d, last)));
        }
    }
    {   // boundary constraints tying the public words to the trace: first * (data[0] - g0), last * (data[wd-1] - g1)
        const size_t b0 = (size_t)cc.J + cc.E + cc.pairs;
        const uint4 m0 = *reinterpret_cast<const uint4*>(mixpows + 4 * b0);
        tot = f4_add(tot, f4_scale(Fp4{{m0.x, m0.y, m0.z, m0.w}}, fp_mul(first, fp_sub(edata[i], zinv.g[0]))));
        if (cc.globals() > 1) {
            const uint4 m1 = *reinterpret_cast<const uint4*>(mixpows + 4 * (b0 + 1));
            const uint32_t last = ecode[(size_t)dom + i];
            tot = f4_add(tot, f4_scale(Fp4{{m1.x, m1.y, m1.z, m1.w}}, fp_mul(last, fp_sub(edata[(size_t)(cc.wd - 1) * dom + i], zinv.g[1]))));
        }
    }
    tot = f4_scale(tot, zinv.v[i & 3u]);
#pragma unroll
    for (int k = 0; k < 4; ++k) check[(size_t)k * dom + i] = tot.c[k];
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side launchers (called by prover.hip); each returns NULL or an error string owned by the ctx
// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned grid_for(size_t n, unsigned bs = 256, size_t cap = 1 << 16) {
    size_t b = (n + bs - 1) / bs;
    return (unsigned)(b > cap ? cap : (b ? b : 1));
}

#define BX_CIRCUIT_DISPATCH(KERNEL, ...)                                              \
    do {                                                                              \
        if (cc.T == 64 && cc.G == 4) hipLaunchKernelGGL((KERNEL<64, 4>), __VA_ARGS__); \
        else if (cc.T == 48 && cc.G == 3) hipLaunchKernelGGL((KERNEL<48, 3>), __VA_ARGS__); \
        else if (cc.T == 16 && cc.G == 3) hipLaunchKernelGGL((KERNEL<16, 3>), __VA_ARGS__); \
        else if (cc.T == 8 && cc.G == 2) hipLaunchKernelGGL((KERNEL<8, 2>), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<0, 0>), __VA_ARGS__);                          \
    } while (0)

const char* circuit_perm_tables(bx_ctx* c, const Circuit& cc, bx_buf offsets, bx_buf index) {
    const size_t n = (size_t)1 << cc.po2;
    BX_REQUIRE(c, offsets.len >= n * cc.pairs && index.len >= n + 1, "circuit_perm_tables: buffers too small");
    hipLaunchKernelGGL(perm_offsets_kernel, dim3(grid_for(n * (cc.pairs ? cc.pairs : 1))), dim3(256), 0, c->stream, (uint32_t*)offsets.dptr,
                       (uint32_t*)index.dptr, cc);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// the code group (public; what bx_prover_control_id commits)
const char* circuit_code(bx_ctx* c, const Circuit& cc, bx_buf code) {
    const size_t n = (size_t)1 << cc.po2;
    BX_REQUIRE(c, code.len == n * cc.wc, "circuit_code: group buffer size mismatch");
    OpScope op(c, "witgen_code", 4.0 * (double)(n * cc.wc));
    hipLaunchKernelGGL(witness_code_kernel, dim3(grid_for(n * cc.wc)), dim3(256), 0, c->stream, (uint32_t*)code.dptr, cc);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// data witness.  `data`/`code` are the groups' column-major N x width buffers (code already filled); the permuted copies go
// through Hal::scatter (one call per pair, one entry per cycle), the derived columns through one thread per row.
const char* circuit_witness(bx_ctx* c, const Circuit& cc, bx_buf code, bx_buf data, uint64_t seed_data, uint64_t seed_noise,
                            bx_buf perm_offsets, bx_buf perm_index) {
    const size_t n = (size_t)1 << cc.po2;
    BX_REQUIRE(c, code.len == n * cc.wc && data.len == n * cc.wd, "circuit_witness: group buffer size mismatch");
    {
        OpScope op(c, "witgen_fill", 4.0 * (double)(n * cc.F));
        hipLaunchKernelGGL(witness_free_kernel, dim3(grid_for(n * cc.F)), dim3(256), 0, c->stream, (uint32_t*)data.dptr, cc, seed_data, seed_noise);
        BX_LAUNCH_CHECK(c);
    }
    for (uint32_t p = 0; p < cc.pairs; ++p) {
        bx_buf values{(uint32_t*)data.dptr + (size_t)(4 * p + 2) * n, n};
        bx_buf offs{(uint32_t*)perm_offsets.dptr + (size_t)p * n, n};
        BX_TRY(bx_scatter(c, data, bx_buf{perm_index.dptr, n + 1}, offs, values));
    }
    if (cc.J) {
        OpScope op(c, "witgen_derive", 4.0 * (double)(n * (cc.J + cc.J + cc.J / 4)));
        BX_CIRCUIT_DISPATCH(witness_derive_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)data.dptr,
                            (const uint32_t*)code.dptr, cc);
        BX_LAUNCH_CHECK(c);
    }
    return nullptr;
}

const char* circuit_accum_gather(bx_ctx* c, const Circuit& cc, bx_buf srcvals, bx_buf data) {
    const size_t n = (size_t)1 << cc.po2;
    if (!cc.E) return nullptr;
    BX_REQUIRE(c, srcvals.len >= n * cc.E, "circuit_accum_gather: buffer too small");
    OpScope op(c, "accum_gather", 8.0 * (double)(n * cc.E));
    hipLaunchKernelGGL(accum_gather_kernel, dim3(grid_for(n * cc.E)), dim3(256), 0, c->stream, (uint32_t*)srcvals.dptr, (const uint32_t*)data.dptr, cc);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// the accumulate step: run = beta_e + x  ->  Hal::prefix_products  ->  accum columns (+ noise columns)
const char* circuit_accumulate(bx_ctx* c, const Circuit& cc, bx_buf accum, bx_buf run, bx_buf srcvals, bx_buf betas_dev, uint64_t seed_accum) {
    const size_t n = (size_t)1 << cc.po2;
    BX_REQUIRE(c, accum.len == n * cc.wa, "circuit_accumulate: group buffer size mismatch");
    if (cc.E) {
        BX_REQUIRE(c, run.len >= 4 * n * cc.E && srcvals.len >= n * cc.E && betas_dev.len >= 4 * cc.E, "circuit_accumulate: buffers too small");
        {
            OpScope op(c, "accum_build", 20.0 * (double)(n * cc.E));
            hipLaunchKernelGGL(accum_build_kernel, dim3(grid_for(n * cc.E)), dim3(256), 0, c->stream, (uint32_t*)run.dptr,
                               (const uint32_t*)srcvals.dptr, (const uint32_t*)betas_dev.dptr, cc);
            BX_LAUNCH_CHECK(c);
        }
        BX_TRY(bx_batch_prefix_products(c, bx_buf{run.dptr, 4 * n * cc.E}, cc.E));
    }
    OpScope op(c, "accum_store", 4.0 * (double)(n * cc.wa) + 16.0 * (double)(n * cc.E));
    hipLaunchKernelGGL(accum_store_kernel, dim3(grid_for(n * (cc.wa ? cc.wa : 1))), dim3(256), 0, c->stream, (uint32_t*)accum.dptr,
                       (const uint32_t*)run.dptr, cc, seed_accum);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// mixpows[i] = poly_mix^i for i < n (canonical), followed by the same table centred (the weights of eval_check's LazyExtAcc)
__global__ void mix_table_kernel(uint32_t* __restrict__ out, Fp4 base, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp4 r = f4_pow(base, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out[4 * i + k] = r.c[k];
        out[4 * (n + i) + k] = (uint32_t)fp_centre_w(r.c[k]);
    }
}
const char* circuit_mix_table(bx_ctx* c, const Circuit& cc, bx_buf mixpows, const uint32_t poly_mix[4]) {
    const uint32_t n = (uint32_t)cc.constraints();
    BX_REQUIRE(c, mixpows.len >= 8 * (size_t)n, "circuit_mix_table: table too small");
    if (!n) return nullptr;
    hipLaunchKernelGGL(mix_table_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, (uint32_t*)mixpows.dptr,
                       Fp4{{poly_mix[0], poly_mix[1], poly_mix[2], poly_mix[3]}}, n);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// eval_check over the committed 4N evaluations; `check` receives the four ext planes of check(x) over the domain
const char* circuit_eval_check(bx_ctx* c, const Circuit& cc, bx_buf check, bx_buf ecode, bx_buf edata, bx_buf eacc, bx_buf mixpows, bx_buf betas_dev,
                               const uint32_t zinv[4], const uint32_t* globals) {
    const size_t dom = (size_t)4 << cc.po2;
    BX_REQUIRE(c, check.len == 4 * dom && ecode.len == dom * cc.wc && edata.len == dom * cc.wd && eacc.len == dom * cc.wa,
               "circuit_eval_check: buffer size mismatch");
    BX_REQUIRE(c, mixpows.len >= 8 * cc.constraints(), "circuit_eval_check: mix power table too small");
    ZInv z;
    for (int m = 0; m < 4; ++m) z.v[m] = zinv[m];
    z.g[0] = globals[0];
    z.g[1] = cc.globals() > 1 ? globals[1] : 0u;
    // every committed evaluation is read once (plus the one-row-back taps), the check planes are written once
    OpScope op(c, "eval_check", 4.0 * (double)dom * (cc.wc + cc.wd + cc.J / 4.0 + 2.0 * cc.wa + 4.0));
    BX_CIRCUIT_DISPATCH(eval_check_kernel, dim3((unsigned)((dom + 255) / 256)), dim3(256), 0, c->stream, (uint32_t*)check.dptr,
                        (const uint32_t*)ecode.dptr, (const uint32_t*)edata.dptr, (const uint32_t*)eacc.dptr, cc, (const uint32_t*)mixpows.dptr,
                        (const uint32_t*)mixpows.dptr + 4 * cc.constraints(), (const uint32_t*)betas_dev.dptr, z);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}

// beta_e = beta^(floor(e/2)+1): the two accumulators of a pair share their challenge
__global__ void beta_table_kernel(uint32_t* __restrict__ out, Fp4 beta, uint32_t n) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fp4 r = f4_pow(beta, e / 2 + 1);
    out[4 * e + 0] = r.c[0]; out[4 * e + 1] = r.c[1]; out[4 * e + 2] = r.c[2]; out[4 * e + 3] = r.c[3];
}

// ---------------------------------------------------------------------------------------------------------------------
// the synthetic circuit as a bx_circuit_ops table (include/bx_circuit.h): what bx_prover_create plugs in by default
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t GOLDEN64 = 0x9E3779B97F4A7C15ull;
struct SynthState {
    Circuit cc;
    uint64_t seed = 0;  // of the segment being proved (decoded by witgen from the segment's bytes; accumulate's noise columns use it)
    uint64_t noise_seed = 0;
    bool noise_set = false;  // bx_circuit_ops::set_noise_seed was called for the next witgen
    bx_buf perm_offsets{nullptr, 0}, perm_index{nullptr, 0}, acc_src{nullptr, 0}, acc_run{nullptr, 0}, betas{nullptr, 0}, mixpows{nullptr, 0};
};
void synth_destroy(void*, void* state) {
    auto* st = (SynthState*)state;
    if (!st) return;
    for (bx_buf* b : {&st->perm_offsets, &st->perm_index, &st->acc_src, &st->acc_run, &st->betas, &st->mixpows})
        if (b->dptr) (void)hipFree(b->dptr);
    delete st;
}
const char* synth_create(void*, bx_ctx* c, const bx_segment_params* shape, void** state) {
    auto* st = new (std::nothrow) SynthState();
    BX_REQUIRE(c, st != nullptr, "synthetic circuit: out of host memory");
    st->cc = circuit_of(shape);
    const Circuit& cc = st->cc;
    const size_t n = (size_t)1 << cc.po2;
    const char* e = nullptr;
    if (!e) e = raw_alloc(c, 8 * (cc.constraints() + 1), &st->mixpows);  // canonical table + centred copy
    if (!e) e = raw_alloc(c, n * (cc.pairs ? cc.pairs : 1), &st->perm_offsets);
    if (!e) e = raw_alloc(c, n + 1, &st->perm_index);
    if (!e) e = raw_alloc(c, n * (cc.E ? cc.E : 1), &st->acc_src);
    if (!e) e = raw_alloc(c, 4 * n * (cc.E ? cc.E : 1), &st->acc_run);
    if (!e) e = raw_alloc(c, 4 * (cc.E ? cc.E : 1), &st->betas);
    if (!e) e = circuit_perm_tables(c, cc, st->perm_offsets, st->perm_index);
    if (e) {
        synth_destroy(nullptr, st);
        return e;
    }
    *state = st;
    return nullptr;
}
__global__ void globals_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ data, Circuit cc) {
    const size_t n = (size_t)1 << cc.po2;
    if (threadIdx.x == 0) out[0] = data[0];                                  // data[0][0]
    if (threadIdx.x == 1) out[1] = data[(size_t)(cc.wd - 1) * n + (cc.active_rows() - 1)];  // data[wd-1][last active row]
}
const char* synth_code_group(void*, void* state, bx_ctx* c, bx_buf code) { return circuit_code(c, ((SynthState*)state)->cc, code); }
// The synthetic segment is "BXSYNSEG" | index | po2 | seed | payload (bx_prover.h): the witness is a function of the seed; the
// payload stands for the preflight trace (it is uploaded like one: `segment_dev`) and is not read.
const char* synth_witgen(void*, void* state, bx_ctx* c, bx_buf code, bx_buf data, const uint8_t* segment, size_t segment_len, bx_buf /*segment_dev*/,
                         uint32_t* globals_out) {
    auto* st = (SynthState*)state;
    uint64_t seed = 0;
    uint32_t seg_po2 = 0;
    if (const char* e = bx_segment_decode(segment, segment_len, nullptr, &seg_po2, &seed)) return set_msg(c, e);
    if (seg_po2 != st->cc.po2) {
        snprintf(c->err, sizeof c->err, "prove_segment: the segment has po2 %u, this prover was created for po2 %u", seg_po2, st->cc.po2);
        return c->err;
    }
    st->seed = seed;
    // the ZK rows' generator: given through set_noise_seed, else a function of the seed (bx_prover.h, "seeds")
    const uint64_t noise = st->noise_set ? st->noise_seed : splitmix64(seed ^ 0x5A4B4E4F49534521ull);
    st->noise_set = false;
    BX_TRY(circuit_witness(c, st->cc, code, data, seed + GOLDEN64 * 2, noise + GOLDEN64 * 2, st->perm_offsets, st->perm_index));
    BX_TRY(circuit_accum_gather(c, st->cc, st->acc_src, data));  // the prover interpolates `data` in place next
    // the statement's public words come out of the witness (one small copy; the derived cell is only known on the device)
    hipLaunchKernelGGL(globals_kernel, dim3(1), dim3(64), 0, c->stream, (uint32_t*)st->betas.dptr, (const uint32_t*)data.dptr, st->cc);
    BX_LAUNCH_CHECK(c);
    uint32_t g[2] = {0, 0};
    BX_TRY(bx_d2h(c, g, bx_buf{st->betas.dptr, 2}, 2));  // betas is written by accumulate later: free to borrow here
    for (uint32_t i = 0; i < st->cc.globals(); ++i) globals_out[i] = g[i];
    return nullptr;
}
void synth_set_noise_seed(void*, void* state, uint64_t noise_seed) {
    auto* st = (SynthState*)state;
    st->noise_seed = noise_seed;
    st->noise_set = true;
}
const char* synth_betas(bx_ctx* c, SynthState* st, const uint32_t mix[4]) {
    if (!st->cc.E) return nullptr;
    hipLaunchKernelGGL(beta_table_kernel, dim3((st->cc.E + 63) / 64), dim3(64), 0, c->stream, (uint32_t*)st->betas.dptr,
                       Fp4{{mix[0], mix[1], mix[2], mix[3]}}, st->cc.E);
    BX_LAUNCH_CHECK(c);
    return nullptr;
}
const char* synth_accumulate(void*, void* state, bx_ctx* c, bx_buf accum, const uint32_t mix[4]) {
    auto* st = (SynthState*)state;
    const uint64_t gseed = (st->seed + GOLDEN64 * 3) ^ (((uint64_t)mix[0] << 32) | mix[1]);
    BX_TRY(synth_betas(c, st, mix));
    return circuit_accumulate(c, st->cc, accum, st->acc_run, st->acc_src, st->betas, gseed);
}
const char* synth_eval_check(void*, void* state, bx_ctx* c, bx_buf check, bx_buf ecode, bx_buf edata, bx_buf eacc, const uint32_t poly_mix[4],
                             const uint32_t mix[4], const uint32_t* globals) {
    auto* st = (SynthState*)state;
    const Circuit& cc = st->cc;
    BX_TRY(synth_betas(c, st, mix));
    BX_TRY(circuit_mix_table(c, cc, st->mixpows, poly_mix));
    // 1 / ((3x)^N - 1) takes four values on the domain x = w_4N^row: (3x)^N = 3^N w_4^(row mod 4)
    uint32_t zinv[4];
    const uint32_t t3n = fp_pow(MONT_THREE, (uint64_t)1 << cc.po2), w4 = fp_pow(fp_encode(137u), (uint64_t)1 << 25);  // ROU_FWD[2]
    uint32_t cur = MONT_ONE;
    for (int m = 0; m < 4; ++m) {
        zinv[m] = fp_inv(fp_sub(fp_mul(t3n, cur), MONT_ONE));
        cur = fp_mul(cur, w4);
    }
    return circuit_eval_check(c, cc, check, ecode, edata, eacc, st->mixpows, st->betas, zinv, globals);
}
}  // namespace

const char* synthetic_constraints_at(void*, const bx_segment_params* shape, const bx_tap_reader* taps, const uint32_t poly_mix[4],
                                     const uint32_t mix[4], const uint32_t* globals, uint32_t out[4]);  // verify.cpp (host arithmetic only)

}  // namespace bx

extern "C" const bx_circuit_ops* bx_synthetic_circuit(void) {
    static const bx_circuit_ops ops = {nullptr,
                                       "bx-synthetic-air",
                                       bx::synth_normalize,
                                       bx::synth_taps,
                                       bx::synth_n_globals,
                                       bx::synth_create,
                                       bx::synth_destroy,
                                       bx::synth_code_group,
                                       bx::synth_witgen,
                                       bx::synth_accumulate,
                                       bx::synth_eval_check,
                                       bx::synthetic_constraints_at,
                                       bx::synth_set_noise_seed,
                                       bx::synth_check_code};
    return &ops;
}
