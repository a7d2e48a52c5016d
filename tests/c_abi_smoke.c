/* Plain-C consumer of the drop-in boundary (include/bx_hal.h, include/bx_prover.h): what a cgo / FFI binding would do.
 * CPU test: compiles with gcc -std=c99 -fsyntax-only (the headers are valid C).  GPU test: built, linked against
 * libbx_hip_hal.so and run: init -> HAL round trip -> prove one small segment -> verify the seal. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bx_hal.h"
#include "bx_prover.h"
#include "bx_agent.h"
#include "bx_circuit.h"

#define CHECK(expr)                                              \
    do {                                                         \
        const char* m_ = (expr);                                 \
        if (m_) {                                                \
            fprintf(stderr, "FAILED %s: %s\n", #expr, m_);       \
            return 1;                                            \
        }                                                        \
    } while (0)

int main(void) {
    bx_ctx* ctx = NULL;
    CHECK(bx_init(0, &ctx));
    /* evaluate(interpolate(x)) == x through the C ABI */
    enum { N = 1 << 10 };
    uint32_t* host = (uint32_t*)malloc(N * 4);
    uint32_t* back = (uint32_t*)malloc(N * 4);
    for (uint32_t i = 0; i < N; ++i) host[i] = (i * 2654435761u) % BX_P;
    bx_buf buf;
    CHECK(bx_alloc(ctx, N, &buf));
    CHECK(bx_h2d(ctx, buf, host, N));
    CHECK(bx_batch_interpolate_ntt(ctx, buf, 1));
    CHECK(bx_batch_evaluate_ntt(ctx, buf, 1, 0));
    CHECK(bx_d2h(ctx, back, buf, N));
    if (memcmp(host, back, N * 4) != 0) {
        fprintf(stderr, "NTT round trip mismatch\n");
        return 1;
    }
    /* an error is a value, not an abort */
    bx_buf bad = buf;
    bad.len = 24;
    if (bx_batch_interpolate_ntt(ctx, bad, 1) == NULL) {
        fprintf(stderr, "expected an error string for a non power-of-two length\n");
        return 1;
    }
    CHECK(bx_release(ctx, buf));
    /* prove + verify one small synthetic segment */
    bx_segment_params shape = {10, 4, 8, 4, 0, 0};
    bx_prover* prover = NULL;
    CHECK(bx_prover_create(ctx, &shape, &prover));
    size_t cap = bx_prover_seal_words(prover), n = 0;
    uint32_t* seal = (uint32_t*)malloc(cap * 4);
    CHECK(bx_prove_segment(prover, 1234u, seal, cap, &n));
    CHECK(bx_verify_segment(seal, n));
    {   /* the boundary takes what the reference passes — the segment's bytes (prove.rs:36-49) — staged and uploaded by the
         * prover; the payload of the stand-in is carried along and does not enter the synthetic witness */
        enum { PAYLOAD = 1 << 16 };
        uint8_t* blob = (uint8_t*)malloc(BX_SEGMENT_WIRE_BYTES + PAYLOAD);
        uint32_t* seal2 = (uint32_t*)malloc(cap * 4);
        size_t n2 = 0, up_bytes = 0;
        double up_ms = 0;
        uint32_t id[8];
        bx_verifier_ctx* vctx = NULL;
        bx_segment_encode(0, 10, 1234u, blob);
        for (size_t i = 0; i < PAYLOAD; ++i) blob[BX_SEGMENT_WIRE_BYTES + i] = (uint8_t)(i * 31u);
        CHECK(bx_prove_segment_bytes(prover, blob, BX_SEGMENT_WIRE_BYTES + PAYLOAD, seal2, cap, &n2));
        if (n2 != n || memcmp(seal, seal2, n * 4) != 0) {
            fprintf(stderr, "the bytes entry point gave another seal than the seed form\n");
            return 1;
        }
        CHECK(bx_prover_last_upload(prover, &up_ms, &up_bytes));
        if (up_bytes != BX_SEGMENT_WIRE_BYTES + PAYLOAD) {
            fprintf(stderr, "the segment's bytes were not uploaded\n");
            return 1;
        }
        /* two deep: segment k+1 goes up while segment k is proved */
        CHECK(bx_prover_submit_segment(prover, blob, BX_SEGMENT_WIRE_BYTES));
        CHECK(bx_prover_submit_segment(prover, blob, BX_SEGMENT_WIRE_BYTES + PAYLOAD));
        if (bx_prover_submit_segment(prover, blob, BX_SEGMENT_WIRE_BYTES) == NULL) {
            fprintf(stderr, "a third outstanding segment was accepted\n");
            return 1;
        }
        CHECK(bx_prove_submitted(prover, seal2, cap, &n2));
        CHECK(bx_prove_submitted(prover, seal2, cap, &n2));
        if (n2 != n || memcmp(seal, seal2, n * 4) != 0 || bx_prove_submitted(prover, seal2, cap, &n2) == NULL) return 1;
        /* VerifierContext: the control ID of the shape, computed by the device, is what a seal's code root must be */
        CHECK(bx_prover_control_id(prover, id));
        CHECK(bx_verifier_ctx_create(&vctx));
        CHECK(bx_verifier_ctx_add_control_id(vctx, 10, id));
        CHECK(bx_verify_segment_with_context(seal, n, NULL, vctx));
        bx_verifier_ctx_destroy(vctx);
        CHECK(bx_verifier_ctx_create(&vctx));
        id[0] ^= 1u;
        CHECK(bx_verifier_ctx_add_control_id(vctx, 10, id));
        if (bx_verify_segment_with_context(seal, n, NULL, vctx) == NULL) {
            fprintf(stderr, "a seal was accepted against a context that does not hold its control ID\n");
            return 1;
        }
        bx_verifier_ctx_destroy(vctx);
        free(blob);
        free(seal2);
    }
    seal[n / 2] ^= 1u;
    if (bx_verify_segment(seal, n) == NULL) {
        fprintf(stderr, "tampered seal was accepted\n");
        return 1;
    }
    CHECK(bx_prover_destroy(prover));
    CHECK(bx_free(ctx));
    /* the native prove agent: 4 segments through the in-memory hot store / task db, 2 prover lanes */
    bx_mem_store* store = NULL;
    bx_mem_taskdb* db = NULL;
    CHECK(bx_mem_store_create(&store));
    CHECK(bx_mem_taskdb_create(&db));
    bx_hot_store_ops sops = bx_mem_store_ops(store);
    bx_taskdb_ops tops = bx_mem_taskdb_ops(db);
    for (int i = 0; i < 4; ++i) {
        uint8_t wire[BX_SEGMENT_WIRE_BYTES];
        char key[64], task[32], def[64];
        bx_segment_encode((uint64_t)i, 10, 0xB0D1E550000ull + (uint64_t)i, wire);
        snprintf(key, sizeof key, "job:c-smoke:segments:%d", i);
        snprintf(task, sizeof task, "prove-%d", i);
        snprintf(def, sizeof def, "{\"Prove\":{\"index\":%d}}", i);
        if (sops.set_ex(sops.user, key, wire, sizeof wire, 600, NULL, 0) != 0) return 1;
        CHECK(bx_mem_taskdb_create_task(db, "prove", "c-smoke", task, def, 3));
    }
    bx_agent_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.inflight = 2;
    cfg.w_code = 4, cfg.w_data = 8, cfg.w_accum = 4;
    cfg.poll_time = 0.01;
    cfg.synthetic = 1; /* the built-in prover proves the synthetic circuit and says so in its key names */
    bx_agent* agent = NULL;
    CHECK(bx_agent_create(&cfg, &sops, &tops, NULL, &agent));
    uint64_t done = 0;
    CHECK(bx_agent_poll_work(agent, 2, &done));
    if (done != 4 || bx_mem_taskdb_count(db, BX_TASK_DONE) != 4 || bx_mem_store_key_count(store) != 4) {
        fprintf(stderr, "agent: done=%llu\n", (unsigned long long)done);
        return 1;
    }
    CHECK(bx_agent_destroy(agent));
    bx_mem_taskdb_destroy(db);
    bx_mem_store_destroy(store);
    printf("c_abi_smoke ok: seal words %zu\n", n);
    free(host);
    free(back);
    free(seal);
    return 0;
}
