/*
 * bx_rest.h — the next-generation Bento worker protocol as the agent's two callback tables.
 *
 * In `prover/` (the next-gen copy of `bento/`) GPU workers no longer talk to Postgres/Redis; they claim tasks and move
 * hot blobs through the API over REST.  This file is the C++ counterpart of that worker-side client, shaped as the
 * `bx_taskdb_ops` / `bx_hot_store_ops` tables of bx_agent.h, so that `bx_agent_create(cfg, &hot, &db, ...)` turns the native
 * feed loop into a worker of a real next-gen Bento cluster without any other change.
 *
 * Reference interfaces restated here
 * ----------------------------------
 *   server routes   prover/crates/api/src/lib.rs:922-1040
 *       POST   /worker/gpu/tasks/claim/:task_stream?wait_timeout_secs=N   -> JSON null | {job_id,task_id,task_def,prereqs,max_retries}
 *       POST   /worker/gpu/tasks/:job_id/:task_id/done      {"output": <json>}   -> {"updated": bool}
 *       POST   /worker/gpu/tasks/:job_id/:task_id/failed    {"error": "<text>"}  -> {"updated": bool}
 *       POST   /worker/gpu/tasks/:job_id/:task_id/retry                          -> {"updated": bool}
 *       GET    /worker/gpu/tasks/:job_id/:task_id/retries-running                -> {"retries": n | null}
 *       GET    /worker/hot/{key...}           -> 200 bytes | 404 (HotDataMissing, lib.rs:248)
 *       PUT    /worker/hot/{key...}?ttl_secs=N   body = bytes -> 204
 *       DELETE /worker/hot/{key...}              -> 204
 *   worker client   prover/crates/workflow/src/assets.rs:88-120 (URLs), :193-420 (calls, error_for_status, JSON shapes)
 *   its use         prover/crates/workflow/src/lib.rs:353-365 (claim with wait_timeout_secs = poll_time), :371-420
 *
 * Transport: plain HTTP/1.1 over TCP, blocking (the agent's lanes are threads).  Connections are kept alive and pooled
 * (up to 16 idle ones; the reference shares one pooling reqwest::Client, assets.rs:76); an idle connection the server has
 * dropped is replaced once, transparently, when no byte of an answer had arrived.  A body with a Content-Length is received
 * straight into the buffer `get` returns: an 80 MB segment is written once by the kernel and never copied by this client.  The reference's workers reach the API inside the cluster network; TLS, if any, terminates in
 * front of it.  A client may be used from any number of threads (the pool is its only mutable state).
 * Errors follow bx_agent.h's callback convention (negative return + message in errbuf); HTTP status >= 400 is an error
 * except 404 on a hot-store GET, which is "key not found" (return 1) as redis nil is for the in-memory store.
 */
#ifndef BX_REST_H
#define BX_REST_H
#include "bx_agent.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct bx_rest_client bx_rest_client;

/* base_url = "http://host[:port][/prefix]".  claim_wait_secs = the wait_timeout_secs sent with every claim (the server
 * blocks up to that long for work; the reference sends its poll_time).  io_timeout_secs bounds connect/send/receive of any
 * one call beyond that wait (0 = 30). */
const char* bx_rest_client_create(const char* base_url, uint64_t claim_wait_secs, uint64_t io_timeout_secs, bx_rest_client** out);
void bx_rest_client_destroy(bx_rest_client* c);
/* The returned tables borrow the client; destroy it after the agent. */
bx_taskdb_ops bx_rest_taskdb_ops(bx_rest_client* c);
bx_hot_store_ops bx_rest_hot_store_ops(bx_rest_client* c);
/* Number of HTTP requests issued so far (all threads), and the number of TCP connections opened for them. */
uint64_t bx_rest_client_requests(const bx_rest_client* c);
uint64_t bx_rest_client_connects(const bx_rest_client* c);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
