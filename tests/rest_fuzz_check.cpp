// rest_fuzz_check.cpp — the REST worker client (boundless_amd/csrc/rest_worker.cpp) under AddressSanitizer / UBSan against a
// server that answers with malformed, truncated and hostile responses.
//
// Built and run by tests/test_verifier_sanitizers_cpu.py.  A thread of this process listens on 127.0.0.1 and answers every
// connection with the next canned byte string; the client's six calls are made against each of them.  Every call must
// return (an error, "not found", or a decoded value) — none may read out of bounds, recurse without bound, loop forever or
// leak.  Two well-formed answers at the end check that the harness is not rejecting everything.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../include/bx_rest.h"

static std::string current;       // what the server answers right now
static std::atomic<int> lfd{-1};
static std::atomic<bool> quit{false};

static void serve() {
    for (;;) {
        int fd = accept(lfd.load(), nullptr, nullptr);
        if (fd < 0) {
            if (quit.load()) return;
            continue;
        }
        // read the request head (and whatever body comes with it) until the blank line
        std::string req;
        char buf[4096];
        while (req.find("\r\n\r\n") == std::string::npos) {
            ssize_t k = recv(fd, buf, sizeof buf, 0);
            if (k <= 0) break;
            req.append(buf, (size_t)k);
        }
        const std::string ans = current;
        size_t off = 0;
        while (off < ans.size()) {
            ssize_t k = send(fd, ans.data() + off, ans.size() - off, MSG_NOSIGNAL);
            if (k <= 0) break;
            off += (size_t)k;
        }
        shutdown(fd, SHUT_RDWR);
        close(fd);
    }
}

static std::string ok_json(const std::string& body) {
    return "HTTP/1.1 200 OK\r\nContent-Type: application/json\r\nContent-Length: " + std::to_string(body.size()) + "\r\n\r\n" + body;
}

int main() {
    int s = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = 0;
    if (bind(s, (sockaddr*)&a, sizeof a) != 0 || listen(s, 16) != 0) return 2;
    socklen_t al = sizeof a;
    getsockname(s, (sockaddr*)&a, &al);
    lfd.store(s);
    std::thread srv(serve);

    bx_rest_client* c = nullptr;
    const std::string url = "http://127.0.0.1:" + std::to_string(ntohs(a.sin_port));
    if (const char* e = bx_rest_client_create(url.c_str(), 0, 2, &c)) {
        printf("create: %s\n", e);
        return 2;
    }
    bx_taskdb_ops db = bx_rest_taskdb_ops(c);
    bx_hot_store_ops hot = bx_rest_hot_store_ops(c);

    std::vector<std::string> hostile = {
        "",
        "garbage\r\n\r\n",
        "HTTP/1.1\r\n\r\n",
        "HTTP/1.1 200 OK\r\n",
        "HTTP/1.1 200 OK\r\nContent-Length: 100\r\n\r\nshort",
        "HTTP/1.1 200 OK\r\nContent-Length: 18446744073709551615\r\n\r\nx",
        "HTTP/1.1 200 OK\r\nContent-Length: -5\r\n\r\nxxxxx",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nFFFFFFFFFFFFFFFA\r\nabc\r\n0\r\n\r\n",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nFFFFFFFFFFFFFFFF\r\nabc\r\n0\r\n\r\n",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nzz\r\nabc\r\n0\r\n\r\n",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n5\r\nab",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n3\r\nabc",
        ok_json("{"),
        ok_json("{\"job_id\":"),
        ok_json("{\"job_id\":\"a\\u12"),
        ok_json("{\"job_id\":\"a\\"),
        ok_json("[1,2"),
        ok_json(std::string(100000, '[')),
        ok_json("{\"job_id\":" + std::string(100000, '{') + "}"),
        ok_json("{\"job_id\":5,\"task_id\":\"t\",\"task_def\":{},\"max_retries\":\"x\"}"),
        ok_json("{\"job_id\":\"" + std::string(10000, 'j') + "\",\"task_id\":\"t\",\"task_def\":{\"Prove\":{\"index\":1}},\"prereqs\":[],\"max_retries\":3}"),
        ok_json("{\"job_id\":\"j\",\"task_id\":\"t\",\"task_def\":\"" + std::string(5000, 'd') + "\",\"prereqs\":[],\"max_retries\":3}"),
        ok_json("{\"job_id\":\"j\",\"task_id\":\"t\",\"task_def\":{\"Prove\":{\"index\":1}},\"prereqs\":[],\"max_retries\":99999999999999999999}"),
        ok_json("{\"updated\":"),
        ok_json("{\"retries\":\"many\"}"),
        ok_json("{\"retries\":-99999999999999999999}"),
        ok_json("nul"),
        "HTTP/1.1 500 Internal Server Error\r\nContent-Length: 300000\r\n\r\n" + std::string(300000, 'E'),
        "HTTP/1.1 999999999999999999999 Weird\r\nContent-Length: 0\r\n\r\n",
        std::string("HTTP/1.1 200 OK\r\nContent-Length: 4\r\n\r\n\0\0\0\0", 42),
        // the pooling transport: bytes past the body, oversized heads, chunk extensions, trailers, EOF-delimited bodies
        "HTTP/1.1 200 OK\r\nContent-Length: 2\r\nConnection: keep-alive\r\n\r\nokHTTP/1.1 200 OK\r\nContent-Length: 4\r\n\r\nnull",
        "HTTP/1.1 200 OK\r\nX-Pad: " + std::string(70000, 'a') + "\r\nContent-Length: 4\r\n\r\nnull",
        "HTTP/1.1 200 OK\r\n" + [] { std::string h; for (int i = 0; i < 9000; ++i) h += "X-" + std::to_string(i) + ": y\r\n"; return h; }() + "\r\nnull",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4;ext=\"a\"\r\nnull\r\n0\r\nTrailer: x\r\n\r\n",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4\r\nnullXX0\r\n\r\n",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n" + std::string(2000, '0') + "4\r\nnull\r\n0\r\n\r\n",
        "HTTP/1.0 200 OK\r\n\r\nnull",
        "HTTP/1.1 204 No Content\r\nContent-Length: 50\r\n\r\n",
        "HTTP/1.1 200 OK\nContent-Length: 4\n\nnull",
        "HTTP/1.1 200 OK\r\nContent-Length: 4\r\nContent-Length: 99999999999999999999\r\n\r\nnull",
        // interim responses: none followed by a final one, an endless run of them, a 101 nobody asked for; trailers without end
        "HTTP/1.1 100 Continue\r\n\r\n",
        [] { std::string h; for (int i = 0; i < 64; ++i) h += "HTTP/1.1 100 Continue\r\n\r\n"; return h + "HTTP/1.1 200 OK\r\nContent-Length: 4\r\n\r\nnull"; }(),
        "HTTP/1.1 101 Switching Protocols\r\nUpgrade: h2c\r\n\r\nnull",
        "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4\r\nnull\r\n0\r\n" + [] { std::string h; for (int i = 0; i < 20000; ++i) h += "T-" + std::to_string(i) + ": y\r\n"; return h; }() + "\r\n",
    };
    char eb[256];
    long calls = 0;
    for (const std::string& h : hostile) {
        current = h;
        bx_ready_task t;
        int32_t retries = 0;
        uint8_t* val = nullptr;
        size_t len = 0;
        (void)db.request_work(db.user, "prove", &t, eb, sizeof eb);
        (void)db.update_task_done(db.user, "job", "task", "null", eb, sizeof eb);
        (void)db.update_task_failed(db.user, "job", "task", "an \"error\"\n", eb, sizeof eb);
        (void)db.update_task_retry(db.user, "job", "task", eb, sizeof eb);
        (void)db.current_retries(db.user, "job", "task", &retries, eb, sizeof eb);
        if (hot.get(hot.user, "job:x:segments:0", &val, &len, eb, sizeof eb) == 0 && val) hot.free_value(hot.user, val);
        (void)hot.set_ex(hot.user, "k", (const uint8_t*)"v", 1, 60, eb, sizeof eb);
        (void)hot.unlink(hot.user, "k", eb, sizeof eb);
        calls += 8;
    }
    // sanity: well-formed answers are understood
    int bad = 0;
    bx_ready_task t;
    current = ok_json("null");
    if (db.request_work(db.user, "prove", &t, eb, sizeof eb) != 0) bad |= 1;
    current = ok_json("{\"job_id\":\"0b1e55\",\"task_id\":\"t-1\",\"task_def\":{\"Prove\":{\"index\":7}},\"prereqs\":[],\"max_retries\":3}");
    if (db.request_work(db.user, "prove", &t, eb, sizeof eb) != 1 || strcmp(t.job_id, "0b1e55") || strcmp(t.task_id, "t-1") ||
        strcmp(t.task_def, "{\"Prove\":{\"index\":7}}") || t.max_retries != 3)
        bad |= 2;
    current = "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4\r\n{\"up\r\nc\r\ndated\":true}\r\n0\r\n\r\n";
    if (db.update_task_retry(db.user, "job", "task", eb, sizeof eb) != 1) bad |= 4;
    current = "HTTP/1.1 404 Not Found\r\nContent-Length: 0\r\n\r\n";
    uint8_t* val = nullptr;
    size_t len = 0;
    if (hot.get(hot.user, "missing", &val, &len, eb, sizeof eb) != 1) bad |= 8;

    // a body delimited by the end of the connection, a chunk extension and a trailer, a bare-LF head
    current = "HTTP/1.0 200 OK\r\n\r\nnull";
    if (db.request_work(db.user, "prove", &t, eb, sizeof eb) != 0) bad |= 512;
    current = "HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\n4;ext=\"a\"\r\nnull\r\n0\r\nTrailer: x\r\n\r\n";
    if (db.request_work(db.user, "prove", &t, eb, sizeof eb) != 0) bad |= 1024;
    // a 5 MB value arrives whole, read straight into the buffer `get` returns; the server of this harness closes every
    // connection without saying so, so each of these calls also exercises the replacement of a dead pooled connection
    {
        std::string big(5u << 20, '\0');
        for (size_t i = 0; i < big.size(); ++i) big[i] = (char)(i * 2654435761u >> 24);
        current = "HTTP/1.1 200 OK\r\nContent-Type: application/octet-stream\r\nContent-Length: " + std::to_string(big.size()) + "\r\n\r\n" + big;
        uint8_t* v2 = nullptr;
        size_t l2 = 0;
        if (hot.get(hot.user, "big", &v2, &l2, eb, sizeof eb) != 0 || l2 != big.size() || memcmp(v2, big.data(), l2) != 0) bad |= 2048;
        if (v2) hot.free_value(hot.user, v2);
        current = "HTTP/1.1 200 OK\r\nContent-Length: 0\r\n\r\n";
        v2 = nullptr, l2 = 7;
        if (hot.get(hot.user, "empty", &v2, &l2, eb, sizeof eb) != 0 || l2 != 0 || !v2) bad |= 4096;
        if (v2) hot.free_value(hot.user, v2);
    }
    if (bx_rest_client_connects(c) >= bx_rest_client_requests(c) + 1 || bx_rest_client_connects(c) == 0) bad |= 8192;

    // interim 1xx responses precede the answer and are not it
    current = "HTTP/1.1 100 Continue\r\n\r\nHTTP/1.1 103 Early Hints\r\nLink: </x>\r\n\r\n" + ok_json("{\"updated\":true}");
    if (db.update_task_retry(db.user, "job", "task", eb, sizeof eb) != 1) bad |= 16384;

    // serde would refuse these: a string or an out-of-range number where an i32 belongs is a decode error, not a zero
    int32_t rr = 7;
    current = ok_json("{\"retries\":\"many\"}");
    if (db.current_retries(db.user, "job", "task", &rr, eb, sizeof eb) >= 0) bad |= 16;
    current = ok_json("{\"retries\":2147483648}");
    if (db.current_retries(db.user, "job", "task", &rr, eb, sizeof eb) >= 0) bad |= 32;
    current = ok_json("{\"retries\":2}");
    if (db.current_retries(db.user, "job", "task", &rr, eb, sizeof eb) != 1 || rr != 2) bad |= 64;
    current = ok_json("{\"retries\":null}");
    if (db.current_retries(db.user, "job", "task", &rr, eb, sizeof eb) != 0) bad |= 128;
    current = ok_json("{\"job_id\":\"j\",\"task_id\":\"t\",\"task_def\":{\"Prove\":{\"index\":1}},\"prereqs\":[],\"max_retries\":\"3\"}");
    if (db.request_work(db.user, "prove", &t, eb, sizeof eb) >= 0) bad |= 256;

    quit.store(true);
    shutdown(s, SHUT_RDWR);
    close(s);
    srv.join();
    bx_rest_client_destroy(c);
    if (bad) {
        printf("well-formed answers misread: %d (%s)\n", bad, eb);
        return 1;
    }
    printf("rest_fuzz_check ok (%ld calls against hostile answers)\n", calls);
    return 0;
}
