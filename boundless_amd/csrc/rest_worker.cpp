// rest_worker.cpp — worker-side client of the next-generation Bento REST protocol, as the agent's callback tables
// (include/bx_rest.h).  Restates prover/crates/workflow/src/assets.rs:88-420 (URLs, request/response shapes, status handling)
// against the routes of prover/crates/api/src/lib.rs:922-1040.  Host code only: POSIX sockets, no third-party HTTP stack
// (the image has no libcurl headers); connections are kept alive and pooled, bodies are read once into their final buffer.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bx_rest.h"

struct bx_rest_client {
    std::string host, port, prefix;  // prefix: path part of the base URL without a trailing '/'
    uint64_t claim_wait = 0, io_timeout = 30;
    std::atomic<uint64_t> requests{0}, connects{0};
    std::mutex mu;          // guards `idle`
    std::vector<int> idle;  // kept-alive connections not in use
    ~bx_rest_client() {
        for (int fd : idle) close(fd);
    }
};

namespace {

thread_local std::string tl_err;
const char* fail(const std::string& m) {
    tl_err = m;
    return tl_err.c_str();
}
void put_err(char* buf, size_t cap, const std::string& m) {
    if (buf && cap) snprintf(buf, cap, "%s", m.c_str());
}

// RFC 3986 path-segment encoding; ':' and '@' stay (keys are "job:{uuid}:segments:{i}"), '/' stays when keep_slash (wildcard routes)
std::string enc_path(const std::string& s, bool keep_slash) {
    static const char* hex = "0123456789ABCDEF";
    std::string o;
    for (unsigned char ch : s) {
        const bool ok = (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '-' || ch == '.' ||
                        ch == '_' || ch == '~' || ch == ':' || ch == '@' || (keep_slash && ch == '/');
        if (ok) {
            o += (char)ch;
        } else {
            o += '%';
            o += hex[ch >> 4];
            o += hex[ch & 15];
        }
    }
    return o;
}
std::string json_escape(const char* s) {
    std::string o = "\"";
    for (const unsigned char* p = (const unsigned char*)s; *p; ++p) {
        switch (*p) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            default:
                if (*p < 0x20) {
                    char b[8];
                    snprintf(b, sizeof b, "\\u%04x", *p);
                    o += b;
                } else if (*p < 0x80) {
                    o += (char)*p;
                } else {
                    // error texts come from the hot store, the prover and the OS: arbitrary bytes.  Pass well-formed UTF-8
                    // sequences through, replace anything else by U+FFFD so that the body is always valid JSON
                    int len = (*p >= 0xC2 && *p <= 0xDF) ? 2 : (*p >= 0xE0 && *p <= 0xEF) ? 3 : (*p >= 0xF0 && *p <= 0xF4) ? 4 : 0;
                    bool ok = len != 0;
                    for (int k = 1; ok && k < len; ++k) ok = (p[k] & 0xC0) == 0x80;  // p[k] == 0 ends the string and fails here
                    if (ok && len == 3) ok = !(p[0] == 0xE0 && p[1] < 0xA0) && !(p[0] == 0xED && p[1] >= 0xA0);  // overlong, surrogates
                    if (ok && len == 4) ok = !(p[0] == 0xF0 && p[1] < 0x90) && !(p[0] == 0xF4 && p[1] >= 0x90);
                    if (ok) {
                        o.append((const char*)p, (size_t)len);
                        p += len - 1;
                    } else {
                        o += "\\ufffd";
                    }
                }
        }
    }
    return o + "\"";
}

// ---- a JSON reader that only needs to find the members of one object and hand back their raw text ----
struct JScan {
    const char* p;
    const char* end;
    bool ok = true;
    int depth = 0;  // nesting of the value being skipped (bounded: the recursion below is on the C stack)
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool str(std::string* out) {  // at '"': decodes the escapes that can occur in ids and error strings
        if (p >= end || *p != '"') return ok = false;
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return ok = false;
                switch (*p) {
                    case 'n': if (out) *out += '\n'; break;
                    case 't': if (out) *out += '\t'; break;
                    case 'r': if (out) *out += '\r'; break;
                    case 'b': if (out) *out += '\b'; break;
                    case 'f': if (out) *out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) return ok = false;
                        unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        if (out) {  // BMP code point -> UTF-8
                            if (v < 0x80) *out += (char)v;
                            else if (v < 0x800) { *out += (char)(0xC0 | (v >> 6)); *out += (char)(0x80 | (v & 63)); }
                            else { *out += (char)(0xE0 | (v >> 12)); *out += (char)(0x80 | ((v >> 6) & 63)); *out += (char)(0x80 | (v & 63)); }
                        }
                        p += 4;
                        break;
                    }
                    default: if (out) *out += *p;
                }
                ++p;
            } else {
                if (out) *out += *p;
                ++p;
            }
        }
        if (p >= end) return ok = false;
        ++p;
        return true;
    }
    bool skip() {  // any value
        ws();
        if (p >= end) return ok = false;
        if (*p == '"') return str(nullptr);
        if (*p == '{' || *p == '[') {
            if (depth >= 64) return ok = false;
            struct Nest {
                int& d;
                explicit Nest(int& x) : d(x) { ++d; }
                ~Nest() { --d; }
            } nest(depth);
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            ws();
            if (p < end && *p == close) { ++p; return true; }
            for (;;) {
                ws();
                if (open == '{') {
                    if (!str(nullptr)) return false;
                    ws();
                    if (p >= end || *p != ':') return ok = false;
                    ++p;
                }
                if (!skip()) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == close) { ++p; return true; }
                return ok = false;
            }
        }
        const char* s = p;
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;
        return p > s ? true : (ok = false);
    }
};
// members of the top-level object of `text` as (key, raw value text); false when `text` is not an object
bool json_members(const std::string& text, std::vector<std::pair<std::string, std::string>>* out) {
    JScan j{text.data(), text.data() + text.size()};
    j.ws();
    if (j.p >= j.end || *j.p != '{') return false;
    ++j.p;
    j.ws();
    if (j.p < j.end && *j.p == '}') return true;
    for (;;) {
        j.ws();
        std::string key;
        if (!j.str(&key)) return false;
        j.ws();
        if (j.p >= j.end || *j.p != ':') return false;
        ++j.p;
        j.ws();
        const char* v0 = j.p;
        if (!j.skip()) return false;
        out->emplace_back(key, std::string(v0, j.p));
        j.ws();
        if (j.p < j.end && *j.p == ',') { ++j.p; continue; }
        if (j.p < j.end && *j.p == '}') return true;
        return false;
    }
}
const std::string* member(const std::vector<std::pair<std::string, std::string>>& m, const char* key) {
    for (auto& kv : m)
        if (kv.first == key) return &kv.second;
    return nullptr;
}
// an i32 as serde reads one: an integer literal in range, nothing else (a string, a fraction or an overflow is a decode error)
bool json_i32_value(const std::string& raw, int32_t* out) {
    if (raw.empty() || raw.size() > 11) return false;
    size_t i = raw[0] == '-' ? 1 : 0;
    if (i == raw.size()) return false;
    for (size_t k = i; k < raw.size(); ++k)
        if (raw[k] < '0' || raw[k] > '9') return false;
    long long v = strtoll(raw.c_str(), nullptr, 10);
    if (v < INT32_MIN || v > INT32_MAX) return false;
    *out = (int32_t)v;
    return true;
}
bool json_string_value(const std::string& raw, std::string* out) {
    JScan j{raw.data(), raw.data() + raw.size()};
    return j.str(out);
}

// ---- one HTTP/1.1 exchange ----
// a segment blob is ~80 MB (bento/crates/workflow/src/tasks/executor.rs:45); nothing this client fetches comes near the cap
constexpr size_t MAX_RESPONSE_BYTES = (size_t)1 << 31;
constexpr size_t MAX_HEADER_BYTES = (size_t)1 << 16;
constexpr size_t MAX_IDLE_CONNECTIONS = 16;

// Response body: one malloc'ed buffer that the socket is read into directly and that a hot-store GET hands to the caller as it
// is (a segment is received once and never copied); JSON answers are moved into `body` for the decoders.
struct Response {
    int status = 0;
    std::string body;
    uint8_t* data = nullptr;
    size_t len = 0, cap = 0;
    Response() = default;
    Response(const Response&) = delete;
    Response& operator=(const Response&) = delete;
    ~Response() { free(data); }
    // room for `want` bytes in total, growing geometrically (a hostile Content-Length commits nothing before bytes arrive)
    bool reserve(size_t want, size_t limit) {
        if (want <= cap) return true;
        size_t c = cap ? cap : (size_t)1 << 16;
        while (c < want) c *= 2;
        if (c > limit) c = limit;
        uint8_t* q = (uint8_t*)realloc(data, c ? c : 1);
        if (!q) return false;
        data = q, cap = c;
        return true;
    }
    uint8_t* release() {
        uint8_t* p = data;
        data = nullptr, cap = 0;
        return p;
    }
};

bool send_all(int fd, const char* p, size_t n) {
    while (n) {
        ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}

// buffered reads for the head and the chunk framing; body bytes beyond what the buffer already holds go straight to their place
struct Reader {
    int fd;
    char buf[1 << 14];
    size_t lo = 0, hi = 0;
    bool any = false;  // a response byte was seen (a reused connection that dies before that is retried on a fresh one)
    int err = 0;       // errno of the recv that failed (0 = the peer closed)
    explicit Reader(int f) : fd(f) {}
    bool fill() {
        if (lo == hi) lo = hi = 0;
        if (hi == sizeof buf) return false;
        for (;;) {
            ssize_t k = recv(fd, buf + hi, sizeof buf - hi, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return err = k < 0 ? errno : 0, false;
            hi += (size_t)k;
            any = true;
            return true;
        }
    }
    // one line without its CRLF (a bare LF ends a line too); false on EOF, error or a line longer than `max`
    bool line(std::string* out, size_t max) {
        out->clear();
        for (;;) {
            while (lo < hi) {
                char ch = buf[lo++];
                if (ch == '\n') {
                    if (!out->empty() && out->back() == '\r') out->pop_back();
                    return true;
                }
                if (out->size() >= max) return false;
                out->push_back(ch);
            }
            if (!fill()) return false;
        }
    }
    bool exact(uint8_t* dst, size_t n) {
        size_t have = hi - lo < n ? hi - lo : n;
        memcpy(dst, buf + lo, have);
        lo += have, dst += have, n -= have;
        while (n) {
            ssize_t k = recv(fd, dst, n, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return err = k < 0 ? errno : 0, false;
            any = true;
            dst += k, n -= (size_t)k;
        }
        return true;
    }
};

int dial(bx_rest_client* c, std::string* err) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    int gai = getaddrinfo(c->host.c_str(), c->port.c_str(), &hints, &res);
    if (gai != 0) return *err = std::string("resolve ") + c->host + ": " + gai_strerror(gai), -1;
    int fd = -1;
    std::string last = "no address";
    for (addrinfo* ai = res; ai; ai = ai->ai_next) {
        fd = socket(ai->ai_family, ai->ai_socktype, ai->ai_protocol);
        if (fd < 0) continue;
        timeval tv{(time_t)c->io_timeout, 0};
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);  // also bounds connect()
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        if (connect(fd, ai->ai_addr, ai->ai_addrlen) == 0) break;
        last = strerror(errno);
        close(fd);
        fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) return *err = "connect " + c->host + ":" + c->port + ": " + last, -1;
    c->connects.fetch_add(1);
    return fd;
}

bool all_digits(const std::string& s, int base) {
    if (s.empty()) return false;
    for (unsigned char ch : s)
        if (!(base == 16 ? isxdigit(ch) : isdigit(ch))) return false;
    return true;
}
std::string trimmed(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
    return s.substr(a, b - a);
}

// One request on `fd`; "" or the transport error.  *reusable = the connection is positioned after a complete response and the
// server did not ask to close it; *dead_before_answer = nothing of a response arrived (send failed or the peer had closed).
std::string exchange(int fd, const std::string& head, const uint8_t* body, size_t body_len, Response* out, bool* reusable,
                     bool* dead_before_answer) {
    *reusable = false;
    *dead_before_answer = false;
    // "the idle connection was gone" means exactly that: the send hit a closed socket, or the peer's FIN / RST is all that came back.
    // A receive TIMEOUT (EAGAIN from SO_RCVTIMEO) is this request's own failure: re-sending a non-idempotent POST (the long-poll claim)
    // on a fresh socket would double the effective timeout and could claim a task twice.
    auto closed = [](int e) { return e == 0 || e == ECONNRESET || e == EPIPE || e == ENOTCONN; };
    if (!send_all(fd, head.data(), head.size()) || (body_len && !send_all(fd, (const char*)body, body_len))) {
        const int se = errno;
        *dead_before_answer = closed(se) && se != 0;
        return std::string("send: ") + strerror(se);
    }
    Reader rd(fd);
    std::string ln;
    int status = 0;
    bool keep = false, chunked = false, have_len = false;
    size_t content_len = 0;
    for (int interim = 0;; ++interim) {  // 1xx responses (100 Continue, 103 Early Hints) precede the final one on the same connection
      if (!rd.line(&ln, MAX_HEADER_BYTES)) {
        *dead_before_answer = !rd.any && closed(rd.err);
        if (rd.any) return "malformed HTTP response";
        if (rd.err == EAGAIN || rd.err == EWOULDBLOCK || rd.err == ETIMEDOUT) return "receive: timed out waiting for the response";
        return std::string("receive: ") + (rd.err ? strerror(rd.err) : "connection closed");
      }
      if (ln.compare(0, 5, "HTTP/") != 0) return "malformed HTTP response";
      keep = ln.compare(0, 8, "HTTP/1.1") == 0;
      size_t sp = ln.find(' ');
      status = 0;
      if (sp != std::string::npos)
        for (size_t k = sp + 1; k < ln.size() && k < sp + 10 && ln[k] >= '0' && ln[k] <= '9'; ++k) status = status * 10 + (ln[k] - '0');
      out->status = status;
      chunked = have_len = false;
      content_len = 0;
      size_t head_bytes = ln.size();
      for (;;) {
        if (!rd.line(&ln, MAX_HEADER_BYTES)) return "malformed HTTP response";
        if (ln.empty()) break;
        if ((head_bytes += ln.size() + 2) > MAX_HEADER_BYTES) return "response header larger than 64 KiB";
        size_t colon = ln.find(':');
        if (colon == std::string::npos) continue;
        std::string name = ln.substr(0, colon), value = trimmed(ln.substr(colon + 1));
        for (auto& ch : name) ch = (char)tolower((unsigned char)ch);
        for (auto& ch : value) ch = (char)tolower((unsigned char)ch);
        if (name == "content-length") {
            if (!all_digits(value, 10) || value.size() > 18) return "malformed Content-Length";
            content_len = strtoull(value.c_str(), nullptr, 10);
            have_len = true;
        } else if (name == "transfer-encoding") {
            if (value.find("chunked") != std::string::npos) chunked = true;
        } else if (name == "connection") {
            if (value.find("close") != std::string::npos) keep = false;
            else if (value.find("keep-alive") != std::string::npos) keep = true;
        }
      }
      if (status >= 100 && status < 200 && status != 101) {  // interim: the real answer follows
        if (interim >= 8) return "too many interim (1xx) responses";
        continue;
      }
      break;
    }
    const std::string too_large = "response larger than " + std::to_string(MAX_RESPONSE_BYTES >> 20) + " MiB";
    if (status == 204 || status == 304 || status == 101) {
        // no body by definition
        if (status == 101) keep = false;  // a protocol switch this client never asked for: do not reuse the socket
    } else if (chunked) {
        for (;;) {
            if (!rd.line(&ln, 1024)) return "malformed chunked body";
            size_t semi = ln.find(';');  // chunk extensions are ignored
            std::string num = trimmed(semi == std::string::npos ? ln : ln.substr(0, semi));
            if (!all_digits(num, 16)) return "malformed chunked body";  // no hex digits where a chunk size belongs
            if (num.size() > 15) return "truncated chunked body";       // a size no body of this client can have
            size_t n = strtoull(num.c_str(), nullptr, 16);
            if (n == 0) {  // trailers: bounded like the header they belong to
                size_t trailer_bytes = 0;
                for (;;) {
                    if (!rd.line(&ln, MAX_HEADER_BYTES)) return "malformed chunked body";
                    if (ln.empty()) break;
                    if ((trailer_bytes += ln.size() + 2) > MAX_HEADER_BYTES) return "chunked trailers larger than 64 KiB";
                }
                break;
            }
            if (n > MAX_RESPONSE_BYTES - out->len) return too_large;
            if (!out->reserve(out->len + n, MAX_RESPONSE_BYTES)) return "out of memory";
            uint8_t crlf[2];
            if (!rd.exact(out->data + out->len, n)) return "truncated chunked body";
            out->len += n;
            if (!rd.exact(crlf, 2) || crlf[0] != '\r' || crlf[1] != '\n') return "truncated chunked body";
        }
    } else if (have_len) {
        if (content_len > MAX_RESPONSE_BYTES) return too_large;
        // grow as bytes arrive: 1 MiB steps doubling, so a lying Content-Length costs nothing
        while (out->len < content_len) {
            size_t step = out->cap > out->len ? out->cap - out->len : 0;
            if (!step) {
                if (!out->reserve(out->len + 1, content_len)) return "out of memory";
                step = out->cap - out->len;
            }
            if (step > content_len - out->len) step = content_len - out->len;
            if (!rd.exact(out->data + out->len, step)) return "truncated body";
            out->len += step;
        }
    } else {
        keep = false;  // delimited by the end of the connection
        for (;;) {
            if (out->len == out->cap) {
                if (out->len >= MAX_RESPONSE_BYTES) return too_large;
                if (!out->reserve(out->len + 1, MAX_RESPONSE_BYTES)) return "out of memory";
            }
            size_t have = rd.hi - rd.lo;
            if (have) {
                size_t take = have < out->cap - out->len ? have : out->cap - out->len;
                memcpy(out->data + out->len, rd.buf + rd.lo, take);
                rd.lo += take, out->len += take;
                continue;
            }
            ssize_t k = recv(fd, out->data + out->len, out->cap - out->len, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k < 0) return std::string("receive: ") + strerror(errno);
            if (k == 0) break;
            out->len += (size_t)k;
        }
    }
    *reusable = keep && rd.lo == rd.hi;
    return "";
}

// returns "" or the transport error.  Connections are kept alive and reused (the reference's client is one shared, pooling reqwest::Client,
// assets.rs:76); a reused connection that turns out to be dead before any answer byte is replaced by a fresh one, once.
// `blob`: leave the body in out->data (hot-store values); otherwise it is moved to out->body for the JSON decoders.
std::string http_call(bx_rest_client* c, const char* method, const std::string& path_and_query, const char* content_type, const uint8_t* body,
                      size_t body_len, uint64_t extra_wait, Response* out, bool blob = false) {
    c->requests.fetch_add(1);
    std::string head = std::string(method) + " " + c->prefix + path_and_query + " HTTP/1.1\r\nHost: " + c->host + ":" + c->port +
                       "\r\nConnection: keep-alive\r\nAccept: */*\r\n";
    if (body || !strcmp(method, "POST") || !strcmp(method, "PUT")) {
        if (content_type) head += std::string("Content-Type: ") + content_type + "\r\n";
        head += "Content-Length: " + std::to_string(body_len) + "\r\n";
    }
    head += "\r\n";
    for (int attempt = 0;; ++attempt) {
        int fd = -1;
        bool reused = false;
        if (attempt == 0) {
            std::lock_guard<std::mutex> g(c->mu);
            if (!c->idle.empty()) {
                fd = c->idle.back();
                c->idle.pop_back();
                reused = true;
            }
        }
        if (fd < 0) {
            std::string e;
            fd = dial(c, &e);
            if (fd < 0) return e;
        }
        timeval tv{(time_t)(c->io_timeout + extra_wait), 0};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
        bool reusable = false, dead = false;
        out->status = 0, out->len = 0;
        std::string e = exchange(fd, head, body, body_len, out, &reusable, &dead);
        if (e.empty() && reusable) {
            std::lock_guard<std::mutex> g(c->mu);
            if (c->idle.size() < MAX_IDLE_CONNECTIONS) {
                c->idle.push_back(fd);
                fd = -1;
            }
        }
        if (fd >= 0) close(fd);
        if (!e.empty() && reused && dead) continue;  // the server had dropped the idle connection: not this request's failure
        if (!e.empty()) return e;
        if (!blob) {
            out->body.assign((const char*)out->data, out->len);
            free(out->release());
            out->len = 0;
        }
        return "";
    }
}

std::string task_url(const char* job, const char* task, const char* action) {
    return "/worker/gpu/tasks/" + enc_path(job, false) + "/" + enc_path(task, false) + "/" + action;
}
// POST .../{action} -> {"updated": bool}; 1 / 0 / -1
int task_update(bx_rest_client* c, const char* job, const char* task, const char* action, const std::string* body, char* eb, size_t cap) {
    Response r;
    std::string e = http_call(c, "POST", task_url(job, task, action), body ? "application/json" : nullptr,
                              body ? (const uint8_t*)body->data() : nullptr, body ? body->size() : 0, 0, &r);
    if (!e.empty()) return put_err(eb, cap, std::string("task ") + action + " " + job + ":" + task + ": " + e), -1;
    if (r.status >= 400 || r.status < 200)
        return put_err(eb, cap, std::string("task ") + action + " update failed for " + job + ":" + task + ": HTTP " + std::to_string(r.status) + " " + r.body.substr(0, 160)), -1;
    std::vector<std::pair<std::string, std::string>> m;
    const std::string* u = json_members(r.body, &m) ? member(m, "updated") : nullptr;
    if (!u) return put_err(eb, cap, std::string("failed to decode task ") + action + " response for " + job + ":" + task), -1;
    return *u == "true" ? 1 : 0;
}

// ---- bx_taskdb_ops ----
int rest_request_work(void* user, const char* stream, bx_ready_task* out, char* eb, size_t cap) {
    auto* c = (bx_rest_client*)user;
    try {
        Response r;
        std::string path = "/worker/gpu/tasks/claim/" + enc_path(stream, false) + "?wait_timeout_secs=" + std::to_string(c->claim_wait);
        std::string e = http_call(c, "POST", path, nullptr, nullptr, 0, c->claim_wait, &r);
        if (!e.empty()) return put_err(eb, cap, std::string("failed to claim GPU work for stream ") + stream + ": " + e), -1;
        if (r.status >= 400 || r.status < 200)
            return put_err(eb, cap, std::string("GPU work claim failed for stream ") + stream + ": HTTP " + std::to_string(r.status) + " " + r.body.substr(0, 160)), -1;
        JScan ws{r.body.data(), r.body.data() + r.body.size()};
        ws.ws();
        if ((size_t)(ws.end - ws.p) >= 4 && !strncmp(ws.p, "null", 4)) return 0;  // Json(None)
        std::vector<std::pair<std::string, std::string>> m;
        if (!json_members(r.body, &m)) return put_err(eb, cap, "failed to decode GPU work claim response"), -1;
        const std::string *job = member(m, "job_id"), *task = member(m, "task_id"), *def = member(m, "task_def"), *mr = member(m, "max_retries");
        std::string job_s, task_s;
        if (!job || !task || !def || !mr || !json_string_value(*job, &job_s) || !json_string_value(*task, &task_s))
            return put_err(eb, cap, "GPU work claim response lacks job_id / task_id / task_def / max_retries"), -1;
        if (job_s.size() >= sizeof out->job_id || task_s.size() >= sizeof out->task_id || def->size() >= sizeof out->task_def)
            return put_err(eb, cap, "claimed task " + job_s + ":" + task_s + " does not fit bx_ready_task"), -1;
        int32_t max_retries = 0;
        if (!json_i32_value(*mr, &max_retries)) return put_err(eb, cap, "failed to decode GPU work claim response: max_retries is not an i32"), -1;
        memset(out, 0, sizeof *out);
        memcpy(out->job_id, job_s.data(), job_s.size());
        memcpy(out->task_id, task_s.data(), task_s.size());
        memcpy(out->task_def, def->data(), def->size());  // the raw JSON text of task_def, e.g. {"Prove":{"index":3}}
        out->max_retries = max_retries;
        return 1;
    } catch (const std::exception& ex) {
        return put_err(eb, cap, std::string("claim: ") + ex.what()), -1;
    }
}
int rest_done(void* user, const char* job, const char* task, const char* output_json, char* eb, size_t cap) {
    try {
        std::string body = std::string("{\"output\":") + (output_json && *output_json ? output_json : "null") + "}";
        return task_update((bx_rest_client*)user, job, task, "done", &body, eb, cap);
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}
int rest_failed(void* user, const char* job, const char* task, const char* error, char* eb, size_t cap) {
    try {
        std::string body = "{\"error\":" + json_escape(error ? error : "") + "}";
        return task_update((bx_rest_client*)user, job, task, "failed", &body, eb, cap);
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}
int rest_retry(void* user, const char* job, const char* task, char* eb, size_t cap) {
    try {
        return task_update((bx_rest_client*)user, job, task, "retry", nullptr, eb, cap);
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}
int rest_current_retries(void* user, const char* job, const char* task, int32_t* retries, char* eb, size_t cap) {
    auto* c = (bx_rest_client*)user;
    try {
        Response r;
        std::string e = http_call(c, "GET", task_url(job, task, "retries-running"), nullptr, nullptr, 0, 0, &r);
        if (!e.empty()) return put_err(eb, cap, std::string("failed to fetch retries for task ") + job + ":" + task + ": " + e), -1;
        if (r.status >= 400 || r.status < 200) return put_err(eb, cap, std::string("task retries fetch failed for ") + job + ":" + task + ": HTTP " + std::to_string(r.status)), -1;
        std::vector<std::pair<std::string, std::string>> m;
        const std::string* v = json_members(r.body, &m) ? member(m, "retries") : nullptr;
        if (!v) return put_err(eb, cap, std::string("failed to decode retries-running response for ") + job + ":" + task), -1;
        if (*v == "null") return 0;  // Option::None: no running row
        if (!json_i32_value(*v, retries)) return put_err(eb, cap, std::string("failed to decode retries-running response for ") + job + ":" + task), -1;
        return 1;
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}

// ---- bx_hot_store_ops ----
int rest_hot_get(void* user, const char* key, uint8_t** value, size_t* len, char* eb, size_t cap) {
    auto* c = (bx_rest_client*)user;
    try {
        Response r;
        std::string e = http_call(c, "GET", "/worker/hot/" + enc_path(key, true), nullptr, nullptr, 0, 0, &r, /*blob=*/true);
        if (!e.empty()) return put_err(eb, cap, std::string("failed to fetch hot-store key ") + key + ": " + e), -1;
        if (r.status == 404) return 1;  // AppError::HotDataMissing
        if (r.status >= 400 || r.status < 200) return put_err(eb, cap, std::string("hot-store fetch failed for key ") + key + ": HTTP " + std::to_string(r.status)), -1;
        if (!r.data && !r.reserve(1, 1)) return put_err(eb, cap, "out of memory"), -1;  // an empty value is still a value
        *len = r.len;
        *value = r.release();  // the buffer the socket was read into
        return 0;
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}
void rest_hot_free(void*, uint8_t* v) { free(v); }
int rest_hot_set(void* user, const char* key, const uint8_t* value, size_t len, uint64_t ttl, char* eb, size_t cap) {
    auto* c = (bx_rest_client*)user;
    try {
        Response r;
        std::string path = "/worker/hot/" + enc_path(key, true);
        if (ttl) path += "?ttl_secs=" + std::to_string(ttl);
        std::string e = http_call(c, "PUT", path, "application/octet-stream", value, len, 0, &r);
        if (!e.empty()) return put_err(eb, cap, std::string("failed to write hot-store key ") + key + ": " + e), -1;
        if (r.status >= 400 || r.status < 200) return put_err(eb, cap, std::string("hot-store write failed for key ") + key + ": HTTP " + std::to_string(r.status)), -1;
        return 0;
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}
int rest_hot_unlink(void* user, const char* key, char* eb, size_t cap) {
    auto* c = (bx_rest_client*)user;
    try {
        Response r;
        std::string e = http_call(c, "DELETE", "/worker/hot/" + enc_path(key, true), nullptr, nullptr, 0, 0, &r);
        if (!e.empty()) return put_err(eb, cap, std::string("failed to delete hot-store key ") + key + ": " + e), -1;
        if (r.status >= 400 || r.status < 200) return put_err(eb, cap, std::string("hot-store delete failed for key ") + key + ": HTTP " + std::to_string(r.status)), -1;
        return 0;
    } catch (const std::exception& ex) {
        return put_err(eb, cap, ex.what()), -1;
    }
}

}  // namespace

extern "C" {

const char* bx_rest_client_create(const char* base_url, uint64_t claim_wait_secs, uint64_t io_timeout_secs, bx_rest_client** out) {
    if (!base_url || !out) return "bx_rest_client_create: NULL argument";
    try {
        std::string u = base_url;
        while (!u.empty() && u.back() == '/') u.pop_back();
        if (u.empty()) return "bx_rest_client_create: API URL must not be empty";  // assets.rs:69-71
        if (u.compare(0, 7, "http://") != 0) return fail("bx_rest_client_create: only http:// URLs are supported (TLS terminates in front of the API): " + u);
        std::string rest = u.substr(7), hostport = rest, prefix;
        size_t slash = rest.find('/');
        if (slash != std::string::npos) {
            hostport = rest.substr(0, slash);
            prefix = rest.substr(slash);
        }
        if (hostport.empty()) return fail("bx_rest_client_create: failed to parse API URL: " + u);
        auto* c = new bx_rest_client();
        size_t colon = hostport.rfind(':');
        if (colon != std::string::npos && hostport.find(']') == std::string::npos) {
            c->host = hostport.substr(0, colon);
            c->port = hostport.substr(colon + 1);
        } else {
            c->host = hostport;
            c->port = "80";
        }
        c->prefix = prefix;
        c->claim_wait = claim_wait_secs;
        c->io_timeout = io_timeout_secs ? io_timeout_secs : 30;
        if (c->host.empty() || c->port.empty() || c->port.find_first_not_of("0123456789") != std::string::npos) {
            delete c;
            return fail("bx_rest_client_create: failed to parse API URL: " + u);
        }
        *out = c;
        return nullptr;
    } catch (const std::exception& ex) {
        return fail(std::string("bx_rest_client_create: ") + ex.what());
    }
}
void bx_rest_client_destroy(bx_rest_client* c) { delete c; }
bx_taskdb_ops bx_rest_taskdb_ops(bx_rest_client* c) {
    return bx_taskdb_ops{c, rest_request_work, rest_done, rest_failed, rest_retry, rest_current_retries};
}
bx_hot_store_ops bx_rest_hot_store_ops(bx_rest_client* c) {
    return bx_hot_store_ops{c, rest_hot_get, rest_hot_free, rest_hot_set, rest_hot_unlink};
}
uint64_t bx_rest_client_requests(const bx_rest_client* c) { return c ? c->requests.load() : 0; }
uint64_t bx_rest_client_connects(const bx_rest_client* c) { return c ? c->connects.load() : 0; }

}  // extern "C"
