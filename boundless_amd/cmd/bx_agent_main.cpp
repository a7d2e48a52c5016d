// bx-agent — the GPU worker process of a prover node, on the C ABI of this library.
//
// What the reference starts once per GPU (`agent -t prove`, compose.yml:113): parse flags / environment, build the Agent, serve
// Prometheus metrics, poll for work until SIGTERM.
//   bento/crates/workflow/src/bin/agent.rs:13-36            main: Args::parse, Agent::new, start_metrics_exporter, poll_work
//   prover/crates/workflow/src/bin/agent.rs:13-29           the same for the next-generation worker (work comes over the Bento API)
//   prover/crates/workflow/src/lib.rs:56-175                Args: names, short flags, environment variables and defaults kept here
//   bento/crates/workflow/src/lib.rs:268-272                create_sig_monitor: SIGTERM / SIGINT set a flag the loops poll
//   bento/crates/workflow-common/src/metrics.rs:184-203     start_metrics_exporter: PROMETHEUS_METRICS_ADDR, default 0.0.0.0:9090
// Work is claimed and blobs move over the next-generation worker protocol (include/bx_rest.h); the feed loop, the lanes, the
// prover and the verifier are the library's (include/bx_agent.h).  Host code only: this file contains no arithmetic.
//
// The built-in prover proves the SYNTHETIC circuit (include/bx_prover.h), so the process refuses to start without --synthetic:
// pointed at a production API it would otherwise claim rv32im segments it cannot prove.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bx_agent.h"
#include "bx_rest.h"

namespace {

struct Options {
    std::string task_stream;                      // -t, --task-stream        TASK_STREAM          (required)
    double poll_time = 1;                         // -p, --poll-time          POLL_TIME            seconds
    std::string api_url = "http://localhost:8081";  //   --api-url            BENTO_API_URL
    uint64_t redis_ttl = 8 * 60 * 60;             //     --redis-ttl          REDIS_TTL            seconds
    bool monitor_requeue = false;                 //     --monitor-requeue    MONITOR_REQUEUE
    double requeue_poll_interval = 5;             //     --requeue-poll-interval REQUEUE_POLL_INTERVAL
    std::string metrics_addr = "0.0.0.0:9090";    //     --metrics-addr       PROMETHEUS_METRICS_ADDR   ("off" = no exporter)
    // ---- not in the reference: what one process per node, several lanes per GPU and the synthetic circuit need
    bool synthetic = false;                       //     --synthetic          BX_SYNTHETIC
    std::vector<int> devices;                     //     --devices 0,1,..     BX_DEVICES           default: device 0 (after HIP_VISIBLE_DEVICES)
    uint32_t inflight = 0;                        //     --inflight           BX_INFLIGHT          lanes per GPU (0 = 3)
    uint32_t widths[3] = {0, 0, 0};               //     --widths c,d,a       BX_WIDTHS            0 = 16,256,64
    uint32_t po2_min = 0, po2_max = 0;            //     --po2-min/--po2-max  BX_PO2_MIN/MAX
    uint32_t join_po2 = 0, lift_po2 = 0;          //     --join-po2/--lift-po2
    std::string also_streams;                     //     --also-streams       BX_ALSO_STREAMS
    bool prefetch = false;                        //     --prefetch           BX_PREFETCH
    bool no_verify = false;                       //     --no-verify
    uint64_t claim_wait_secs = 0;                 //     --claim-wait-secs    BX_CLAIM_WAIT_SECS   long-poll of the claim route
    uint64_t io_timeout_secs = 0;                 //     --io-timeout-secs    BX_IO_TIMEOUT_SECS
    int64_t max_idle_polls = -1;                  //     --max-idle-polls     exit after this many empty polls per lane (batch runs, tests)
    bool print_config = false;                    //     --print-config       print the parsed options as JSON and exit
};

const char* USAGE =
    "bx-agent: MI355X segment-prove worker (synthetic circuit) for a Bento API (next-generation worker protocol)\n"
    "\n"
    "usage: bx-agent -t <stream> --synthetic [options]\n"
    "\n"
    "  -t, --task-stream <s>          worker type to claim from: prove, join, ...            [env TASK_STREAM]\n"
    "  -p, --poll-time <secs>         sleep between empty claims (default 1)                  [env POLL_TIME]\n"
    "      --api-url <url>            Bento API base URL (default http://localhost:8081)      [env BENTO_API_URL]\n"
    "      --redis-ttl <secs>         expiry of stored receipts (default 28800)               [env REDIS_TTL]\n"
    "      --monitor-requeue          run the requeue monitor (needs a task db that has one)  [env MONITOR_REQUEUE]\n"
    "      --requeue-poll-interval <secs>  (default 5)                                        [env REQUEUE_POLL_INTERVAL]\n"
    "      --metrics-addr <ip:port>   Prometheus exporter (default 0.0.0.0:9090, 'off' = none) [env PROMETHEUS_METRICS_ADDR]\n"
    "      --synthetic                accept the synthetic segment format (required)          [env BX_SYNTHETIC]\n"
    "      --devices <a,b,..>         HIP device ordinals served by this process (default 0)  [env BX_DEVICES]\n"
    "      --inflight <n>             proofs in flight per device (default 3)                 [env BX_INFLIGHT]\n"
    "      --widths <code,data,accum> group widths (default 16,256,64)                        [env BX_WIDTHS]\n"
    "      --po2-min <n> --po2-max <n>  segment sizes accepted (default 9..22)                [env BX_PO2_MIN, BX_PO2_MAX]\n"
    "      --join-po2 <n> --lift-po2 <n>  stand-in recursion proof sizes (default 18, off)\n"
    "      --also-streams <a,b>       worker types claimed from when the main one is empty    [env BX_ALSO_STREAMS]\n"
    "      --prefetch                 claim one task ahead per lane and download under the current proof [env BX_PREFETCH]\n"
    "      --no-verify                do not verify seals before storing them\n"
    "      --claim-wait-secs <n>      long-poll the claim route (default 0)                   [env BX_CLAIM_WAIT_SECS]\n"
    "      --io-timeout-secs <n>      HTTP timeout (default: the library's)                   [env BX_IO_TIMEOUT_SECS]\n"
    "      --max-idle-polls <n>       exit after n consecutive empty polls per lane (default: run until SIGTERM)\n"
    "      --print-config             print the parsed options as JSON and exit\n"
    "  -h, --help\n";

[[noreturn]] void usage_error(const std::string& m) {
    fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", m.c_str());
    exit(2);
}

bool parse_u64(const std::string& s, uint64_t* out) {
    if (s.empty() || s[0] == '-' || s[0] == '+') return false;
    errno = 0;
    char* end = nullptr;
    unsigned long long v = strtoull(s.c_str(), &end, 10);
    if (errno || *end) return false;
    *out = v;
    return true;
}
bool parse_f64(const std::string& s, double* out) {
    if (s.empty()) return false;
    errno = 0;
    char* end = nullptr;
    double v = strtod(s.c_str(), &end);
    if (errno || *end || !(v >= 0) || v > 1e9) return false;
    *out = v;
    return true;
}
bool parse_bool(const std::string& s, bool* out) {  // clap's bool env parsing: true/false (also 1/0, yes/no, on/off)
    static const char* yes[] = {"true", "1", "yes", "on", "y", "t"};
    static const char* no[] = {"false", "0", "no", "off", "n", "f", ""};
    std::string l;
    for (char c : s) l.push_back((char)tolower((unsigned char)c));
    for (auto* y : yes)
        if (l == y) return *out = true, true;
    for (auto* n : no)
        if (l == n) return *out = false, true;
    return false;
}
std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) {
        if (c == sep) {
            out.push_back(cur);
            cur.clear();
        } else {
            cur.push_back(c);
        }
    }
    out.push_back(cur);
    return out;
}

// one setter per option: used for the environment first, then for the command line (which wins, as with clap)
struct Setter {
    const char* flag;   // long flag
    const char* env;    // environment variable or nullptr
    bool takes_value;
    bool (*set)(Options&, const std::string&);
};
bool set_u32(uint32_t* dst, const std::string& v, uint64_t max) {
    uint64_t x;
    if (!parse_u64(v, &x) || x > max) return false;
    *dst = (uint32_t)x;
    return true;
}
const Setter SETTERS[] = {
    {"--task-stream", "TASK_STREAM", true, [](Options& o, const std::string& v) { return !v.empty() && v.size() < 64 ? (o.task_stream = v, true) : false; }},
    {"--poll-time", "POLL_TIME", true, [](Options& o, const std::string& v) { return parse_f64(v, &o.poll_time); }},
    {"--api-url", "BENTO_API_URL", true, [](Options& o, const std::string& v) { return !v.empty() ? (o.api_url = v, true) : false; }},
    {"--redis-ttl", "REDIS_TTL", true, [](Options& o, const std::string& v) { return parse_u64(v, &o.redis_ttl); }},
    {"--monitor-requeue", "MONITOR_REQUEUE", false, [](Options& o, const std::string& v) { return parse_bool(v, &o.monitor_requeue); }},
    {"--requeue-poll-interval", "REQUEUE_POLL_INTERVAL", true, [](Options& o, const std::string& v) { return parse_f64(v, &o.requeue_poll_interval); }},
    {"--metrics-addr", "PROMETHEUS_METRICS_ADDR", true, [](Options& o, const std::string& v) { return !v.empty() ? (o.metrics_addr = v, true) : false; }},
    {"--synthetic", "BX_SYNTHETIC", false, [](Options& o, const std::string& v) { return parse_bool(v, &o.synthetic); }},
    {"--devices", "BX_DEVICES", true,
     [](Options& o, const std::string& v) {
         std::vector<int> d;
         for (auto& p : split(v, ',')) {
             uint64_t x;
             if (!parse_u64(p, &x) || x > 1023) return false;
             for (int seen : d)
                 if (seen == (int)x) return false;
             d.push_back((int)x);
         }
         if (d.empty() || d.size() > 16) return false;
         o.devices = d;
         return true;
     }},
    {"--inflight", "BX_INFLIGHT", true, [](Options& o, const std::string& v) { return set_u32(&o.inflight, v, 64); }},
    {"--widths", "BX_WIDTHS", true,
     [](Options& o, const std::string& v) {
         auto p = split(v, ',');
         if (p.size() != 3) return false;
         uint32_t w[3];
         for (int i = 0; i < 3; ++i)
             if (!set_u32(&w[i], p[i], 65535) || w[i] == 0) return false;
         memcpy(o.widths, w, sizeof w);
         return true;
     }},
    {"--po2-min", "BX_PO2_MIN", true, [](Options& o, const std::string& v) { return set_u32(&o.po2_min, v, 24); }},
    {"--po2-max", "BX_PO2_MAX", true, [](Options& o, const std::string& v) { return set_u32(&o.po2_max, v, 24); }},
    {"--join-po2", "BX_JOIN_PO2", true, [](Options& o, const std::string& v) { return set_u32(&o.join_po2, v, 24); }},
    {"--lift-po2", "BX_LIFT_PO2", true, [](Options& o, const std::string& v) { return set_u32(&o.lift_po2, v, 24); }},
    {"--also-streams", "BX_ALSO_STREAMS", true, [](Options& o, const std::string& v) { return v.size() < 128 ? (o.also_streams = v, true) : false; }},
    {"--prefetch", "BX_PREFETCH", false, [](Options& o, const std::string& v) { return parse_bool(v, &o.prefetch); }},
    {"--no-verify", nullptr, false, [](Options& o, const std::string& v) { return parse_bool(v, &o.no_verify); }},
    {"--claim-wait-secs", "BX_CLAIM_WAIT_SECS", true, [](Options& o, const std::string& v) { return parse_u64(v, &o.claim_wait_secs); }},
    {"--io-timeout-secs", "BX_IO_TIMEOUT_SECS", true, [](Options& o, const std::string& v) { return parse_u64(v, &o.io_timeout_secs); }},
    {"--max-idle-polls", nullptr, true,
     [](Options& o, const std::string& v) {
         uint64_t x;
         if (!parse_u64(v, &x) || x > (uint64_t)1 << 40) return false;
         o.max_idle_polls = (int64_t)x;
         return true;
     }},
    {"--print-config", nullptr, false, [](Options& o, const std::string& v) { return parse_bool(v, &o.print_config); }},
};

Options parse(int argc, char** argv) {
    Options o;
    for (const Setter& s : SETTERS) {
        const char* e = s.env ? getenv(s.env) : nullptr;
        if (e && !s.set(o, e)) usage_error(std::string("invalid value '") + e + "' for environment variable " + s.env);
    }
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], inline_value;
        bool has_inline = false;
        if (a == "-h" || a == "--help") {
            fputs(USAGE, stdout);
            exit(0);
        }
        if (a == "-t") a = "--task-stream";
        else if (a == "-p") a = "--poll-time";
        else if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            if (eq != std::string::npos) {
                inline_value = a.substr(eq + 1);
                a.resize(eq);
                has_inline = true;
            }
        }
        const Setter* hit = nullptr;
        for (const Setter& s : SETTERS)
            if (a == s.flag) hit = &s;
        if (!hit) usage_error("unexpected argument '" + std::string(argv[i]) + "' found");
        std::string value = "true";
        if (hit->takes_value) {
            if (has_inline) value = inline_value;
            else if (i + 1 < argc) value = argv[++i];
            else usage_error(std::string("a value is required for '") + hit->flag + "' but none was supplied");
        } else if (has_inline) {
            value = inline_value;
        }
        if (!hit->set(o, value)) usage_error("invalid value '" + value + "' for '" + hit->flag + "'");
    }
    if (o.task_stream.empty()) usage_error("the following required arguments were not provided:\n  --task-stream <TASK_STREAM>");
    if (o.po2_min && o.po2_max && o.po2_min > o.po2_max) usage_error("--po2-min is larger than --po2-max");
    return o;
}

std::string json_escape(const std::string& s) {
    std::string out;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') out.push_back('\\'), out.push_back((char)c);
        else if (c < 0x20) {
            char b[8];
            snprintf(b, sizeof b, "\\u%04x", c);
            out += b;
        } else out.push_back((char)c);
    }
    return out;
}
void print_config(const Options& o) {
    std::string dev;
    for (size_t i = 0; i < o.devices.size(); ++i) dev += (i ? "," : "") + std::to_string(o.devices[i]);
    printf("{\"task_stream\":\"%s\",\"poll_time\":%g,\"api_url\":\"%s\",\"redis_ttl\":%llu,\"monitor_requeue\":%s,"
           "\"requeue_poll_interval\":%g,\"metrics_addr\":\"%s\",\"synthetic\":%s,\"devices\":[%s],\"inflight\":%u,"
           "\"widths\":[%u,%u,%u],\"po2_min\":%u,\"po2_max\":%u,\"join_po2\":%u,\"lift_po2\":%u,\"also_streams\":\"%s\","
           "\"prefetch\":%s,\"no_verify\":%s,\"claim_wait_secs\":%llu,\"io_timeout_secs\":%llu,\"max_idle_polls\":%lld}\n",
           json_escape(o.task_stream).c_str(), o.poll_time, json_escape(o.api_url).c_str(), (unsigned long long)o.redis_ttl,
           o.monitor_requeue ? "true" : "false", o.requeue_poll_interval, json_escape(o.metrics_addr).c_str(), o.synthetic ? "true" : "false",
           dev.c_str(), o.inflight, o.widths[0], o.widths[1], o.widths[2], o.po2_min, o.po2_max, o.join_po2, o.lift_po2,
           json_escape(o.also_streams).c_str(), o.prefetch ? "true" : "false", o.no_verify ? "true" : "false",
           (unsigned long long)o.claim_wait_secs, (unsigned long long)o.io_timeout_secs, (long long)o.max_idle_polls);
}

// ---------------------------------------------------------------------------------------------------- signals
std::atomic<int> g_term{0};
void on_signal(int) { g_term.store(1); }

// ---------------------------------------------------------------------------------------------------- metrics exporter
// GET /metrics (any path, as prometheus_exporter answers) -> the agent's Prometheus exposition.  One request per connection.
struct Exporter {
    int fd = -1;
    uint16_t port = 0;
    std::thread th;
    std::atomic<int> quit{0};

    // "ip:port" -> bound, listening socket; returns "" or the error
    std::string start(const std::string& addr, bx_agent* agent) {
        size_t colon = addr.rfind(':');
        uint64_t p = 0;
        sockaddr_in sa;
        memset(&sa, 0, sizeof sa);
        sa.sin_family = AF_INET;
        if (colon == std::string::npos || !parse_u64(addr.substr(colon + 1), &p) || p > 65535 ||
            inet_pton(AF_INET, addr.substr(0, colon).c_str(), &sa.sin_addr) != 1)
            return "not an IPv4 socket address: " + addr;
        sa.sin_port = htons((uint16_t)p);
        fd = socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) return std::string("socket: ") + strerror(errno);
        int one = 1;
        setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (bind(fd, (sockaddr*)&sa, sizeof sa) != 0 || listen(fd, 16) != 0) {
            std::string e = std::string("bind ") + addr + ": " + strerror(errno);
            close(fd);
            fd = -1;
            return e;
        }
        socklen_t len = sizeof sa;
        getsockname(fd, (sockaddr*)&sa, &len);
        port = ntohs(sa.sin_port);
        timeval tv{0, 200000};  // accept() wakes up five times a second to look at `quit`
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        th = std::thread([this, agent] { serve(agent); });
        return "";
    }
    void serve(bx_agent* agent) {
        std::vector<char> body(1 << 16);
        while (!quit.load()) {
            int c = accept(fd, nullptr, nullptr);
            if (c < 0) continue;
            timeval tv{2, 0};
            setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
            char req[2048];
            size_t got = 0;
            while (got < sizeof req - 1) {  // read the request head; its content does not matter
                ssize_t k = recv(c, req + got, sizeof req - 1 - got, 0);
                if (k <= 0) break;
                got += (size_t)k;
                req[got] = 0;
                if (strstr(req, "\r\n\r\n")) break;
            }
            size_t n = bx_agent_metrics(agent, body.data(), body.size());
            if (n >= body.size()) {  // truncated: size the buffer for the full text and ask again
                body.resize(n + 1024);
                n = bx_agent_metrics(agent, body.data(), body.size());
                if (n >= body.size()) n = body.size() - 1;
            }
            char head[160];
            int h = snprintf(head, sizeof head,
                             "HTTP/1.1 200 OK\r\nContent-Type: text/plain; version=0.0.4\r\nContent-Length: %zu\r\nConnection: close\r\n\r\n", n);
            std::string out(head, (size_t)h);
            out.append(body.data(), n);
            size_t sent = 0;
            while (sent < out.size()) {
                ssize_t k = send(c, out.data() + sent, out.size() - sent, MSG_NOSIGNAL);
                if (k <= 0) break;
                sent += (size_t)k;
            }
            close(c);
        }
    }
    void stop() {
        quit.store(1);
        if (th.joinable()) th.join();
        if (fd >= 0) close(fd);
        fd = -1;
    }
};

}  // namespace

int main(int argc, char** argv) {
    Options o = parse(argc, argv);
    if (o.print_config) {
        print_config(o);
        return 0;
    }
    if (!o.synthetic) {
        fprintf(stderr,
                "Error: [BENTO-AGENT-001] Failed to initialize Agent: the built-in prover proves the synthetic circuit only "
                "(include/bx_prover.h); start with --synthetic against an API that serves synthetic segments\n");
        return 1;
    }
    // create_sig_monitor (lib.rs:268-272)
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);

    bx_rest_client* rest = nullptr;
    if (const char* e = bx_rest_client_create(o.api_url.c_str(), o.claim_wait_secs, o.io_timeout_secs, &rest)) {
        fprintf(stderr, "Error: [BENTO-AGENT-001] Failed to initialize Agent: %s\n", e);
        return 1;
    }
    bx_taskdb_ops tops = bx_rest_taskdb_ops(rest);
    bx_hot_store_ops sops = bx_rest_hot_store_ops(rest);

    bx_agent_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.inflight = o.inflight;
    cfg.w_code = o.widths[0], cfg.w_data = o.widths[1], cfg.w_accum = o.widths[2];
    cfg.redis_ttl = o.redis_ttl;
    cfg.poll_time = o.poll_time > 0 ? o.poll_time : 1e-3;  // the library reads <= 0 as "default"; 0 here means "do not sleep"
    cfg.no_verify = o.no_verify;
    snprintf(cfg.task_stream, sizeof cfg.task_stream, "%s", o.task_stream.c_str());
    cfg.synthetic = 1;
    if (o.devices.size() == 1) {
        cfg.device = o.devices[0];
    } else if (o.devices.size() > 1) {
        cfg.n_devices = (uint32_t)o.devices.size();
        for (size_t i = 0; i < o.devices.size(); ++i) cfg.devices[i] = o.devices[i];
    }
    cfg.po2_min = o.po2_min, cfg.po2_max = o.po2_max;
    cfg.join_po2 = o.join_po2, cfg.lift_po2 = o.lift_po2;
    snprintf(cfg.also_streams, sizeof cfg.also_streams, "%s", o.also_streams.c_str());
    cfg.prefetch = o.prefetch;
    cfg.monitor_requeue = o.monitor_requeue;
    cfg.requeue_poll_interval = o.requeue_poll_interval;
    if (o.monitor_requeue && !tops.requeue_tasks)
        fprintf(stderr, "warning: --monitor-requeue: the Bento API requeues timed-out tasks itself; no monitor is started here\n");

    bx_agent* agent = nullptr;
    if (const char* e = bx_agent_create(&cfg, &sops, &tops, nullptr, &agent)) {
        fprintf(stderr, "Error: [BENTO-AGENT-001] Failed to initialize Agent: %s\n", e);
        bx_rest_client_destroy(rest);
        return 1;
    }
    Exporter exporter;
    if (o.metrics_addr != "off") {
        std::string e = exporter.start(o.metrics_addr, agent);
        if (!e.empty()) fprintf(stderr, "Failed to start metrics server: %s\n", e.c_str());  // the reference logs and carries on
        else fprintf(stderr, "metrics exporter listening on port %u\n", exporter.port);
    }
    fprintf(stderr, "bx-agent: stream '%s', %u lane(s), API %s\n", o.task_stream.c_str(), bx_agent_lane_count(agent), o.api_url.c_str());

    // the signal flag is forwarded to the library's stop flag by a watcher: bx_agent_stop is not async-signal-safe
    std::atomic<int> poll_done{0};
    std::thread watcher([&] {
        while (!poll_done.load()) {
            if (g_term.load()) {
                bx_agent_stop(agent);
                return;
            }
            usleep(20000);
        }
    });
    uint64_t done = 0;
    const char* err = bx_agent_poll_work(agent, o.max_idle_polls, &done);
    poll_done.store(1);
    watcher.join();
    if (g_term.load()) fprintf(stderr, "Handled SIGTERM, shutting down...\n");
    fprintf(stderr, "bx-agent: %llu task(s) completed\n", (unsigned long long)done);
    int rc = 0;
    if (err) {
        fprintf(stderr, "Error: [BENTO-AGENT-002] Exiting agent polling: %s\n", err);
        rc = 1;
    }
    exporter.stop();
    bx_agent_destroy(agent);
    bx_rest_client_destroy(rest);
    return rc;
}
