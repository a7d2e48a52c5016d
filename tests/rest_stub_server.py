"""A local stand-in for the next-gen Bento API's worker routes (prover/crates/api/src/lib.rs:922-1040), for tests.

Test infrastructure only: an in-memory task table + hot store behind the eight routes a GPU worker uses, with the reference's
status codes (404 HotDataMissing, 400 InvalidGpuWorkerStream, 204 for PUT/DELETE) and JSON shapes (WorkerTask, TaskUpdateRes,
TaskRetriesRunningRes).  State transitions follow taskdb (ready -> running -> done | failed, retry puts a running task back).
"""
import json
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from urllib.parse import parse_qs, unquote, urlparse

GPU_STREAMS = {"prove", "join", "coproc", "snark"}


class State:
    def __init__(self):
        self.mu = threading.RLock()  # handlers answer while holding it
        self.hot = {}      # key -> (bytes, deadline or None)
        self.tasks = []    # dicts: stream, job_id, task_id, task_def, max_retries, retries, state, error, output
        self.log = []      # (method, path)
        self.fail_next = 0  # respond 500 to this many upcoming requests (fault injection)
        self.keep_alive = True   # HTTP/1.1 persistent connections, as the reference's axum server keeps them
        self.drop_every = 0      # close the connection after every n-th response ...
        self.drop_silently = False  # ... without announcing it ("Connection: close" left out)
        self.served = 0
        self.stall_next = 0.0    # the next request is read and logged, then the handler sleeps this long before answering (a hung upstream)
        self.get_delay = 0.0     # every GET of a hot key takes this long before the body is sent (an ~80 MB segment download)
        self.interim_next = 0    # answer the next request with this many "100 Continue" interim responses before the real one

    def create_task(self, stream, job_id, task_id, task_def, max_retries=0):
        with self.mu:
            self.tasks.append(dict(stream=stream, job_id=job_id, task_id=task_id, task_def=task_def, max_retries=max_retries,
                                   retries=0, state="ready", error="", output=None))

    def find(self, job_id, task_id):
        for t in self.tasks:
            if t["job_id"] == job_id and t["task_id"] == task_id:
                return t
        return None


def make_handler(st):
    class H(BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"
        wbufsize = -1  # headers and body leave in ONE segment (handle_one_request flushes): unbuffered, the body waits ~40 ms for the
        #                client's delayed ACK of the headers (Nagle), which no real server does

        def log_message(self, *a):
            pass

        def _send(self, code, body=b"", ctype="application/json"):
            self.send_response(code)
            self.send_header("Content-Type", ctype)
            self.send_header("Content-Length", str(len(body)))
            with st.mu:
                st.served += 1
                drop = not st.keep_alive or (st.drop_every and st.served % st.drop_every == 0)
                silent = drop and st.drop_silently
            if drop and not silent:
                self.send_header("Connection", "close")
            self.end_headers()
            if body:
                self.wfile.write(body)
            if drop:
                self.close_connection = True  # silent: what a server's idle timeout looks like to a pooling client

        def _json(self, obj, code=200):
            self._send(code, json.dumps(obj).encode())

        def _body(self):
            n = int(self.headers.get("Content-Length") or 0)
            return self.rfile.read(n) if n else b""

        def _route(self, method):
            u = urlparse(self.path)
            path, q = unquote(u.path), parse_qs(u.query)
            body = self._body()
            with st.mu:
                st.log.append((method, path))
                stall, st.stall_next = st.stall_next, 0.0
                interim, st.interim_next = st.interim_next, 0
            if stall:
                time.sleep(stall)
            for _ in range(interim):
                self.wfile.write(b"HTTP/1.1 100 Continue\r\nX-Interim: 1\r\n\r\n")
                self.wfile.flush()
            with st.mu:
                if st.fail_next > 0:
                    st.fail_next -= 1
                    return self._send(500, b'{"type":"InternalErr","msg":"injected"}')
            parts = path.strip("/").split("/")
            if parts[:2] == ["worker", "hot"]:
                key = "/".join(parts[2:])
                if method == "GET" and st.get_delay:
                    time.sleep(st.get_delay)  # outside the lock: other requests are served meanwhile
                with st.mu:
                    if method == "GET":
                        v = st.hot.get(key)
                        if v and v[1] is not None and v[1] < time.time():
                            del st.hot[key]
                            v = None
                        if v is None:
                            return self._json({"type": "HotDataMissing", "msg": key}, 404)
                        return self._send(200, v[0], "application/octet-stream")
                    if method == "PUT":
                        ttl = q.get("ttl_secs")
                        st.hot[key] = (body, time.time() + int(ttl[0]) if ttl else None)
                        return self._send(204)
                    if method == "DELETE":
                        st.hot.pop(key, None)
                        return self._send(204)
            if parts[:4] == ["worker", "gpu", "tasks", "claim"] and method == "POST" and len(parts) == 5:
                stream = parts[4]
                if stream not in GPU_STREAMS:
                    return self._json({"type": "InvalidGpuWorkerStream", "msg": stream}, 400)
                deadline = time.time() + int(q.get("wait_timeout_secs", ["0"])[0])
                while True:
                    with st.mu:
                        # taskdb::request_work: the oldest ready task of the OLDEST job (9_request_work.sql:139-141)
                        first = {}
                        for i, t in enumerate(st.tasks):
                            first.setdefault(t["job_id"], i)
                        ready = [(first[t["job_id"]], i) for i, t in enumerate(st.tasks) if t["state"] == "ready" and t["stream"] == stream]
                        if ready:
                            t = st.tasks[min(ready)[1]]
                            t["state"] = "running"
                            return self._json(dict(job_id=t["job_id"], task_id=t["task_id"], task_def=t["task_def"], prereqs=[],
                                                   max_retries=t["max_retries"]))
                    if time.time() >= deadline:
                        return self._send(200, b"null")
                    time.sleep(0.01)
            if parts[:3] == ["worker", "gpu", "tasks"] and len(parts) == 6:
                job_id, task_id, action = parts[3], parts[4], parts[5]
                with st.mu:
                    t = st.find(job_id, task_id)
                    if action == "retries-running" and method == "GET":
                        return self._json({"retries": t["retries"] if t and t["state"] == "running" else None})
                    if method != "POST":
                        return self._send(405)
                    if action == "done":
                        ok = bool(t) and t["state"] in ("ready", "running")
                        if ok:
                            t["state"], t["output"] = "done", json.loads(body)["output"]
                        return self._json({"updated": ok})
                    if action == "failed":
                        ok = bool(t) and t["state"] in ("ready", "running")
                        if ok:
                            t["state"], t["error"] = "failed", json.loads(body)["error"]
                        return self._json({"updated": ok})
                    if action == "retry":
                        ok = bool(t) and t["state"] == "running"
                        if ok:
                            t["retries"] += 1
                            t["state"] = "ready"
                            if t["retries"] > t["max_retries"]:
                                t["state"], t["error"], ok = "failed", "retry max hit", False
                        return self._json({"updated": ok})
            return self._send(404, b'{"type":"NotFound"}')

        def do_GET(self):
            self._route("GET")

        def do_POST(self):
            self._route("POST")

        def do_PUT(self):
            self._route("PUT")

        def do_DELETE(self):
            self._route("DELETE")

    return H


class StubServer:
    def __init__(self):
        self.state = State()
        self.httpd = ThreadingHTTPServer(("127.0.0.1", 0), make_handler(self.state))
        self.httpd.daemon_threads = True
        self.port = self.httpd.server_address[1]
        self.url = f"http://127.0.0.1:{self.port}"
        self._t = threading.Thread(target=self.httpd.serve_forever, daemon=True)
        self._t.start()

    def close(self):
        self.httpd.shutdown()
        self.httpd.server_close()
