"""GPU: `bx-agent` as a worker process of a Bento API — started the way `agent -t prove` is (compose.yml:113), against the local
stub of the API's worker routes: claims over HTTP, proves on the GPU, verifies, stores, reports; serves Prometheus metrics
meanwhile; leaves on SIGTERM.  Seals are compared with the CPU oracle's word for word."""
import os
import signal
import subprocess
import sys
import time
import urllib.request

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rest_stub_server import StubServer  # noqa: E402

from boundless_amd import agent as ag  # noqa: E402
from boundless_amd import build  # noqa: E402
from boundless_amd.prover import Segment, verify_seal  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

pytestmark = pytest.mark.gpu
JOB = "0b1e55ed-0000-4000-8000-0000000000b1"


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_the_worker_process_serves_a_bento_api_and_leaves_on_sigterm():
    build.build(verbose=False)
    po2, widths, n = 12, (4, 12, 4), 7
    srv = StubServer()
    port = free_port()
    try:
        for i in range(n):
            srv.state.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=po2)), None)
            srv.state.create_task("prove", JOB, f"prove-{i}", {"Prove": {"index": i}}, max_retries=1)
        env = dict(os.environ, BENTO_API_URL=srv.url, PROMETHEUS_METRICS_ADDR=f"127.0.0.1:{port}", BX_WIDTHS=",".join(map(str, widths)))
        p = subprocess.Popen([build.AGENT_BIN, "-t", "prove", "-p", "0.02", "--synthetic", "--inflight", "2", "--prefetch"], env=env,
                             stderr=subprocess.PIPE, text=True)
        try:
            t0 = time.monotonic()
            while time.monotonic() - t0 < 120 and not all(t["state"] == "done" for t in srv.state.tasks):
                assert p.poll() is None, p.stderr.read()
                time.sleep(0.05)
            assert [t["state"] for t in srv.state.tasks] == ["done"] * n
            text = urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=10).read().decode()
            assert f'task_operations_total{{task_name="prove",operation_type="complete",status="success"}} {n}' in text
            assert 'task_claims_total{task_stream="prove",result="claimed"}' in text
            assert p.poll() is None  # still polling: it runs until it is told to stop
            p.send_signal(signal.SIGTERM)
            rc = p.wait(timeout=30)
            err = p.stderr.read()
            assert rc == 0 and "Handled SIGTERM, shutting down..." in err and f"{n} task(s) completed" in err, err
        finally:
            if p.poll() is None:
                p.kill()
        assert sorted(srv.state.hot) == sorted(f"job:{JOB}:synthetic_receipts:prove-{i}" for i in range(n))
        for i in range(n):
            rec = ag.deserialize_receipt(srv.state.hot[f"job:{JOB}:synthetic_receipts:prove-{i}"][0])
            want, _ = ol.prove_segment(po2, *widths, Segment.synthetic(i, po2=po2).seed)
            assert np.array_equal(rec.seal, want)
            verify_seal(rec.seal)
    finally:
        srv.close()


def test_a_batch_run_ends_by_itself_and_an_unreachable_api_is_a_fatal_error():
    build.build(verbose=False)
    srv = StubServer()
    try:
        for i in range(3):
            srv.state.hot[f"job:{JOB}:segments:{i}"] = (ag.serialize_segment(Segment.synthetic(i, po2=11)), None)
            srv.state.create_task("prove", JOB, f"prove-{i}", {"Prove": {"index": i}}, max_retries=1)
        r = subprocess.run([build.AGENT_BIN, "-t", "prove", "-p", "0.01", "--synthetic", "--inflight", "1", "--widths", "2,6,2", "--api-url", srv.url,
                            "--metrics-addr", "off", "--max-idle-polls", "3"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "3 task(s) completed" in r.stderr, r.stderr
        assert [t["state"] for t in srv.state.tasks] == ["done"] * 3
    finally:
        srv.close()
    r = subprocess.run([build.AGENT_BIN, "-t", "prove", "--synthetic", "--inflight", "1", "--api-url", "http://127.0.0.1:9", "--metrics-addr", "off",
                        "--max-idle-polls", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "[BENTO-AGENT-002] Exiting agent polling" in r.stderr and "[BENTO-WF-107] Failed to request_work" in r.stderr
