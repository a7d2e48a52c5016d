"""CPU: `bx-agent`, the worker process (boundless_amd/cmd/bx_agent_main.cpp) — its command line and environment.

Reference: `agent -t prove` (bento/crates/workflow/src/bin/agent.rs:13-36, prover/crates/workflow/src/bin/agent.rs:13-29) with the
clap `Args` of prover/crates/workflow/src/lib.rs:56-175: same flag names, short flags, environment variables and defaults for the
options the two have in common; the command line wins over the environment; a missing --task-stream is a usage error (exit 2).
Without a GPU the process must fail loudly at start-up (no CPU fallback); the run against an API is tests/test_agent_binary_gpu.py.
"""
import json
import os
import subprocess

import pytest

from boundless_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe():
    build.build(verbose=False)
    assert os.access(build.AGENT_BIN, os.X_OK)
    return build.AGENT_BIN


def run(exe, *args, env=None):
    e = {k: v for k, v in os.environ.items() if not k.startswith(("BX_", "BENTO_")) and k not in ("TASK_STREAM", "POLL_TIME", "REDIS_TTL",
         "MONITOR_REQUEUE", "REQUEUE_POLL_INTERVAL", "PROMETHEUS_METRICS_ADDR")}
    e.update(env or {})
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=60, env=e)


def test_defaults_are_the_reference_agents(exe):
    r = run(exe, "-t", "prove", "--print-config")
    assert r.returncode == 0, r.stderr
    c = json.loads(r.stdout)
    assert c["task_stream"] == "prove" and c["poll_time"] == 1 and c["api_url"] == "http://localhost:8081"  # lib.rs:62,68,101-102
    assert c["redis_ttl"] == 8 * 60 * 60 and c["monitor_requeue"] is False and c["requeue_poll_interval"] == 5  # :89-90,105,154
    assert c["metrics_addr"] == "0.0.0.0:9090"  # workflow-common metrics.rs:193-196
    assert c["synthetic"] is False and c["max_idle_polls"] == -1 and c["devices"] == [] and c["inflight"] == 0


def test_environment_then_command_line(exe):
    env = {"TASK_STREAM": "join", "POLL_TIME": "7", "BENTO_API_URL": "http://api:8081", "REDIS_TTL": "60", "MONITOR_REQUEUE": "true",
           "PROMETHEUS_METRICS_ADDR": "127.0.0.1:9191", "BX_DEVICES": "0,1,2", "BX_INFLIGHT": "2", "BX_SYNTHETIC": "1", "BX_WIDTHS": "4,12,4"}
    c = json.loads(run(exe, "--print-config", env=env).stdout)
    assert (c["task_stream"], c["poll_time"], c["api_url"], c["redis_ttl"], c["monitor_requeue"]) == ("join", 7, "http://api:8081", 60, True)
    assert (c["metrics_addr"], c["devices"], c["inflight"], c["synthetic"], c["widths"]) == ("127.0.0.1:9191", [0, 1, 2], 2, True, [4, 12, 4])
    c = json.loads(run(exe, "-t", "prove", "-p", "0.25", "--api-url=http://other:1", "--devices", "3", "--prefetch", "--max-idle-polls", "5",
                       "--print-config", env=env).stdout)
    got = (c["task_stream"], c["poll_time"], c["api_url"], c["devices"], c["prefetch"], c["max_idle_polls"])
    assert got == ("prove", 0.25, "http://other:1", [3], True, 5)
    assert c["redis_ttl"] == 60 and c["inflight"] == 2  # untouched by the command line: still the environment's


@pytest.mark.parametrize("args,needle", [
    ((), "required arguments were not provided"),
    (("-t", "prove", "--bogus"), "unexpected argument '--bogus'"),
    (("-t", "prove", "--inflight"), "a value is required for '--inflight'"),
    (("-t", "prove", "--inflight", "many"), "invalid value 'many' for '--inflight'"),
    (("-t", "prove", "--devices", "0,0"), "invalid value '0,0' for '--devices'"),
    (("-t", "prove", "--widths", "16,256"), "invalid value '16,256' for '--widths'"),
    (("-t", "prove", "--po2-min", "20", "--po2-max", "12"), "--po2-min is larger"),
    (("-t", "x" * 64), "invalid value"),
])
def test_usage_errors_exit_2_with_claps_wording(exe, args, needle):
    r = run(exe, *args)
    assert r.returncode == 2 and needle in r.stderr and "--help" in r.stderr and r.stdout == ""


def test_a_bad_environment_value_names_the_variable(exe):
    r = run(exe, "-t", "prove", env={"POLL_TIME": "soon"})
    assert r.returncode == 2 and "POLL_TIME" in r.stderr


def test_help_lists_every_option_with_its_environment_variable(exe):
    r = run(exe, "--help")
    assert r.returncode == 0
    for flag in ("--task-stream", "--poll-time", "--api-url", "--redis-ttl", "--monitor-requeue", "--metrics-addr", "--synthetic", "--devices",
                 "--inflight", "--prefetch", "--max-idle-polls"):
        assert flag in r.stdout
    for env in ("TASK_STREAM", "POLL_TIME", "BENTO_API_URL", "REDIS_TTL", "PROMETHEUS_METRICS_ADDR"):
        assert env in r.stdout


def test_refuses_to_serve_a_production_api_and_fails_loudly_without_a_gpu(exe):
    r = run(exe, "-t", "prove", "--metrics-addr", "off")
    assert r.returncode == 1 and "[BENTO-AGENT-001] Failed to initialize Agent" in r.stderr and "--synthetic" in r.stderr
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the start-up failure below cannot be provoked")
    r = run(exe, "-t", "prove", "--synthetic", "--metrics-addr", "off")
    assert r.returncode == 1
    assert "[BENTO-AGENT-001] Failed to initialize Agent" in r.stderr and "no HIP device visible" in r.stderr and "no CPU fallback" in r.stderr
