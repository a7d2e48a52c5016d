"""Host CPU budget of the prover: does one GPU's share of the box's CPUs keep its lanes fed?

The GPU boxes give a container 16 CPUs (cgroup cpu.max) for 8 GPUs = 2 CPUs per GPU, against 3 lane threads + finishers per GPU.
This runs bench.py (one GPU) with the process restricted to 2 CPUs and unrestricted, under both wait policies
(BX_WAIT=block: host threads sleep on a blocking event; spin: hipStreamSynchronize busy-polls), reads `value` and
`host_cpu_s_per_proof`, and the agent-mode figures with and without CPU verification of every seal.

    python tools/host_budget.py [--steps 10] > profiles/r03_host_budget.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundless_amd.build import csrc_hash, device_code_hash  # noqa: E402


def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", STEPS, "--warmup", "3", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": (r.stderr or r.stdout)[-300:]}
    j = json.loads(lines[-1])
    out = {"segment_proofs_per_s": round(j["value"], 3), "host_cpu_s_per_proof": j.get("host_cpu_s_per_proof"),
           "cpus_busy_avg": (j.get("host") or {}).get("cpus_busy_avg")}
    am = j.get("agent_mode") or {}
    for k in ("verify_on", "verify_off"):
        if k in am:
            out["agent_" + k] = {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in am[k].items() if kk in
                                 ("segment_proofs_per_s", "host_cpu_s_per_proof", "cpus_busy_avg")}
    return out


if __name__ == "__main__":
    STEPS = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "10"
    res = {"device_code_sha": device_code_hash(), "csrc_sha": csrc_hash(), "what": __doc__.split("\n\n")[0], "command": f"bench.py --steps {STEPS} --warmup 3 --no-cpu-baseline [--cpus 2] --wait block|spin",
           "cpus_allowed": len(os.sched_getaffinity(0)), "runs": {}}
    for cpus in (0, 2, 1):
        for wait in ("block", "poll", "spin"):
            key = f"cpus={'all' if not cpus else cpus},wait={wait}"
            res["runs"][key] = run((["--cpus", str(cpus)] if cpus else []) + ["--wait", wait])
            print(key, res["runs"][key], file=sys.stderr)
    base = res["runs"]["cpus=all,wait=block"].get("segment_proofs_per_s")
    two = res["runs"]["cpus=2,wait=block"].get("segment_proofs_per_s")
    if base and two:
        res["loss_at_2_cpus_blocking_pct"] = round(100 * (1 - two / base), 2)
    two_s = res["runs"]["cpus=2,wait=spin"].get("segment_proofs_per_s")
    base_s = res["runs"]["cpus=all,wait=spin"].get("segment_proofs_per_s")
    if base_s and two_s:
        res["loss_at_2_cpus_spinning_pct"] = round(100 * (1 - two_s / base_s), 2)
    print(json.dumps(res, indent=1))
