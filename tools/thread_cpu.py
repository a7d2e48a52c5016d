"""Which threads of an agent-mode run use the CPU?  Per-thread utime+stime from /proc/self/task around the timed run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def snap():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except (OSError, ValueError):
            pass
    return out


def plain():
    import threading

    from boundless_amd.prover import HipProverServer, Segment

    servers = [HipProverServer(device=0, po2=20, widths=(16, 256, 64)) for _ in range(3)]

    def work(sv, n, base):
        for i in range(n):
            sv.prove_segment(Segment.synthetic(index=base + i, po2=20))

    import resource

    for phase, n in (("warm", 1), ("timed", 10)):
        s0 = snap()
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(sv, n, 100 * k)) for k, sv in enumerate(servers)]
        [t.start() for t in ts]
        mid = None
        time.sleep(0.3 if phase == "timed" else 0.0)
        mid = snap()
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        s1 = snap()
        r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"rusage: user {r1.ru_utime - r0.ru_utime:.2f}s sys {r1.ru_stime - r0.ru_stime:.2f}s; BX_WAIT={os.environ.get('BX_WAIT')}")
    print(f"plain bench flow: wall {dt:.2f}s for 30 proofs; per-thread CPU of the persistent threads (lane threads measured up to 0.3 s before the end are in 'mid')")
    rows = sorted(((s1[t][1] - s0.get(t, (None, 0))[1], t) for t in s1), reverse=True)
    for cpu, tid in rows[:6]:
        print(f"  {cpu:6.2f}s tid {tid}")
    rows = sorted(((mid[t][1] - s0.get(t, (None, 0))[1], t) for t in mid if t not in s0), reverse=True)
    for cpu, tid in rows[:6]:
        print(f"  lane thread after 0.3 s: {cpu:6.2f}s tid {tid}")


def main():
    if "--plain" in sys.argv:
        return plain()
    from boundless_amd import agent as ag
    from boundless_amd.prover import Segment

    verify = "--no-verify" not in sys.argv
    a = ag.Agent(prover=None, device=0, inflight=3, poll_time=0.001, verify=verify)

    def enqueue(job, n):
        for i in range(n):
            a.store.set_key_with_expiry(f"job:{job}:segments:{i}", ag.serialize_segment(Segment.synthetic(i, po2=20)), 600)
            a.taskdb.create_task(job, f"prove-{i}", {"Prove": {"index": i}})

    enqueue("warm", 3)
    a.poll_work(max_idle_polls=1)
    enqueue("timed", 30)
    s0 = snap()
    t0 = time.perf_counter()
    a.poll_work(max_idle_polls=1)
    dt = time.perf_counter() - t0
    s1 = snap()
    rows = sorted(((s1[t][1] - s0.get(t, (None, 0))[1], s1[t][0], t) for t in s1), reverse=True)
    print(f"verify={verify} wall {dt:.2f}s for 30 proofs")
    for cpu, comm, tid in rows[:12]:
        print(f"  {cpu:6.2f}s  {comm:20s} tid {tid}{'  (new)' if tid not in s0 else ''}")
    a.close()


if __name__ == "__main__":
    main()
