"""CPU: the in-memory task db against a row-by-row restatement of the reference's SQL, on random operation sequences.

Reference: bento/crates/taskdb/migrations/1_taskdb.sql — create_task (:197-228), request_work (as replaced by 9_request_work.sql
:118-167: the oldest ready task of the OLDEST job, `ORDER BY job_created_at ASC, created_at ASC`), update_task_done (:278-314), update_task_failed (:316-347), update_task_retry (:361-391),
the job row's state and error (:287-311, :333-340).  `Model` below is that SQL with Python lists for tables; the library
(csrc/agent.cpp: bx_mem_taskdb) must agree with it after every operation — on every row, every job and every return value.

One documented difference, not generated here: a prerequisite listed TWICE.  The SQL counts it twice at creation and releases it
once (UPDATE ... FROM joins a target row once), leaving the task pending for ever; the library releases it twice.  The planner
never emits one.
"""
import ctypes as C
import random

import pytest

from boundless_amd import agent as ag
from boundless_amd.hal import HalError

_UPD3 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)
_UPD2 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)
_REQ = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(ag._ReadyTask), C.c_char_p, C.c_size_t)
_CUR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t)


class Lib:
    """The library's table through the same callback table the agent uses."""

    def __init__(self):
        self.db = ag.TaskDb()
        o = self.db.ops
        self.u = o.user
        self._done, self._failed = _UPD3(o.update_task_done), _UPD3(o.update_task_failed)
        self._retry, self._req, self._cur = _UPD2(o.update_task_retry), _REQ(o.request_work), _CUR(o.current_retries)

    def create_task(self, stream, job, task, pre, max_retries, timeout=None):
        try:
            self.db.create_task(job, task, {"Prove": {"index": 0}}, max_retries=max_retries, stream=stream, prerequisites=pre,
                                timeout_secs=timeout)
            return True
        except HalError:
            return False

    def advance(self, seconds):
        self.db.advance_clock(seconds)

    def requeue(self, limit=100):
        return self.db.requeue_tasks(limit)

    def request_work(self, stream):
        out = ag._ReadyTask()
        rc = self._req(self.u, stream.encode(), C.byref(out), None, 0)
        assert rc in (0, 1)
        return (out.job_id.decode(), out.task_id.decode(), out.max_retries) if rc else None

    def done(self, job, task, output):
        return self._done(self.u, job.encode(), task.encode(), output.encode(), None, 0) == 1

    def failed(self, job, task, error):
        return self._failed(self.u, job.encode(), task.encode(), error.encode(), None, 0) == 1

    def retry(self, job, task):
        return self._retry(self.u, job.encode(), task.encode(), None, 0) == 1

    def clear_completed_jobs(self):
        return self.db.clear_completed_jobs()

    def current_retries(self, job, task):
        r = C.c_int32(-1)
        rc = self._cur(self.u, job.encode(), task.encode(), C.byref(r), None, 0)
        return r.value if rc == 1 else None


class Model:
    """1_taskdb.sql with lists for tables.  A stream is a worker type here (one stream per type: include/bx_agent.h)."""

    def __init__(self):
        self.tasks = []  # dict rows, insertion order = created_at order
        self.deps = []   # (job, pre, post)
        self.jobs = {}   # job -> {"state", "error", "created"}; dicts keep insertion order = job_created_at order
        self.now = 0.0   # moved by advance() only: the real time a test takes is far below the margins the generator leaves

    def _row(self, job, task):
        for r in self.tasks:
            if r["job"] == job and r["task"] == task:
                return r
        return None

    def advance(self, seconds):
        self.now += seconds

    def requeue(self, limit=100):
        """requeue_tasks, bento/crates/taskdb/src/lib.rs:328-358 (GREATEST ignores a NULL updated_at)."""
        timed_out = [r for r in self.tasks if r["state"] == "running" and r["timeout"] is not None
                     and r["timeout"] < self.now - max(r["started"], r["updated"] or 0.0)][:limit]
        for r in timed_out:
            self.retry(r["job"], r["task"])
        return len(timed_out)

    def create_task(self, stream, job, task, pre, max_retries, timeout=None):
        if self._row(job, task) is not None:  # PRIMARY KEY (job_id, task_id)
            return False
        if any(self._row(job, p) is None for p in pre):  # FOREIGN KEY (job_id, pre_task_id)
            return False
        if job not in self.jobs:
            self.n_jobs_ever = getattr(self, "n_jobs_ever", 0) + 1
            self.jobs[job] = {"state": "running", "error": "", "created": self.n_jobs_ever}  # the job row comes with its first task
        row = dict(stream=stream, job=job, task=task, state="pending", waiting_on=0, retries=0, max_retries=max_retries, error="", output="",
                   timeout=timeout, started=0.0, updated=None)
        self.tasks.append(row)
        for p in pre:
            self.deps.append((job, p, task))
        not_done = sum(1 for (j, p, post) in self.deps if j == job and post == task and self._row(job, p)["state"] != "done")
        row["waiting_on"] = not_done
        row["state"] = "ready" if not_done == 0 else "pending"
        return True

    def request_work(self, stream):
        ready = [(self.jobs[r["job"]]["created"], i) for i, r in enumerate(self.tasks) if r["stream"] == stream and r["state"] == "ready"]
        if not ready:
            return None
        r = self.tasks[min(ready)[1]]  # ORDER BY job_created_at ASC, created_at ASC LIMIT 1
        r["state"], r["started"] = "running", self.now
        return (r["job"], r["task"], r["max_retries"])

    def clear_completed_jobs(self):
        gone = [j for j, row in self.jobs.items() if row["state"] == "done"]
        self.tasks = [t for t in self.tasks if t["job"] not in gone]
        self.deps = [d for d in self.deps if d[0] not in gone]
        for j in gone:
            del self.jobs[j]
        return len(gone)

    def done(self, job, task, output):
        r = self._row(job, task)
        if r is None or r["state"] not in ("ready", "running"):
            return False
        r["state"], r["output"], r["updated"] = "done", output, self.now
        for (j, p, post) in self.deps:
            if j == job and p == task:
                d = self._row(job, post)
                if d["state"] != "failed":
                    d["state"] = "ready" if d["waiting_on"] == 1 else "pending"
                    d["waiting_on"] -= 1
        if all(t["state"] == "done" for t in self.tasks if t["job"] == job):
            self.jobs[job]["state"] = "done"
        return True

    def failed(self, job, task, error):
        r = self._row(job, task)
        if r is None or r["state"] not in ("ready", "running", "pending"):
            return False
        r["state"], r["error"], r["updated"] = "failed", error, self.now
        if self.jobs[job]["state"] != "failed":
            self.jobs[job].update(state="failed", error=error)
        return True

    def retry(self, job, task):
        r = self._row(job, task)
        if r is None or r["state"] != "running":
            return False
        r["retries"] += 1
        r["state"], r["error"], r["updated"] = "ready", "", self.now
        if r["retries"] > r["max_retries"]:
            self.failed(job, task, "retry max hit")
            return False
        return True

    def current_retries(self, job, task):
        r = self._row(job, task)
        return r["retries"] if r is not None and r["state"] == "running" else None


def compare(lib, model, trail):
    for r in model.tasks:
        got = lib.db.task(r["job"], r["task"])
        want = (r["state"], r["waiting_on"], r["retries"], r["max_retries"], r["error"], r["output"], r["timeout"] or 0x7FFFFFFF)
        assert (got.state, got.waiting_on, got.retries, got.max_retries, got.error, got.output, got.timeout_secs) == want, (r["job"], r["task"], trail[-8:])
    for job, j in model.jobs.items():
        rows = [t for t in model.tasks if t["job"] == job]
        got = lib.db.job(job)
        assert got["state"] == j["state"] and got["error"] == j["error"], (job, got, j, trail[-8:])
        assert got["tasks"] == len(rows)
        for s in ("pending", "ready", "running", "done", "failed"):
            assert got[s] == sum(1 for t in rows if t["state"] == s), (job, s, trail[-8:])
    for s in ag.TASK_STATES:
        assert lib.db.count(s) == sum(1 for t in model.tasks if t["state"] == s)


@pytest.mark.parametrize("seed", range(40))
def test_random_operation_sequences_agree_with_the_sql(seed):
    rng = random.Random(0xDB0000 + seed)
    lib, model, trail = Lib(), Model(), []
    jobs, streams = ["J0", "J1", "J2"][: 1 + seed % 3], ["prove", "join", "aux"]
    names = {j: [] for j in jobs}
    for step in range(220):
        kind = rng.choices(["create", "request", "done", "failed", "retry", "current", "clear", "advance", "requeue"],
                           weights=[30, 25, 25, 3 if seed % 4 else 0, 10, 5, 2 if seed % 2 else 0, 6, 6])[0]
        job = rng.choice(jobs)
        known = names[job]
        pick = (lambda: rng.choice(known)) if known else (lambda: "none")
        if kind == "create":
            task = f"t{len(known)}" if rng.random() < 0.95 else pick()  # now and then a duplicate id
            pre = rng.sample(known, k=min(len(known), rng.choice([0, 0, 1, 2, 2, 3])))
            if rng.random() < 0.03:
                pre = pre + ["ghost"]  # a prerequisite that does not exist
            # timeouts 2 / 5 / 9 s against a clock that moves in multiples of 0.37 s: never within 0.1 s of a timeout
            args = (rng.choice(streams), job, task, pre, rng.choice([0, 0, 1, 2]), rng.choice([None, 2, 5, 9]))
            a, b = lib.create_task(*args), model.create_task(*args)
            if b and task not in known:
                known.append(task)
        elif kind == "request":
            args = (rng.choice(streams),)
            a, b = lib.request_work(*args), model.request_work(*args)
        elif kind == "done":
            args = (job, pick(), rng.choice(["null", '{"x":1}']))
            a, b = lib.done(*args), model.done(*args)
        elif kind == "failed":
            args = (job, pick(), f"boom {step}")
            a, b = lib.failed(*args), model.failed(*args)
        elif kind == "retry":
            args = (job, pick())
            a, b = lib.retry(*args), model.retry(*args)
        elif kind == "advance":
            args = (0.37 * rng.randint(1, 8),)
            a, b = lib.advance(*args), model.advance(*args)
        elif kind == "requeue":
            args = (100,)
            a, b = lib.requeue(*args), model.requeue(*args)
        elif kind == "clear":
            args = ()
            a, b = lib.clear_completed_jobs(), model.clear_completed_jobs()
            for j in jobs:
                if j not in model.jobs:
                    names[j] = []  # the job is gone: its ids are free again, and a new task re-creates the job row (youngest)
                    with pytest.raises(HalError, match="no such job"):
                        lib.db.job(j)
        else:
            args = (job, pick())
            a, b = lib.current_retries(*args), model.current_retries(*args)
        trail.append((kind, args, a, b))
        assert a == b, trail[-8:]
        compare(lib, model, trail)
    assert sum(1 for t in trail if t[0] == "request" and t[2]) > 5  # the walk did claim things


def test_a_task_created_in_a_finished_job_does_not_reopen_it():
    """create_task's own TODO (1_taskdb.sql:207-208): nothing stops a task being added to a done job, and the job row stays
    'done' until the next update_task_done re-counts."""
    lib, model = Lib(), Model()
    for t in (lib, model):
        assert t.create_task("prove", "J", "a", [], 0)
        assert t.done("J", "a", "null")
        assert t.create_task("prove", "J", "b", ["a"], 0)
    compare(lib, model, [])
    assert lib.db.job("J")["state"] == "done" and lib.db.task("J", "b").state == "ready"


def test_the_jobs_error_is_the_first_failure_in_time_not_in_creation_order():
    lib, model = Lib(), Model()
    for t in (lib, model):
        for name in "abc":
            assert t.create_task("prove", "J", name, [], 0)
        assert t.failed("J", "c", "third task, first failure")
        assert t.failed("J", "a", "first task, second failure")
    compare(lib, model, [])
    assert lib.db.job("J")["error"] == "third task, first failure"


def test_a_job_of_65536_segments_is_planned_and_drained_in_linear_time():
    """2^17 + 1 rows through plan_job, request_work and update_task_done: every operation is O(log rows) (per-stream ready sets, a
    hash index, dependants lists), so the whole walk takes a second — with the table scanned per operation it took minutes, under
    the one mutex every lane of every device shares."""
    import time

    lib = Lib()
    k = 1 << 16
    t0 = time.monotonic()
    ids = lib.db.plan_job("big", k, aux_stream="prove")
    assert len(ids) == 2 * k - 1 + 2
    order = []
    while True:
        w = lib.request_work("prove")
        if w is None:
            break
        order.append(w[1])
        assert lib.done(w[0], w[1], "null")
    wall = time.monotonic() - t0
    job = lib.db.job("big")
    assert job["state"] == "done" and job["done"] == len(ids) == len(order)
    assert order[:k] == [str(i) for i in range(k)] or set(order[:k]) <= set(ids)  # proves first: they were created first and ready
    assert order[-2:] == ["resolve", "finalize"]
    assert wall < 30, wall  # ~1.5 s here; the bound only separates linear from quadratic


def test_request_work_is_job_level_fifo():
    """9_request_work.sql:139-141: a job's tasks go before a younger job's, whenever they were created — here A's join is created
    (and becomes ready) after all of B's proves, and is still claimed before them."""
    lib, model = Lib(), Model()
    for t in (lib, model):
        assert t.create_task("prove", "A", "0", [], 0) and t.create_task("prove", "A", "1", [], 0)
        assert t.create_task("prove", "B", "0", [], 0) and t.create_task("prove", "B", "1", [], 0)
        assert t.create_task("prove", "A", "2", ["0", "1"], 0)
        got = []
        while (w := t.request_work("prove")) is not None:
            got.append(w[:2])
            assert t.done(w[0], w[1], "null")
        assert got == [("A", "0"), ("A", "1"), ("A", "2"), ("B", "0"), ("B", "1")]
    compare(lib, model, [])


def test_clear_completed_jobs_drops_done_jobs_only_and_the_rest_keeps_working():
    lib = Lib()
    for job in ("done-1", "run", "done-2", "bad"):
        lib.db.plan_job(job, 3, aux_stream="prove")
    for job in ("done-1", "done-2"):
        for t in ["0", "1", "2", "3", "4", "resolve", "finalize"]:
            assert lib.done(job, t, "null")
    assert lib.failed("bad", "1", "boom")
    assert lib.done("run", "0", "null") and lib.done("run", "1", "null")  # its first join (task 3) is ready now
    before = {t: lib.db.task("run", t) for t in ["0", "1", "2", "3", "4", "resolve", "finalize"]}
    assert len(lib.db.rows()) == 4 * 7
    assert lib.clear_completed_jobs() == 2 and lib.clear_completed_jobs() == 0
    assert sorted({r.job_id for r in lib.db.rows()}) == ["bad", "run"] and len(lib.db.rows()) == 2 * 7
    for job in ("done-1", "done-2"):
        with pytest.raises(HalError, match="no such job"):
            lib.db.job(job)
        with pytest.raises(HalError, match="no such task"):
            lib.db.task(job, "0")
    after = {t: lib.db.task("run", t) for t in before}
    assert {t: (r.state, r.waiting_on) for t, r in after.items()} == {t: (r.state, r.waiting_on) for t, r in before.items()}
    assert lib.db.job("bad")["state"] == "failed" and lib.db.job("run")["state"] == "running"
    assert lib.db.count("done") == 2 and lib.db.count("failed") == 1
    # the surviving job runs to its end: dependants lists, ready sets and the index were rebuilt consistently
    order = []
    while (w := lib.request_work("prove")) is not None:
        order.append(w[:2])
        assert lib.done(w[0], w[1], "null")
    assert [t for j, t in order if j == "run"] == ["2", "3", "4", "resolve", "finalize"]
    assert lib.db.job("run")["state"] == "done"
    lib.db.plan_job("done-1", 1, aux_stream="prove")  # a cleared job id may be used again
    assert lib.db.job("done-1")["tasks"] == 3


def test_requeue_tasks_retries_what_ran_past_its_timeout_and_fails_what_has_no_retries_left():
    """requeue_tasks (bento/crates/taskdb/src/lib.rs:328-358; its own test is :1254): the clock a task is measured against restarts
    at every claim and every update (GREATEST(started_at, updated_at))."""
    lib = Lib()
    db = lib.db
    db.create_task("J", "a", {"Prove": {"index": 0}}, max_retries=1, timeout_secs=10)
    db.create_task("J", "b", {"Prove": {"index": 1}}, max_retries=0, timeout_secs=10)
    db.create_task("J", "c", {"Prove": {"index": 2}}, max_retries=5)  # no timeout: never requeued
    db.create_task("J", "d", {"Prove": {"index": 3}}, max_retries=5, timeout_secs=10)  # stays ready: not running, not requeued
    assert [lib.request_work("prove")[1] for _ in range(3)] == ["a", "b", "c"]
    assert db.requeue_tasks() == 0
    db.advance_clock(9.0)
    assert db.requeue_tasks() == 0  # 9 s < 10 s
    db.advance_clock(2.0)
    assert db.requeue_tasks() == 2  # a and b ran 11 s
    a, b, c, d = (db.task("J", t) for t in "abcd")
    assert (a.state, a.retries) == ("ready", 1)
    assert (b.state, b.retries, b.error) == ("failed", 1, "retry max hit") and db.job("J")["error"] == "retry max hit"
    assert (c.state, c.retries) == ("running", 0) and (d.state, d.retries) == ("ready", 0)
    assert db.requeue_tasks() == 0
    assert lib.request_work("prove")[1] == "a"  # a is older than d: claimed again, its clock restarts
    db.advance_clock(9.0)
    assert db.requeue_tasks() == 0
    db.advance_clock(2.0)
    assert db.requeue_tasks() == 1 and db.task("J", "a").state == "failed" and db.task("J", "a").retries == 2
    # the limit bounds one sweep
    for i in range(5):
        db.create_task("K", f"t{i}", {"Prove": {"index": i}}, max_retries=9, timeout_secs=1)
    while lib.request_work("prove"):
        pass
    db.advance_clock(2.0)
    assert db.requeue_tasks(limit=3) == 3 and db.requeue_tasks(limit=3) == 2 and db.requeue_tasks(limit=3) == 0
    with pytest.raises(HalError, match="forward"):
        db.advance_clock(-1.0)
