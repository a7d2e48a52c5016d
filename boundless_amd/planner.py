"""Join-tree planner — ctypes binding of the native planner (boundless_amd/csrc/planner.cpp, include/bx_agent.h).

Mirrors `taskdb::planner::Planner` (bento/crates/taskdb/src/planner/mod.rs:20-252): method names, return values and error
text are the reference's, so tests/test_planner_agent_cpu.py restates its unit tests (mod.rs:254-453) one for one.
All logic is in C++; this file only marshals.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

from .hal import load_library

SEGMENT, KECCAK, JOIN, UNION, FINALIZE = "Segment", "Keccak", "Join", "Union", "Finalize"
_COMMANDS = (SEGMENT, KECCAK, JOIN, UNION, FINALIZE)


class PlannerError(Exception):
    pass


class PlanNotStarted(PlannerError):  # PlannerErr::PlanNotStartedString
    pass


class PlanFinalized(PlannerError):  # PlannerErr::PlanFinalized
    pass


class _PlanTask(C.Structure):
    _fields_ = [("task_number", C.c_uint64), ("task_height", C.c_uint32), ("command", C.c_uint32),
                ("n_depends_on", C.c_uint32), ("n_keccak_depends_on", C.c_uint32),
                ("depends_on", C.c_uint64 * 2), ("keccak_depends_on", C.c_uint64 * 2)]


@dataclass
class Task:
    task_number: int
    task_height: int
    command: str
    depends_on: List[int] = field(default_factory=list)
    keccak_depends_on: List[int] = field(default_factory=list)


def _declare(lib):
    if getattr(lib, "_bx_planner_declared", False):
        return
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    for name, args in {
        "bx_planner_create": [C.POINTER(C.c_void_p)],
        "bx_planner_enqueue_segment": [vp, u64p],
        "bx_planner_enqueue_keccak": [vp, u64p],
        "bx_planner_finish": [vp, u64p],
        "bx_planner_next_task": [vp, C.POINTER(_PlanTask), C.POINTER(C.c_int)],
        "bx_planner_get_task": [vp, C.c_uint64, C.POINTER(_PlanTask)],
    }.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p
    lib.bx_planner_destroy.argtypes = [vp]
    lib.bx_planner_destroy.restype = None
    lib.bx_planner_task_count.argtypes = [vp]
    lib.bx_planner_task_count.restype = C.c_size_t
    lib._bx_planner_declared = True


def _raise(msg):
    text = msg.decode()
    if text == "Planning not yet started":
        raise PlanNotStarted(text)
    if text == "Cannot add segment to finished plan":
        raise PlanFinalized(text)
    raise PlannerError(text)


def _export(t):
    return Task(t.task_number, t.task_height, _COMMANDS[t.command], list(t.depends_on[: t.n_depends_on]),
                list(t.keccak_depends_on[: t.n_keccak_depends_on]))


class Planner:
    def __init__(self):
        self._lib = load_library()
        _declare(self._lib)
        h = C.c_void_p()
        msg = self._lib.bx_planner_create(C.byref(h))
        if msg:
            _raise(msg)
        self._h = h

    def _call(self, fn):
        n = C.c_uint64()
        msg = fn(self._h, C.byref(n))
        if msg:
            _raise(msg)
        return n.value

    def enqueue_segment(self):
        return self._call(self._lib.bx_planner_enqueue_segment)

    def enqueue_keccak(self):
        return self._call(self._lib.bx_planner_enqueue_keccak)

    def finish(self):
        return self._call(self._lib.bx_planner_finish)

    def task_count(self):
        return self._lib.bx_planner_task_count(self._h)

    def get_task(self, task_number):
        t = _PlanTask()
        if self._lib.bx_planner_get_task(self._h, task_number, C.byref(t)):
            raise IndexError(f"Invalid task number {task_number}")  # the reference panics with this text
        return _export(t)

    def next_task(self):
        t, has = _PlanTask(), C.c_int()
        msg = self._lib.bx_planner_next_task(self._h, C.byref(t), C.byref(has))
        if msg:
            _raise(msg)
        return _export(t) if has.value else None

    @property
    def tasks(self):
        return [self.get_task(i) for i in range(self.task_count())]

    def __del__(self):
        try:
            if self._h:
                self._lib.bx_planner_destroy(self._h)
                self._h = None
        except Exception:
            pass
