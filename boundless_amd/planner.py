"""Join-tree planner: which segment proofs get joined with which, as segments stream out of the executor.

Behavioural restatement of `taskdb::planner::Planner` (bento/crates/taskdb/src/planner/mod.rs:20-252, tasks
planner/task.rs:8-81), the component that turns "segment i is ready" events into Prove/Join/Finalize tasks for the queue
the prove agents pull from.  The shape is a binary counter over "peaks": a new leaf merges with
the smallest peak while their heights are equal, `finish` folds the remaining peaks from the smallest upwards and appends
the Finalize task.  Keccak leaves form a second, independent forest of Union nodes that Finalize also depends on.
Task numbers, heights and dependency lists match the reference's unit tests (mod.rs:254-453) one for one.
"""
from dataclasses import dataclass, field
from typing import List, Optional

SEGMENT, KECCAK, JOIN, UNION, FINALIZE = "Segment", "Keccak", "Join", "Union", "Finalize"


class PlannerError(Exception):
    pass


class PlanNotStarted(PlannerError):
    def __init__(self):
        super().__init__("Planning not yet started")  # PlannerErr::PlanNotStartedString


class PlanFinalized(PlannerError):
    def __init__(self):
        super().__init__("Cannot add segment to finished plan")  # PlannerErr::PlanFinalized


@dataclass
class Task:
    task_number: int
    task_height: int
    command: str
    depends_on: List[int] = field(default_factory=list)
    keccak_depends_on: List[int] = field(default_factory=list)


class Planner:
    def __init__(self):
        self.tasks: List[Task] = []
        self._peaks: List[int] = []         # Segment/Join roots nobody depends on yet, tallest first
        self._keccak_peaks: List[int] = []  # same for Keccak/Union
        self._cursor = 0
        self._last: Optional[int] = None

    # -- building -----------------------------------------------------------------------------------------------
    def _push(self, command, height=0, deps=(), kdeps=()):
        t = Task(len(self.tasks), height, command, list(deps), list(kdeps))
        self.tasks.append(t)
        return t.task_number

    def _merge(self, forest, leaf, node_command):
        """binary-counter carry: absorb equal-height peaks, smallest first"""
        top = leaf
        while forest and self.tasks[forest[-1]].task_height == self.tasks[top].task_height:
            left = forest.pop()
            height = 1 + max(self.tasks[left].task_height, self.tasks[top].task_height)
            if node_command == JOIN:
                top = self._push(JOIN, height, deps=(left, top))
            else:
                top = self._push(UNION, height, kdeps=(left, top))
        forest.append(top)

    def enqueue_segment(self):
        if self._last is not None:
            raise PlanFinalized()
        n = self._push(SEGMENT)
        self._merge(self._peaks, n, JOIN)
        return n

    def enqueue_keccak(self):
        if self._last is not None:
            raise PlanFinalized()
        n = self._push(KECCAK)
        self._merge(self._keccak_peaks, n, UNION)
        return n

    def finish(self):
        if not self._peaks:
            raise PlanNotStarted()
        # unions: fold from the tallest pair downwards (the reference pops from the front of its deque)
        kdeps = []
        if self._keccak_peaks:
            while len(self._keccak_peaks) >= 2:
                p0, p1 = self._keccak_peaks.pop(0), self._keccak_peaks.pop(0)
                h = 1 + max(self.tasks[p0].task_height, self.tasks[p1].task_height)
                self._keccak_peaks.insert(0, self._push(UNION, h, kdeps=(p1, p0)))
            kdeps = [self._keccak_peaks[0]]
        if self._last is None:
            while len(self._peaks) >= 2:  # joins: fold from the smallest pair upwards
                p0, p1 = self._peaks.pop(), self._peaks.pop()
                h = 1 + max(self.tasks[p0].task_height, self.tasks[p1].task_height)
                self._peaks.append(self._push(JOIN, h, deps=(p1, p0)))
            height = 1 + self.tasks[self._peaks[0]].task_height
            if kdeps:
                height = max(height, 1 + self.tasks[max(kdeps)].task_height)
            self._last = self._push(FINALIZE, height, deps=(self._peaks[0],), kdeps=kdeps)
        return self._last

    # -- consuming ----------------------------------------------------------------------------------------------
    def task_count(self):
        return len(self.tasks)

    def get_task(self, task_number):
        if not 0 <= task_number < len(self.tasks):
            raise IndexError(f"Invalid task number {task_number}")
        return self.tasks[task_number]

    def next_task(self):
        if self._cursor < len(self.tasks):
            t = self.tasks[self._cursor]
            self._cursor += 1
            return t
        return None
