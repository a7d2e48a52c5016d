// Join-tree planner: behavioural restatement of taskdb::planner::Planner
// (bento/crates/taskdb/src/planner/mod.rs:20-252; task constructors planner/task.rs:28-81).
//
// Segments stream out of the executor one at a time; each becomes a leaf.  The segment forest is a binary counter over
// "peaks" (roots nobody depends on yet): a new leaf is joined with the most recent peak for as long as both have the same
// height.  `finish` folds what is left — joins from the smallest peak upwards, unions from the oldest pair downwards —
// and appends the Finalize task that depends on the single join root and the single union root.  Task numbers, heights
// and dependency lists reproduce the reference's unit tests (mod.rs:254-453), restated in tests/test_planner_agent_cpu.py.
#include <exception>
#include <deque>
#include <string>
#include <vector>

#include "../../include/bx_agent.h"

namespace {
struct Node {
    uint64_t number;
    uint32_t height;
    uint32_t command;
    std::vector<uint64_t> deps, kdeps;
};
}  // namespace

struct bx_planner {
    std::vector<Node> tasks;
    std::deque<uint64_t> peaks, keccak_peaks;  // tallest/oldest first
    size_t cursor = 0;
    bool finished = false;
    uint64_t last = 0;
    std::string err;

    uint64_t push(uint32_t command, uint32_t height, std::vector<uint64_t> deps, std::vector<uint64_t> kdeps) {
        tasks.push_back(Node{(uint64_t)tasks.size(), height, command, std::move(deps), std::move(kdeps)});
        return tasks.back().number;
    }
    uint32_t height(uint64_t n) const { return tasks[n].height; }
    // binary-counter carry: absorb equal-height peaks, most recent first
    void merge(std::deque<uint64_t>& forest, uint64_t leaf, bool join) {
        uint64_t top = leaf;
        while (!forest.empty() && height(forest.back()) == height(top)) {
            uint64_t left = forest.back();
            forest.pop_back();
            uint32_t h = 1 + std::max(height(left), height(top));
            top = join ? push(BX_PLAN_JOIN, h, {left, top}, {}) : push(BX_PLAN_UNION, h, {}, {left, top});
        }
        forest.push_back(top);
    }
    const char* fail(const char* m) {
        err = m;
        return err.c_str();
    }
};

static void export_task(const Node& n, bx_plan_task* out) {
    out->task_number = n.number;
    out->task_height = n.height;
    out->command = n.command;
    out->n_depends_on = (uint32_t)n.deps.size();
    out->n_keccak_depends_on = (uint32_t)n.kdeps.size();
    for (size_t i = 0; i < 2; ++i) {
        out->depends_on[i] = i < n.deps.size() ? n.deps[i] : 0;
        out->keccak_depends_on[i] = i < n.kdeps.size() ? n.kdeps[i] : 0;
    }
}

extern "C" {

const char* bx_planner_create(bx_planner** out) {
    if (!out) return "bx_planner_create: out is NULL";
    try {
        *out = new bx_planner();
    } catch (...) {
        return "bx_planner_create: out of memory";
    }
    return nullptr;
}

void bx_planner_destroy(bx_planner* p) { delete p; }

const char* bx_planner_enqueue_segment(bx_planner* p, uint64_t* task_number) {
    try {
        if (!p) return "bx_planner: NULL planner";
        if (p->finished) return p->fail("Cannot add segment to finished plan");  // PlannerErr::PlanFinalized
        uint64_t n = p->push(BX_PLAN_SEGMENT, 0, {}, {});
        p->merge(p->peaks, n, true);
        if (task_number) *task_number = n;
        return nullptr;
    } catch (const std::exception&) {
        return "bx_planner_enqueue_segment: out of memory";
    }
}

const char* bx_planner_enqueue_keccak(bx_planner* p, uint64_t* task_number) {
    try {
        if (!p) return "bx_planner: NULL planner";
        if (p->finished) return p->fail("Cannot add segment to finished plan");
        uint64_t n = p->push(BX_PLAN_KECCAK, 0, {}, {});
        p->merge(p->keccak_peaks, n, false);
        if (task_number) *task_number = n;
        return nullptr;
    } catch (const std::exception&) {
        return "bx_planner_enqueue_keccak: out of memory";
    }
}

const char* bx_planner_finish(bx_planner* p, uint64_t* task_number) {
    try {
        if (!p) return "bx_planner: NULL planner";
        if (p->peaks.empty()) return p->fail("Planning not yet started");  // PlannerErr::PlanNotStartedString
        if (!p->finished) {
            std::vector<uint64_t> kdeps;
            if (!p->keccak_peaks.empty()) {
                while (p->keccak_peaks.size() >= 2) {  // unions: oldest pair first
                    uint64_t p0 = p->keccak_peaks.front();
                    p->keccak_peaks.pop_front();
                    uint64_t p1 = p->keccak_peaks.front();
                    p->keccak_peaks.pop_front();
                    uint32_t h = 1 + std::max(p->height(p0), p->height(p1));
                    p->keccak_peaks.push_front(p->push(BX_PLAN_UNION, h, {}, {p1, p0}));
                }
                kdeps.push_back(p->keccak_peaks.front());
            }
            while (p->peaks.size() >= 2) {  // joins: smallest pair first
                uint64_t p0 = p->peaks.back();
                p->peaks.pop_back();
                uint64_t p1 = p->peaks.back();
                p->peaks.pop_back();
                uint32_t h = 1 + std::max(p->height(p0), p->height(p1));
                p->peaks.push_back(p->push(BX_PLAN_JOIN, h, {p1, p0}, {}));
            }
            uint32_t h = 1 + p->height(p->peaks.front());
            if (!kdeps.empty()) h = std::max(h, 1 + p->height(kdeps[0]));
            p->last = p->push(BX_PLAN_FINALIZE, h, {p->peaks.front()}, kdeps);
            p->finished = true;
        }
        if (task_number) *task_number = p->last;
        return nullptr;
    } catch (const std::exception&) {
        return "bx_planner_finish: out of memory";
    }
}

const char* bx_planner_next_task(bx_planner* p, bx_plan_task* out, int* has) {
    if (!p || !out || !has) return "bx_planner_next_task: NULL argument";
    if (p->cursor < p->tasks.size()) {
        export_task(p->tasks[p->cursor++], out);
        *has = 1;
    } else {
        *has = 0;
    }
    return nullptr;
}

size_t bx_planner_task_count(const bx_planner* p) { return p ? p->tasks.size() : 0; }

const char* bx_planner_get_task(bx_planner* p, uint64_t task_number, bx_plan_task* out) {
    if (!p || !out) return "bx_planner_get_task: NULL argument";
    if (task_number >= p->tasks.size()) return p->fail("Invalid task number");
    export_task(p->tasks[task_number], out);
    return nullptr;
}

}  // extern "C"
