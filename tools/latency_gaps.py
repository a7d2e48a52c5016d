"""Where a single proof's latency goes beyond the VALU floor: from a rocprofv3 kernel trace of `bench.py --inflight 1`.

usage: python tools/latency_gaps.py <kernel_trace.csv> <out.json>
Splits the busiest stream's timeline of the last proof-sized window into: time inside kernels that fill the chip
(>= 1024 workgroups... by grid size), time inside under-filled kernels (fewer workgroups than 4 per CU), and idle gaps
between consecutive kernels (host round trips, launch latency)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.build import csrc_hash, device_code_hash  # noqa: E402


def main(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            try:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            except (KeyError, ValueError):
                continue
            wg = 1
            for d in ("X", "Y", "Z"):
                g, w = int(r.get(f"Grid_Size_{d}", 1) or 1), int(r.get(f"Workgroup_Size_{d}", 1) or 1)
                wg *= max(1, g // max(1, w))
            rows.append((s, e, r["Kernel_Name"], wg))
    rows.sort()
    # proofs are separated by the longest idle gaps; take the kernels between occurrences of witness_code_kernel
    starts = [i for i, r in enumerate(rows) if "witness_code_kernel" in r[2]]
    res = {"device_code_sha": device_code_hash(), "csrc_sha": csrc_hash(), "note": "one segment in flight; per proof (median over the proofs of the run)", "proofs": []}
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        busy_full = busy_small = idle = 0
        small = defaultdict(float)
        edges = [5, 10, 20, 50, 100, 200]  # us
        hist = [[0, 0.0] for _ in range(len(edges) + 1)]
        gaps = []
        for i, (s, e, name, wg) in enumerate(seg):
            d = (e - s) / 1e6
            if wg >= 1024:
                busy_full += d
            else:
                busy_small += d
                small[name.split("(")[0][:60]] += d
            if i + 1 < len(seg):
                g = max(0, seg[i + 1][0] - e) / 1e3  # us
                idle += g / 1e3
                b = sum(g >= x for x in edges)
                hist[b][0] += 1
                hist[b][1] += g
                gaps.append((g, name.split("(")[0][:40], seg[i + 1][2].split("(")[0][:40]))
        res["proofs"].append({"kernels": len(seg), "ms_in_chip_filling_kernels": round(busy_full, 3), "ms_in_underfilled_kernels": round(busy_small, 3),
                              "ms_idle_between_kernels": round(idle, 3), "ms_first_to_last": round((seg[-1][1] - seg[0][0]) / 1e6, 3),
                              "underfilled_by_kernel_ms": {k: round(v, 3) for k, v in sorted(small.items(), key=lambda kv: -kv[1])[:12]},
                              "idle_gap_histogram_us": {("<%d" % edges[0] if k == 0 else (">=%d" % edges[-1] if k == len(edges) else "%d-%d" % (edges[k - 1], edges[k]))):
                                                        {"gaps": h[0], "ms": round(h[1] / 1e3, 3)} for k, h in enumerate(hist)},
                              "largest_gaps_us": [{"us": round(g, 1), "after": a_, "before": b_} for g, a_, b_ in sorted(gaps, reverse=True)[:14]]})
    if res["proofs"]:
        res["proofs"] = [sorted(res["proofs"], key=lambda p: p["ms_first_to_last"])[len(res["proofs"]) // 2]]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
