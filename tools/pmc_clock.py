"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE pass: busy GPU cycles / dispatch duration.

    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d out -o pmc -- python bench.py --inflight 1 ...
    python tools/pmc_clock.py out/.../pmc_counter_collection.csv <out.json> [kernel-substring,...]

The MI355X settles its clock under the package power limit, differently for each kernel's mix of VALU / LDS / HBM activity; this is
the figure that turns "ns per wave-instruction" into cycles.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    subs = sys.argv[3].split(",") if len(sys.argv) > 3 else []
    rows = list(csv.DictReader(open(src)))
    cols = rows[0].keys() if rows else []
    start = next((c for c in cols if c.lower().startswith("start")), None)
    end = next((c for c in cols if c.lower().startswith("end")), None)
    cyc = collections.defaultdict(float)
    dur = collections.defaultdict(float)
    n = collections.Counter()
    for r in rows:
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        k = r["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        cyc[k] += float(r["Counter_Value"])
        if start and end:
            dur[k] += float(r[end]) - float(r[start])
        n[k] += 1
    from boundless_amd.build import device_code_hash

    out = {"device_code_sha": device_code_hash(), "columns": list(cols), "kernels": {}}
    for k in sorted(cyc, key=lambda q: -cyc[q]):
        d = {"dispatches": n[k], "gui_active_cycles": cyc[k]}
        if dur.get(k):
            d["duration_ns"] = dur[k]
            d["effective_clock_GHz"] = round(cyc[k] / dur[k] / 8.0, 3)  # the counter is summed over the chip's 8 XCDs
        out["kernels"][k[:100]] = d
        print(k[:80], d)
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
