"""Per-kernel SQ stall-side counters from rocprofv3 --pmc passes (one or more counter_collection.csv files) -> JSON.

    python tools/pmc_stalls.py <out.json> <kernel-substring,...> <counter_collection.csv> [...]

For every kernel whose name contains one of the substrings: the sum of each counter over all dispatches and, where the pass has them,
the shares of SQ_WAVE_CYCLES spent issuing (ACTIVE_INST_ANY), parked at s_waitcnt / s_barrier (WAIT_ANY) and stalled at issue
(WAIT_INST_ANY) — MI355X_MICROARCH.md §rocprofv3 PMC slots: the three are disjoint and sum to ~WAVE_CYCLES (units: quad-cycles).
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_path, subs, files = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in subs):
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r.get("Dispatch_Id")))
    res = {}
    for k, c in tot.items():
        d = {"dispatches_seen": len(disp[k]), "counters": dict(c)}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            sh = {}
            for name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                         "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM"):
                if name in c:
                    sh[name + "/WAVE_CYCLES"] = round(c[name] / wc, 4)
            d["shares_of_wave_cycles"] = sh
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
            d["valu_active_per_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"], 4)
        if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
            d["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        res[k] = d
    try:
        from boundless_amd.build import device_code_hash

        stamp = device_code_hash()
    except Exception:  # noqa: BLE001
        stamp = None
    json.dump({"device_code_sha": stamp, "note": "sums over all dispatches of the run; SQ cycle counters are in quad-cycles", "kernels": res},
              open(out_path, "w"), indent=1)
    for k, d in res.items():
        print(k[:70], json.dumps(d.get("shares_of_wave_cycles", {})))


if __name__ == "__main__":
    main()
