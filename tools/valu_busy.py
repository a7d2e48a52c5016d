"""rocprofv3 --pmc VALUBusy VALUUtilization counter CSV -> per-kernel averages (weighted by dispatch count).

    rocprofv3 --pmc VALUBusy VALUUtilization --output-format csv -d out -o pmc -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode
    python tools/valu_busy.py out/.../pmc_counter_collection.csv profiles/r02_kernel_valu_busy.json

VALUBusy = 100 * sum(SQ_ACTIVE_INST_VALU) / CU_NUM / GRBM_GUI_ACTIVE (share of the kernel's GPU-active cycles in which the vector
ALUs of a CU are processing instructions, summed over the CU's SIMDs by the counter's definition); VALUUtilization = share of
active lanes in the issued VALU instructions.  Both are rocprofv3's derived metrics (rocprofv3 --list-avail).
"""
import collections
import csv
import json
import sys


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        n = max(len(x) for x in v.values())
        out[k[:90]] = {"dispatches": n, **{c: round(sum(x) / len(x), 2) for c, x in v.items()}}
    json.dump({"note": "rocprofv3 --pmc VALUBusy VALUUtilization on bench.py --inflight 1 (per-dispatch averages per kernel)", "kernels": out},
              open(dst, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("VALUBusy", 0))[:14]:
        print(f"{k[:70]:70s} {d}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
