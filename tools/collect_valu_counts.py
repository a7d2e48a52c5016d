"""rocprofv3 --pmc counter CSV -> profiles/r01_kernel_valu_counts.json (per-kernel, per-dispatch averages).

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv \
              -d out -o pmc -- python tools/opbench.py --po2 20 --cols 64 --reps 1
    python tools/collect_valu_counts.py out/pmc_counter_collection.csv profiles/r01_kernel_valu_counts.json
"""
import collections
import csv
import json
import sys

NOTE = ("rocprofv3 --pmc on tools/opbench.py --po2 20 --cols 64 (per-dispatch averages); hash kernels: 4 permutations per "
        "wave-lane in hash_rows at 64 columns, 1 in hash_fold; NTT kernels: 16 elements per lane (multi-column pass A: "
        "16 elements x 8 columns per lane)")


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur = collections.Counter(), collections.defaultdict(float)
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, v in agg.items():
        n = cnt[k] or 1
        d = {"dispatches": n, "avg_duration_us_profiled": dur[k] / n / 1e3}
        d.update({c: x / n for c, x in v.items()})
        if d.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        out[k] = d
    json.dump({"note": NOTE, "kernels": out}, open(dst, "w"), indent=1)
    for k, d in out.items():
        if "hash" in k or "ntt" in k:
            print(k[:70], round(d.get("valu_insts_per_wave", 0)), int(d["SQ_WAVES"]), round(d["avg_duration_us_profiled"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
