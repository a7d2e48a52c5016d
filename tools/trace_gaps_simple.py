"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: count, median, mean (us).  usage: trace_gaps_simple.py <csv>"""
import csv, sys, statistics
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    try:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]))
    except (KeyError, ValueError):
        pass
rows.sort()
g = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
g = [x for x in g if x < 500]
print({"kernels": len(rows), "gaps": len(g), "median_us": round(statistics.median(g), 2), "mean_us": round(sum(g) / len(g), 2), "p90_us": round(sorted(g)[int(0.9 * len(g))], 2)})
