"""Summarise a rocprofv3 marker trace of BX_TRACE=2 into per-stage milliseconds per proof.

usage: python tools/stage_trace.py <marker_api_trace.csv> <out.json>
The csv has one row per roctx range (columns include Function = the range's message, Start_Timestamp, End_Timestamp in ns).
Only the "bx:*" ranges are prover stages (device time at level 2); the nested HAL entry-point ranges are host-side enqueue
time (the calls are asynchronous) and are summed separately.
"""
import csv
import json
import sys
from collections import defaultdict


def main(path, out):
    stage, op = defaultdict(list), defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Function") or row.get("Name") or ""
            try:
                ms = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            except (KeyError, ValueError):
                continue
            (stage if name.startswith("bx:") else op)[name].append(ms)
    proofs = len(stage.get("bx:prove_segment", [])) or 1
    # the first proofs are warm-up (kernel load, pool growth): report the median proof
    def med(v):
        v = sorted(v)
        return v[len(v) // 2]

    res = {
        "note": "BX_TRACE=2 roctx ranges, rocprofv3 --marker-trace; median over the proofs of the run; one segment in flight",
        "proofs": proofs,
        "stage_ms_median": {k: round(med(v), 3) for k, v in sorted(stage.items())},
        "hal_entry_host_enqueue_ms_per_proof": {k: round(sum(v) / proofs, 3) for k, v in sorted(op.items(), key=lambda kv: -sum(kv[1]))},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["stage_ms_median"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
