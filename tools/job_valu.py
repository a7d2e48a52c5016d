"""rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES counter CSV of a bench.py run -> wave-level VALU instructions per segment, per kernel.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d out -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-agent-mode
    python tools/job_valu.py out/.../pmc_counter_collection.csv <segments proved in that run> profiles/r02_job_valu_insts.json
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.build import csrc_hash, device_code_hash  # noqa: E402


def main(src, segments, dst):
    tot = collections.defaultdict(float)
    waves = collections.defaultdict(float)
    proofs = set()
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            if "eval_check_kernel" in r["Kernel_Name"]:
                proofs.add(r.get("Dispatch_Id"))
        elif r["Counter_Name"] == "SQ_WAVES":
            waves[r["Kernel_Name"]] += float(r["Counter_Value"])
    # "auto": every proof launches eval_check exactly once, so the run counts its own segments (the hand-passed number went stale
    # when bench.py's lone-proof section grew from 4 to 8 proofs)
    segments = len(proofs) if segments == "auto" else int(segments)
    total = sum(tot.values())
    out = {"device_code_sha": device_code_hash(), "csrc_sha": csrc_hash(), "note": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES on bench.py ({segments} segments proved in the profiled process, warm-up and "
                   "isolated probe included); wave-level VALU instructions",
           "segments": segments, "valu_wave_insts_total": total, "per_segment": total / segments,
           "per_kernel_per_segment": {k[:60]: v / segments for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
           "valu_insts_per_wave": {k[:60]: tot[k] / waves[k] for k in tot if waves.get(k)}}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in list(out["per_kernel_per_segment"].items())[:12]:
        print(f"{k:60s} {v / 1e9:8.3f} G/segment")
    print("total per segment", out["per_segment"] / 1e9, "G")


if __name__ == "__main__":
    main(*sys.argv[1:4])
