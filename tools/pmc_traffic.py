"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE/WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md §HBM), so fetch bytes are doubled; WRITE_SIZE is taken as is.  Infinity-Cache hits are included in
FETCH_SIZE (it counts the L2's fabric-side requests), so this is an upper bound on DRAM reads.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.build import csrc_hash, device_code_hash  # noqa: E402


def load(path, name):
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def main():
    fetch, fc = load(sys.argv[1], "FETCH_SIZE")
    write, wc = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        n = fc.get(k) or wc.get(k) or 1
        fb = fetch.get(k, 0.0) * 1024 * 2
        wb = write.get(k, 0.0) * 1024
        out[k] = {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / max(wc.get(k, n), 1),
                  "hbm_bytes_per_launch": fb / n + wb / max(wc.get(k, n), 1)}
    json.dump({"device_code_sha": device_code_hash(), "csrc_sha": csrc_hash(), "note": "FETCH_SIZE doubled (gfx950 calibration), WRITE_SIZE as reported; separate --pmc passes", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if "ntt" in k or "hash_rows" in k:
            print(f"{k[:64]:64s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x{v['launches']}")


if __name__ == "__main__":
    main()
