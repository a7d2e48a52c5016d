"""rocprofv3 --kernel-trace --memory-copy-trace CSVs of a `bench.py --segment-bytes N --two-deep` run -> how the segments' uploads
overlap the kernels: size, duration and rate of the large host-to-device copies, and the share of their time during which at
least one kernel was running on the device.

    python tools/copy_overlap.py <memory_copy_trace.csv> <kernel_trace.csv> profiles/r04_h2d_overlap.json [bytes per segment]
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boundless_amd.build import csrc_hash  # noqa: E402


def col(row, *names):
    for k in row:
        if any(n in k.lower() for n in names):
            return k
    raise KeyError(names)


def main(copies_csv, kernels_csv, dst, bytes_each=0):
    bytes_each = int(bytes_each)
    copies = list(csv.DictReader(open(copies_csv)))
    kernels = list(csv.DictReader(open(kernels_csv)))
    cs, ce = col(copies[0], "start"), col(copies[0], "end")
    try:
        cb = col(copies[0], "bytes", "size")
    except KeyError:
        cb = None
    ks, ke = col(kernels[0], "start"), col(kernels[0], "end")
    kiv = sorted((int(k[ks]), int(k[ke])) for k in kernels)
    big = []
    for c in copies:
        s, e = int(c[cs]), int(c[ce])
        n = int(c[cb]) if cb and c[cb] else 0
        if n >= 1_000_000 or (not cb and e - s > 500_000):
            big.append((s, e, n))
    covered_total, dur_total = 0, 0
    for s, e, _ in big:
        # union of kernel intervals clipped to [s, e]
        cov, cur = 0, s
        for a, b in kiv:
            if b <= cur:
                continue
            if a >= e:
                break
            a = max(a, cur)
            if b > a:
                cov += min(b, e) - a
                cur = min(b, e)
        covered_total += cov
        dur_total += e - s
    out = {"csrc_sha": csrc_hash(),
           "note": "rocprofv3 --kernel-trace --memory-copy-trace of bench.py --segment-bytes 80000000 --two-deep: host-to-device copies of at least "
                   "1 MB (the segments' uploads on the provers' copy streams) and the kernels running meanwhile",
           "uploads": len(big), "bytes_each": ((sorted(n for _, _, n in big)[len(big) // 2] or bytes_each) if big else 0),
           "bytes_from": "the trace" if any(n for _, _, n in big) else "the command line (this rocprofv3's CSV has no size column)",
           "avg_ms": round(dur_total / max(len(big), 1) / 1e6, 4),
           "GBps": round(sum((n or bytes_each) for _, _, n in big) / max(dur_total, 1), 2),
           "share_of_upload_time_with_a_kernel_running": round(covered_total / max(dur_total, 1), 4),
           "kernels_in_trace": len(kernels)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:5])
