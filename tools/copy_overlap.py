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
from boundless_amd.build import csrc_hash, device_code_hash  # noqa: E402


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
    # the segments go up in 8 MiB pieces (bx_prover_submit_segment): host-to-device copies that take longer than 50 us are those pieces
    # (the prover's own small uploads — tap points, query positions — take microseconds)
    try:
        cd = col(copies[0], "direction")
    except KeyError:
        cd = None
    big = []
    for c in copies:
        if cd and "HOST_TO_DEVICE" not in c[cd].upper():
            continue
        s, e = int(c[cs]), int(c[ce])
        n = int(c[cb]) if cb and c[cb] else 0
        if n >= 1_000_000 or (not cb and e - s > 50_000):
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
    out = {"device_code_sha": device_code_hash(), "csrc_sha": csrc_hash(),
           "note": "rocprofv3 --kernel-trace --memory-copy-trace of bench.py --segment-bytes 80000000: the 8 MiB pieces of the segments' uploads "
                   "(host-to-device copies longer than 50 us, on the provers' copy streams) and the kernels running meanwhile",
           "pieces": len(big), "piece_bytes": 8 << 20, "pieces_per_segment": (-(-bytes_each // (8 << 20)) if bytes_each else None),
           "segments_uploaded": (round(len(big) / -(-bytes_each // (8 << 20)), 1) if bytes_each else None),
           "segment_bytes": bytes_each, "bytes_from": "the command line (this rocprofv3's CSV has no size column)",
           "dma_ms_per_segment": (round(dur_total / 1e6 / (len(big) / -(-bytes_each // (8 << 20))), 4) if bytes_each and big else None),
           "GBps": (round(bytes_each * (len(big) / -(-bytes_each // (8 << 20))) / max(dur_total, 1), 2) if bytes_each else None),
           "share_of_upload_time_with_a_kernel_running": round(covered_total / max(dur_total, 1), 4),
           "kernels_in_trace": len(kernels)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:5])
