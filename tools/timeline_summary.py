#!/usr/bin/env python3
"""One line per launch-timeline CSV (tools/rocpd_kernel_stats.py --timeline): forward span, summed launch durations by kernel
class, and how much of the span has a recurrence launch (one, two or more) or no kernel at all in flight.

    python tools/timeline_summary.py profiles/r02_bench_b64_mode3_timeline.csv ... > profiles/r02_timeline_summary.txt"""
import csv
import sys


def klass(name):
    if name.startswith("bilstm"):
        return "recurrence"
    if "proj_f16" in name or "EpProjStats" in name:
        return "projection"
    if "HeadConv" in name or "Gate" in name or "MaskDecode" in name or name.startswith("decoder"):
        return "mask head + decoder"
    return "other"


def covered(iv, lo, hi, at_least):
    """length of [lo, hi] where at least `at_least` of the intervals overlap"""
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    n, last, tot = 0, lo, 0.0
    for t, d in ev:
        if n >= at_least:
            tot += t - last
        n += d
        last = t
    return tot


for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    iv = [(float(r["start_us"]), float(r["end_us"]), klass(r["kernel"])) for r in rows]
    lo, hi = min(s for s, _, _ in iv), max(e for _, e, _ in iv)
    span = hi - lo
    by = {}
    for s, e, k in iv:
        by[k] = by.get(k, 0.0) + (e - s)
    rec = [(s, e) for s, e, k in iv if k == "recurrence"]
    allk = [(s, e) for s, e, _ in iv]
    print(f"{path.split('/')[-1]}: forward {span / 1e3:.1f} ms; sum of launch durations by class (ms): "
          + ", ".join(f"{k} {by.get(k, 0.0) / 1e3:.1f}" for k in ("recurrence", "projection", "mask head + decoder", "other"))
          + f"; a recurrence launch in flight {100 * covered(rec, lo, hi, 1) / span:.0f} % of the time (two or more: "
          f"{100 * covered(rec, lo, hi, 2) / span:.0f} %); no kernel in flight {100 * (1 - covered(allk, lo, hi, 1) / span):.1f} %")
