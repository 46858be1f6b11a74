#!/usr/bin/env python3
"""Per-kernel statistics and a launch timeline out of a rocprofv3 (ROCm 7.2) rocpd results database.

    rocprofv3 --kernel-trace --stats -d out -o name -- python3 bench.py ...      ->  out/name_results.db
    python tools/rocpd_kernel_stats.py out/name_results.db [--timeline profiles/x_timeline.csv] > profiles/x_kernel_stats.csv

stdout: CSV of the `top_kernels` view (name, calls, total / average duration in us, share).  --timeline additionally writes one
row per dispatch of the LAST complete forward in the trace (start / end in us relative to its first kernel, stream, grid in
workgroups), the evidence behind the slot-occupancy figures in DESIGN.md."""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("rsplit::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--timeline", default="")
    ap.add_argument("--first-kernel", default="encoder_stats_kernel", help="kernel that opens a forward")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        w.writerow([short(name), calls, round(total, 1), round(avg, 2), round(pct, 2)])       # the view is in microseconds
    if a.timeline:
        rows = list(db.execute("select name, start, end, stream_id, queue_id, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, "
                               "workgroup_z from kernels order by start"))
        starts = [i for i, r in enumerate(rows) if a.first_kernel in r[0]]
        if not starts:
            raise SystemExit("no forward found")
        i0 = starts[-1]
        t0 = rows[i0][1]
        ends = [i for i, r in enumerate(rows) if "decoder_kernel" in r[0] and i > i0]
        i1 = ends[-1] + 1 if ends else len(rows)          # what follows the decoder is the harness (isfinite checks), not the path
        with open(a.timeline, "w", newline="") as f:
            tw = csv.writer(f)
            tw.writerow(["kernel", "start_us", "end_us", "duration_us", "stream", "queue", "workgroups", "threads_per_workgroup"])
            for name, st, en, sid, qid, gx, gy, gz, wx, wy, wz in rows[i0:i1]:
                wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
                tw.writerow([short(name), round((st - t0) / 1e3, 1), round((en - t0) / 1e3, 1), round((en - st) / 1e3, 1), sid, qid,
                             wgs, wx * wy * wz])


if __name__ == "__main__":
    main()
