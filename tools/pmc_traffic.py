#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes:
separate passes, FETCH_SIZE/WRITE_SIZE in KB, gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes (x2).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python3 <cmd>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python3 <cmd>
    python tools/pmc_traffic.py out/fetch out/write "<command description>" > profiles/<name>.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rsplit::", "").replace("void ", "")
            name = re.sub(r"\(.*", "", name)                     # drop the argument list
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, what = sys.argv[1], sys.argv[2], sys.argv[3]
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    out = {"command": what,
           "correction": "hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B: "
                         "MI355X_MICROARCH.md HBM section); hbm_write_bytes = WRITE_SIZE * 1024",
           "per_launch": {}}
    for name in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        n = fc[name]
        out["per_launch"][name] = {
            "launches": n,
            "FETCH_SIZE_KB": round(ft[name] / n, 1),
            "WRITE_SIZE_KB": round(wt.get(name, 0.0) / max(wc.get(name, 1), 1), 1),
            "hbm_read_GB": round(2 * ft[name] / n * 1024 / 1e9, 4),
            "hbm_write_GB": round(wt.get(name, 0.0) / max(wc.get(name, 1), 1) * 1024 / 1e9, 4),
        }
    if len(sys.argv) > 5:
        # recurrence summary for bench.py's roofline.traffic: KEY GROUP_BATCH ROWS_PER_GROUP  (rows of the [rows][128] fp32 slab)
        key, group_batch, rows = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
        unit = rows * 128 * 4 / 1e9
        rec = {k: v for k, v in out["per_launch"].items() if k.startswith("bilstm_split8_kernel") or k.startswith("bilstm_kernel")}
        n = sum(v["launches"] for v in rec.values())
        traffic = sum(v["launches"] * (v["hbm_read_GB"] + v["hbm_write_GB"]) for v in rec.values()) / n
        # with the pending norm folded in (template flag FUSE = true, the 4th argument): X and Z per direction, x' once, H = 7 units;
        # the first launch of a forward has no pending norm: X per direction + H = 4 units
        fused = sum(v["launches"] for k, v in rec.items() if re.match(r"bilstm_split8_kernel<\d+, \d+, \w+, true", k))
        algo = (7 * fused + 4 * (n - fused)) * unit / n
        note = (f"mean over the {n} recurrence launches of the profiled forwards ({fused} with the pending norm folded in: 7 slab "
                f"units of {unit:.3f} GB, the others 4)")
        # Linear folded in (round 5; template argument 8 = PJ): a launch meets every row once -- phase 1 reads x (+ z: pending norm) and
        # writes (x' and) the partial products: 4 / 2 units; phase 2 reads x' and the partial products and writes z: 3 units
        def units(k):
            m = re.match(r"bilstm_split8_kernel<\d+, \d+, \w+, (\w+), \w+, \w+, \w+, (\d)>", k)
            if not m or m.group(2) == "0":
                return None
            return 3 if m.group(2) == "2" else (4 if m.group(1) == "true" else 2)
        if any(units(k) for k in rec):
            algo = sum(v["launches"] * (units(k) or 0) for k, v in rec.items()) * unit / n
            note = (f"mean over the {n} launches of the recurrence with the Linear folded in (phase 1: x, z in, x' and partial products out = 4 slab "
                    f"units of {unit:.3f} GB, 2 without a pending norm; phase 2: x', partial products in, z out = 3)")
        out[key] = {"group_batch": group_batch, "traffic_GB_per_launch": round(traffic, 4), "algorithmic_GB_per_launch": round(algo, 4),
                    "note": note + f"; the packed weight pieces come from L2.  traffic / algorithmic = {traffic / algo:.3f}"}
    if len(sys.argv) > 7:
        # whole forward: FORWARDS = forwards in the profiled run (bench.py --steps 1 --warmup 1: two), PATHS = DPRNN paths per forward
        # (12).  Slab units of 512 B per row and path = (traffic of the recurrence + projection kernels) / (paths x rows x 512 B).
        forwards, paths = int(sys.argv[7]), int(sys.argv[8]) if len(sys.argv) > 8 else 12
        total = sum(v["launches"] * (v["hbm_read_GB"] + v["hbm_write_GB"]) for v in out["per_launch"].values()) / forwards
        path_k = {k: v for k, v in out["per_launch"].items() if k.startswith(("bilstm_split8_kernel", "bilstm_kernel", "proj_f16_kernel"))}
        path_gb = sum(v["launches"] * (v["hbm_read_GB"] + v["hbm_write_GB"]) for v in path_k.values()) / forwards
        batch_rows = int(sys.argv[6]) * (64 // int(sys.argv[5]))
        out["forward"] = {"forwards_profiled": forwards, "total_GB_per_forward": round(total, 2), "dprnn_paths_GB_per_forward": round(path_gb, 2),
                          "slab_units_per_row_and_path": round(path_gb * 1e9 / (paths * batch_rows * 512), 3),
                          "note": "recurrence (+ projection) kernels only in the path figure; SURVEY.md section 8(d) counts 4 units per path with ideal "
                                  "fusion, rounds 2-4 moved 10 (x, z read per direction; x', h per direction, z written; h read back)"}
    bw = {k: v for k, v in out["per_launch"].items() if k.startswith("lstm_bwd8_f16_kernel")}
    if bw:
        # training step: the BPTT launch of one 32-utterance micro-batch (bench.py --mode train: 64 utterances as 2 x 32), for
        # bench.py's training roofline.traffic.  Algorithmic bytes per slab row and both directions: tape 2 x 128 units x 5 floats,
        # dH 256 floats in; dG 1024 four-byte elements (f16 piece pairs in the quad layout) and dX 2 x 128 floats out = 11 264 B
        # (packed gate tape, kernel <.., true>: 2 x 128 units x 4 words = 4096 B of tape, 10 240 B in all)
        n = sum(v["launches"] for v in bw.values())
        traffic = sum(v["launches"] * (v["hbm_read_GB"] + v["hbm_write_GB"]) for v in bw.values()) / n
        packed = all(re.match(r"lstm_bwd8_f16_kernel<\w+, true>", k) for k in bw)
        tape = 4096 if packed else 5120
        algo = 32 * 48500 * (tape + 6144) / 1e9
        out["lstm_bwd8_f16_kernel"] = {"group_batch": 32, "traffic_GB_per_launch": round(traffic, 4), "algorithmic_GB_per_launch": round(algo, 4),
                                       "tape_bytes_per_row": tape,
                                       "note": f"mean over {n} BPTT launches; per slab row and both directions: tape {tape} B + dH 1024 B in, dG 4096 B + "
                                               f"dX 1024 B out; traffic / algorithmic = {traffic / algo:.3f}"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
