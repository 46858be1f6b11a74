#!/bin/bash
# PMC counters of the default recurrence kernel alone (tools/bench_recur.py --gemm-mode 3 --only 2064,1): separate passes, CSV output.
tag=${1:-r04}
export TMPDIR=/tmp
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_recur/$name -- python3 tools/bench_recur.py --gemm-mode 3 --only 2064,1 > gpurun_out/${tag}_pmc_recur_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/${tag}_pmc_recur_$name.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/${tag}_pmc_recur/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bilstm_split8" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k] / cnt[k]:18.0f}   (mean of {cnt[k]} launches)")
PY
rm -rf gpurun_out/${tag}_pmc_recur
