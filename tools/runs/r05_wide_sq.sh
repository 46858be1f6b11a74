#!/bin/bash
# SQ counters of the any-width recurrence launch alone (tools/bench_wide_rnn.py 256 256 64): where its cycles go
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05sq; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python3 tools/bench_wide_rnn.py 256 256 64 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob('gpurun_out/r05sq/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
with open('gpurun_out/r05sq/wide_rnn_sq.txt', 'w') as out:
    for k, c in tot.items():
        if 'wide_rnn_kernel' not in k: continue
        n = max(cnt[k], 1)
        w = c['SQ_WAVE_CYCLES']
        line = (f"{k}: {n} launches; per launch SQ_WAVE_CYCLES {w/n:.3e}; WAIT_ANY {c['SQ_WAIT_ANY']/w:.3f}, WAIT_INST_ANY {c['SQ_WAIT_INST_ANY']/w:.3f}, "
                f"ACTIVE_INST_ANY {c['SQ_ACTIVE_INST_ANY']/w:.3f} of the wave cycles; SQ_VALU_MFMA_BUSY_CYCLES {c['SQ_VALU_MFMA_BUSY_CYCLES']/n:.3e}, "
                f"SQ_BUSY_CYCLES {c['SQ_BUSY_CYCLES']/n:.3e}; LDS bank conflict cycles / LDS active cycles {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.3f}")
        print(line); out.write(line + "\n")
PY
rm -rf $O/pmc
