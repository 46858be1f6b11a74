#!/bin/bash
# Round 4, call T: joules per matrix FLOP by MFMA shape (register-operand loops at the power cap) + fresh PMC passes of the training
# step with the packed gate tape.
set -o pipefail
o=gpurun_out/r04t; mkdir -p $o
timeout -k 10 200 tools/probes/mfma_shape_energy.bin 4 > $o/mfma_shape_energy.txt 2>&1; echo "probe rc=$?"; cat $o/mfma_shape_energy.txt
bash tools/collect_pmc_train.sh r04t && cp gpurun_out/r04t_pmc_traffic_train.json $o/pmc_traffic_train_packed.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r04t/pmc_traffic_train_packed.json'))
print(d.get('lstm_bwd8_f16_kernel'))
for k,v in d['per_launch'].items():
    if k.startswith(('lstm_bwd8','bilstm_split8','wgrad_lstm')): print(k[:80], v['launches'], v['hbm_read_GB'], v['hbm_write_GB'])
P
