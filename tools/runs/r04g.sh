#!/bin/bash
# Round 4, call G: batch-size sweep with the new utterance-group policy, the training step's HBM traffic (PMC), bench line, train line.
set -o pipefail
o=gpurun_out/r04g2; mkdir -p $o
timeout -k 10 600 python tools/latency_modes.py 1:32:1,36:64:4 3,0 > $o/batch_sweep_after.txt 2>&1; cat $o/batch_sweep_after.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inference.py -x -q -m gpu -k "graph or async or repeat or pins or fall_back or separate_many" > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $o/tests.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/collect_pmc_train.sh r04t || echo "pmc train failed"
cp gpurun_out/r04t_pmc_traffic_train.json $o/ 2>/dev/null
timeout -k 10 400 python bench.py --mode train --steps 10 --warmup 3 > $o/train.json 2> $o/train.err; tail -c 600 $o/train.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "import json; d=json.load(open('$o/bench.json')); print(d['value'], d['exact_fp32']['value'], d['host'].get('reserved_GiB'), d['other_configs']['spe_cat_b64']['value'], d['other_configs']['ira_full_length']['value'])"
