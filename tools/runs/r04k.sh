#!/bin/bash
# Round 4, call K: the speaker encoder's 1x1 convs in f16 pieces -- spe / ira / inference / parity tests, the Spe profile, the bench line.
set -o pipefail
o=gpurun_out/r04k2; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_gpu_spe.py tests/test_gpu_parity.py tests/test_gpu_inference.py -x -q -m gpu > $o/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $o/tests.txt
DPRNN_GRAPHS=0 timeout -k 10 300 python tools/bench_spe.py cat 64 > $o/spe_profile.txt 2>&1; head -8 $o/spe_profile.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; python -c "import json; d=json.load(open('$o/bench.json')); print(d['value'], d['exact_fp32']['value'], d['host'].get('reserved_GiB'), d['other_configs']['spe_cat_b64']['value'], d['other_configs']['ira_full_length']['value'])"
