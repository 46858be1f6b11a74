#!/bin/bash
# Round 4, call V (final tree): the driver's sequence on the current tree -- the whole GPU suite, smoke, the default bench line, the training line,
# the N = 2 rehearsal over gloo on one GPU and the one-rank RCCL run under torchrun.
set -o pipefail
o=gpurun_out/r04v2; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $o/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.txt
timeout -k 10 500 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; python -c "import json; d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['exact_fp32']['value'], d['host'].get('reserved_GiB'), d['other_configs']['spe_cat_b64']['value'], d['other_configs']['ira_full_length']['value'], d['cpu_baseline']['value'], d['parity']['rel_l2'])"
timeout -k 10 400 python bench.py --mode train > $o/train.json 2> $o/train.err; echo "train rc=$?"; python -c "import json; d=json.load(open('$o/train.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'])"
DPRNN_BENCH_BACKEND=gloo DPRNN_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 > $o/bench_n2_gloo.json 2> $o/bench_n2.err; echo "n2 rc=$?"; python -c "import json; d=json.load(open('$o/bench_n2_gloo.json')); print(d['value'], d['n_gpus'], d['distributed'])"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-optional-modes > $o/bench_torchrun_n1.json 2> $o/bench_torchrun.err; echo "torchrun rc=$?"; python -c "import json; d=json.load(open('$o/bench_torchrun_n1.json')); print(d['value'], d['distributed'])"
