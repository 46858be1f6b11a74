#!/bin/bash
# round 5, fourth session, final tree: the driver's launch forms -- one RCCL rank under torch.distributed.run, and the N = 2 rehearsal over
# gloo with both ranks on the one GPU of this box (forward line and training line: the flat gradient all-reduce between two processes)
set -o pipefail
o=gpurun_out/r05n; mkdir -p $o
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-optional-modes > $o/bench_torchrun_n1.json 2> $o/bench_torchrun.err || { tail -5 $o/bench_torchrun.err; exit 1; }
python -c "import json; d=json.load(open('$o/bench_torchrun_n1.json')); print('torchrun n1', d['value'], d['distributed'])" &&
DPRNN_BENCH_BACKEND=gloo DPRNN_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 > $o/bench_n2_gloo.json 2> $o/bench_n2.err || { tail -5 $o/bench_n2.err; exit 1; }
python -c "import json; d=json.load(open('$o/bench_n2_gloo.json')); print('n2 gloo', d['value'], d['n_gpus'], d['distributed'])" &&
DPRNN_BENCH_BACKEND=gloo DPRNN_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --mode train --gpus 2 --steps 4 --warmup 2 --batch 16 > $o/train_n2_gloo.json 2> $o/train_n2.err || { tail -8 $o/train_n2.err; exit 1; }
python -c "import json; d=json.load(open('$o/train_n2_gloo.json')); print('train n2 gloo', d['value'], d['ms_per_step'], d['n_gpus'], d.get('distributed'))"
