"""bench.py --gpus N must work as a plain command (no outer torchrun): the parent starts N ranks as child processes
before anything touches the GPU, relays rank 0's JSON line and passes a failing rank's exit code on.  CPU only: the ranks
run a stub instead of the GPU benchmark."""
import json
import os
import subprocess
import sys

from conftest import ROOT

STUB = """
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and "LOCAL_RANK" in os.environ
if rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(3)
if rank == 0:
    print(json.dumps({"world": world, "argv": sys.argv[1:]}), flush=True)
"""
DRIVER = """
import sys
sys.path.insert(0, {root!r})
import bench
assert "torch" not in sys.modules, "the launching parent must not import torch (and so cannot touch the GPU)"
rc = bench._self_launch(2, script={stub!r}, argv=["--gpus", "2", "--fail-rank", sys.argv[1]])
assert "torch" not in sys.modules
sys.exit(rc)
"""


def _run(tmp_path, fail_rank):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER.format(root=ROOT, stub=str(stub)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, str(drv), str(fail_rank)], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_relays_rank0_line(tmp_path):
    r = _run(tmp_path, -1)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"world": 2, "argv": ["--gpus", "2", "--fail-rank", "-1"]}


def test_self_launch_passes_failure_on(tmp_path):
    r = _run(tmp_path, 1)
    assert r.returncode != 0


def test_bench_has_no_module_level_gpu_imports():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def parse_args")]
    assert "import torch" not in head and "import numpy" not in head
    body = src[src.index("def main():"):]
    assert body.index("_self_launch(args.gpus)") < body.index("import torch")
