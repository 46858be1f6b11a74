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


SPY = """
import importlib.abc, os, sys
os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
class Spy(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name == "torch":
            print("AT_TORCH_IMPORT", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), flush=True)
            os._exit(0)         # the rank stops here: this test has no GPU
sys.meta_path.insert(0, Spy())
sys.path.insert(0, {root!r})
sys.argv = ["bench.py", "--gpus", "1"] + {extra!r}
{body}
"""


def _spy(tmp_path, body, extra=()):
    f = tmp_path / "spy.py"
    f.write_text(SPY.format(root=ROOT, extra=list(extra), body=body))
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    return subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=300, env=env)


def test_rank_under_the_drivers_torchrun_has_dmabuf_ipc_before_torch_is_imported(tmp_path):
    """The driver starts the ranks itself (python -m torch.distributed.run ... bench.py --gpus N): every rank must have
    HSA_ENABLE_IPC_MODE_LEGACY=0 in its environment before torch (and with it the HIP runtime / RCCL) is imported, in the
    forward and in the training mode -- not only under bench.py's own self-launch."""
    for extra in ((), ("--mode", "train")):
        r = _spy(tmp_path, "import bench\nbench.main()", extra)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "AT_TORCH_IMPORT 0" in r.stdout, r.stdout + r.stderr[-2000:]


def test_package_import_sets_dmabuf_ipc_before_torch(tmp_path):
    """Users of trainer.DistributedTrainer / trainer.init_process_group import the package, not bench.py."""
    r = _spy(tmp_path, "import tss_with_dprnn_amd")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "AT_TORCH_IMPORT 0" in r.stdout, r.stdout + r.stderr[-2000:]


def test_power_sampler_reads_rocm_smi_and_survives_without_it(tmp_path, monkeypatch):
    """bench.PowerSampler: parses rocm-smi's "Power (W)" / "sclk clock level" lines from a child process while a region runs; no
    rocm-smi on PATH (or unparsable output) means no samples and a null `power` object, never an error."""
    import importlib
    import time
    bench = importlib.import_module("bench")
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncat <<EOT\n"
                    "GPU[0]\t\t: mclk clock level: 3: (2000Mhz)\n"
                    "GPU[0]\t\t: sclk clock level: 1: (1731Mhz)\n"
                    "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1386.0\nEOT\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    ps = bench.PowerSampler(True, 0)
    with ps:
        time.sleep(1.0)
    out = ps.summary()
    assert out is not None and out["socket_W"] == 1386.0 and out["sclk_MHz"] == 1731 and out["samples"] >= 1
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    ps = bench.PowerSampler(True, 0)
    with ps:
        pass
    assert ps.summary() is None
    assert bench.PowerSampler(False, 0).summary() is None
