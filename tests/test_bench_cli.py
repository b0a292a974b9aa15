"""bench.py as the driver runs it, including the N > 1 code path (tile-set dispatch + RCCL all-gather at readback)
forced onto one GPU: the JSON contract, parity fields, and that the distributed path survives its own warm-up gather."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract(built):
    d = run_bench({}, "--steps", "4", "--warmup", "2", "--size", "256")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["value"] > 0 and d["higher_is_better"] is True
    assert "workload" in d["config"] and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["parity"]["differing_f16_words"] == 0 and d["parity"]["rmse_vs_cpu_restatement"] == 0.0


def test_distributed_bench_path_on_one_gpu(built):
    """world_size 1 through torch.distributed + RCCL: warm-up gather, timed steps, final gather."""
    env = {"LUPIN_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    d = run_bench(env, "--steps", "6", "--warmup", "4", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["path_bounces"] > 0


def test_gather_framebuffer_with_device_payloads(built):
    """distributed.gather_framebuffer + HipTileOps (torch CUDA payloads, pack / unpack kernels) for world 3 on one GPU
    (tests/_gather_worker.py; its own process because torch brings its own HIP runtime, which has to load first)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gather_worker.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "GATHER OK" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
