"""GPU: bench.py prints one JSON line with the contract's keys (tiny run: 4 trajectories in 2 lanes, 2 timed steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
                          "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=ROOT)     # (the extra lines: next test)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["trajectories_per_gpu"] == 4 and d["config"]["lanes"] == 2 and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # value = trajectories * steps / wall
    assert abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-2


def test_bench_strong_scaling_and_small_batch_lines():
    """--batch >= --ensemble: the line also carries BASELINE config 3 as written (ONE ensemble of 32 sharded over the GPUs: strong
    scaling) and the small per-GPU batches (B in {1..16})."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "32",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    s = d["ensemble_strong_scaling"]
    assert s["ensemble"] == 32 and s["trajectories_per_gpu"] == 32 and s["scaling"] == "strong" and s["value"] > 0
    assert set(d["small_batch"]) == {"B1", "B4", "B16"} and all(v["value"] > 0 for v in d["small_batch"].values())
    assert d["attention_block"]["frac"] > 0 and d["roofline"]["traffic_source"]
    # round 3: the Conv3d kernel as it runs in the timed two-lane configuration, the fp32-class engine's throughput beside the bf16
    # headline, and the VAE's two ends of sample() with their roofline fractions
    ins = d["roofline"]["in_situ"]
    assert ins["avg_launch_us"] > 0 and abs(ins["frac"] - ins["achieved"] / d["roofline"]["peak"]) < 1e-3 and ins["launches_timed"] > 0
    f32 = d["precision_fp32"]
    assert f32["value"] > 0 and f32["value"] < d["value"] and f32["trajectories_per_gpu"] == 32 and "bf16x3" in f32["dtype"]
    for end in ("encode", "decode"):
        v = d["vae"][end]
        assert v["frames"] > 0 and v["ms"] > 0 and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3
