"""CPU: `bench.py --gpus N` outside a launcher starts N ranks itself (one process per GPU, torch.distributed.run on 127.0.0.1) and
never reports a smaller job as n_gpus = N (the reference runs one process per GPU under DDP,
scripts/prediff/sevirlr/train_sevirlr_prediff.py:648)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""         # also on a GPU box: this test is about the device-count check
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def test_gpus_2_without_devices_fails_loudly():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=_env())
    assert out.returncode != 0
    assert "n_gpus" not in out.stdout            # no JSON line of a smaller job
    assert "--gpus 2" in out.stderr and "visible" in out.stderr


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr and "{" not in out.stdout


def test_self_launch_starts_one_rank_per_gpu():
    """--launch-check: the same re-exec under torch.distributed.run as a real --gpus 2 run, ranks meet in a gloo group instead of
    touching a device; rank 0 reports the world it sees."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2}


def test_ensemble_e2e_sharding_on_a_two_rank_gloo_world():
    """--ensemble-e2e --launch-check: the end-to-end line's member sharding and its gather (prediff_amd.ensemble.sample_ensemble with a
    stand-in sampler) on two gloo ranks started by bench.py itself: every member comes back once, in member order."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--ensemble-e2e", "--ensemble", "7"], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"launch_check": True, "ensemble_e2e": True, "n_gpus": 2, "ranks_seen": 2, "members": 7, "gathered_in_member_order": True}
