"""The driver-facing multi-GPU entry: `python bench.py --gpus N` must itself
become N ranks.  Runs here without a GPU through the hidden `--backend gloo --dry-run` (launcher, rendezvous, process
group, rank / device census, gradient-bucket all-reduce, max-over-ranks timing; no HIP work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _run(argv, env, timeout=300):
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "0"], _env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["dry_run"] is True
    assert sorted(res["collective"]["ranks_devices"]) == [[0, 0], [1, 1]]
    # three summing all-reduces of (1, 2): 3 -> 6 -> 12: the bucket really crossed the two processes
    assert res["allreduce_sum_check"] == 12.0
    assert "launching 2 ranks" in r.stderr


def test_world_size_must_equal_gpus():
    # launched the driver's way for ONE rank but asked for two: refuse instead of printing n_gpus = 1
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run"],
             _env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_real_path_fails_loudly_without_devices_after_spawning():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has two devices: the real path would run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], _env(), timeout=600)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr                 # the ranks were started ...
    assert "has no HIP device" in r.stderr                 # ... and each one said why it cannot run
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
