"""bench.py's own launcher: `python bench.py --gpus N` must start N ranks itself (the driver calls it exactly like that) and
must refuse to run with fewer devices than asked for.  Covered on CPU through --selftest-cpu (gloo, no PatchMatch work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=600, env=env)


def test_gpus_2_spawns_two_ranks_and_reports_the_slowest():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] is None and "selftest" in out
    assert out["config"]["backend"] == "gloo"
    per_rank = out["rank_ms_per_step"]
    assert len(per_rank) == 2 and per_rank[1] > 1.5 * per_rank[0]           # rank 1 sleeps twice as long
    assert out["ms_per_step"] >= per_rank[1] * 0.99                            # MAX over ranks, not rank 0's own time
    assert out["allgather_ms"] is not None


def test_refuses_more_gpus_than_visible():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env_extra={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = _run("--gpus", "1", "--selftest-cpu", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
