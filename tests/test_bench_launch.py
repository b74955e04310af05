"""bench.py's rank launcher (ADVICE r1 / VERDICT r1 item 2): `python bench.py --gpus N` must start N ranks itself and
report the world size the process group actually spans; a WORLD_SIZE that disagrees with --gpus is an error."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_gpus2_spawns_two_ranks_dry_run():
    """CPU tier: the launcher + gloo rendezvous + rank-0 JSON (no kernels: --dry-run reports value null)."""
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["value"] is None


@pytest.mark.timeout(300)
def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpus2_on_one_gpu_gloo_rehearsal():
    """GPU tier: two real ranks (sharing the box's one GPU, gloo in place of RCCL) through the whole bench line, the sharded
    LOSO block included: each rank runs its own SubjectStore, LockstepTrainer fold batches and the metric gather."""
    r = _run(["--gpus", "2", "--batch", "256", "--samples", "256", "--steps", "2", "--warmup", "1", "--cpu-budget", "0.5", "--b64-steps", "20",
              "--loso", "1", "--loso-windows", "24", "--loso-spread", "3", "--loso-epochs", "3", "--profile-steps", "1", "--long-steps", "4"],
             {"MSIG_DIST_BACKEND": "gloo"}, timeout=800)
    assert r.returncode == 0, "\n".join([ln for ln in r.stderr.splitlines() if "[rank0]" in ln][-60:]) + "\n" + r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["parallelism"] == "replica x2"
    assert j["roofline"] is not None and j["b64"]["value"] > 0
    assert j["loso"]["folds"] == 15 and j["loso"]["folds_per_rank"] == [8, 7] and 0.0 < j["loso"]["mean_acc"] <= 1.0
    assert j["cpu_baseline"] is not None and j["cpu_baseline"]["value"] > 0          # reported at every N, not only N = 1
    assert j["long_run"]["steps"] == 4 and j["ms_per_step_spread"]["min"] <= j["ms_per_step_spread"]["max"]
