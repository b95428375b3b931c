"""bench.py's multi-process orchestration on CPU (gloo): `python bench.py --gpus 2 --dry-run` must spawn two
ranks by itself, shard the utterances, broadcast, barrier, time, gather and print ONE JSON line with n_gpus 2
(--gpus used to be ignored).  --dry-run replaces only the engine call with a stub."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1",
                        *extra], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    return json.loads(lines[0])


def test_gpus2_self_spawns_two_ranks_weak():
    out = _run("--gpus", "2")
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["scaling"] == "weak"
    cfg = dict(out["config"])
    coll = cfg.pop("collectives")
    assert cfg == {"global_batch": 64, "utterances_this_rank": 32, "minibatches_per_step": 1}
    assert out["gather_ms"] is not None and out["gather_ms"] >= 0.0      # gather_ragged_to (rank 0) ran on both ranks
    assert out["gather_all_ms"] is not None and out["gather_all_ms"] >= 0.0
    assert "none" in coll["in_the_timed_step"] and "broadcast" in coll["weights"] and "rank 0" in coll["results"]


def test_gpus2_strong_scaling_splits_the_same_batch():
    out = _run("--gpus", "2", "--scaling", "strong", "--global-batch", "24", "--minibatch", "8")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    cfg = {k: v for k, v in out["config"].items() if k != "collectives"}
    assert cfg == {"global_batch": 24, "utterances_this_rank": 12, "minibatches_per_step": 2}


def test_gpus8_the_node_the_scale_record_is_taken_on():
    """VERDICT r4 "next" #7: the first real 8-GPU run must be boring -- 8 gloo ranks, weak (32 utterances per rank = BASELINE
    config 4: 256 utterances) and strong (the same 256 split 8 ways, 32 per rank in one mini-batch; and a ragged split)."""
    env = {"OMP_NUM_THREADS": "1"}
    out = _run("--gpus", "8", env=env)
    cfg = {k: v for k, v in out["config"].items() if k != "collectives"}
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    assert cfg == {"global_batch": 256, "utterances_this_rank": 32, "minibatches_per_step": 1}
    assert out["gather_ms"] is not None and out["gather_all_ms"] is not None
    out = _run("--gpus", "8", "--scaling", "strong", env=env)
    cfg = {k: v for k, v in out["config"].items() if k != "collectives"}
    assert out["n_gpus"] == 8 and out["scaling"] == "strong"
    assert cfg == {"global_batch": 256, "utterances_this_rank": 32, "minibatches_per_step": 1}
    out = _run("--gpus", "8", "--scaling", "strong", "--global-batch", "100", "--minibatch", "8", env=env)
    cfg = {k: v for k, v in out["config"].items() if k != "collectives"}
    assert cfg == {"global_batch": 100, "utterances_this_rank": 13, "minibatches_per_step": 2}    # rank 0 of 100 = 4 x 13 + 4 x 12


def test_world_size_must_match_gpus():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "4"], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"--gpus 4" in r.stderr
