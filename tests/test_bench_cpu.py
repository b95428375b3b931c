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
    last = r.stdout.decode().rstrip("\n").splitlines()[-1]
    assert last == lines[0] and len(last) < 8192            # what the driver parses: the LAST stdout line, and a short one
    return json.loads(last)


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
    # VERDICT r5 "next" #8: the communicator's own rank count and the spread of the per-rank step times are in the line
    assert out["rccl_world_size"] == 8
    assert 0.0 < out["rank_ms_per_step"]["min"] <= out["rank_ms_per_step"]["max"] == out["ms_per_step"]
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


def test_line_is_small_and_last_stdout_line_parses():
    """VERDICT r5 "next" #1: round 5's line was a 20.5 KB document and came back from the driver as `"parsed": null`.  The line is
    now built by bench.short_line from the full record (which goes to profiles/bench_extras_last.json): round 5's own full
    record, stuffed further with prose in every extra, must give a line under 8 KB that carries the contract's fields, the
    roofline with its strict-fp32 companion, the CPU baseline and the parity check -- and no prose."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    for k, v in full["extras"].items():
        if isinstance(v, dict):
            v["what"] = "prose " * 2000
            v["runs_ms"] = list(range(5000))
    full["extras"]["broken"] = {"error": "RuntimeError('x' * 100000)" + "x" * 100000}
    full["dtype_note"] = "note " * 5000
    full["rccl_world_size"] = 8
    full["rank_ms_per_step"] = {"min": 44.1, "max": 45.9, "by_rank": list(range(8))}
    line = bench.short_line(full)
    assert len(line) < 8192 and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"):
        assert k in out, k
    assert out["vs_baseline"] is None and out["config"]["workload"].startswith("FastSpeech2+PWG")
    r = out["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and r["traffic"] > 1e9
    assert r["exact_f32"]["bound"] == "mfma" and 0.9 < r["exact_f32"]["frac_of_fp32_mfma"] <= 1.0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 32
    assert out["rccl_world_size"] == 8 and out["rank_ms_per_step"] == {"min": 44.1, "max": 45.9}
    assert out["others"]["waveflow_c64_batch8_ms"] > 10 and out["others"]["errors_in"] == ["broken"]
    assert max(len(v) for v in _strings(out)) <= 200            # labels, not paragraphs


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)
