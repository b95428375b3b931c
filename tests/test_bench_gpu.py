"""bench.py's RCCL branch on one MI355X: the nccl process group, the flat weight broadcast, the
barriers, the max-over-ranks reduction and `gather_ragged` run at world size 1 under PK_BENCH_FORCE_DIST=1, and the
strong-scaling path (several mini-batches per step, pipeline off) runs on the engine -- so the first execution of that
code is not the driver's 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline", *extra], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_forced_rccl_branch_at_world_1():
    out = _bench(env={"PK_BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 1e6
    assert out["gather_ms"] is not None and out["gather_ms"] >= 0.0
    assert out["pipeline_check"]["bit_identical_to_unpipelined"] is True
    assert out["roofline"]["bound"] == "hbm" and 0.2 < out["roofline"]["frac"] < 1.0


def test_strong_scaling_two_minibatches_on_the_engine():
    out = _bench("--scaling", "strong", "--global-batch", "64", env={"PK_BENCH_FORCE_DIST": "1"})
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 64
    assert out["config"]["utterances_per_gpu"] == 64 and out["config"]["pipeline"] == "none"
    assert out["value"] > 1e6 and out["gather_ms"] is not None
    assert abs(out["ms_per_step"] * 1e-3 * out["value"] - 64 * 163840) < 1.0      # value = whole-job samples / time


def test_under_torch_distributed_run_strong_scaling():
    """the exact launcher path of the driver's multi-GPU runs (and of bench.py's own self_spawn) --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N` --
    at N = 1, not the PK_BENCH_FORCE_DIST shortcut: RANK / WORLD_SIZE / MASTER_* come from the launcher, the RCCL group is
    created from them, the strong-scaling split runs two mini-batches, and `gather_ms` is reported."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PK_BENCH_FORCE_DIST"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-extras", "--no-cpu-baseline", "--scaling", "strong", "--global-batch", "64"]
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["config"]["utterances_per_gpu"] == 64
    assert out["gather_ms"] is not None and out["gather_ms"] >= 0.0 and out["value"] > 1e6
