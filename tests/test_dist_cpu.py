"""Multi-process path on CPU: world_size-2 gloo runs of the sharding / broadcast / gather helpers
that the N>1 benchmark and a multi-GPU deployment use (RCCL on the GPU box)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_balanced_and_complete():
    from parakeet_amd.dist import shard_indices
    rng = np.random.default_rng(0)
    costs = rng.integers(20, 200, size=256).tolist()
    shards = [shard_indices(costs, 8, r) for r in range(8)]
    assert sorted(i for s in shards for i in s) == list(range(256))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)
    # weak-scaling benchmark shape: equal costs -> equal counts
    shards = [shard_indices([128] * 256, 8, r) for r in range(8)]
    assert all(len(s) == 32 for s in shards)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from parakeet_amd import dist as pdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 1. one-collective weight broadcast
    state = {{"a.weight": np.arange(12, dtype=np.float32).reshape(3, 4), "b.bias": np.full(5, 7.0, np.float32)}}
    mine = state if rank == 0 else {{k: np.zeros_like(v) for k, v in state.items()}}
    got = pdist.broadcast_state_dict(mine, src=0)
    assert list(got) == list(state)
    for k in state:
        assert np.array_equal(got[k], state[k]), k
    # 2. utterance sharding + ragged gather of per-rank packed "waveforms"
    costs = [5, 9, 3, 7, 4, 8]
    own = pdist.shard_indices(costs, world, rank)
    lens = [costs[i] * 2 for i in own]
    local = torch.cat([torch.full((n,), float(i)) for i, n in zip(own, lens)]) if own else torch.zeros(0)
    bufs, meta = pdist.gather_ragged(local, lens)
    seen = {{}}
    for r in range(world):
        idx = pdist.shard_indices(costs, world, r)
        assert meta[r] == [costs[i] * 2 for i in idx]
        o = 0
        for i, n in zip(idx, meta[r]):
            seg = bufs[r][o:o + n]
            assert seg.numel() == n and bool((seg == float(i)).all())
            seen[i] = n
            o += n
    assert sorted(seen) == list(range(len(costs)))
    # 3. the same collection on ONE rank (direct sends, exact sizes), any destination
    for dst in (0, world - 1):
        bufs1, meta1 = pdist.gather_ragged_to(local, lens, dst=dst)
        assert meta1 == meta
        if rank == dst:
            assert len(bufs1) == world
            for r in range(world):
                assert torch.equal(bufs1[r], bufs[r]), r
        else:
            assert bufs1 is None
    # a rank with nothing to send (more ranks than utterances) must not hang the others
    empty = torch.zeros(0) if rank == world - 1 else local
    b2, m2 = pdist.gather_ragged_to(empty, [] if rank == world - 1 else lens, dst=0)
    if rank == 0:
        assert b2[world - 1].numel() == 0 and m2[world - 1] == []
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_broadcast_shard_gather(tmp_path, world):
    """world 2, and world 8 = the node the SCALE record will be taken on (6 utterances over 8 ranks: two ranks own nothing)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    bad = [f"--- rank {r} (rc {p.returncode})\n{o[-1500:]}" for r, (p, o) in enumerate(zip(procs, outs)) if p.returncode != 0]
    assert not bad, "\n".join(bad)      # (the rank that failed FIRST is the one whose message is not "connection closed")
    assert all("ok" in o for o in outs)
