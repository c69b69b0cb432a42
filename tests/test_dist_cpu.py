"""N > 1 path on CPU: world_size-2 gloo, frames sharded round-robin, one all-gather of padded records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monoorbslam3_amd import dist as D

CAP = 16


def _fake_extract(frames):
    """deterministic stand-in for the GPU extractor: records derived from the frame bytes"""
    b = len(frames)
    counts = torch.zeros(b, dtype=torch.int32)
    kps = torch.zeros((b, CAP, 28), dtype=torch.uint8)
    desc = torch.zeros((b, CAP, 32), dtype=torch.uint8)
    for i, f in enumerate(frames):
        n = int(f[0, 0]) % CAP
        counts[i] = n
        for k in range(n):
            kps[i, k] = torch.from_numpy(np.full(28, (int(f[0, 1]) + k) % 256, np.uint8))
            desc[i, k] = torch.from_numpy(np.full(32, (int(f[1, 0]) * 3 + k) % 256, np.uint8))
    return counts, kps, desc


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    frames = rng.randint(0, 256, (n_frames, 4, 6)).astype(np.uint8)
    res = D.extract_sharded(frames, _fake_extract, CAP)
    q.put((rank, [(c, k.tobytes(), d.tobytes()) for c, k, d in res]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_root(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(100 + rank)
    frames = rng.randint(0, 256, (3, 4, 6)).astype(np.uint8)
    counts, kps, desc = _fake_extract(frames)
    got = D.gather_records_to_root(counts, kps, desc, dst=0)
    if rank == 0:
        out = []
        for r in range(world):
            c, k, d = D.unpack_records(got[r], 3, CAP)
            out.append((c.numpy().tobytes(), k.numpy().tobytes(), d.numpy().tobytes()))
        q.put(out)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_to_root_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_root, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        rng = np.random.RandomState(100 + r)
        frames = rng.randint(0, 256, (3, 4, 6)).astype(np.uint8)
        c, k, d = _fake_extract(frames)
        assert got[r] == (c.numpy().tobytes(), k.numpy().tobytes(), d.numpy().tobytes())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_frames", [8, 5])
def test_sharded_extract_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.RandomState(0)
    frames = rng.randint(0, 256, (n_frames, 4, 6)).astype(np.uint8)
    counts, kps, desc = _fake_extract(frames)
    ref = [(int(counts[i]), kps[i, :counts[i]].numpy().tobytes(), desc[i, :counts[i]].numpy().tobytes())
           for i in range(n_frames)]
    assert got[0] == ref and got[1] == ref  # every rank ends with every frame's records, in global order


def test_shard_indices():
    assert D.shard_indices(10, 1, 4) == [1, 5, 9]
    assert D.unshard_order(5, 2) == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2)]


def test_c_abi_sharding_arithmetic_matches_the_torch_path():
    """orbd_shard_count / orbd_shard_global_index (include/orbd.h) = monoorbslam3_amd.dist.shard_indices: frame i of a
    global batch goes to rank i % world."""
    from monoorbslam3_amd import dist as D
    for n in (0, 1, 7, 8, 9, 512):
        for world in (1, 2, 3, 8):
            seen = []
            for rank in range(world):
                mine = D.RecordExchange.shard(n, rank, world)
                assert mine == D.shard_indices(n, rank, world)
                seen += mine
            assert sorted(seen) == list(range(n))
            # uneven shards (n % world != 0): the exchange moves equal blocks of orbd_shard_capacity frames, the
            # padding the torch path's extract_sharded applies (per = ceil(n / world), padded counts = 0)
            cap = D.RecordExchange.capacity(n, world)
            assert cap == (n + world - 1) // world == max(len(D.shard_indices(n, r, world)) for r in range(world))
            assert all(cap - len(D.shard_indices(n, r, world)) in (0, 1) for r in range(world))


def test_c_abi_exchange_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    from monoorbslam3_amd import dist as D
    from monoorbslam3_amd._lib import OrbxError
    with pytest.raises(OrbxError, match="no HIP device"):
        D.RecordExchange.unique_id()
    with pytest.raises(OrbxError, match="no HIP device"):
        D.RecordExchange(0, 1, bytes(128))


def test_bench_launches_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts the two ranks itself (children of the
    launcher, before any GPU call), passes rank 0's JSON line through and returns the children's status.  The step is the
    `--stub-step` sleep: what is checked is the launcher and the barrier / max-over-ranks control flow over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-step", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["data"] == "stub"
    assert out["ms_per_step"] >= 2.0  # rank r sleeps r + 1 ms per step between the barriers, the slower rank counts
    # what a scaling curve needs to be read (`scaling_diag`, the same keys as a real N > 1 run prints): every rank's own time
    # before the barrier -- rank 1's about twice rank 0's here --, the gather's time, the world size as the backend reports it
    diag = out["scaling_diag"]
    assert set(diag) >= {"ms_per_step_per_rank", "ms_per_step_rank_min", "ms_per_step_rank_max", "gather_ms_mean", "gather_ms_max",
                         "world", "world_c_abi", "backend", "gather"}
    assert diag["world"] == 2 and len(diag["ms_per_step_per_rank"]) == 2
    assert diag["ms_per_step_per_rank"][1] > 1.5 * diag["ms_per_step_per_rank"][0] >= 1.5
    assert diag["ms_per_step_rank_max"] == max(diag["ms_per_step_per_rank"]) <= out["ms_per_step"] + 0.5
    # a mismatch between --gpus and an existing WORLD_SIZE is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-step"], capture_output=True, text=True,
                        timeout=120, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE" in r2.stderr
