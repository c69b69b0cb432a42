"""RCCL code path on the GPU box: world_size 1 (one GPU here), real nccl backend, device tensors.  Checks that the
collectives used by the batch-of-frames mode accept the payload layout; N > 1 is covered by the gloo tests."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_gather_paths_world1():
    import torch.distributed as dist
    from monoorbslam3_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        b, cap = 5, 12
        rng = np.random.RandomState(1)
        counts = torch.from_numpy(rng.randint(0, cap, b).astype(np.int32)).to(dev)
        kps = torch.from_numpy(rng.randint(0, 256, (b, cap, 28)).astype(np.uint8)).to(dev)
        desc = torch.from_numpy(rng.randint(0, 256, (b, cap, 32)).astype(np.uint8)).to(dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):  # the bench issues the gather on a side stream
            got = D.gather_records_to_root(counts, kps, desc)
        side.synchronize()
        c, k, d = D.unpack_records(got[0], b, cap)
        assert torch.equal(c, counts) and torch.equal(k, kps) and torch.equal(d, desc)
        g_counts, g_kps, g_desc = D.gather_records(counts, kps, desc)
        assert torch.equal(g_counts[0], counts) and torch.equal(g_kps[0], kps) and torch.equal(g_desc[0], desc)
        dist.barrier()
    finally:
        dist.destroy_process_group()
