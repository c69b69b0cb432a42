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
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for attempt in range(5):  # a port that was free a moment ago may have been taken (another run's store in TIME_WAIT): pick again
        os.environ["MASTER_PORT"] = str(_free_port())
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            break
        except (RuntimeError, ValueError):
            if attempt == 4:
                raise
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


def test_c_abi_record_exchange_world1():
    """include/orbd.h: RCCL driven from liborbx.so itself (no torch.distributed): unique id, communicator, gather to
    the root and all-gather of real extraction records on a side stream.  One GPU here, so world = 1; the N > 1 path
    is the same grouped send / recv with more peers."""
    from monoorbslam3_amd import dist as D, synth
    from monoorbslam3_amd.extractor import ORBExtractor
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w, h, B = 640, 240, 3
    frames = torch.from_numpy(synth.make_frames(B, w, h, seed=5)).to(dev)
    ex = ORBExtractor(500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = ex.max_keypoints(w, h)
    d_kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    ex.extract_batch_device(frames.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(),
                            side.cuda_stream)
    xc = D.RecordExchange(0, 1, D.RecordExchange.unique_id(), device=0)
    # both are asked of RCCL (ncclCommUserRank / ncclCommCount), and orbd_create refused a communicator that disagreed
    assert (xc._L.orbd_rank(xc._h), xc._L.orbd_world(xc._h)) == (0, 1)
    assert xc.world_reported() == 1
    assert (xc._L.orbd_rank(None), xc._L.orbd_world(None)) == (-1, 0)
    got = xc.gather(d_n, d_kp, d_desc, root=0, stream=side.cuda_stream)
    allg = xc.allgather(d_n, d_kp, d_desc, stream=side.cuda_stream)
    side.synchronize()
    assert int(d_n.min()) > 100
    for g in (got, allg):
        assert torch.equal(g[0][0], d_n) and torch.equal(g[1][0], d_kp) and torch.equal(g[2][0], d_desc)
    with pytest.raises(Exception):
        xc.gather(d_n, d_kp, d_desc, root=3)
    xc.close()
