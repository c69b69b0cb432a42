"""Batch-of-frames mode: independent frames sharded across ranks, one gather of the results.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  Frame i of a global batch goes to rank i % world
(SURVEY.md section 8e).  Extraction has no cross-frame state, so the only
communication is ONE all-gather per batch of fixed-capacity records
(count:int32, keypoints cap x 28 B, descriptors cap x 32 B): ~120 KB per frame,
latency-bound on the xGMI mesh, no reduction, no ring-sized tuning needed.
The reference has no counterpart (single process, CPU).
"""
import numpy as np
import torch
import torch.distributed as dist

KP_BYTES = 28
DESC_BYTES = 32


def shard_indices(n_frames, rank, world):
    """Global frame indices owned by `rank` (round-robin)."""
    return list(range(rank, n_frames, world))


def unshard_order(n_frames, world):
    """Position of global frame i inside the rank-major gathered layout: (rank, local index)."""
    return [(i % world, i // world) for i in range(n_frames)]


def gather_records(counts, kps, desc, group=None):
    """All-gather fixed-capacity per-frame records.

    counts: int32 [b]; kps: uint8 [b, cap, 28]; desc: uint8 [b, cap, 32] (same b, cap on every rank).
    Returns (counts [world, b], kps [world, b, cap, 28], desc [world, b, cap, 32]).
    """
    world = dist.get_world_size(group)
    b, cap = kps.shape[0], kps.shape[1]
    assert kps.dtype == torch.uint8 and desc.dtype == torch.uint8 and counts.dtype == torch.int32
    # one fused payload per rank -> a single collective per batch
    payload = torch.cat([counts.view(torch.uint8).reshape(-1), kps.reshape(-1), desc.reshape(-1)])
    flat = torch.empty(world * payload.numel(), dtype=torch.uint8, device=payload.device)
    dist.all_gather_into_tensor(flat, payload, group=group)
    out = flat.view(world, payload.numel())
    n0 = b * 4
    n1 = n0 + b * cap * KP_BYTES
    g_counts = out[:, :n0].contiguous().view(torch.int32).reshape(world, b)
    g_kps = out[:, n0:n1].reshape(world, b, cap, KP_BYTES)
    g_desc = out[:, n1:].reshape(world, b, cap, DESC_BYTES)
    return g_counts, g_kps, g_desc


def pack_records(counts, kps, desc):
    """One contiguous uint8 payload per rank: counts | keypoints | descriptors (fixed capacity)."""
    assert kps.dtype == torch.uint8 and desc.dtype == torch.uint8 and counts.dtype == torch.int32
    return torch.cat([counts.view(torch.uint8).reshape(-1), kps.reshape(-1), desc.reshape(-1)])


def unpack_records(payload, b, cap):
    n0 = b * 4
    n1 = n0 + b * cap * KP_BYTES
    return (payload[:n0].contiguous().view(torch.int32), payload[n0:n1].reshape(b, cap, KP_BYTES),
            payload[n1:].reshape(b, cap, DESC_BYTES))


def gather_records_to_root(counts, kps, desc, recv=None, dst=0, group=None):
    """The batch-of-frames exchange: ONE gather of every rank's fixed-capacity records to `dst`
    (RCCL: grouped send/recv, each peer over its own xGMI link; no ring, no reduction).
    `recv` (root only) is an optional preallocated list of world payload tensors.  Returns the list on the
    root, None elsewhere."""
    world = dist.get_world_size(group)
    payload = pack_records(counts, kps, desc)
    if dist.get_rank(group) == dst:
        if recv is None:
            recv = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, gather_list=recv, dst=dst, group=group)
        return recv
    dist.gather(payload, gather_list=None, dst=dst, group=group)
    return None


def extract_sharded(frames, extract_fn, cap, group=None, device="cpu"):
    """Shard a global batch, run `extract_fn` on the local frames, gather everything everywhere.

    frames: uint8 array [n, h, w] (identical on every rank).
    extract_fn(local_frames[b, h, w]) -> (counts int32 [b], kps uint8 [b, cap, 28], desc uint8 [b, cap, 32])
    as torch tensors on `device`.  Returns per-global-frame lists (count, kps[count], desc[count]) as numpy.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(frames)
    per = (n + world - 1) // world
    mine = shard_indices(n, rank, world)
    local = np.zeros((per,) + tuple(frames.shape[1:]), dtype=np.uint8)
    for k, i in enumerate(mine):
        local[k] = frames[i]
    counts, kps, desc = extract_fn(local)
    if len(mine) < per:  # padding frames contribute nothing
        counts = counts.clone()
        counts[len(mine):] = 0
    g_counts, g_kps, g_desc = gather_records(counts.to(device), kps.to(device), desc.to(device), group)
    g_counts, g_kps, g_desc = g_counts.cpu().numpy(), g_kps.cpu().numpy(), g_desc.cpu().numpy()
    out = []
    for r, k in unshard_order(n, world):
        c = int(g_counts[r, k])
        out.append((c, g_kps[r, k, :c].copy(), g_desc[r, k, :c].copy()))
    return out


class RecordExchange:
    """The same exchange through the C ABI (include/orbd.h): RCCL called from liborbx.so itself, for consumers without
    torch.distributed (the reference is C++).  The 128-byte unique id comes from rank 0 (RecordExchange.unique_id())
    and reaches the other ranks by any out-of-band means."""

    def __init__(self, rank, world, unique_id, device=-1):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        L = _lib.lib()
        vp, i32 = C.c_void_p, C.c_int
        for name, res, args in (("orbd_create", i32, [i32, i32, vp, i32, C.POINTER(vp)]), ("orbd_destroy", None, [vp]),
                                ("orbd_rank", i32, [vp]), ("orbd_world", i32, [vp]),
                                ("orbd_gather_records", i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
                                ("orbd_allgather_records", i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        self._L = L
        self._h = vp()
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(L.orbd_create(rank, world, idb, device, C.byref(self._h)))
        self.rank, self.world = rank, world

    def world_reported(self):
        """the communicator's size as RCCL reports it (orbd_world)"""
        return int(self._L.orbd_world(self._h))

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        L.orbd_unique_id.restype, L.orbd_unique_id.argtypes = C.c_int, [C.c_void_p]
        buf = (C.c_uint8 * 128)()
        _lib.check(L.orbd_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def shard(n_frames, rank, world):
        """Global indices of the frames rank `rank` owns (orbd_shard_count / orbd_shard_global_index)."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        for name in ("orbd_shard_count", "orbd_shard_global_index"):
            getattr(L, name).restype, getattr(L, name).argtypes = C.c_int, [C.c_int, C.c_int, C.c_int]
        return [L.orbd_shard_global_index(k, rank, world) for k in range(L.orbd_shard_count(n_frames, rank, world))]

    @staticmethod
    def capacity(n_frames, world):
        """Frames every rank passes to gather / allgather (orbd_shard_capacity = ceil(n_frames / world)): the exchanges
        move equal blocks, so a rank whose shard is shorter pads with frames of count 0."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        L.orbd_shard_capacity.restype, L.orbd_shard_capacity.argtypes = C.c_int, [C.c_int, C.c_int]
        return L.orbd_shard_capacity(n_frames, world)

    def gather(self, counts, kps, desc, root=0, stream=None, out=None):
        """counts int32 [b], kps uint8 [b, cap, 28], desc uint8 [b, cap, 32] device tensors -> on the root
        (counts [world, b], kps [world, b, cap, 28], desc [world, b, cap, 32]), None elsewhere.  `out` (root only): a
        preallocated triple to receive into."""
        b, cap = kps.shape[0], kps.shape[1]
        if self.rank == root and out is None:
            out = (torch.empty((self.world, b), dtype=torch.int32, device=kps.device),
                   torch.empty((self.world, b, cap, KP_BYTES), dtype=torch.uint8, device=kps.device),
                   torch.empty((self.world, b, cap, DESC_BYTES), dtype=torch.uint8, device=kps.device))
        st = stream if stream is not None else torch.cuda.current_stream(kps.device).cuda_stream
        p = (lambda t: t.data_ptr()) if out is not None else (lambda t: None)
        self._lib.check(self._L.orbd_gather_records(self._h, root, b, cap, counts.data_ptr(), kps.data_ptr(), desc.data_ptr(),
                                                    p(out[0]) if out else None, p(out[1]) if out else None,
                                                    p(out[2]) if out else None, st))
        return out

    def allgather(self, counts, kps, desc, stream=None):
        b, cap = kps.shape[0], kps.shape[1]
        out = (torch.empty((self.world, b), dtype=torch.int32, device=kps.device),
               torch.empty((self.world, b, cap, KP_BYTES), dtype=torch.uint8, device=kps.device),
               torch.empty((self.world, b, cap, DESC_BYTES), dtype=torch.uint8, device=kps.device))
        st = stream if stream is not None else torch.cuda.current_stream(kps.device).cuda_stream
        self._lib.check(self._L.orbd_allgather_records(self._h, b, cap, counts.data_ptr(), kps.data_ptr(), desc.data_ptr(),
                                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.orbd_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
