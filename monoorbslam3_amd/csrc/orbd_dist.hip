// Batch-of-frames exchange over RCCL (include/orbd.h): grouped ncclSend / ncclRecv of the fixed-capacity records to a
// root rank, or an all-gather.  librccl is resolved with dlopen so that liborbx.so carries no link-time dependency.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <string>

#include "../../include/orbd.h"
#include "../../include/orbx.h"

int orbx_set_error(int code, const std::string &msg);

namespace {
// the few RCCL entry points used, with the signatures of /opt/rocm/include/rccl/rccl.h
typedef struct { char internal[ORBD_ID_BYTES]; } nccl_id;
typedef void *nccl_comm;
enum { NCCL_UINT8 = 1 }; // ncclUint8 (rccl.h: ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1)
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(nccl_id *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*CommCount)(const nccl_comm, int *) = nullptr;
    int (*CommUserRank)(const nccl_comm, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool load()
    {
        if (so) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (so) break;
        }
        if (!so) { err = std::string("librccl not found: ") + dlerror(); return false; }
#define ORBD_SYM(field, name)                                                        \
    *(void **)(&field) = dlsym(so, name);                                            \
    if (!field) { err = std::string("librccl lacks ") + name; so = nullptr; return false; }
        ORBD_SYM(GetUniqueId, "ncclGetUniqueId") ORBD_SYM(CommInitRank, "ncclCommInitRank")
        ORBD_SYM(CommDestroy, "ncclCommDestroy") ORBD_SYM(CommCount, "ncclCommCount") ORBD_SYM(CommUserRank, "ncclCommUserRank")
        ORBD_SYM(GroupStart, "ncclGroupStart") ORBD_SYM(GroupEnd, "ncclGroupEnd")
        ORBD_SYM(Send, "ncclSend") ORBD_SYM(Recv, "ncclRecv") ORBD_SYM(AllGather, "ncclAllGather")
        ORBD_SYM(GetErrorString, "ncclGetErrorString")
#undef ORBD_SYM
        return true;
    }
};
Rccl g_rccl;

int need_device()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    return ORBX_OK;
}
int rccl_fail(const char *what, int rc)
{
    return orbx_set_error(ORBX_E_NO_DEVICE, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
} // namespace

struct orbd_comm {
    int rank, world, device;
    nccl_comm comm;
};

#define D_NCCL(call, what) do { int rc_ = (call); if (rc_ != 0) return rccl_fail(what, rc_); } while (0)
#define D_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return orbx_set_error(ORBX_E_NO_DEVICE, hipGetErrorString(e_)); } while (0)

extern "C" int orbd_unique_id(uint8_t id[ORBD_ID_BYTES])
{
    if (!id) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (int rc = need_device()) return rc;
    if (!g_rccl.load()) return orbx_set_error(ORBX_E_UNSUPPORTED, g_rccl.err);
    nccl_id u;
    D_NCCL(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, ORBD_ID_BYTES);
    return ORBX_OK;
}

extern "C" int orbd_create(int rank, int world, const uint8_t id[ORBD_ID_BYTES], int device, orbd_t **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return orbx_set_error(ORBX_E_ARG, "bad argument");
    *out = nullptr;
    if (int rc = need_device()) return rc;
    if (!g_rccl.load()) return orbx_set_error(ORBX_E_UNSUPPORTED, g_rccl.err);
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    D_HIP(hipSetDevice(device));
    nccl_id u;
    memcpy(u.internal, id, ORBD_ID_BYTES);
    nccl_comm comm = nullptr;
    D_NCCL(g_rccl.CommInitRank(&comm, world, u, rank), "ncclCommInitRank");
    // rank and world are what the COMMUNICATOR reports (ncclCommUserRank / ncclCommCount), not what the caller passed: a short
    // world shows in orbd_world(), and a communicator that disagrees with the arguments is refused here
    int c_world = -1, c_rank = -1;
    int rc = g_rccl.CommCount(comm, &c_world);
    if (rc == 0) rc = g_rccl.CommUserRank(comm, &c_rank);
    if (rc != 0 || c_world != world || c_rank != rank) {
        (void)g_rccl.CommDestroy(comm);
        if (rc != 0) return rccl_fail("ncclCommCount / ncclCommUserRank", rc);
        return orbx_set_error(ORBX_E_NO_DEVICE, "RCCL reports rank " + std::to_string(c_rank) + " of " + std::to_string(c_world) +
                                                    ", orbd_create was given rank " + std::to_string(rank) + " of " + std::to_string(world));
    }
    orbd_comm *c = new orbd_comm{c_rank, c_world, device, comm};
    *out = c;
    return ORBX_OK;
}

extern "C" void orbd_destroy(orbd_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}
// asked of RCCL on every call (ncclCommUserRank / ncclCommCount); -1 / 0 without a handle or when RCCL refuses
extern "C" int orbd_rank(const orbd_t *c)
{
    int r = -1;
    return (c && c->comm && g_rccl.CommUserRank && g_rccl.CommUserRank(c->comm, &r) == 0) ? r : -1;
}
extern "C" int orbd_world(const orbd_t *c)
{
    int n = 0;
    return (c && c->comm && g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) == 0) ? n : 0;
}

extern "C" int orbd_shard_count(int n_frames, int rank, int world)
{
    if (n_frames < 0 || world < 1 || rank < 0 || rank >= world) return 0;
    return (n_frames - rank + world - 1) / world; // frames rank, rank + world, ... below n_frames
}
extern "C" int orbd_shard_global_index(int k, int rank, int world) { return rank + k * world; }
extern "C" int orbd_shard_capacity(int n_frames, int world)
{
    if (n_frames < 0 || world < 1) return 0;
    return (n_frames + world - 1) / world; // = orbd_shard_count of rank 0, the largest shard
}

extern "C" int orbd_gather_records(orbd_t *c, int root, int n_frames, int cap, const int32_t *d_n, const orbx_kp *d_kp,
                                   const uint8_t *d_desc, int32_t *d_n_all, orbx_kp *d_kp_all, uint8_t *d_desc_all,
                                   void *stream)
{
    if (!c || !d_n || !d_kp || !d_desc || n_frames < 0 || cap < 0 || root < 0 || root >= c->world)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (c->rank == root && (!d_n_all || !d_kp_all || !d_desc_all)) return orbx_set_error(ORBX_E_ARG, "the root needs the receive buffers");
    if (n_frames == 0) return ORBX_OK;
    D_HIP(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t bn = (size_t)n_frames * sizeof(int32_t), bk = (size_t)n_frames * cap * sizeof(orbx_kp), bd = (size_t)n_frames * cap * 32;
    if (c->rank == root) {
        // the root's own block is a device copy; every peer's three arrays arrive over that peer's link
        D_HIP(hipMemcpyAsync((uint8_t *)d_n_all + (size_t)root * bn, d_n, bn, hipMemcpyDeviceToDevice, s));
        if (bk) D_HIP(hipMemcpyAsync((uint8_t *)d_kp_all + (size_t)root * bk, d_kp, bk, hipMemcpyDeviceToDevice, s));
        if (bd) D_HIP(hipMemcpyAsync(d_desc_all + (size_t)root * bd, d_desc, bd, hipMemcpyDeviceToDevice, s));
        if (c->world > 1) {
            // inside a group every call is made even after a failure and the group is always closed: an early return
            // between ncclGroupStart and ncclGroupEnd would leave the communicator with an open group
            D_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
            int bad = 0;
            for (int r = 0; r < c->world && !bad; ++r) {
                if (r == root) continue;
                bad |= g_rccl.Recv((uint8_t *)d_n_all + (size_t)r * bn, bn, NCCL_UINT8, r, c->comm, s);
                if (bk && !bad) bad |= g_rccl.Recv((uint8_t *)d_kp_all + (size_t)r * bk, bk, NCCL_UINT8, r, c->comm, s);
                if (bd && !bad) bad |= g_rccl.Recv(d_desc_all + (size_t)r * bd, bd, NCCL_UINT8, r, c->comm, s);
            }
            const int end = g_rccl.GroupEnd();
            if (bad) return rccl_fail("ncclRecv", bad);
            D_NCCL(end, "ncclGroupEnd");
        }
    } else {
        D_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
        int bad = g_rccl.Send(d_n, bn, NCCL_UINT8, root, c->comm, s);
        if (bk && !bad) bad |= g_rccl.Send(d_kp, bk, NCCL_UINT8, root, c->comm, s);
        if (bd && !bad) bad |= g_rccl.Send(d_desc, bd, NCCL_UINT8, root, c->comm, s);
        const int end = g_rccl.GroupEnd();
        if (bad) return rccl_fail("ncclSend", bad);
        D_NCCL(end, "ncclGroupEnd");
    }
    return ORBX_OK;
}

extern "C" int orbd_allgather_records(orbd_t *c, int n_frames, int cap, const int32_t *d_n, const orbx_kp *d_kp,
                                      const uint8_t *d_desc, int32_t *d_n_all, orbx_kp *d_kp_all, uint8_t *d_desc_all,
                                      void *stream)
{
    if (!c || !d_n || !d_kp || !d_desc || !d_n_all || !d_kp_all || !d_desc_all || n_frames < 0 || cap < 0)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (n_frames == 0) return ORBX_OK;
    D_HIP(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t bn = (size_t)n_frames * sizeof(int32_t), bk = (size_t)n_frames * cap * sizeof(orbx_kp), bd = (size_t)n_frames * cap * 32;
    D_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    int bad = g_rccl.AllGather(d_n, d_n_all, bn, NCCL_UINT8, c->comm, s);
    if (bk && !bad) bad |= g_rccl.AllGather(d_kp, d_kp_all, bk, NCCL_UINT8, c->comm, s);
    if (bd && !bad) bad |= g_rccl.AllGather(d_desc, d_desc_all, bd, NCCL_UINT8, c->comm, s);
    const int end = g_rccl.GroupEnd();
    if (bad) return rccl_fail("ncclAllGather", bad);
    D_NCCL(end, "ncclGroupEnd");
    return ORBX_OK;
}
