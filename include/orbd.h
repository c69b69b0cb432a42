/*
 * orbd.h -- C ABI of the batch-of-frames mode: frames sharded over the GPUs of one node, one exchange of the
 * fixed-capacity records per batch over RCCL / xGMI (liborbx.so).
 *
 * The reference has no counterpart: it is one process extracting one frame at a time on the CPU
 * (modules/System.cpp:96 -> modules/BasicObject/Frame.cpp:20).  This is the scale-out row of SURVEY.md section 8(e):
 * frame i of a global batch goes to rank i % world, every rank runs orbx_extract_batch_device on its shard, and
 * the per-frame records { int32 count; orbx_kp[cap]; uint8 desc[cap][32] } -- the layout orbx_extract_batch_device
 * writes -- are gathered to one rank (or to all).  There is no reduction and no ring: every peer sends its block
 * straight to the root over its own xGMI link (grouped ncclSend / ncclRecv), so the step is latency-bound and can run
 * on a side stream under the next batch's kernels.
 *
 * One process per GPU.  librccl is loaded at run time (dlopen "librccl.so.1"): liborbx.so has no link-time
 * dependency on it, and a process that already holds RCCL (PyTorch) shares that copy.
 * Returns 0 or a negative ORBX_E_* code (orbx.h); text in orbx_last_error().
 */
#ifndef ORBD_H
#define ORBD_H

#include <stddef.h>
#include <stdint.h>

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBD_ID_BYTES 128 /* NCCL_UNIQUE_ID_BYTES (rccl.h) */

typedef struct orbd_comm orbd_t;

/* ncclGetUniqueId: call on ONE rank and hand the 128 bytes to the others by any out-of-band means (a file, MPI,
 * a torch.distributed store, a socket) before orbd_create. */
int orbd_unique_id(uint8_t id[ORBD_ID_BYTES]);

/* ncclCommInitRank on `device` (a HIP ordinal; -1 = the current one).  Collective: every rank calls it.  Fails
 * (ORBX_E_NO_DEVICE, the text names both pairs) when the communicator RCCL built reports another rank or size than the
 * arguments (ncclCommUserRank / ncclCommCount). */
int orbd_create(int rank, int world, const uint8_t id[ORBD_ID_BYTES], int device, orbd_t **out);
void orbd_destroy(orbd_t *c);
/* What the communicator itself reports -- ncclCommUserRank / ncclCommCount asked on every call, not the arguments of
 * orbd_create: a short world on a real node shows here.  -1 / 0 for a NULL handle or when RCCL refuses. */
int orbd_rank(const orbd_t *c);
int orbd_world(const orbd_t *c);

/* Round-robin sharding of a global batch: number of frames rank `rank` owns, and the global index of its
 * k-th frame.  Pure host arithmetic. */
int orbd_shard_count(int n_frames, int rank, int world);
int orbd_shard_global_index(int k, int rank, int world);
/* ceil(n_frames / world): the size of the largest shard.  When n_frames % world != 0 the shards are uneven, but the
 * exchanges below move equal blocks: every rank passes n_frames = orbd_shard_capacity(...) records and sets the count of
 * the frames it does not own (k >= orbd_shard_count) to 0 -- d_n[k] = 0 is all the padding a record needs. */
int orbd_shard_capacity(int n_frames, int world);

/* Gather to `root`.  Every rank passes its own records (device pointers: d_n [n_frames], d_kp [n_frames][cap],
 * d_desc [n_frames][cap][32]; n_frames and cap MUST be the same on every rank -- pass orbd_shard_capacity, not
 * orbd_shard_count, when the batch does not divide evenly: mismatched sizes would hang the exchange).  On the root the three *_all buffers
 * (device, world x the local sizes, rank-major) receive them; other ranks may pass NULL.  Enqueued on `stream`
 * (hipStream_t; NULL: orbx.h, "Streams" -- no handle stream here: stream 0 itself); no host synchronisation. */
int orbd_gather_records(orbd_t *c, int root, int n_frames, int cap, const int32_t *d_n, const orbx_kp *d_kp,
                        const uint8_t *d_desc, int32_t *d_n_all, orbx_kp *d_kp_all, uint8_t *d_desc_all, void *stream);
/* The same to every rank (ncclAllGather on each of the three arrays). */
int orbd_allgather_records(orbd_t *c, int n_frames, int cap, const int32_t *d_n, const orbx_kp *d_kp,
                           const uint8_t *d_desc, int32_t *d_n_all, orbx_kp *d_kp_all, uint8_t *d_desc_all, void *stream);

#ifdef __cplusplus
}
#endif
#endif
