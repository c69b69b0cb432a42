/*
 * orbx.h -- C ABI of the MI355X ORB extractor (liborbx.so).
 *
 * Drop-in boundary for the reference's ORBExtractor
 * (modules/ORB/ORBExtractor.h:27-122, modules/ORB/ORBExtractor.cpp:424-638):
 * plain pointers and sizes only, no C++ or torch types.  The header-only shim
 * monoorbslam3_amd/compat/ORBExtractor.h re-creates the reference class on top
 * of these entry points.
 *
 * Every entry point returns 0 on success or a negative ORBX_E_* code; the text
 * of the last failure on the calling thread is in orbx_last_error().
 * There is no CPU fallback: without a HIP device the calls fail with
 * ORBX_E_NO_DEVICE.
 *
 * Streams and threads -- ONE rule for every entry point of orbx.h, orbm.h, orbf.h, orbv.h, orbba.h and orbd.h:
 *  (1) A `*_device` entry point (device pointers, a `void *stream` argument) enqueues on exactly the stream it is given.  The
 *      argument is a hipStream_t, and NULL is the legacy default stream (stream 0) ITSELF, as in the HIP runtime, with or without
 *      a handle.  A caller that fills its device buffers on the null stream (hipMemcpy, kernels on stream 0, torch's default
 *      stream), calls with NULL and reads the results on the null stream needs no synchronisation of its own
 *      (tests/cpp/null_stream.cpp); a caller that works on a NON-BLOCKING stream (hipStreamNonBlocking, any torch.cuda.Stream)
 *      passes that stream.  No device entry point waits on the host (first use and growth of a handle's scratch excepted).  The
 *      call uses the handle's scratch: one device call in flight per handle.
 *  (2) A host-pointer entry point (host arrays in and out, no stream argument) is synchronous: it uploads, computes, downloads
 *      and waits on a stream the library owns -- the handle's own, or one leased for the call by the handle-less orbba_* entry
 *      points and by orbv_transform -- and that stream is NON-BLOCKING: it is ordered neither with stream 0 nor with any other
 *      handle, and steady-state calls neither allocate nor touch the legacy stream.  So two host threads that work through
 *      handles of their own never wait for each other on the device -- the reference's Tracking and LocalMapping threads
 *      (System.cpp:55; LocalMapping.cpp:45-52, 168, 282, 301), tests/cpp/two_threads.cpp.  The one coupling the library adds
 *      itself: a host-pointer call on a handle whose NULL-stream device call may still be in flight first waits for stream 0.
 *  A handle is single-threaded (one call at a time); handles are independent of each other; orbv_transform on ONE shared
 *  vocabulary handle is re-entrant (the reference's vocabulary is a singleton both threads use).  Stream 0 is a process-wide
 *  resource: NULL-stream device calls of two threads are serialised by the runtime as any legacy-stream work is, so a second
 *  thread that wants device entry points passes a stream of its own.  The library's internal side streams are forked from and
 *  joined into the stream of the call with events; the caller sees one in-order stream.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_E_ARG (-1)       /* bad argument / image larger than the handle was created for */
#define ORBX_E_NO_DEVICE (-2) /* no HIP device, or a HIP runtime error */
#define ORBX_E_CAPACITY (-3)  /* caller's output capacity too small (n_out still reports the need) */
#define ORBX_E_UNSUPPORTED (-4)

#define ORBX_MAX_LEVELS 16

/* Same layout as cv::KeyPoint (28 bytes): the shim memcpy's between the two.
 * Replaces the std::vector<cv::KeyPoint>& out-parameter of
 * ORBExtractor::operator() (modules/ORB/ORBExtractor.h:38-39). */
typedef struct orbx_kp {
    float x, y;      /* pt, level-0 pixel coordinates (ORBExtractor.cpp:537-542) */
    float size;      /* scale_factors[octave]            (ORBExtractor.cpp:631)   */
    float angle;     /* degrees in [0,360]               (ORBExtractor.cpp:41)    */
    float response;  /* FAST score                                                */
    int32_t octave;  /* pyramid level                    (ORBExtractor.cpp:630)   */
    int32_t class_id;/* always -1                                                  */
} orbx_kp;

/* Constructor arguments of ORBExtractor(int nFeatures, float scaleFactor,
 * int nLevels, int iniThFast, int minThFast) (modules/ORB/ORBExtractor.h:29-30)
 * plus the device-side sizing the CPU class does not need. */
typedef struct orbx_cfg {
    int32_t n_features;
    float scale_factor;
    int32_t n_levels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t max_width;    /* largest level-0 image the handle must accept */
    int32_t max_height;
    int32_t max_batch;    /* frames resident per call (>= 1) */
    int32_t blur_variant; /* 0: 7-tap set summing to 256 (default), 1: plain-rounded set (sum 257) */
    int32_t device;       /* HIP device ordinal, -1 = current device */
} orbx_cfg;

typedef struct orbx_ctx orbx_t;

/* ORBExtractor::ORBExtractor(...)  (modules/ORB/ORBExtractor.cpp:424-475) */
int orbx_create(const orbx_cfg *cfg, orbx_t **out);
/* ORBExtractor::ORBExtractor(int nFeatures, const ORBExtractor&) (ORBExtractor.cpp:477-493):
 * same pyramid, new per-level quotas (the 2N "initial" extractor, Tracking.cpp:24). */
int orbx_create_requota(const orbx_t *other, int n_features, orbx_t **out);
void orbx_destroy(orbx_t *h);

/* Static scale tables and per-instance quotas
 * (modules/ORB/ORBExtractor.h:44-86, :109-118).  Any pointer may be NULL. */
int orbx_tables(const orbx_t *h, int *n_levels, float *scale_factors, float *inv_scale_factors,
                float *square_sigmas, float *inv_square_sigmas, float *log_scale_factor,
                int32_t *n_features_per_level, int32_t *u_max16);
/* Pyramid level size for a level-0 size (ORBExtractor.cpp:563-564) */
int orbx_level_size(const orbx_t *h, int w0, int h0, int level, int *w, int *ht);
/* Upper bound on keypoints one frame can yield (sum over levels of max(quota+3, 4*nIni)) */
int orbx_max_keypoints(const orbx_t *h, int w0, int h0);

/* ORBExtractor::operator()(image, keyPoints, descriptors)
 * (modules/ORB/ORBExtractor.cpp:495-547).  Host pointers.  img is 8UC1 with
 * `stride` bytes per row.  On success *n_out keypoints and n_out*32 descriptor
 * bytes are written, level-major, in the reference's order.  Empty image or
 * zero keypoints: *n_out = 0 and the outputs are untouched (ORBExtractor.cpp:497,:512). */
int orbx_extract(orbx_t *h, const uint8_t *img, int width, int height, int stride,
                 orbx_kp *out_kp, uint8_t *out_desc, int cap, int *n_out);

/* Batch of equally sized frames, host pointers.  Frame f starts at
 * imgs + f*frame_stride.  Outputs: frame f owns out_kp[f*cap .. ], out_desc[f*cap*32 .. ],
 * n_out[f]. */
int orbx_extract_batch(orbx_t *h, const uint8_t *imgs, int n_frames, int width, int height,
                       int stride, size_t frame_stride, orbx_kp *out_kp, uint8_t *out_desc, int cap,
                       int32_t *n_out);

/* Same, but every pointer is DEVICE memory (inputs already resident in HBM, the
 * bench path) and the work is enqueued on `stream` (NULL: see "Streams" at the top of this header) without a host
 * synchronisation.  There is no host-visible
 * status for a frame that found more key points than `cap` (the host-pointer calls
 * return ORBX_E_CAPACITY): d_n_out[f] then holds the FULL count, > cap, and only the
 * first cap records are written -- a consumer must use min(d_n_out[f], cap), as every
 * device entry point of this library does, or size cap with orbx_max_keypoints(),
 * which no frame can exceed. */
int orbx_extract_batch_device(orbx_t *h, const uint8_t *d_imgs, int n_frames, int width, int height,
                              int stride, size_t frame_stride, orbx_kp *d_out_kp, uint8_t *d_out_desc,
                              int cap, int32_t *d_n_out, void *stream);
/* Scheduling aid for a caller that pipelines other GPU work beside the extraction (no reference counterpart): makes `stream`
 * (hipStream_t) wait until the FAST stage of the most recently enqueued orbx_extract_batch_device call has finished.  FAST
 * saturates the vector ALUs; the quadtree and the orientation that follow are latency-bound and leave them mostly idle, so
 * ALU-heavy work of the caller -- bench.py: the Hamming match of the previous batch -- costs least when it starts there. */
int orbx_stream_wait_fast(orbx_t *h, void *stream);
/* Block until everything enqueued on the handle's stream (host-pointer calls, NULL-stream device calls) has finished. */
int orbx_synchronize(orbx_t *h);
/* Optional, for a caller whose frames live in a long-lived buffer (a capture ring, a cv::Mat that is reused): page-lock that
 * buffer once (hipHostRegister) so that orbx_extract's copy of the frame is a plain DMA that does not block the calling thread
 * -- the launches that follow are then issued while the frame is still on its way (1920x1080: the 52 us in which a pageable
 * frame blocks the host).  The range must stay allocated until orbx_host_unregister; registering is slow (milliseconds): once per
 * buffer, never per frame.  No reference counterpart. */
int orbx_host_register(void *ptr, size_t bytes);
int orbx_host_unregister(void *ptr);

/* ---- stage taps: copy intermediate results of the LAST extract call to host
 * memory (parity tests compare every stage with the oracle). ---- */
/* pyramid level (blurred = 0) or its 7x7 Gaussian-blurred copy (blurred = 1), tightly packed w*h */
int orbx_tap_level(orbx_t *h, int frame, int level, int blurred, uint8_t *out, size_t out_bytes);
/* FAST candidates of one level, unordered: x,y relative to (19,19), response.  Returns count via n_out. */
int orbx_tap_candidates(orbx_t *h, int frame, int level, uint16_t *xs, uint16_t *ys, uint8_t *resp,
                        int cap, int *n_out);
/* per-level keypoint counts after the quadtree */
int orbx_tap_level_counts(orbx_t *h, int frame, int32_t *counts);

/* the device's evaluation of `cos(angle)`, `sin(angle)` of computeOrbDescriptor (modules/ORB/ORBExtractor.cpp:53-54,
 * i.e. glibc cosf / sinf of angle_deg * (float)(CV_PI / 180.f)) on n host angles; cos_sin receives n (cos, sin) pairs */
int orbx_tap_sincos(orbx_t *h, const float *angles_deg, int n, float *cos_sin);

/* ---- kernel-choice switches (no reference counterpart) ----
 * Every stage has one default kernel per call size and at least one parity twin that produces the same bytes; these
 * select them per handle (tests run every twin, tools measure them).  They replace the environment variables earlier
 * builds read: an exported variable in a user's shell can no longer change which kernel Tracking runs.  A new value
 * applies from the next extract call; an unknown switch or a value outside its range is ORBX_E_ARG. */
#define ORBX_VAR_FAST 0         /* 0 by call size (default: strips of cells from 24 frames), 1 one wave per cell, 2 strips */
#define ORBX_VAR_BLUR 1         /* 1 by call size (default: matrix pipe from 8 frames, levels >= 160 px), 0 VALU kernels, 2 matrix pipe */
#define ORBX_VAR_RESIZE_LDS 2   /* 1 by call size (default: source tile through LDS for resident batches), 0 never, 2 always */
#define ORBX_VAR_RESIZE2 3      /* 1 by call size (default: two levels per launch below 24 frames), 0 never, 2 always */
#define ORBX_VAR_SIDE_BLUR 4    /* blur pass on a side stream: 1 beside FAST (default), 2 beside the quadtree, 3 beside the orientation, 0 in line */
#define ORBX_VAR_EARLY_FAST 5   /* level 0's FAST beside the pyramid: -1 by pyramid kernel (default), 0 no, 1 yes, 2 and its blur */
#define ORBX_VAR_SPLIT_LEVEL0 6 /* synchronous calls with a few frames: first level of the main chain, 1 .. n_levels - 1 (default 1), 0 = one chain */
#define ORBX_VAR_STREAMS 7      /* frame ranges of a batch on 1..8 internal streams (default 1) */
#define ORBX_VAR_ZERO_COPY 8    /* 1 small calls write their records into pinned host memory (default), 0 copy them back */
#define ORBX_VAR_DESC 9         /* 0 by call size (default), 1 separate blur pass + k_orient_desc, 2 k_blur_desc (blur and
                                 * descriptors in one pass over the raw level, no blurred level in memory) */
#define ORBX_VAR_FAST_CELL_GROUP 10 /* calls with a few frames: FAST cells (one wave each) per workgroup -- the group reserves its
                                     * place in the candidate list with one atomic: 1, 4, 8 (default) or 16 */
#define ORBX_N_VARIANTS 11
int orbx_set_variant(orbx_t *h, int which, int value);
int orbx_get_variant(const orbx_t *h, int which, int *value);

/* ---- per-kernel timing (HIP events on the handle's stream) ---- */
#define ORBX_STAGE_RESIZE 0
#define ORBX_STAGE_FAST 1
#define ORBX_STAGE_BLUR 2   /* the separate blur pass (zero when k_blur_desc describes every level) */
#define ORBX_STAGE_OCTREE 3
#define ORBX_STAGE_ORIENT 4 /* k_orient + k_angle */
#define ORBX_STAGE_DESC 5   /* k_desc_bins + k_blur_desc and / or k_orient_desc */
#define ORBX_N_STAGES 6
/* enable = 1: every later extract call records HIP events around each stage, with every kernel on ONE stream (isolated
 * stage times).  enable = 2: the call keeps its streams (FAST, blur and the caller's other work overlap as in production) and
 * events on each stage's own launch stream bracket its launches: orbx_stage_times_in_step_ms then gives every stage's time as it
 * runs beside the others.  Mode 2 times calls that run as ONE frame range (ORBX_VAR_STREAMS = 1, else ORBX_E_UNSUPPORTED from
 * the query below); a synchronous call with a few frames that takes the split order (ORBX_VAR_SPLIT_LEVEL0) brackets only its FAST,
 * blur, orientation and descriptor launches -- resize and quadtree read 0 there.  The orientation bracket includes k_desc_bins;
 * the descriptor bracket opens behind the wait for a side-stream blur. */
int orbx_set_stage_timing(orbx_t *h, int enable);
/* timing mode 2: per stage, the sum over the last extract call's launch groups of that stage (FAST: one or two, level 0 may
 * start early on the side stream) of the time between the events around each group.  Synchronises on those events. */
int orbx_stage_times_in_step_ms(orbx_t *h, float *ms /* ORBX_N_STAGES */);
/* timing mode 2, FAST only (kept for callers of the earlier interface): sum and number of its launch groups */
int orbx_fast_times_in_step_ms(orbx_t *h, float *ms_sum, int *n_launches);
/* timing mode 1: milliseconds each stage took in the last extract call (synchronises) */
int orbx_stage_times_ms(orbx_t *h, float *ms /* ORBX_N_STAGES */);

const char *orbx_last_error(void);
const char *orbx_version(void);

#ifdef __cplusplus
}
#endif
#endif
