#!/usr/bin/env python3
"""bench.py -- ORB extract+match frames/s at 1242x375 / 2000 features on N MI355X.

One "step" = one pass of the hot path over one resident batch of synthetic frames per GPU:
pyramid -> FAST cells -> blur -> quadtree -> orientation + rBRIEF for every frame, then the
256-bit Hamming best-2 brute force (the SearchByBow inner loop) of every frame's descriptors
against its neighbour's.  Inputs live in HBM before the timed region.  With N > 1 each rank
processes its own batch (weak scaling) and every step issues one RCCL gather of the fixed-capacity
keypoint/descriptor records to rank 0 on its own stream (monoorbslam3_amd/dist.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (HIP-event timing on the launch stream) and `cpu_baseline` (the C oracle
timed on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A step uses four streams of its own (extraction, the extractor's side stream, match, and with N > 1 the gather) and RCCL
# adds its own; the runtime's default of four hardware queues makes streams beyond that share a queue and serialise
# (tools/overlap_sweep.sh).  Eight queues cost nothing at N = 1 (103.3 k vs 103.6 k frames/s) and keep the streams apart.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue rates measured on MI355X (tools/microbench/valu_ops2.hip, valu_ops3.hip -> profiles/r02_valu_ops2.txt,
# r03_valu_ops3.txt), G wave-instructions/s over the chip: the plain 32-bit and / or / xor / add / sub / lshr / mov /
# v_bitop3 class against everything else (VOP3, packed, SDWA, DPP, min / max, compares, shifts left, scalar operands).
# A kernel's issue floor is the sum over its instructions of 1 / rate(class): tools/isa_mix.py does that for FAST
# (profiles/fast_mix.json); one number for every stage would be wrong by up to 1.7 x either way.
VALU_RATE_CHEAP_GWINST = 850.0
VALU_RATE_SLOW_GWINST = 540.0
# dense FP4 matrix rate: v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 4) issues every 32 cycles per SIMD (measured,
# tools/microbench/fp4_hamming.hip) = 2048 multiply-adds = 4096 operations per cycle per SIMD; x 1024 SIMDs x 2.4 GHz =
# 10.07 POP/s at the nominal clock (MI355X_MICROARCH.md: ~10 PF dense FP4)
MFMA_FP4_PEAK_TOPS = 10066.0


def baseline_metric():
    """BASELINE.json's metric string, verbatim (falls back to the same wording when the file is not shipped)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "ORB extract+match frames/s @1242\u00d7375, 2000 feat; 1/2/4/8 GPU + %HBM roofline"


def level_bytes(ex, w, h):
    sizes = [ex.level_size(w, h, l) for l in range(ex.n_levels)]
    return [a * b for a, b in sizes]


def algorithmic_bytes(ex, w, h, kp_per_frame, one_pass):
    """Per-frame compulsory HBM bytes of each stage.  one_pass: the blur and the descriptors run as
    k_blur_desc (the raw pyramid read once, nothing blurred written); else the blur pass (read P, write P) + k_orient_desc."""
    lb = level_bytes(ex, w, h)
    P, L0, Llast = sum(lb), lb[0], lb[-1]
    return {
        "resize": (P - Llast) + (P - L0),
        "fast": P,
        "blur": 0 if one_pass else 2 * P,
        "octree": 0,
        "orient": min(kp_per_frame * 961, P),
        "desc": (P if one_pass else min(kp_per_frame * 1369, P)) + kp_per_frame * 60,
    }, P


def cpu_baseline(frames, n_features, threads, match):
    """the C oracle (restatement of the reference's CPU extractor and best/second-best loop) on the
    host cores, doing the same work as a GPU step per frame -- extract, then match against the predecessor.
    (i) one thread, 20 warm-up + 200 timed frames, per-frame median / p10 / p90; (ii) all `threads` threads, one frame
    stream per thread.  Built -O3 -march=native on this box when gcc is present (oracle/Makefile `native`), else the
    portable -O2 build."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orb_ref_py
    build = orb_ref_py.build_native() if hasattr(orb_ref_py, "build_native") else "-O2"
    orb_ref_py.build()

    def run(orc, idx, warm):
        lat, prev = [], None
        for k, i in enumerate(idx):
            t0 = time.perf_counter()
            _, desc, _ = orc.extract(frames[i % len(frames)])
            if match and prev is not None:
                orb_ref_py.best2(prev, desc)
            prev = desc
            if k >= warm:
                lat.append(time.perf_counter() - t0)
        return lat

    def pct(lat):
        a = np.sort(np.asarray(lat)) * 1e3
        return {"ms_median": round(float(np.median(a)), 2), "ms_p10": round(float(np.percentile(a, 10)), 2),
                "ms_p90": round(float(np.percentile(a, 90)), 2), "frames": len(a)}

    one = run(orb_ref_py.Oracle(n_features, 1.2, 8, 20, 7), range(220), 20)
    one_stats = pct(one)
    one_stats["frames_per_s"] = round(len(one) / sum(one), 2)
    per = max(8, int(4.0 / max(np.median(one), 1e-3)))  # about 4 s of wall time on every thread
    orcs = [orb_ref_py.Oracle(n_features, 1.2, 8, 20, 7) for _ in range(threads)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:  # ctypes releases the GIL during the C call
        lats = list(pool.map(lambda t: run(orcs[t], range(t * per, t * per + per + 2), 2), range(threads)))
    wall = time.perf_counter() - t0
    allc = pct([x for lat in lats for x in lat])
    done = sum(len(lat) for lat in lats)
    allc["frames_per_s"] = round(done / max(sum(lat) for lat in lats), 2)  # timed frames over the slowest thread's timed span
    return {"value": allc["frames_per_s"], "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "C oracle (restatement of the reference's CPU code, not the OpenCV-backed original), %s; per frame: "
                      "extraction%s; one thread: 20 warm-up + %d timed frames; %d threads: %d timed frames each, %.1f s wall"
                      % (build, " + best/second-best match against the previous frame" if match else "", len(one),
                         threads, per, wall),
            "one_thread": one_stats, "all_threads": allc}


def self_launch(n):
    """Start `n` ranks of this script under torch.distributed.run on this node (one per GPU, rendezvous on 127.0.0.1 and
    a free port), pass their output through and return the launcher's exit status.  Called before any GPU call."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    return subprocess.call(cmd, env=env)


def stub_run(args, rank, world):
    """`--stub-step`: the contract's control flow (warm-up, barrier + timing of exactly K steps, MAX over ranks, one JSON
    line on rank 0) with the step replaced by a sleep, over gloo -- what a CPU test can check of the N > 1 launcher path."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        time.sleep(1e-3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank))  # (rank r sleeps r + 1 ms per step: the per-rank times of `scaling_diag` must tell them apart)
    t_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    diag = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the same keys as the real run's `scaling_diag` (per-rank time before the barrier, gather time, world as the backend reports it)
        mine = torch.tensor([t_local / args.steps * 1e3], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(x.item()), 4) for x in every]
        diag = {"ms_per_step_per_rank": per_rank, "ms_per_step_rank_min": min(per_rank), "ms_per_step_rank_max": max(per_rank),
                "gather_ms_mean": None, "gather_ms_max": None, "gather_timed_by": "no gather in the stub step",
                "world": dist.get_world_size(), "world_c_abi": None, "backend": "gloo", "gather": None}
    if rank == 0:
        print(json.dumps({"metric": baseline_metric(), "value": round(world * (args.batch or 1) * args.steps / dt, 2), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "scaling_diag": diag, "vs_baseline": None, "dtype": "u8", "data": "stub",
                          "config": {"workload": "launcher rehearsal: the step is a 1 ms sleep, nothing is measured"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# VALU popcount peak for the 256-bit Hamming distance ("fraction of VALU popcount peak"): a pair costs 8 v_xor_b32 +
# 8 v_bcnt_u32_b32 (the count accumulates in the instruction's second operand, so there is no add tree), 64 pairs per
# wave-instruction; v_xor issues in the cheap class (850 G wave-instr/s), v_bcnt_u32_b32 at 568.4 G wave-instr/s
# (tools/microbench/valu_ops2.hip -> profiles/r02_valu_ops2.txt).  The key update (shift-or, med3, min) is not in the peak.
VALU_BCNT_GWINST = 568.4
VALU_POPCOUNT_PEAK_GPAIRS = 64.0 / (8.0 / VALU_RATE_CHEAP_GWINST + 8.0 / VALU_BCNT_GWINST)


def bench_config3(dev, device_index):
    """BASELINE config 3 (ORBmatcher::SearchByBoW Hamming BF, 2000 x 2000 256-bit descriptors): the dense best / second-best
    search on the FP4 matrix path and on its VALU twin -- one problem (latency) and 256 problems in one launch (rate) --, and
    the whole SearchByBow on device-resident records with every feature in one vocabulary node.  Device time by HIP events on the
    stream the calls are enqueued on, median of 15."""
    from monoorbslam3_amd import synth, _lib
    from monoorbslam3_amd.extractor import KP_DTYPE
    from monoorbslam3_amd.matcher import MatcherHandle, ORBMatcher, _mlib
    n, P = 2000, 256
    a, b, _ = synth.make_descriptor_pair(n, seed=1)
    ML = _mlib()
    st = torch.cuda.Stream(device=dev)
    d_a = torch.from_numpy(a).to(dev).unsqueeze(0).repeat(P, 1, 1).contiguous()
    d_b = torch.from_numpy(b).to(dev).unsqueeze(0).repeat(P, 1, 1).contiguous()
    d_na = torch.full((P,), n, dtype=torch.int32, device=dev)
    d_bi = torch.zeros((P, n), dtype=torch.int32, device=dev)
    d_bd = torch.zeros((P, n), dtype=torch.int16, device=dev)
    d_sd = torch.zeros((P, n), dtype=torch.int16, device=dev)

    def timed(fn, reps=15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for r in range(reps + 2):
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                e0.record(st)
                fn()
                e1.record(st)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    out = {"descriptors": "%d x %d, 256 bit" % (n, n), "timed_by": "HIP events on the calls' stream, median of 15"}
    ref = None
    for variant in ("fp4", "valu"):
        mh = MatcherHandle(device=device_index)
        mh.set_variant("best2", variant)
        call = lambda np_: _lib.check(ML.orbm_best2_device(mh._h, np_, d_a.data_ptr(), n, d_na.data_ptr(), n, d_b.data_ptr(), n,  # noqa: E731
                                                           d_na.data_ptr(), n, None, None, d_bi.data_ptr(), d_bd.data_ptr(),
                                                           d_sd.data_ptr(), st.cuda_stream))
        ms1 = timed(lambda: call(1))
        msP = timed(lambda: call(P))
        got = (d_bi[0].cpu().numpy().copy(), d_bd[0].cpu().numpy().copy(), d_sd[0].cpu().numpy().copy())
        if ref is None:
            ref = got
        same = all(np.array_equal(x, y) for x, y in zip(ref, got))
        assert same, "the VALU and the FP4 best-2 kernels disagree"
        pairs = float(n) * n
        r = {"one_problem_ms": round(ms1, 4), "one_problem_gpairs_per_s": round(pairs / (ms1 * 1e-3) / 1e9, 1),
             "problems_per_launch": P, "launch_ms": round(msP, 4), "gpairs_per_s": round(P * pairs / (msP * 1e-3) / 1e9, 1)}
        if variant == "fp4":
            tops = P * pairs * 512 / (msP * 1e-3) / 1e12
            r.update({"kernel": "k_best2_fp4 (v_mfma_f32_32x32x64_f8f6f4)", "achieved_TOPS": round(tops, 1), "peak_TOPS": MFMA_FP4_PEAK_TOPS,
                      "frac_of_dense_fp4_peak": round(tops / MFMA_FP4_PEAK_TOPS, 3)})
        else:
            r.update({"kernel": "k_best2 (v_xor_b32 + v_bcnt_u32_b32, the parity twin)", "peak_gpairs_per_s": round(VALU_POPCOUNT_PEAK_GPAIRS, 1),
                      "frac_of_valu_popcount_peak": round(P * pairs / (msP * 1e-3) / 1e9 / VALU_POPCOUNT_PEAK_GPAIRS, 3),
                      "peak_note": "64 pairs per (8 v_xor at 850 + 8 v_bcnt at 568.4 G wave-instr/s), profiles/r02_valu_ops2.txt; "
                                   "the kernel also issues a shift-or, a med3 and a min per pair for the two running keys",
                      "equal_to_fp4_outputs": same})
        out["best2_" + variant] = r
    # the whole SearchByBow (node join, top-8 lists, greedy resolve, rotation histogram) with all features in ONE node
    rng = np.random.RandomState(0)
    k1, k2 = np.zeros(n, KP_DTYPE), np.zeros(n, KP_DTYPE)
    k1["angle"], k2["angle"] = rng.uniform(0, 360, n).astype(np.float32), rng.uniform(0, 360, n).astype(np.float32)
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    kp = lambda k: torch.from_numpy(np.frombuffer(k.tobytes(), np.uint8).copy()).to(dev)  # noqa: E731

    def dev_fv(fv):
        nodes, off, idx = fv
        pad = lambda x, dt, m: torch.from_numpy(np.concatenate([np.asarray(x, dt), np.zeros(max(m - len(x), 0), dt)])).to(dev)  # noqa: E731
        return (pad(nodes, np.uint32, n).view(torch.int32), pad(off, np.int32, n + 1), pad(idx, np.uint32, n).view(torch.int32),
                torch.tensor([len(nodes)], dtype=torch.int32, device=dev))
    d = dict(desc1=up(a), kps1=kp(k1), kf_mp_ok=up(np.ones(n, np.uint8)), fv1=dev_fv(synth.feature_vector_by_prefix(a, 0)), desc2=up(b),
             kps2=kp(k2), frame_mp=torch.full((n,), -1, dtype=torch.int32, device=dev), fv2=dev_fv(synth.feature_vector_by_prefix(b, 0)),
             result=torch.zeros(8, dtype=torch.int32, device=dev))
    m = ORBMatcher(0.7, True, handle=MatcherHandle(device=device_index))

    def bow():
        d["frame_mp"].fill_(-1)
        m.SearchByBowDevice(d, n, n, stream=st.cuda_stream)
    ms = timed(bow)
    res = d["result"].cpu().numpy()
    out["search_by_bow_device"] = {"ms": round(ms, 4), "matches": int(res[0]), "sweeps": int(res[2]),
                                   "gpairs_per_s": round(float(n) * n / (ms * 1e-3) / 1e9, 1),
                                   "note": "orbm_search_by_bow_device, one vocabulary node holding all 2000 x 2000 pairs "
                                           "(the clearing of frame_mp included); ORBMatcher.cpp:118-201"}
    return out


def bench_config5(device_index):
    """BASELINE config 5 (local BA, 20 key frames x 3000 map points): one linearisation -- per-edge reprojection residual and
    Jacobians (k_ba_edges) + the block J^T W J reductions (k_ba_reduce_pose / k_ba_reduce_point) -- by the HIP events of
    orbba_linearize, against SURVEY 8(d)'s byte count; and the whole Optimize::localBundleAdjustment loop."""
    from monoorbslam3_amd import ba, synth
    pr = synth.make_ba_problem(20, 3000)
    args5 = (pr["cam"], pr["pose_R"], pr["pose_t"], pr["pose_fixed"], pr["points"], pr["edge_pose"], pr["edge_point"], pr["edge_z"],
             pr["edge_inv_sigma2"])
    ne, n_p, n_l = len(pr["edge_pose"]), 20, 3000
    ks, ws = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        g = ba.linearize(*args5)
        ws.append((time.perf_counter() - t0) * 1e3)
        ks.append(g["kernel_ms"])
    k_ms = sorted(ks[2:])[len(ks[2:]) // 2]
    # per edge read 8 (indices) + 16 (z) + 8 (inv sigma^2) + gathers 96 (pose) + 24 (point), write H_pl 144;
    # reductions 20 x (36 + 6) x 8 + 3000 x (9 + 3) x 8
    alg = ne * (8 + 16 + 8 + 96 + 24 + 144) + n_p * 42 * 8 + n_l * 12 * 8
    gbs = alg / (k_ms * 1e-3) / 1e9
    lb = []
    for _ in range(4):
        t0 = time.perf_counter()
        got = ba.local_bundle_adjustment(*args5)
        lb.append(((time.perf_counter() - t0) * 1e3, got["device_ms"]))
    lb = sorted(lb[1:])[1]
    return {"poses": n_p, "points": n_l, "edges": int(ne),
            "linearize_kernels_ms": round(k_ms, 4), "linearize_call_ms": round(sorted(ws[2:])[len(ws[2:]) // 2], 3),
            "algorithmic_bytes_per_launch": int(alg), "achieved_GBps": round(gbs, 1), "peak_GBps": HBM_PEAK_GBS,
            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "dtype": "f64",
            "bound_note": "three dependent launches of %d / %d / %d workgroups of 256 threads on 256 CUs: launch- and latency-bound, the "
                          "machine is never full" % ((ne + 255) // 256, n_p, (n_l + 255) // 256),
            "local_bundle_adjustment_call_ms": round(lb[0], 3), "local_bundle_adjustment_device_ms": round(lb[1], 3),
            "timed_by": "kernels: HIP events inside orbba_linearize on the call's stream (median of 5); calls: host clock, host arrays in and out"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="resident frames per GPU per step (default 512 for config 2: "
                                                            "throughput saturates there; 1 for config 4)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="BASELINE.json configuration: 2 = KITTI 1242x375 / 2000 features, a resident batch per GPU (the "
                         "metric's configuration, default); 4 = 1920x1080 stream, ONE frame per GPU per step (8 concurrent "
                         "frames sharded 1/GPU at --gpus 8), gathered to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive measurement")
    ap.add_argument("--no-match", action="store_true", help="time extraction only")
    ap.add_argument("--clock-warmup-steps", type=int, default=150,
                    help="untimed steps run before the --warmup steps so that the device's clocks are at their sustained level (0 = none)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the extra (untimed) keys config3 (2000 x 2000 Hamming best-2 / SearchByBow) and config5 (local BA 20 x 3000)")
    ap.add_argument("--no-density-sweep", action="store_true",
                    help="skip the extra (untimed) extraction runs on frames with about a tenth and a third of the headline workload's corner density")
    ap.add_argument("--match-placement", default=None, choices=["after-fast", "eager"],
                    help="when the match of a batch starts: behind the next batch's FAST stage (orbx_stream_wait_fast), beside its "
                         "latency-bound quadtree / k_desc_bins / orientation (default; with --best2-resident 1: 2.18 ms per step, "
                         "two kernels in flight 33-38 %% of the step), or as soon as its own batch is extracted, beside the next "
                         "batch's pyramid (eager: 2.33 ms; profiles/r05_overlap.md).  One frame per step (config 4) always takes eager: "
                         "its match reads the previous step's records, which the next extraction overwrites")
    ap.add_argument("--records", action="store_true",
                    help="also build complete frame records per step: undistortion + grid (orbf) and bag of words on a "
                         "synthetic ORBvoc-sized vocabulary (orbv); not the headline configuration")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --share-device rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank computes on cuda:0")
    ap.add_argument("--stub-step", action="store_true",
                    help="launcher rehearsal without a GPU: every rank replaces the step by a 1 ms sleep and runs the barrier / "
                         "max-over-ranks timing over gloo (tests/test_dist_cpu.py drives `bench.py --gpus 2 --stub-step`)")
    ap.add_argument("--variant", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-choice switch of the extractor handle (orbx_set_variant; names in monoorbslam3_amd/extractor.py "
                         "VARIANTS, e.g. --variant side_blur=2 --variant desc=separate); repeatable")
    ap.add_argument("--best2", default="fp4", choices=["fp4", "i8", "valu"], help="dense best / second-best kernel (orbm_set_variant)")
    ap.add_argument("--best2-resident", type=int, default=None, choices=[0, 1, 2],
                    help="k_best2_fp4 as this many workgroups per CU walking the query blocks (ORBM_VAR_BEST2_RESIDENT): the match then "
                         "holds a fixed share of every CU -- half the registers, 37 KB of LDS at 1 -- and the extraction's kernels "
                         "beside it always find room; 0 = one workgroup per block of queries.  Default: 1 with --match-placement "
                         "after-fast, 0 with eager")
    ap.add_argument("--gather", default="torch", choices=["torch", "c-abi"],
                    help="N > 1: the record gather through torch.distributed (default) or through the library's own RCCL "
                         "entry point orbd_gather_records (include/orbd.h)")
    args = ap.parse_args()
    if args.config == 4:
        dw, dh, db = 1920, 1080, 1
    else:
        dw, dh, db = 1242, 375, 512
    args.width = args.width or dw
    args.height = args.height or dh
    args.batch = args.batch or db
    if args.match_placement is None or args.batch == 1:
        args.match_placement = "after-fast" if args.batch > 1 else "eager"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a plain shell: start the N ranks ourselves.  Nothing in this process has touched
        # the GPU yet (importing torch does not initialise it), and the ranks are CHILD processes -- a process that has
        # initialised the GPU is never replaced by another program.
        sys.exit(self_launch(args.gpus))
    if args.gpus != world:
        sys.exit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or from a plain shell"
                 % (args.gpus, world, args.gpus))
    if args.stub_step:
        return stub_run(args, rank, world)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")  # gloo rehearsal: collectives on host copies
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from monoorbslam3_amd import synth
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.matcher import MatcherHandle, _mlib
    from monoorbslam3_amd import _lib
    from monoorbslam3_amd.dist import gather_records_to_root, pack_records

    W, H, B, NF = args.width, args.height, args.batch, args.features
    # ---- synthetic resident batch: a few dozen distinct frames, replicated with per-copy noise
    n_distinct = min(B, 32)
    base = synth.make_frames(n_distinct, W, H, seed=synth.DEFAULT_SEED + 101 * rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    frames = torch.from_numpy(base).to(dev)
    if B > n_distinct:
        reps = (B + n_distinct - 1) // n_distinct
        frames = frames.repeat(reps, 1, 1)[:B].contiguous()
        noise = torch.randint(-2, 3, frames.shape, generator=g, dtype=torch.int16).to(dev)
        noise[:n_distinct] = 0
        frames = (frames.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8).contiguous()
    variants = {}
    for kv in args.variant:
        k, _, v = kv.partition("=")
        variants[k] = int(v) if v.lstrip("-").isdigit() else v
    ex = ORBExtractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local_rank, variants=variants)
    cap = ex.max_keypoints(W, H)
    # two output sets: the best-2 match of batch k runs on its own stream while batch k+1 is being extracted
    NBUF = 2
    d_kp = [torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    d_desc = [torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    d_n = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(NBUF)]
    d_bidx = [torch.zeros((B, cap), dtype=torch.int32, device=dev) for _ in range(NBUF)]
    d_bd = [torch.zeros((B, cap), dtype=torch.int16, device=dev) for _ in range(NBUF)]
    d_sd = [torch.zeros((B, cap), dtype=torch.int16, device=dev) for _ in range(NBUF)]
    mh = MatcherHandle(device=local_rank)
    mh.set_variant("best2", args.best2)
    if args.best2_resident is None:
        args.best2_resident = 1 if (args.match_placement == "after-fast" and args.best2 == "fp4" and not args.no_match) else 0
    mh.set_variant("best2_resident", args.best2_resident)
    ML = _mlib()
    rec = None
    if args.records:  # SURVEY 8f rows 2 and 3 chained behind the extraction, all on device buffers
        from monoorbslam3_amd.frame import FramePost
        from monoorbslam3_amd.vocabulary import ORBVocabulary
        fpost = FramePost(W, H, 718.856, 718.856, W / 2.0, H / 2.0, dist=(-0.2834, 0.0739, 1.9e-4, 1.8e-5), device=local_rank)
        voc = ORBVocabulary.from_arrays(synth.make_vocabulary(10, 6, seed=1, p_early_leaf=0.0, p_stop=0.0), device=local_rank)
        z = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=dev) for _ in range(NBUF)]  # noqa: E731
        rec = dict(fpost=fpost, voc=voc, kp_un=z((B, cap, 28), torch.uint8), cell_start=z((B, fpost.n_cells + 1), torch.int32),
                   cell_items=z((B, cap), torch.int32), bow_ids=z((B, cap), torch.int32), bow_vals=z((B, cap), torch.float64),
                   n_words=z((B,), torch.int32), fv_nodes=z((B, cap), torch.int32), fv_off=z((B, cap + 1), torch.int32),
                   fv_idx=z((B, cap), torch.int32), n_fv=z((B,), torch.int32))
    # dedicated non-blocking streams: the extraction and the match overlap only on streams of their own (a NULL stream is the
    # legacy stream 0 itself for every device entry point, include/orbx.h "Streams"), and the RCCL gather below must be
    # ordered behind the kernels it depends on
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    mstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0 and mstream.cuda_stream != 0
    ev_extracted = [torch.cuda.Event() for _ in range(NBUF)]
    ev_matched = [torch.cuda.Event() for _ in range(NBUF)]
    ev_gathered = [torch.cuda.Event() for _ in range(NBUF)]
    step_no = [0]
    # N > 1: one gather of the fixed-capacity records to rank 0 per step, on its own stream so that the xGMI
    # transfer of batch k overlaps the extraction of batch k+1 (everything is drained before the clock stops)
    cstream = torch.cuda.Stream(device=dev) if world > 1 else None
    xc, xc_out = None, None
    if world > 1 and args.gather == "c-abi" and args.backend == "nccl":
        from monoorbslam3_amd.dist import RecordExchange
        uid = [RecordExchange.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        xc = RecordExchange(rank, world, uid[0], device=local_rank)
        if rank == 0:
            xc_out = [(torch.empty((world, B), dtype=torch.int32, device=dev),
                       torch.empty((world, B, cap, 28), dtype=torch.uint8, device=dev),
                       torch.empty((world, B, cap, 32), dtype=torch.uint8, device=dev)) for _ in range(NBUF)]
    recv = None
    if world > 1 and rank == 0 and xc is None:
        nbytes = pack_records(d_n[0], d_kp[0], d_desc[0]).numel()
        recv = [[torch.empty(nbytes, dtype=torch.uint8, device=coll_dev) for _ in range(world)] for _ in range(NBUF)]

    def match(i, st):
        # frame f against frame f+1 (B-1 problems), and the last frame against frame 0
        if B > 1:
            _lib.check(ML.orbm_best2_device(mh._h, B - 1, d_desc[i].data_ptr(), cap, d_n[i].data_ptr(), cap,
                                            d_desc[i].data_ptr() + cap * 32, cap, d_n[i].data_ptr() + 4, cap, None, None,
                                            d_bidx[i].data_ptr(), d_bd[i].data_ptr(), d_sd[i].data_ptr(), st))
        # the batch's last frame: against frame 0, or (one frame per step, config 4) against the previous step's frame
        o = i if B > 1 else (i + 1) % NBUF
        _lib.check(ML.orbm_best2_device(mh._h, 1, d_desc[i].data_ptr() + (B - 1) * cap * 32, cap,
                                        d_n[i].data_ptr() + 4 * (B - 1), cap, d_desc[o].data_ptr(), cap, d_n[o].data_ptr(),
                                        cap, None, None, d_bidx[i].data_ptr() + 4 * (B - 1) * cap,
                                        d_bd[i].data_ptr() + 2 * (B - 1) * cap, d_sd[i].data_ptr() + 2 * (B - 1) * cap, st))

    # The match of batch k is ALU- and matrix-pipe-bound; the quadtree, k_desc_bins and the orientation of an extraction are
    # latency-bound and leave those units mostly idle.  --match-placement after-fast (default) therefore starts the match not when
    # its batch is extracted (beside the next batch's pyramid: both stretch) but behind the FAST stage of the NEXT batch
    # (orbx_stream_wait_fast), and as a grid of ONE workgroup per CU (ORBM_VAR_BEST2_RESIDENT): it then holds half of every CU's
    # registers and 37 KB of its LDS for as long as it runs, and the quadtree (two of its four workgroups per CU), k_desc_bins and
    # k_orient (four of its eight waves per SIMD) run beside it.  A step is still one extraction + one match, the match belonging
    # to the batch before; flush() -- inside the timed region -- runs the last one.  Round 4 measured this placement with the
    # match at full occupancy (no gain: whichever kernel came first kept the other out); with the fixed share the step goes
    # from 2.33 to 2.18 ms.
    pending = [None]
    gather_evs = []  # N > 1: (start, end) events around every gather of the timed region, on the gather's stream
    lagged = args.match_placement == "after-fast" and not args.no_match

    def launch_match(i):
        mstream.wait_event(ev_extracted[i])
        match(i, mstream.cuda_stream)
        ev_matched[i].record(mstream)

    def flush():
        if pending[0] is not None:
            launch_match(pending[0])
            pending[0] = None

    def step():
        i = step_no[0] % NBUF
        step_no[0] += 1
        side.wait_event(ev_matched[i])          # buffer set i is free once its previous match ...
        if world > 1:
            side.wait_event(ev_gathered[i])     # ... and its previous gather have finished
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, d_kp[i].data_ptr(), d_desc[i].data_ptr(), cap,
                                d_n[i].data_ptr(), stream)
        if rec:
            rec["fpost"].post_device(B, d_kp[i].data_ptr(), d_n[i].data_ptr(), cap, rec["kp_un"][i].data_ptr(),
                                     rec["cell_start"][i].data_ptr(), rec["cell_items"][i].data_ptr(), stream)
            rec["voc"].transform_device(B, d_desc[i].data_ptr(), d_n[i].data_ptr(), cap, 4, rec["bow_ids"][i].data_ptr(),
                                        rec["bow_vals"][i].data_ptr(), rec["n_words"][i].data_ptr(),
                                        rec["fv_nodes"][i].data_ptr(), rec["fv_off"][i].data_ptr(),
                                        rec["fv_idx"][i].data_ptr(), rec["n_fv"][i].data_ptr(), stream)
        ev_extracted[i].record(side)
        if args.no_match:
            ev_matched[i].record(mstream)
        elif lagged:
            if pending[0] is not None:
                ex.stream_wait_fast(mstream.cuda_stream)   # the previous batch's match: behind THIS batch's FAST
                launch_match(pending[0])
            pending[0] = i
        else:
            launch_match(i)
        if world > 1:
            cstream.wait_event(ev_extracted[i])
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gather_evs.append((g0, g1))
            with torch.cuda.stream(cstream):
                g0.record(cstream)
                if xc is not None:
                    xc.gather(d_n[i], d_kp[i], d_desc[i], root=0, stream=cstream.cuda_stream, out=xc_out[i] if rank == 0 else None)
                elif args.backend == "nccl":
                    gather_records_to_root(d_n[i], d_kp[i], d_desc[i], recv[i] if rank == 0 else None)
                else:  # rehearsal on a 1-GPU box: host copies through gloo
                    gather_records_to_root(d_n[i].to(coll_dev), d_kp[i].to(coll_dev), d_desc[i].to(coll_dev),
                                           recv[i] if rank == 0 else None)
                g1.record(cstream)
                ev_gathered[i].record(cstream)

    def sync():
        """drain this rank, then meet the others; returns the time at which THIS rank's own work was done"""
        flush()
        torch.cuda.synchronize()
        t_local = time.perf_counter()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return t_local

    # Steady-state clocks before anything is timed: a device that sat idle while the inputs were built ramps its clocks over tens of
    # milliseconds of load, and W = 3-5 warm-up steps are 7-11 ms of it -- the first timed steps then run slow (20 timed steps:
    # 2.168 ms per step behind 3 warm-up steps, 2.141 behind 50; a 3000-step run sustains 2.12: profiles/r06_steps_sweep.txt).  A fixed
    # number of untimed steps (the same on every rank: the N > 1 step holds a collective), then the W warm-up steps of the contract.
    for _ in range(args.clock_warmup_steps):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    gather_evs.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_local = sync() - t0
    dt = time.perf_counter() - t0
    scaling_diag = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # What a scaling curve needs to be read: every rank's own time for its K steps (before the barrier -- a slow rank shows
        # here, a slow collective does not), the gather's device time by events on its stream, and the world size as the
        # collective library reports it.
        mine = torch.tensor([t_local / args.steps * 1e3], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(x.item()), 4) for x in every]
        g_ms = [a.elapsed_time(b2) for a, b2 in gather_evs] if gather_evs else []
        scaling_diag = {"ms_per_step_per_rank": per_rank, "ms_per_step_rank_min": min(per_rank), "ms_per_step_rank_max": max(per_rank),
                        "gather_ms_mean": round(sum(g_ms) / len(g_ms), 4) if g_ms else None,
                        "gather_ms_max": round(max(g_ms), 4) if g_ms else None,
                        "gather_timed_by": "HIP events on the gather's own stream (rank 0's side of the exchange)",
                        "world": dist.get_world_size(), "world_c_abi": xc.world_reported() if xc is not None else None,
                        "backend": args.backend, "gather": args.gather}
    fps = world * B * args.steps / dt

    # ---- PCIe-inclusive rate (SURVEY 8d "report both"; never `value`): the same steps, but every batch arrives from
    # pinned host memory (H2D of level 0) and its records -- counts, 28-byte key points, 32-byte descriptors at their fixed
    # capacity -- go back to pinned host memory (D2H).  Two input / output sets, copies on their own stream, so the
    # transfers of batch k+1 / k-1 overlap the kernels of batch k.
    e2e = None
    if world == 1 and not args.no_end_to_end:
        h_in = torch.from_numpy(np.ascontiguousarray(frames.cpu().numpy())).pin_memory()
        d_in = [torch.empty_like(frames) for _ in range(NBUF)]
        h_out = [(torch.empty((B,), dtype=torch.int32).pin_memory(), torch.empty((B, cap, 28), dtype=torch.uint8).pin_memory(),
                  torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory()) for _ in range(NBUF)]
        cp_in, cp_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        ev_in = [torch.cuda.Event() for _ in range(NBUF)]
        ev_free = [torch.cuda.Event() for _ in range(NBUF)]
        ev_done = [torch.cuda.Event() for _ in range(NBUF)]

        def e2e_step(k):
            i = k % NBUF
            cp_in.wait_event(ev_free[i])                      # the extraction that last read d_in[i] has finished
            with torch.cuda.stream(cp_in):
                d_in[i].copy_(h_in, non_blocking=True)
                ev_in[i].record(cp_in)
            side.wait_event(ev_in[i])
            side.wait_event(ev_matched[i])
            side.wait_event(ev_done[i])                       # the D2H that last read output set i has finished
            ex.extract_batch_device(d_in[i].data_ptr(), B, W, H, W, W * H, d_kp[i].data_ptr(), d_desc[i].data_ptr(), cap,
                                    d_n[i].data_ptr(), stream)
            ev_free[i].record(side)
            ev_extracted[i].record(side)
            if not args.no_match:
                mstream.wait_event(ev_extracted[i])
                match(i, mstream.cuda_stream)
            ev_matched[i].record(mstream)
            cp_out.wait_event(ev_extracted[i])
            with torch.cuda.stream(cp_out):
                h_out[i][0].copy_(d_n[i], non_blocking=True)
                h_out[i][1].copy_(d_kp[i], non_blocking=True)
                h_out[i][2].copy_(d_desc[i], non_blocking=True)
                ev_done[i].record(cp_out)

        for k in range(args.warmup):
            e2e_step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            e2e_step(k)
        torch.cuda.synchronize()
        dte = time.perf_counter() - t0
        h2d, d2h = B * W * H, B * (4 + cap * 60)
        # each direction alone and both at once with no kernel running: says whether the step is bound by max(H2D, D2H) -- the
        # copies overlap -- or by their sum -- one engine or one link direction at a time
        def copies_ms(do_in, do_out, n=4):
            torch.cuda.synchronize()
            t0c = time.perf_counter()
            for k in range(n):
                i = k % NBUF
                if do_in:
                    with torch.cuda.stream(cp_in):
                        d_in[i].copy_(h_in, non_blocking=True)
                if do_out:
                    with torch.cuda.stream(cp_out):
                        h_out[i][0].copy_(d_n[i], non_blocking=True)
                        h_out[i][1].copy_(d_kp[i], non_blocking=True)
                        h_out[i][2].copy_(d_desc[i], non_blocking=True)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0c) / n * 1e3
        copies_ms(True, True, 1)
        h2d_ms, d2h_ms, both_ms = copies_ms(True, False), copies_ms(False, True), copies_ms(True, True)
        e2e_ms = dte / args.steps * 1e3
        overlap = both_ms < 0.5 * (max(h2d_ms, d2h_ms) + h2d_ms + d2h_ms)   # nearer to the maximum than to the sum
        step_ms = dt / args.steps * 1e3
        hidden, serial = max(both_ms, step_ms), both_ms + step_ms           # copies fully beside the kernels / one after the other
        copies_beside_kernels = e2e_ms < 0.5 * (hidden + serial)
        e2e = {"value": round(B * args.steps / dte, 2), "unit": "frames/s", "ms_per_step": round(e2e_ms, 4),
               "h2d_bytes_per_step": int(h2d), "d2h_bytes_per_step": int(d2h),
               "pcie_GBps": round((h2d + d2h) * args.steps / dte / 1e9, 2),
               "h2d_alone": {"ms_per_step": round(h2d_ms, 4), "GBps": round(h2d / (h2d_ms * 1e-3) / 1e9, 2)},
               "d2h_alone": {"ms_per_step": round(d2h_ms, 4), "GBps": round(d2h / (d2h_ms * 1e-3) / 1e9, 2)},
               "both_directions_no_kernels": {"ms_per_step": round(both_ms, 4), "GBps": round((h2d + d2h) / (both_ms * 1e-3) / 1e9, 2),
                                              "copies_overlap": bool(overlap)},
               "copies_overlap_kernels": bool(copies_beside_kernels),
               "explained": ("ms_per_step %.2f.  Copies alone: H2D %.2f, D2H %.2f, both at once %.2f ms (%s).  HBM-resident step %.2f ms.  "
                             "Copies fully beside the kernels would be %.2f ms, one after the other %.2f: %s" % (
                                 e2e_ms, h2d_ms, d2h_ms, both_ms,
                                 "the two directions overlap" if overlap else "the two directions do NOT overlap: one copy at a time",
                                 step_ms, hidden, serial,
                                 "the copies ran beside the kernels; the step is bound by the PCIe link" if copies_beside_kernels else
                                 "on this box the copies did NOT run beside the kernels (profiles/r06_e2e_overlap.txt has both cases: the "
                                 "same double-buffered loop, copy streams and events give either behaviour from one box to the next)")),
               "note": "inputs from pinned host memory, records (fixed capacity) back to pinned host memory, double-buffered "
                       "on copy streams; PCIe-bound -- `value` is the HBM-resident rate"}
        assert int(h_out[(args.steps - 1) % NBUF][0][0]) > 0

    # ---- per-stage HIP-event timing on the launch stream (untimed extra steps).  The end-to-end phase above leaves the device waiting for
    # PCIe most of the time and its clocks low: a few full steps first, or the isolated stage times read 5-10 % long (FAST 0.77 ms
    # against 0.70 inside the timed steps of the same run)
    for _ in range(12):
        step()
    sync()
    ex.set_stage_timing(True)
    acc = {}
    n_prof = 5
    ev_m0, ev_m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    match_ms = 0.0
    for _ in range(n_prof):
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, d_kp[0].data_ptr(), d_desc[0].data_ptr(), cap,
                                d_n[0].data_ptr(), stream)
        for k, v in ex.stage_times_ms().items():
            acc[k] = acc.get(k, 0.0) + v / n_prof
        ev_m0.record()
        match(0, stream)
        ev_m1.record()
        torch.cuda.synchronize()
        match_ms += ev_m0.elapsed_time(ev_m1) / n_prof
    ex.set_stage_timing(False)
    # ---- the same kernels timed INSIDE overlapped steps (the streams of the timed region): every extractor stage by events on
    # the stream its kernels are launched on (orbx_set_stage_timing(2)), the match by events on the match stream
    torch.cuda.synchronize()
    ex.set_stage_timing(2)
    n_in = 6
    acc_in = {}
    evm = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_in + 2)]
    prev = None
    for k in range(n_in + 2):
        i = step_no[0] % NBUF
        step_no[0] += 1
        side.wait_event(ev_matched[i])
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, d_kp[i].data_ptr(), d_desc[i].data_ptr(), cap,
                                d_n[i].data_ptr(), stream)
        ev_extracted[i].record(side)
        if args.no_match:
            ev_matched[i].record(mstream)
        else:
            j = prev if lagged else i      # the same placement as the timed steps
            if j is not None:
                if lagged:
                    ex.stream_wait_fast(mstream.cuda_stream)
                mstream.wait_event(ev_extracted[j])
                evm[k][0].record(mstream)
                match(j, mstream.cuda_stream)
                evm[k][1].record(mstream)
                ev_matched[j].record(mstream)
            prev = i
        if k >= 2:  # the stage events of this step are read before the next step re-records them
            for name, v in ex.stage_times_in_step_ms().items():
                acc_in[name] = acc_in.get(name, 0.0) + v / n_in
    if lagged and prev is not None:
        launch_match(prev)
    torch.cuda.synchronize()
    match_in_step_ms = None
    if not args.no_match:
        match_in_step_ms = sum(evm[k][0].elapsed_time(evm[k][1]) for k in range(2, n_in + 2)) / n_in
    ex.set_stage_timing(False)
    counts = d_n[0].cpu().numpy()
    # ---- the timed match is checked, not only timed: a sample of frame 0's rows against frame 1 by a numpy popcount scan
    match_check = None
    if B > 1 and not args.no_match:
        n0, n1 = int(counts[0]), int(counts[1])
        a0 = d_desc[0][0, :n0].cpu().numpy()
        b1 = d_desc[0][1, :n1].cpu().numpy()
        rows = np.random.RandomState(7).choice(n0, size=min(48, n0), replace=False)
        dm = np.unpackbits(a0[rows][:, None, :] ^ b1[None, :, :], axis=2).sum(axis=2)
        want_idx = dm.argmin(axis=1)
        part = np.partition(dm, 1, axis=1)
        got_idx = d_bidx[0][0, :n0].cpu().numpy()[rows]
        got_bd = d_bd[0][0, :n0].cpu().numpy().view(np.uint16)[rows]
        got_sd = d_sd[0][0, :n0].cpu().numpy().view(np.uint16)[rows]
        ok = bool((got_idx == want_idx).all() and (got_bd == part[:, 0]).all() and (got_sd == part[:, 1]).all())
        match_check = {"rows": int(len(rows)), "equal_to_numpy_popcount_scan": ok}
        assert ok, "the best/second-best match of the timed step differs from a numpy popcount scan"
    kp_mean = float(counts.mean())
    one_pass = acc.get("blur", 0.0) < 0.03 * max(acc.get("desc", 0.0), 1e-9)   # no separate blur pass ran: k_blur_desc
    alg, P = algorithmic_bytes(ex, W, H, int(round(kp_mean)), one_pass)
    stage_gbs = {k: (alg[k] * B / (acc[k] * 1e-3) / 1e9 if acc[k] > 0 and alg[k] > 0 else None) for k in acc}
    acc_all = dict(acc)
    acc_all["match_best2"] = match_ms
    alg["match_best2"] = int(2 * round(kp_mean) * 32 + round(kp_mean) * 8)
    stage_gbs["match_best2"] = alg["match_best2"] * B / (match_ms * 1e-3) / 1e9 if match_ms > 0 else None
    # `roofline` is reported for the stage that takes the most of a step's TIME: the step's wall time is the chain of the
    # extraction's stages on its stream (their in-step times and the gaps between them add up to ms_per_step), so the rule is
    # "the extraction stage with the largest time inside an overlapped step", by HIP events on the stream(s) its kernels are
    # launched on.  The match is not in that chain: it runs on its own stream BESIDE the quadtree / orientation with a fixed
    # share of every CU (after-fast placement) or beside the next pyramid (eager), and its elapsed time there -- reported in
    # stages_ms_in_step and roofline_match -- is as long as the window it hides in, not step time.  It would only be the
    # dominant stage if it outlasted the whole extraction (then the step would be waiting for it), which the rule checks.
    acc = acc_all
    in_step = {k: v for k, v in acc_in.items() if v and v > 0}
    dominant = max(in_step, key=lambda k: in_step[k])
    if match_in_step_ms:
        if match_in_step_ms > sum(in_step.values()):
            dominant = "match_best2"
        in_step["match_best2"] = match_in_step_ms
    # HBM bytes / VALU instructions per stage from separate rocprofv3 --pmc passes (tools/profile_round.sh).  They are
    # REPLAYED from files under profiles/, not measured in this run: each carries the hash of the kernel sources it was
    # collected on, and is dropped from the line when the sources have changed since.
    from monoorbslam3_amd._lib import kernels_sha16
    src_hash = kernels_sha16()

    def replayed(name):
        path = os.path.join(ROOT, "profiles", name)
        try:
            j = json.load(open(path))
        except Exception:
            return None, None
        if j.get("batch") != B:
            return None, "profiles/%s is for batch %s" % (name, j.get("batch"))
        if j.get("kernels_sha16") != src_hash:
            return None, "profiles/%s is stale (kernel sources changed since it was collected)" % name
        return j, "profiles/%s (rocprofv3 --pmc pass of an earlier run of this command on the same kernel sources %s; replayed)" % (name, src_hash)

    tj, traffic_source = replayed("pmc_traffic.json")
    traffic = tj["bytes_per_launch"].get(dominant) if tj else None
    vj, valu_source = replayed("pmc_valu.json")
    valu = None
    if vj:
        valu = {k: {"wave_instr_per_step": int(n), "achieved_gwinst_s": round(n / (acc[k] * 1e-3) / 1e9, 1)}
                for k, n in vj.get("wave_instr_per_step", {}).items() if k in acc and acc[k] > 0}
    # FAST's mix-weighted issue floor (tools/isa_mix.py: instruction classes from the ISA x measured stage counts, priced
    # with the measured per-class issue rates); replayed like the counters, for this batch size only
    fast_mix = None
    try:
        fm = json.load(open(os.path.join(ROOT, "profiles", "fast_mix.json")))
        if fm.get("frames") == B and fm.get("kernels_sha16") == src_hash and (W, H, NF) == (1242, 375, 2000):
            fast_mix = fm
    except Exception:
        pass
    fb_ms = acc["fast"] + acc["orient"] + acc["desc"] + acc["blur"]   # FAST + everything rBRIEF needs (orientation, blur, sampling)
    fast_brief_gbs = 2 * P * B / (fb_ms * 1e-3) / 1e9
    pairs = float((counts.astype(np.float64) * np.roll(counts, -1)).sum())

    # the dense match runs on the FP4 matrix path: 256 multiply-adds per descriptor pair
    mfma = None
    if match_ms > 0:
        tops = pairs * 512 / (match_ms * 1e-3) / 1e12
        mfma = {"pairs_per_step": int(pairs), "gpairs_per_s": round(pairs / (match_ms * 1e-3) / 1e9, 2),
                "pipe": "v_mfma_f32_32x32x64_f8f6f4 (FP4 operands)" if args.best2 == "fp4" else args.best2,
                "achieved_TOPS": round(tops, 1), "peak_TOPS": MFMA_FP4_PEAK_TOPS, "frac_of_dense_fp4_peak": round(tops / MFMA_FP4_PEAK_TOPS, 3),
                "note": "2 x 256 operations per 256-bit pair; peak = 4096 operations per cycle per SIMD x 1024 SIMDs x 2.4 GHz "
                        "(32 cycles per 32x32x64 instruction, measured by tools/microbench/fp4_hamming.hip)"}

    def roof(k):
        """Achieved algorithmic HBM bytes per second of stage k against the 8 TB/s peak, from the isolated launch time and --
        where measured -- from the time inside an overlapped step.  Every stage here is bounded by instruction issue before
        it is bounded by HBM; `limiter` says by how much where the instruction mix was priced."""
        g = stage_gbs.get(k)
        r = {"kernel": k, "bound": "hbm", "achieved": round(g, 1) if g else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(g / HBM_PEAK_GBS, 4) if g else None,
             "traffic": (tj["bytes_per_launch"].get(k) if tj else None), "traffic_source": traffic_source,
             "algorithmic_bytes_per_launch": int(alg[k] * B), "launch_ms": round(acc[k], 4)}
        if in_step.get(k):
            gi = alg[k] * B / (in_step[k] * 1e-3) / 1e9
            r["launch_ms_in_step"] = round(in_step[k], 4)
            r["achieved_in_step"] = round(gi, 1)
            r["frac_in_step"] = round(gi / HBM_PEAK_GBS, 4)
            r["in_step_note"] = ("HIP events on the stage's own launch stream(s) inside overlapped steps issued one at a time "
                                 "(the previous step's match still in flight); a stage = its launch groups summed")
        if k == "fast" and fast_mix:
            r["limiter"] = {"kind": "valu_issue_mix_weighted", "issue_floor_ms": fast_mix["issue_floor_ms_per_launch"],
                            "frac_of_floor": round(fast_mix["issue_floor_ms_per_launch"] / acc[k], 3),
                            "cheap_class_fraction": fast_mix.get("cheap_fraction_marked"),
                            "source": "profiles/fast_mix.json (tools/isa_mix.py: ISA instruction classes x stage counts of "
                                      "tools/fast_mix.py x issue rates of profiles/r02_valu_ops2.txt, r03_valu_ops3.txt; replayed)"}
        if k == "match_best2" and mfma:
            # the dense match is a matrix-pipe kernel (74 MB of descriptors per step: HBM says nothing about it): when it is the
            # stage reported, it is reported against the dense FP4 rate, isolated and inside the step, the HBM view kept beside it
            hbm = {f: r[f] for f in ("achieved", "peak", "unit", "frac", "achieved_in_step", "frac_in_step") if f in r}
            r.update({"bound": "mfma", "achieved": mfma["achieved_TOPS"], "peak": MFMA_FP4_PEAK_TOPS, "unit": "TFLOP/s",
                      "frac": mfma["frac_of_dense_fp4_peak"], "hbm_view": hbm,
                      "unit_note": "operations of the FP4 matrix path (2 x 256 per descriptor pair), counted as the contract's TFLOP/s"})
            if in_step.get(k):
                ti = pairs * 512 / (in_step[k] * 1e-3) / 1e12
                r["achieved_in_step"] = round(ti, 1)
                r["frac_in_step"] = round(ti / MFMA_FP4_PEAK_TOPS, 4)
                r["in_step_note"] += ("; the match runs on its own stream beside the next extraction's quadtree / orientation with one "
                                      "workgroup per CU (after-fast) or beside its pyramid (eager): its elapsed time there is the window it "
                                      "hides in (0.43 ms alone in the one-workgroup form, 0.30 at full occupancy), not step time")
        return r

    # ---- corner-density sweep (extra key; the headline workload is unchanged): the synthetic frames are corner-rich by design
    # (every level must exceed its quota) -- 5.5 % of all pyramid pixels are FAST corners at threshold 20 -- and
    # FAST's sparse stages scale with that.  The same batch size with about a tenth and a third of the shapes shows what the
    # stage costs at a density closer to a street scene's.  Isolated stage times (one stream), three steps each.
    density = None
    if world == 1 and not args.no_density_sweep and (W, H, NF) == (1242, 375, 2000):
        density = []
        for n_shapes in (50, 150, None):
            fb = synth.make_frames(n_distinct, W, H, seed=synth.DEFAULT_SEED + 101 * rank, n_shapes=n_shapes)
            fr = torch.from_numpy(fb).to(dev)
            if B > n_distinct:
                fr = fr.repeat((B + n_distinct - 1) // n_distinct, 1, 1)[:B].contiguous()
            ex.set_stage_timing(True)
            accd = {}
            for it in range(4):
                ex.extract_batch_device(fr.data_ptr(), B, W, H, W, W * H, d_kp[0].data_ptr(), d_desc[0].data_ptr(), cap, d_n[0].data_ptr(), stream)
                if it:
                    for k, v in ex.stage_times_ms().items():
                        accd[k] = accd.get(k, 0.0) + v / 3
            torch.cuda.synchronize()
            ex.set_stage_timing(False)
            total = sum(accd.values())
            density.append({"shapes_per_frame": n_shapes if n_shapes is not None else 500,
                            "fast9_corner_fraction_level0": round(synth.fast9_corner_fraction(fb[0]), 5),
                            "keypoints_per_frame": round(float(d_n[0].float().mean().item()), 1),
                            "stages_ms": {k: round(v, 4) for k, v in accd.items()},
                            "extract_frames_per_s_isolated_stages": round(B / (total * 1e-3), 1),
                            "fast_algorithmic_GBps": round(P * B / (accd["fast"] * 1e-3) / 1e9, 1),
                            "fast_frac_of_hbm_peak": round(P * B / (accd["fast"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            del fr

    out = {
        "metric": baseline_metric(),
        "value": round(fps, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "clock_warmup_steps": args.clock_warmup_steps,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "scaling_diag": scaling_diag,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": ("KITTI 1242x375, 2000 feat, 8-level pyramid scale 1.2, FAST 20/7; extract + "
                                "2000x2000 Hamming best-2 per frame" + (" + frame records (undistort, grid, bag of words)" if args.records else "")
                                if (W, H, NF) == (1242, 375, 2000) else
                                "Batched synthetic 1920x1080 stream, 2000 feat, %d concurrent frame(s) sharded %d/GPU, RCCL gather "
                                "over xGMI; extract + Hamming best-2 against the previous frame" % (world * B, B)
                                if (W, H, NF) == (1920, 1080, 2000) else "%dx%d, %d feat" % (W, H, NF)),
                   "baseline_config": args.config,
                   "frames_per_gpu_per_step": B, "width": W, "height": H, "n_features": NF,
                   "match": not args.no_match, "records": bool(args.records),
                   "match_placement": None if args.no_match else args.match_placement, "best2_resident": args.best2_resident,
                   "parallelism": ("frames sharded %d per GPU per step over %d GPU(s), one gather of the fixed-capacity records "
                                   "to rank 0 per step" % (B, world)) if world > 1 else
                                  "%d resident frame(s) on one GPU, no collective" % B},
        "keypoints_per_frame": round(kp_mean, 1),
        "records": ({"words_per_frame": round(float(rec["n_words"][0].float().mean().item()), 1),
                     "fv_nodes_per_frame": round(float(rec["n_fv"][0].float().mean().item()), 1),
                     "grid_items_per_frame": round(float(rec["cell_start"][0][:, -1].float().mean().item()), 1)}
                    if rec else None),
        "stages_ms": dict({k: round(v, 4) for k, v in acc.items()}, orient_desc=round(acc["orient"] + acc["desc"], 4)),
        "descriptor_path": "k_blur_desc (blur + descriptors in one pass, no blurred level written)" if one_pass else "blur pass + k_orient_desc",
        "match_ms": round(match_ms, 4),
        "match_check": match_check,
        "stage_algorithmic_GBps": {k: (round(v, 1) if v else None) for k, v in stage_gbs.items()},
        "stages_ms_note": "HIP events around each stage with every kernel on ONE stream (orbx_set_stage_timing), extra untimed "
                          "steps; the timed steps overlap FAST / blur / match on three streams, so the stages sum to more "
                          "than ms_per_step",
        "stages_ms_in_step": {k: round(v, 4) for k, v in in_step.items()},
        "dominant_rule": "the extraction stage with the largest time inside an overlapped step (stages_ms_in_step); the match runs "
                         "beside the extraction on its own stream and counts only if it outlasts the whole extraction",
        "dominant_rule_history": "round 5 changed the rule (rounds 1-4: the stage with the largest in-step time, the match counted like "
                                 "any other stage); roofline_longest_in_step_stage keeps that older rule's answer beside `roofline`",
        "roofline_longest_in_step_stage": roof(max(in_step, key=lambda k: in_step[k])),
        "density_sweep": density,
        "roofline": roof(dominant),
        "roofline_fast": roof("fast"),
        "roofline_match": dict(roof("match_best2"), mfma=mfma),
        "roofline_fast_plus_brief": {"bound": "hbm", "achieved": round(fast_brief_gbs, 1), "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": round(fast_brief_gbs / HBM_PEAK_GBS, 4),
                                     "algorithmic_bytes_per_frame": 2 * P},
        "valu_issue": {"rates_gwinst_s": {"cheap_class": VALU_RATE_CHEAP_GWINST, "slow_class": VALU_RATE_SLOW_GWINST},
                       "stages": valu, "source": valu_source},
    }
    if e2e is not None:
        out["value_end_to_end"] = e2e
    if world == 1 and not args.no_extra_configs:
        # BASELINE configs 3 and 5 beside the headline: extra keys, outside the timed region
        del frames
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        out["config3"] = bench_config3(dev, local_rank)
        out["config5"] = bench_config5(local_rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(base, NF, min(16, os.cpu_count() or 1), not args.no_match)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
