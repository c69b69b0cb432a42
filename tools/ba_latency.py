import sys, time
sys.path.insert(0, '.')
import numpy as np
sys.path.insert(0, 'tests')
import torch
from monoorbslam3_amd import ba
import test_ba
pr, args = test_ba._perturbed(20, 3000, 5)
for rep in range(4):
    t0 = time.perf_counter(); got = ba.local_bundle_adjustment(*args); dt = time.perf_counter() - t0
    print("local BA: wall %.2f ms, device %.2f ms, %d its / %d solves, %d outliers" % (dt * 1e3, got["device_ms"], got["iterations"], got["trials"], got["outlier"].sum()))
for rep in range(3):
    t0 = time.perf_counter(); g = ba.linearize(pr["cam"], *args[1:9]); dt = time.perf_counter() - t0
    print("linearize: wall %.2f ms, kernels %.3f ms" % (dt * 1e3, g["kernel_ms"]))
cam, R0, t0_, off, P, Z, W = test_ba._pose_frames([1000] * 64, 5)
for rep in range(3):
    t0 = time.perf_counter(); g = ba.pose_optimize_batch(cam, R0, t0_, off, P, Z, W); dt = time.perf_counter() - t0
    print("pose_optimize_batch(64 x 1000): wall %.2f ms, kernel %.3f ms" % (dt * 1e3, g["kernel_ms"]))
