"""Register / scratch budgets of the kernels whose occupancy the measured numbers depend on, read from the gfx950 ISA that
hipcc emits for the shipped sources (no GPU needed).  A few more VGPRs are invisible in a diff and cost a whole workgroup per CU:
the batch quadtree went from four to three workgroups per CU (0.23 -> 0.27 ms per 512 frames) on a 125 -> 133 VGPR step."""
import hashlib
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monoorbslam3_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--offload-arch=gfx950", "-mllvm",
         "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S"]

# kernel (substring of the mangled name) -> (max VGPRs, max scratch bytes, why)
BUDGET = {
    "k_fast_strip": (96, 0, "5 waves per SIMD (512 / 5 = 102); LDS allows 18 workgroups per CU"),
    "k_blur_descILi256": (102, 0, "amdgpu_waves_per_eu 5"),
    "k_blur_descILi257": (102, 0, "amdgpu_waves_per_eu 5"),
    "oct_batch12k_octree_lds": (128, 12, "four 256-thread workgroups per CU; the 12 B of scratch are ONE 8-byte address spilled before the "
                                         "pass loop and reloaded once per final-phase sweep ahead of the rank scatter of the bitonic-sort path "
                                         "(orbx_octree.h:975, only levels whose final phase holds more than 320 nodes): outside every inner loop"),
    "oct_huge12k_octree_lds": (128, 0, "16 waves per workgroup: 128 VGPRs is all a thread can have; spills stalled the level for 30 us"),
    "k_fast_cells_waveILi8E": (128, 0, "8 waves per workgroup, several workgroups per CU"),
    "k_orientILb0E": (64, 0, "8 waves per SIMD"),
    "k_blur_mfmaILi256": (96, 0, "5 waves per SIMD"),
}


def _isa(source="orbx_kernels"):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    out_dir = os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "%s_%s.s" % (source, h.hexdigest()[:16]))
    if not os.path.exists(out):
        subprocess.run([hipcc] + FLAGS + ["-o", out, os.path.join(CSRC, source + ".hip")], check=True, cwd=CSRC,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(isa):
    """mangled name -> (VGPRs, scratch bytes, static LDS bytes)"""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", isa, re.S):
        body = m.group(2)
        out[m.group(1)] = (int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)),
                           int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)),
                           int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1)))
    return out


def test_occupancy_critical_kernels_stay_within_their_register_budgets():
    s = _isa()
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S):
        name, body = m.group(1), m.group(2)
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        for key in BUDGET:
            if key in name:
                found[key] = (vg, sc)
    assert set(found) == set(BUDGET), "kernels renamed? missing: %s" % sorted(set(BUDGET) - set(found))
    over = {k: (found[k], BUDGET[k]) for k in BUDGET if found[k][0] > BUDGET[k][0] or found[k][1] > BUDGET[k][1]}
    assert not over, "over budget (got (vgpr, scratch), budget (vgpr, scratch, why)): %s" % over


def test_the_match_and_its_partners_fit_one_cu_together():
    """bench.py's default step runs the previous batch's match -- k_best2_fp4 as ONE 8-wave workgroup per CU
    (ORBM_VAR_BEST2_RESIDENT = 1) -- beside the quadtree, k_desc_bins and k_orient of the next extraction (DESIGN 4f,
    profiles/r05_overlap.md).  That only works while the shares add up on a CU of 4 SIMDs x 512 VGPRs and 160 KB of LDS: the
    match's two waves per SIMD, two quadtree workgroups (one wave per SIMD each), four k_orient waves per SIMD."""
    x, m = _kernels(_isa("orbx_kernels")), _kernels(_isa("orbm_matcher"))
    pick = lambda d, key: next(v for k, v in d.items() if key in k)  # noqa: E731
    alloc = lambda vg: (vg + 7) // 8 * 8                            # VGPRs are allocated in blocks of 8 on gfx950  # noqa: E731
    match_vg, match_scratch, match_lds = pick(m, "k_best2_fp4ILb1E")
    plain_vg, plain_scratch, _ = pick(m, "k_best2_fp4ILb0E")
    oct_vg, _, oct_lds = pick(x, "oct_batch12k_octree_lds")
    ori_vg, _, ori_lds = pick(x, "k_orientILb0E")
    assert match_vg <= 128 and plain_vg <= 128 and plain_scratch == 0      # two workgroups per CU at full occupancy, no spill in the hot form
    # round 6: the walking form carried popcount(query) and a few addresses across its tile loop in scratch (36 B, stored before and
    # reloaded after a block's 63 tiles); the popcount now comes from a second read of the row after the loop: no scratch at all
    assert match_scratch == 0
    assert match_lds <= 40 * 1024
    oct_dyn = 34 * 1024                                                    # orbx_octree_lds_bytes for 1242x375 / 2000 features (33.3 KB)
    # the match (8 waves: 2 per SIMD) + two quadtree workgroups (4 waves: 1 per SIMD each)
    assert 2 * alloc(match_vg) + 2 * alloc(oct_vg) <= 512
    assert match_lds + 2 * (oct_lds + oct_dyn) <= 160 * 1024
    # the match + four k_orient waves per SIMD (workgroups of 4 waves, 1 per SIMD: four workgroups)
    assert 2 * alloc(match_vg) + 4 * alloc(ori_vg) <= 512
    assert match_lds + 4 * ori_lds <= 160 * 1024
