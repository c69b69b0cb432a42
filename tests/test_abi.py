"""The C-ABI library loads and exports every symbol include/*.h declares; no compute without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from monoorbslam3_amd import _lib
    return _lib


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orb(?:x|m|ba|f|v|d)_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(built):
    L = C.CDLL(built.LIB_PATH)
    names = _declared("orbx.h") + _declared("orbm.h") + _declared("orbba.h") + _declared("orbf.h") + _declared("orbv.h") + \
        _declared("orbd.h")
    assert "orbba_linearize" in names
    assert len(names) >= 60
    for prefix in ("orbx_", "orbm_", "orbba_", "orbf_", "orbv_", "orbd_"):
        assert any(n.startswith(prefix) for n in names), prefix
    for n in names:
        assert hasattr(L, n), "liborbx.so does not export %s" % n


def test_python_bindings_resolve(built):
    built.lib()
    from monoorbslam3_amd.matcher import _mlib
    _mlib()
    assert built.lib().orbx_version().startswith(b"orbx")


def test_kp_record_layout_matches_cv_keypoint():
    from monoorbslam3_amd.extractor import KP_DTYPE
    assert KP_DTYPE.itemsize == 28
    assert [KP_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_fails_loudly_without_a_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from monoorbslam3_amd._lib import OrbxError
    from monoorbslam3_amd.extractor import ORBExtractor
    from monoorbslam3_amd.matcher import MatcherHandle
    with pytest.raises(OrbxError) as e:
        ORBExtractor(1000)
    assert e.value.code == -2 and "no CPU path" in str(e.value)
    with pytest.raises(OrbxError):
        MatcherHandle()


def test_bad_arguments_are_rejected_before_touching_the_device(built):
    L = built.lib()
    h = C.c_void_p()
    cfg = built.OrbxCfg(1000, 1.2, 99, 20, 7, 0, 0, 1, 0, -1)
    assert L.orbx_create(C.byref(cfg), C.byref(h)) == -1 and b"n_levels" in L.orbx_last_error()
    cfg = built.OrbxCfg(1000, 2.0, 8, 20, 7, 0, 0, 1, 0, -1)
    assert L.orbx_create(C.byref(cfg), C.byref(h)) == -4
    cfg = built.OrbxCfg(1000, 1.2, 8, 0, 7, 0, 0, 1, 0, -1)
    assert L.orbx_create(C.byref(cfg), C.byref(h)) == -1


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/"""
    pkg = os.path.join(ROOT, "monoorbslam3_amd")
    bad = re.compile(r"(from|import)\s+oracle|orb_ref|oracle/|oracle\.")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(txt), "%s references the oracle" % os.path.join(dirpath, f)
    out = subprocess.run(["ldd", os.path.join(pkg, "lib", "liborbx.so")], capture_output=True, text=True).stdout
    assert "orb_ref" not in out


def test_compat_shims_compile_against_mock_opencv(built):
    """the reference-signature shims (compat/ORBExtractor.h, ORBMatcher.h) are header-only C++;
    they are syntax-checked here against the minimal cv:: mirror types in compat/cv_mirror.h"""
    src = os.path.join(ROOT, "monoorbslam3_amd", "compat", "shim_check.cpp")
    if not os.path.exists(src):
        pytest.skip("shim not built yet")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src])


def test_every_handle_type_fails_loudly_without_a_gpu():
    """orbf / orbv / orbba entry points: no device -> a negative status and a message, never a CPU fallback."""
    import ctypes as C
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    from monoorbslam3_amd import ba, frame, vocabulary
    from monoorbslam3_amd._lib import OrbxError
    with pytest.raises(OrbxError, match="no HIP device"):
        frame.FramePost(640, 480, 500.0, 500.0, 320.0, 240.0)
    voc = dict(k=2, L=1, scoring=0, weighting=0, parent=np.zeros(3, np.int32), is_leaf=np.array([0, 1, 1], np.uint8),
               desc=np.zeros((3, 32), np.uint8), weight=np.array([0.0, 1.0, 1.0]))
    with pytest.raises(OrbxError, match="no HIP device"):
        vocabulary.ORBVocabulary.from_arrays(voc)
    args = ((500.0, 500.0, 320.0, 240.0), np.eye(3)[None].repeat(2, 0), np.zeros((2, 3)), np.array([1, 0], np.uint8),
            np.ones((3, 3)), np.array([0, 1, 1], np.int32), np.array([0, 1, 2], np.int32), np.zeros((3, 2)), np.ones(3))
    for fn in (ba.linearize, ba.optimize, ba.local_bundle_adjustment):
        with pytest.raises(OrbxError, match="no HIP device"):
            fn(*args)
    with pytest.raises(OrbxError, match="no HIP device"):
        ba.pose_optimize_batch(args[0], np.eye(3)[None], np.zeros((1, 3)), np.array([0, 3], np.int32), np.ones((3, 3)),
                               np.zeros((3, 2)), np.ones(3))


def test_issue_class_pricing_of_the_fast_floor():
    """tools/isa_mix.py prices every VALU instruction of k_fast_strip by opcode AND operand form (bench.py reports the sum as
    FAST's `limiter`): plain 32-bit logic / add / right shifts and v_bitop3 are the cheap class, a scalar-register operand
    or any other encoding is not; LDS, scalar and memory instructions are counted apart."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    kind = lambda line: m.classify(line)[0]  # noqa: E731
    assert kind("v_and_b32_e32 v53, 0x3f3f3f3f, v53") == "valu_cheap"
    assert kind("v_bitop3_b32 v48, v52, v49, v48 bitop3:0xc8") == "valu_cheap"
    assert kind("v_lshrrev_b32_e32 v48, 2, v48") == "valu_cheap" and kind("v_lshlrev_b32_e32 v49, 2, v49") == "valu_slow"
    assert kind("v_add_u32_e32 v49, s35, v28") == "valu_slow"            # the same opcode with a scalar operand
    assert kind("v_alignbit_b32 v49, v49, v51, 26") == "valu_slow" and kind("v_cmp_ne_u32_e32 vcc, 0, v48") == "valu_slow"
    assert kind("ds_read2_b32 v[48:49], v46 offset0:1 offset1:104") == "lds" and kind("s_bcnt1_i32_b64 s8, vcc") == "salu"
    assert kind("global_load_dwordx2 v[2:3], v[0:1], off") == "vmem"
    cheap, slow = m.classify("v_xor_b32_e32 v1, v2, v3")[1], m.classify("v_perm_b32 v1, v2, v3, v4")[1]
    assert 700 < cheap < 1100 and 450 < slow < 650 and m.CHEAP > 1.4 * m.SLOW


def test_quadtree_build_choice_across_the_lds_boundaries(built):
    """orbx_launch_octree's host logic (orbx_dev_octree_plan, no device): the 1024-thread count-pyramid build is only taken
    when the node-list arrays -- laid out for the largest level's quota -- fit the LDS it is launched with; between that size and
    the limit of the LDS-resident list the 512-thread build serves, above it the global-scratch build.  Quotas are swept across both
    boundaries at 1920x1080 (levels 0 and 1 are above the megapixel threshold)."""
    L = C.CDLL(built.LIB_PATH)
    fn = L.orbx_dev_octree_plan
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(built.OrbxCfg), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    HUGE, LIMIT = 160 * 1024 - 1024, 160 * 1024 - 256

    def plan(nf, frames=1, lb=0, le=8, requota=0, w=1920, h=1080):
        cfg = built.OrbxCfg(nf, 1.2, 8, 20, 7, w, h, frames, 0, -1)
        out = (C.c_int64 * 4)()
        rc = fn(C.byref(cfg), requota, w, h, frames, lb, le, out)
        assert rc in (0, -4)   # -4: a per-level quota above 7680 is refused (orbx_create refuses it the same way)
        return tuple(out) if rc == 0 else None

    kinds = {}
    prev_lds = 0
    for nf in sorted(set(range(500, 14000, 100)) | set(range(10300, 10700, 2))):   # (the 512-thread window is 21 quotas wide)
        got = plan(nf)
        if got is None:   # level 0's quota above 4096: its sort keys no longer fit the global-scratch build's LDS either
            assert nf > 12000
            break
        kind, lds, huge, hend = got
        assert lds >= prev_lds  # grows with the quota
        prev_lds = lds
        kinds.setdefault(kind, []).append(lds)
        if kind == 2:      # the list arrays fit what the build is launched with, and it is launched with the stated size
            assert lds <= huge == HUGE and hend == 8
        elif kind == 1:    # too large for the count-pyramid build's launch size, still LDS-resident
            assert HUGE < lds <= LIMIT
        else:
            assert kind == 3 and lds > LIMIT
    assert set(kinds) == {1, 2, 3}, "the sweep must cross both boundaries: %s" % {k: (min(v), max(v)) for k, v in kinds.items()}
    # the default tracker quota and its 2N initial extractor take the count-pyramid build for a single frame
    assert plan(1000)[0] == 2 and plan(1000, requota=2000)[0] == 2
    # more workgroups than CUs: only the megapixel levels keep the whole-CU build, the small ones go to the 512-thread build
    kind, _, _, hend = plan(1000, frames=40)
    assert kind == 2 and hend == 2
    assert plan(1000, frames=8)[3] == 8
    # a range without a megapixel level (the main chain of a split call at 1242x375), and a resident batch
    assert plan(2000, w=1242, h=375)[0] == 1 and plan(2000, frames=512, w=1242, h=375)[0] == 0
    assert plan(1000, lb=2, le=8)[0] == 1
