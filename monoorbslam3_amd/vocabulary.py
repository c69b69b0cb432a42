"""Host mirror of the reference's ORBVocabulary (modules/ORB/ORBVocabulary.h:12-24, a DBoW2 TemplatedVocabulary<FORB>)
over the C ABI of include/orbv.h: the vocabulary tree lives on the GPU; transform() is Frame::computeBow."""
import ctypes as C

import numpy as np

from . import _lib

NO_NODE = 0xFFFFFFFF
MAX_FEATURES = 8192

_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp, i32 = C.c_void_p, C.c_int
        L.orbv_create.restype = i32
        L.orbv_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, C.POINTER(vp)]
        L.orbv_load_text.restype = i32
        L.orbv_load_text.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        L.orbv_destroy.restype = None
        L.orbv_destroy.argtypes = [vp]
        L.orbv_info.restype = i32
        L.orbv_info.argtypes = [vp] + [C.POINTER(i32)] * 6
        L.orbv_nodes.restype = i32
        L.orbv_nodes.argtypes = [vp, vp, vp, vp, vp]
        L.orbv_transform_features_device.restype = i32
        L.orbv_transform_features_device.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
        L.orbv_transform_device.restype = i32
        L.orbv_transform_device.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orbv_transform.restype = i32
        L.orbv_transform.argtypes = [vp, vp, i32, i32, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
        _bound = True
    return L


class ORBVocabulary:
    def __init__(self, handle):
        self._h = handle
        v = [C.c_int() for _ in range(6)]
        _lib.check(_L().orbv_info(self._h, *[C.byref(x) for x in v]))
        self.k, self.L, self.scoring, self.weighting, self.n_nodes, self.n_words = (x.value for x in v)

    @classmethod
    def from_arrays(cls, voc, device=0):
        """voc: dict(k, L, scoring, weighting, parent, is_leaf, desc, weight), nodes in text-file order."""
        parent = np.ascontiguousarray(voc["parent"], dtype=np.int32)
        leaf = np.ascontiguousarray(voc["is_leaf"], dtype=np.uint8)
        desc = np.ascontiguousarray(voc["desc"], dtype=np.uint8)
        weight = np.ascontiguousarray(voc["weight"], dtype=np.float64)
        if not (len(parent) == len(leaf) == len(desc) == len(weight)) or desc.shape[1:] != (32,):
            raise ValueError("vocabulary arrays disagree in length")
        h = C.c_void_p()
        _lib.check(_L().orbv_create(voc["k"], voc["L"], voc["scoring"], voc["weighting"], len(parent), parent.ctypes.data,
                                    leaf.ctypes.data, desc.ctypes.data, weight.ctypes.data, device, C.byref(h)))
        return cls(h)

    @classmethod
    def load_text(cls, path, device=0):
        """ORBVocabulary::createORBVocabulary -> loadFromTextFile (ORBVocabulary.cpp:10-21)."""
        h = C.c_void_p()
        _lib.check(_L().orbv_load_text(str(path).encode(), device, C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            try:
                _L().orbv_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    __del__ = close

    def nodes(self):
        parent = np.zeros(self.n_nodes, np.int32)
        leaf = np.zeros(self.n_nodes, np.uint8)
        desc = np.zeros((self.n_nodes, 32), np.uint8)
        weight = np.zeros(self.n_nodes, np.float64)
        _lib.check(_L().orbv_nodes(self._h, parent.ctypes.data, leaf.ctypes.data, desc.ctypes.data, weight.ctypes.data))
        return dict(k=self.k, L=self.L, scoring=self.scoring, weighting=self.weighting, parent=parent, is_leaf=leaf,
                    desc=desc, weight=weight)

    def transform(self, desc, levelsup=4):
        """Frame::computeBow for one frame: returns (bow_ids, bow_vals, (fv_nodes, fv_off, fv_idx)); the last tuple is
        the CSR FeatureVector ORBMatcher.SearchByBow takes."""
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        n = len(desc)
        m = max(n, 1)
        bi, bv = np.zeros(m, np.uint32), np.zeros(m, np.float64)
        fn, fo, fi = np.zeros(m, np.uint32), np.zeros(m + 1, np.int32), np.zeros(m, np.uint32)
        nw, nf = C.c_int(), C.c_int()
        _lib.check(_L().orbv_transform(self._h, desc.ctypes.data, n, levelsup, bi.ctypes.data, bv.ctypes.data, C.byref(nw),
                                       fn.ctypes.data, fo.ctypes.data, fi.ctypes.data, C.byref(nf)))
        return bi[: nw.value], bv[: nw.value], (fn[: nf.value], fo[: nf.value + 1], fi[: fo[nf.value]])

    def transform_features_device(self, d_desc, n, levelsup, d_word, d_node, d_weight, stream=None):
        _lib.check(_L().orbv_transform_features_device(self._h, d_desc, n, levelsup, d_word, d_node, d_weight, _lib.stream_arg(stream)))

    def transform_device(self, n_frames, d_desc, d_n, cap, levelsup, d_bow_ids, d_bow_vals, d_n_words, d_fv_nodes, d_fv_off,
                         d_fv_idx, d_n_fv, stream=None):
        _lib.check(_L().orbv_transform_device(self._h, n_frames, d_desc, d_n, cap, levelsup, d_bow_ids, d_bow_vals,
                                              d_n_words, d_fv_nodes, d_fv_off, d_fv_idx, d_n_fv, _lib.stream_arg(stream)))
