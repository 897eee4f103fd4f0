"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from os1_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liborb_oracle.so')
# the reference's own DBoW2 BowVector.cpp + FeatureVector.cpp, built by oracle/Makefile where /root/reference exists
DBOW2_REF_SO = os.path.join(_HERE, '_ref', 'libdbow2_vec.so')

KP_DTYPE = np.dtype([('x', 'f4'), ('y', 'f4'), ('size', 'f4'), ('angle', 'f4'), ('response', 'f4'),
                     ('octave', 'i4'), ('class_id', 'i4')])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self):
        if not os.path.exists(_SO):
            build()
        L = self.L = C.CDLL(_SO)
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_sincos_host.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extractor_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_gauss_kernel.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gauss7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_gauss7_u8_variant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_gauss_kernel_variant.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.orc_extractor_set_gauss_variant.argtypes = [C.c_void_p, C.c_int]
        L.orc_extractor_set_tie_rule.argtypes = [C.c_void_p, C.c_int]
        L.orc_tie_stats.argtypes = [C.c_void_p, C.c_int]
        L.orc_fast9.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_int]
        L.orc_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                               C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_float, C.c_float, C.c_void_p]
        L.orc_search_by_projection_uv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p]
        L.orc_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
        L.orc_cvt_gray.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_int]
        L.orc_search_for_triangulation.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                                                  C.c_void_p, C.c_int, C.c_void_p]
        L.orc_search_projected.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + \
            [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_vocab_create.restype = C.c_void_p
        L.orc_vocab_create.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int]
        L.orc_vocab_destroy.argtypes = [C.c_void_p]
        L.orc_use_dbow2_ref.argtypes = [C.c_char_p]
        L.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 2 + \
            [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + [C.POINTER(C.c_int)] + [C.c_void_p] * 2
        L.orc_search_by_bow.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_bow_kf.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_undistort_equidistant.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_undistort_pinhole.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.orc_image_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                       C.c_int, C.c_void_p]

    # ---- reference-built pieces --------------------------------------------------------------
    def have_dbow2_ref(self):
        return os.path.exists(DBOW2_REF_SO)

    def use_dbow2_ref(self, on=True):
        """Accumulate BowVector / FeatureVector through the reference's own DBoW2 code (oracle/_ref/libdbow2_vec.so)
        instead of the restatement.  Process-wide switch of the oracle library; returns True if now active."""
        if not on:
            self.L.orc_use_dbow2_ref(None)
            return False
        if not self.have_dbow2_ref():
            return False
        return self.L.orc_use_dbow2_ref(DBOW2_REF_SO.encode()) == 0

    def bow_accumulator_is_reference(self):
        return bool(self.L.orc_bow_accumulator_is_reference())

    # ---- primitives -------------------------------------------------------------------------
    def fast_atan2(self, y, x):
        return float(self.L.orc_fast_atan2(y, x))

    def sincos(self, angle_deg):
        a, b = C.c_float(), C.c_float()
        self.L.orc_sincos_host(angle_deg, C.byref(a), C.byref(b))
        return a.value, b.value

    def gauss_kernel(self, n=7, sigma=2.0, variant=0):
        """variant 0: error-diffused 8.8 taps (OpenCV >= 4.1.1); 1: plainly rounded taps (4.0.0 - 4.1.0)."""
        k = np.zeros(n, np.int32)
        self.L.orc_gauss_kernel_variant(n, sigma, variant, _p(k))
        return k

    def tie_stats(self, reset=True):
        """H1 bookkeeping of every quadtree run since the last reset: dict(sorts, sorted_nodes, nodes_in_ties, breaks, breaks_inside_a_tie)."""
        out = np.zeros(5, np.int64)
        self.L.orc_tie_stats(_p(out), int(reset))
        return dict(zip(('sorts', 'sorted_nodes', 'nodes_in_ties', 'breaks', 'breaks_inside_a_tie'), (int(v) for v in out)))

    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        self.L.orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
        return dst

    def gauss7(self, src, variant=0):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros_like(src)
        self.L.orc_gauss7_u8_variant(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0], variant)
        return dst

    def fast9(self, img, threshold, nms=True):
        img = np.ascontiguousarray(img, np.uint8)
        cap = img.size
        out = np.zeros((max(cap, 1), 3), np.int32)
        n = self.L.orc_fast9(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nms), _p(out), cap)
        return out[:n].copy()

    def fast_score_bruteforce(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return self.L.orc_fast_score_bruteforce(_p(img), img.strides[0], x, y)

    def distribute_octtree(self, kps, minX, maxX, minY, maxY, N):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        out = np.zeros(max(len(kps), 1), KP_DTYPE)
        n = self.L.orc_distribute_octtree(_p(kps), len(kps), minX, maxX, minY, maxY, N, _p(out), len(out))
        return out[:n].copy()

    def hamming(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return self.L.orc_hamming(_p(a), _p(b))

    def get_features_in_area(self, kps, bounds, x, y, r, minLevel, maxLevel):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        b = np.asarray(bounds, np.float32)
        out = np.zeros(max(len(kps), 1), np.int32)
        n = self.L.orc_get_features_in_area(_p(kps), len(kps), _p(b), x, y, r, minLevel, maxLevel, _p(out), len(out))
        return out[:n].copy()

    def search_for_initialization(self, kps1, desc1, kps2, desc2, bounds, prev_xy, window=100, nnratio=0.9,
                                  check_ori=True):
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.asarray(bounds, np.float32)
        prev = np.ascontiguousarray(prev_xy, np.float32).copy()
        m12 = np.full(max(len(kps1), 1), -1, np.int32)
        n = self.L.orc_search_for_initialization(_p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2),
                                                 _p(b), _p(prev), _p(m12), window, nnratio, int(check_ori))
        return n, m12[:len(kps1)], prev

    def search_by_projection(self, kps, desc, bounds, scale_factors, kp_occupied, mp_xy, mp_level, mp_viewcos,
                             mp_flags, mp_desc, th, nnratio):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        b = np.asarray(bounds, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        occ = np.ascontiguousarray(kp_occupied, np.uint8)
        mp_xy = np.ascontiguousarray(mp_xy, np.float32)
        mp_level = np.ascontiguousarray(mp_level, np.int32)
        mp_viewcos = np.ascontiguousarray(mp_viewcos, np.float32)
        mp_flags = np.ascontiguousarray(mp_flags, np.uint8)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        assigned = np.full(max(len(kps), 1), -1, np.int32)
        n = self.L.orc_search_by_projection(_p(kps), _p(desc), len(kps), _p(b), _p(sf), _p(occ), _p(mp_xy),
                                            _p(mp_level), _p(mp_viewcos), _p(mp_flags), _p(mp_desc), len(mp_level),
                                            th, nnratio, _p(assigned))
        return n, assigned[:len(kps)]

    def search_by_projection_uv(self, kps, desc, bounds, scale_factors, kp_occupied, src_uv, src_level, src_angle,
                                src_flags, src_valid, src_desc, th, max_dist, skip_any_occupied, check_ori):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        b = np.asarray(bounds, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        occ = np.ascontiguousarray(kp_occupied, np.uint8)
        src_uv = np.ascontiguousarray(src_uv, np.float32)
        src_level = np.ascontiguousarray(src_level, np.int32)
        src_angle = np.ascontiguousarray(src_angle, np.float32)
        src_flags = np.ascontiguousarray(src_flags, np.uint8)
        src_valid = np.ascontiguousarray(src_valid, np.uint8)
        src_desc = np.ascontiguousarray(src_desc, np.uint8)
        assigned = np.full(max(len(kps), 1), -1, np.int32)
        n = self.L.orc_search_by_projection_uv(_p(kps), _p(desc), len(kps), _p(b), _p(sf), _p(occ), _p(src_uv),
                                               _p(src_level), _p(src_angle), _p(src_flags), _p(src_valid),
                                               _p(src_desc), len(src_level), th, max_dist, int(skip_any_occupied),
                                               int(check_ori), _p(assigned))
        return n, assigned[:len(kps)]

    def distinctive_descriptor(self, descs):
        descs = np.ascontiguousarray(descs, np.uint8)
        return self.L.orc_distinctive_descriptor(_p(descs), len(descs))

    def cvt_gray(self, img, rgb_order=True, variant=0):
        """(H, W, 3|4) uint8 -> (H, W) uint8 like cv::cvtColor(RGB2GRAY / BGR2GRAY)."""
        img = np.ascontiguousarray(img, np.uint8)
        H, W, ch = img.shape
        out = np.zeros((H, W), np.uint8)
        self.L.orc_cvt_gray(_p(img), H, W, W * ch, ch, int(rgb_order), variant, _p(out), W)
        return out

    def search_for_triangulation(self, kps1, desc1, has_mp1, fv1, kps2, desc2, has_mp2, fv2, F12, ex, ey, scale2, sigma2,
                                 check_ori=True):
        kps1 = np.ascontiguousarray(kps1)
        kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        h1 = np.ascontiguousarray(has_mp1, np.uint8)
        h2 = np.ascontiguousarray(has_mp2, np.uint8)
        f1 = [np.ascontiguousarray(a, np.uint32) for a in fv1]
        f2 = [np.ascontiguousarray(a, np.uint32) for a in fv2]
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale2, np.float32)
        sg = np.ascontiguousarray(sigma2, np.float32)
        pairs = np.full((max(len(kps1), 1), 2), -1, np.int32)
        nm = self.L.orc_search_for_triangulation(_p(kps1), _p(desc1), _p(h1), len(kps1), _p(f1[0]), _p(f1[1]), _p(f1[2]),
                                                 len(f1[0]), _p(kps2), _p(desc2), _p(h2), len(kps2), _p(f2[0]), _p(f2[1]),
                                                 _p(f2[2]), len(f2[0]), _p(F), C.c_float(ex), C.c_float(ey), _p(sc), _p(sg),
                                                 int(check_ori), _p(pairs))
        return nm, pairs[:nm]

    def search_projected(self, kps, desc, bounds, uv, radius, level, valid, sdesc, kp_skip=None, claim=False,
                         inv_sigma2=None, chi2=5.99, max_dist=50):
        kps = np.ascontiguousarray(kps)
        desc = np.ascontiguousarray(desc, np.uint8)
        b = np.asarray(bounds, np.float32)
        uv = np.ascontiguousarray(uv, np.float32)
        radius = np.ascontiguousarray(radius, np.float32)
        level = np.ascontiguousarray(level, np.int32)
        valid = np.ascontiguousarray(valid, np.uint8)
        sdesc = np.ascontiguousarray(sdesc, np.uint8)
        ns = len(radius)
        skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
        inv = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
        bi = np.full(max(ns, 1), -1, np.int32)
        bd = np.full(max(ns, 1), -1, np.int32)
        nm = self.L.orc_search_projected(_p(kps), _p(desc), len(kps), _p(b), ns, _p(uv), _p(radius), _p(level), _p(valid),
                                         _p(sdesc), None if skip is None else _p(skip), int(claim),
                                         None if inv is None else _p(inv), C.c_double(chi2), max_dist, _p(bi), _p(bd))
        return nm, bi[:ns], bd[:ns]

    def vocabulary(self, image):
        return OracleVocabulary(self, image)

    def search_by_bow(self, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio=0.7, check_ori=True):
        """valid2 None: (KeyFrame, Frame) overload, result converted to matches12; else (KeyFrame, KeyFrame)."""
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        angle1 = np.ascontiguousarray(angle1, np.float32)
        angle2 = np.ascontiguousarray(angle2, np.float32)
        valid1 = np.ascontiguousarray(valid1, np.uint8)
        f1 = [np.ascontiguousarray(a, np.uint32) for a in fv1]
        f2 = [np.ascontiguousarray(a, np.uint32) for a in fv2]
        n1, n2 = len(desc1), len(desc2)
        if valid2 is None:
            m21 = np.full(max(n2, 1), -1, np.int32)
            nm = self.L.orc_search_by_bow(_p(desc1), _p(angle1), _p(valid1), n1, _p(f1[0]), _p(f1[1]), _p(f1[2]),
                                          len(f1[0]), _p(desc2), _p(angle2), n2, _p(f2[0]), _p(f2[1]), _p(f2[2]),
                                          len(f2[0]), nnratio, int(check_ori), _p(m21))
            m12 = np.full(n1, -1, np.int32)
            for i2 in range(n2):
                if m21[i2] >= 0:
                    assert m12[m21[i2]] == -1
                    m12[m21[i2]] = i2
            return nm, m12
        valid2 = np.ascontiguousarray(valid2, np.uint8)
        m12 = np.full(max(n1, 1), -1, np.int32)
        nm = self.L.orc_search_by_bow_kf(_p(desc1), _p(angle1), _p(valid1), n1, _p(f1[0]), _p(f1[1]), _p(f1[2]),
                                         len(f1[0]), _p(desc2), _p(angle2), _p(valid2), n2, _p(f2[0]), _p(f2[1]),
                                         _p(f2[2]), len(f2[0]), nnratio, int(check_ori), _p(m12))
        return nm, m12[:n1]

    def cv_small(self, op, A, b, alpha=1.0, c=None, beta=0.0):
        """The restated small-matrix OpenCV arithmetic: op 'gemm' (alpha*A*b + beta*c), 'gemmT' (alpha*A.t()*b), 'norm' (of b),
        'dot' (A[:3] . b)."""
        A = np.ascontiguousarray(A, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        cc = None if c is None else np.ascontiguousarray(c, np.float32)
        out3, out1 = np.zeros(3, np.float32), C.c_double(0)
        self.L.orc_cv_small.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p,
                                        C.POINTER(C.c_double)]
        self.L.orc_cv_small({'gemm': 0, 'gemmT': 1, 'norm': 2, 'dot': 3}[op], _p(A), _p(b), alpha, None if cc is None else _p(cc),
                            beta, _p(out3), C.byref(out1))
        return out3 if op in ('gemm', 'gemmT') else out1.value

    def undistort_pinhole(self, xy, fx, fy, cx, cy, dist):
        xy = np.ascontiguousarray(xy, np.float32).copy()
        d = np.ascontiguousarray(dist, np.float32)
        self.L.orc_undistort_pinhole(_p(xy), len(xy), fx, fy, cx, cy, _p(d) if len(d) else None, len(d))
        return xy

    def image_bounds(self, cols, rows, mode, fx, fy, cx, cy, dist=()):
        d = np.ascontiguousarray(dist, np.float32)
        b = np.zeros(4, np.float32)
        self.L.orc_image_bounds(cols, rows, mode, fx, fy, cx, cy, _p(d) if len(d) else None, len(d), _p(b))
        return b

    def undistort_equidistant(self, xy, fx, fy, cx, cy):
        xy = np.ascontiguousarray(xy, np.float32).copy()
        self.L.orc_undistort_equidistant(_p(xy), len(xy), fx, fy, cx, cy)
        return xy


class OracleVocabulary:
    """DBoW2 vocabulary + transform(features, v, fv, levelsup) restated on the CPU."""

    def __init__(self, oracle, image):
        self.o = oracle
        buf = np.frombuffer(image, np.uint8)
        self.h = oracle.L.orc_vocab_create(int(buf[0]), int(buf[1]), int(buf[2]), int(buf[3]), _p(buf[4:].copy()),
                                           (buf.size - 4) // 45)

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        ids = np.zeros(max(n, 1), np.uint32)
        vals = np.zeros(max(n, 1), np.float64)
        fvn = np.zeros(max(n, 1), np.uint32)
        fvo = np.zeros(n + 1, np.uint32)
        fvf = np.zeros(max(n, 1), np.uint32)
        wof = np.zeros(max(n, 1), np.uint32)
        nof = np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int(0), C.c_int(0)
        self.o.L.orc_bow_transform(self.h, _p(desc), n, levelsup, _p(ids), _p(vals), C.byref(nw), _p(fvn), _p(fvo),
                                   _p(fvf), C.byref(nn), _p(wof), _p(nof))
        nw, nn = nw.value, nn.value
        return ids[:nw], vals[:nw], (fvn[:nn], fvo[:nn + 1], fvf[:int(fvo[nn])]), wof[:n], nof[:n]

    def __del__(self):
        try:
            self.o.L.orc_vocab_destroy(self.h)
        except Exception:
            pass


class OracleExtractor:
    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, oracle=None):
        self.o = oracle or Oracle()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.o.L.orc_extractor_create(nfeatures, scale, nlevels, ini_th, min_th)

    def __del__(self):
        try:
            self.o.L.orc_extractor_destroy(self.h)
        except Exception:
            pass

    def set_gauss_variant(self, variant):
        """0: GaussianBlur of OpenCV >= 4.1.1 (error-diffused taps, the default); 1: of OpenCV 4.0.0 - 4.1.0 (rounded taps, saturating)."""
        self.o.L.orc_extractor_set_gauss_variant(self.h, int(variant))

    def set_tie_rule(self, rule):
        """H1: order of equal-sized quadtree nodes in the final phase's sort.  0 creation order (pinned), 1 reversed, 2 real heap
        addresses of this process, >= 3 seeded random."""
        self.o.L.orc_extractor_set_tie_rule(self.h, int(rule))

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.o.L.orc_extractor_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(sf=sf, isf=isf, s2=s2, is2=is2, nfeat=nf, umax=um)

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        # a level returns at most max(N_l + 2, 4 * roots) keypoints (ORBextractor.cc:620-773): strips many roots wide with a tiny quota
        # return far more than nfeatures (found by tools/sweep_debug.py, round 5)
        cap = self.nfeatures + 64 + 520 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.o.L.orc_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), _p(desc), cap)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def level(self, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        self.o.L.orc_level_size(self.h, level, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.o.L.orc_level_copy(self.h, level, int(blurred), _p(out))
        return out

    def candidates(self, level):
        n = self.o.L.orc_level_candidates(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.o.L.orc_level_candidates(self.h, level, _p(out), n)
        return out[:n].copy()
