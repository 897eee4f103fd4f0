"""Live cross-check of the oracle's OpenCV-side primitives against a REAL OpenCV -- the only route from "parity
unpinned" to pinned (SURVEY.md s4 "optional live cross-check").  The reference calls cv::FAST
(ORBextractor.cc:848,854), cv::resize (:984), cv::GaussianBlur (:950), cv::fastAtan2 (:112) and cv::cvtColor
(Tracking.cc:96-109) from an un-vendored, un-pinned OpenCV 4; oracle/orb_oracle.cpp restates them from the published
algorithms.  Wherever `cv2` can be imported this test compares the two on the seeded frames and on the FAST cell ROIs
of every pyramid level, and prints cv2.__version__ (quote the result in DESIGN.md s2).  It is SKIPPED when OpenCV is
absent (this image and the GPU box have none: no cv2, no opencv4 headers, no network).

It exists twice -- unmarked and `gpu`-marked -- so that it gets its chance both in the CPU suite and on the GPU box.
"""
import numpy as np
import pytest

from oracle.pyoracle import OracleExtractor
from os1_amd.synth import synth


def _gauss_variant(cv2):
    """Which 8-bit GaussianBlur this OpenCV has (include/orbfe.h, orbfe_extractor_set_blur_variant): 1 = taps rounded one by one
    (4.0.0 - 4.1.0 and 3.4.2 - 3.4.6), 0 = error-diffused taps (every later release).  The same rule as include/orbfe/ORBextractor.h."""
    v = tuple(int(''.join(ch for ch in x if ch.isdigit()) or 0) for x in cv2.__version__.split('.')[:3])
    if v[0] == 4 and (v[1] == 0 or (v[1] == 1 and v[2] == 0)):
        return 1
    if v[0] == 3 and v[1] == 4 and 2 <= v[2] <= 6:
        return 1
    return 0


def _cv2():
    try:
        import cv2
    except Exception:
        pytest.skip('OpenCV (cv2) is not installed: the oracle stays "parity unpinned"')
    return cv2


def _restated_ops():
    """orbfe::detail::RestatedOps (include/orbfe/orb_shim.hpp) behind a C entry point, compiled on the spot."""
    import ctypes as C
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tempfile.mkdtemp(), 'librestated_ops.so')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-I' + os.path.join(root, 'include'),
                           os.path.join(root, 'tests', 'cpp', 'restated_ops_export.cpp'), '-o', so])
    L = C.CDLL(so)
    L.shim_cv_small.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.POINTER(C.c_double),
                                C.c_int]

    def call(op, A, b, alpha=1.0, c=None, beta=0.0):
        A = np.ascontiguousarray(A, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        cc = None if c is None else np.ascontiguousarray(c, np.float32)
        n = A.size
        out, out1 = np.zeros(max(n, 3), np.float32), C.c_double(0)
        L.shim_cv_small({'gemm': 0, 'gemmT': 1, 'norm': 2, 'dot': 3, 'scale': 4, 'divide': 5}[op], A.ctypes.data, b.ctypes.data, alpha,
                        None if cc is None else cc.ctypes.data, beta, out.ctypes.data, C.byref(out1), n)
        if op in ('gemm', 'gemmT'):
            return out[:3]
        return out[:n] if op in ('scale', 'divide') else out1.value
    return call


def _check_against_opencv(oracle):
    cv2 = _cv2()
    print('OpenCV version under test:', cv2.__version__)
    info = cv2.getBuildInformation()
    for line in info.splitlines():      # what decides bit-exactness besides the version: IPP, the CPU baseline / dispatch lists
        if any(key in line for key in ('IPP', 'Baseline', 'Dispatched', 'requested')):
            print('  build:', ' '.join(line.split()))
    # optimisation paths (IPP / OpenCL) are not bit-exact by OpenCV's own account; the reference links the plain library,
    # compare against the generic code paths
    cv2.setUseOptimized(True)
    try:
        cv2.ocl.setUseOpenCL(False)
    except Exception:
        pass
    rng = np.random.default_rng(7)
    report = {}

    # ---- cv::fastAtan2 (ORBextractor.cc:112): every (m01, m10) on a coarse grid plus random moments, bit-exact float32
    ys = np.concatenate([np.arange(-300, 301, 7), rng.integers(-2000000, 2000000, 4000)]).astype(np.float32)
    xs = np.concatenate([np.arange(-300, 301, 7)[::-1], rng.integers(-2000000, 2000000, 4000)]).astype(np.float32)
    bad = 0
    for y, x in zip(ys, xs):
        a = np.float32(oracle.fast_atan2(float(y), float(x)))
        b = np.float32(cv2.fastAtan2(float(y), float(x)))
        bad += a.tobytes() != b.tobytes()
    report['fastAtan2'] = bad

    # ---- cv::cvtColor RGB2GRAY / BGR2GRAY, 3 and 4 channels (Tracking.cc:96-109)
    col = rng.integers(0, 256, (120, 173, 4), dtype=np.uint8)
    bad = 0
    for ch in (3, 4):
        img = np.ascontiguousarray(col[:, :, :ch])
        for rgb, code in ((True, cv2.COLOR_RGB2GRAY if ch == 3 else cv2.COLOR_RGBA2GRAY),
                          (False, cv2.COLOR_BGR2GRAY if ch == 3 else cv2.COLOR_BGRA2GRAY)):
            want = cv2.cvtColor(img, code)
            got15, got14 = oracle.cvt_gray(img, rgb, 0), oracle.cvt_gray(img, rgb, 1)
            bad += not ((got15 == want).all() or (got14 == want).all())
    report['cvtColor'] = bad

    # ---- the small-matrix arithmetic of the pose-driven searches (ORBmatcher.cc:293-298, 322-348, 1326-1343): the
    # reference's cv::Mat expressions as OpenCV evaluates them -- `Rcw*x3Dw+tcw` = gemm(A, b, 1, c, 1), `-Rcw.t()*tcw` =
    # gemm(A, b, -1, noArray, 0, GEMM_1_T), `-sR21*t12` = gemm(A, b, -1, ...), cv::norm -- against the oracle's restatement
    # (cvGemm3 / cvGemmT3 / cvNorm3) AND the shim's (orbfe::detail::RestatedOps).  Mat::dot and MatExpr scale / divide have no
    # Python binding (three-float loops; pinned only by an integrated build, whose CvOps evaluates them with its own OpenCV).
    shim = _restated_ops()
    bad = {'gemm': 0, 'gemm_neg': 0, 'gemmT': 0, 'norm': 0}
    for it in range(4000):
        sc = 10.0 ** rng.integers(-2, 3)
        A = (rng.standard_normal((3, 3)) * sc).astype(np.float32)
        b = (rng.standard_normal((3, 1)) * sc).astype(np.float32)
        c = (rng.standard_normal((3, 1)) * sc).astype(np.float32)
        want = cv2.gemm(A, b, 1.0, c, 1.0).astype(np.float32).ravel()
        bad['gemm'] += want.tobytes() != oracle.cv_small('gemm', A, b, 1.0, c, 1.0).tobytes()
        bad['gemm'] += want.tobytes() != shim('gemm', A, b, 1.0, c, 1.0).tobytes()
        want = cv2.gemm(A, b, -1.0, None, 0.0).astype(np.float32).ravel()
        bad['gemm_neg'] += want.tobytes() != oracle.cv_small('gemm', A, b, -1.0, None, 0.0).tobytes()
        bad['gemm_neg'] += want.tobytes() != shim('gemm', A, b, -1.0, None, 0.0).tobytes()
        want = cv2.gemm(A, b, -1.0, None, 0.0, flags=cv2.GEMM_1_T).astype(np.float32).ravel()
        bad['gemmT'] += want.tobytes() != oracle.cv_small('gemmT', A, b, -1.0).tobytes()
        bad['gemmT'] += want.tobytes() != shim('gemmT', A, b, -1.0).tobytes()
        want = float(cv2.norm(b))
        bad['norm'] += want != oracle.cv_small('norm', A, b) or want != shim('norm', A, b)
    report.update({'cv_' + k_: v_ for k_, v_ in bad.items()})

    # ---- cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) as Frame::UndistortKeyPoints / ComputeImageBounds call it
    # (Frame.cc:286-353): 4-, 5- and 8-coefficient models, points over and beyond the image, against the oracle AND the product's
    # host routine (orbfe_undistort_pinhole, the same double arithmetic; loads without a GPU)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    pts = np.stack([rng.uniform(-40, 800, 3000), rng.uniform(-40, 520, 3000)], 1).astype(np.float32)
    bad = 0
    for dist in ([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], [-0.28, 0.07, 2e-4, 2e-5, -0.01],
                 [0.1, -0.05, 1e-3, -1e-3, 0.01, 0.02, -0.01, 0.003]):
        d = np.array(dist, np.float32)
        want = cv2.undistortPoints(pts.reshape(-1, 1, 2), K, d, None, K).reshape(-1, 2).astype(np.float32)
        got = oracle.undistort_pinhole(pts, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), d)
        bad += int((got.view(np.uint32) != want.view(np.uint32)).any(axis=1).sum())
        try:
            from os1_amd import api
            got2 = api.undistort_pinhole(pts, np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), d)
            bad += int((got2.view(np.uint32) != want.view(np.uint32)).any(axis=1).sum())
        except OSError:
            pass                         # liborbfe.so needs the HIP runtime to load; the oracle comparison stands
    report['undistortPoints'] = bad

    for seed, W, H, N in ((1, 640, 480, 1000), (2, 1920, 1080, 2000)):
        img = synth(seed, W, H)
        ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
        ox.extract(img)
        levels = [ox.level(l) for l in range(8)]
        # ---- cv::resize INTER_LINEAR, level l from level l-1 (ORBextractor.cc:984)
        bad = 0
        for l in range(1, 8):
            h, w = levels[l].shape
            want = cv2.resize(levels[l - 1], (w, h), interpolation=cv2.INTER_LINEAR)
            bad += int((want != levels[l]).sum())
            bad += int((oracle.resize(levels[l - 1], w, h) != want).sum())
        report['resize_%dx%d' % (W, H)] = bad
        # ---- cv::GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 on a clone of every level (ORBextractor.cc:949-950)
        bad = 0
        for l in range(8):
            want = cv2.GaussianBlur(levels[l].copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
            bad += int((oracle.gauss7(levels[l], _gauss_variant(cv2)) != want).sum())
        report['GaussianBlur_%dx%d' % (W, H)] = bad
        report['GaussianBlur_variant_expected_for_this_cv2'] = _gauss_variant(cv2)
        # ---- cv::FAST(ROI, th, nms=true) on the reference's cell ROIs of every level (ORBextractor.cc:826-856)
        bad = 0
        det = {t: cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
               for t in (20, 7)}
        for l in range(8):
            lv = levels[l]
            h, w = lv.shape
            width, height = float(w - 32), float(h - 32)
            ncols, nrows = int(width / 30), int(height / 30)
            wc, hc = int(np.ceil(width / ncols)), int(np.ceil(height / nrows))
            cells = [(i, j) for i in range(nrows) for j in range(ncols)]
            for i, j in [cells[k] for k in rng.choice(len(cells), min(40, len(cells)), replace=False)]:
                iy, ix = 16 + i * hc, 16 + j * wc
                if iy >= h - 16 - 3 or ix >= w - 16 - 6:
                    continue
                roi = np.ascontiguousarray(lv[iy:min(iy + hc + 6, h - 16), ix:min(ix + wc + 6, w - 16)])
                for t in (20, 7):
                    kps = det[t].detect(roi)
                    want = sorted((int(k.pt[1]), int(k.pt[0]), int(k.response)) for k in kps)
                    got = sorted((int(y), int(x), int(s)) for x, y, s in oracle.fast9(roi, t, True))
                    bad += want != got
        report['FAST_%dx%d' % (W, H)] = bad
    print('OpenCV %s cross-check, mismatches per primitive: %r' % (cv2.__version__, report))
    assert not any(report.values()), (cv2.__version__, report)


def test_oracle_primitives_against_live_opencv(oracle):
    _check_against_opencv(oracle)


@pytest.mark.gpu
def test_oracle_primitives_against_live_opencv_on_gpu_box(oracle):
    _check_against_opencv(oracle)
