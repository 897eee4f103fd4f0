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


def _cv2():
    try:
        import cv2
    except Exception:
        pytest.skip('OpenCV (cv2) is not installed: the oracle stays "parity unpinned"')
    return cv2


def _check_against_opencv(oracle):
    cv2 = _cv2()
    print('OpenCV version under test:', cv2.__version__)
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
            bad += int((oracle.gauss7(levels[l]) != want).sum())
        report['GaussianBlur_%dx%d' % (W, H)] = bad
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
