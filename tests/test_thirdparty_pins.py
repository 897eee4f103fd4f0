"""Third-party cross-checks of the oracle's OpenCV-side primitives (the reference's pixel arithmetic lives in an
OpenCV that is absent from the image, so the oracle's restatement of it is 'unpinned').  What the image DOES hold is an
Anaconda python3.9 with scikit-image and an older scipy: implementations of the same published algorithms written by
somebody else.  They are not OpenCV, so they pin the ALGORITHM (the FAST-9 segment test, the score as the largest passing
threshold, an integer 7x7 convolution with reflect-101 borders), not OpenCV's choice of variant.

The other interpreter runs as a child process on .npy files; skipped where it (or scikit-image) is absent."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

PY39 = '/opt/conda/bin/python3.9'


def _run39(script, tmp_path, **arrays):
    if not os.path.exists(PY39):
        pytest.skip('no %s in this image' % PY39)
    for k, a in arrays.items():
        np.save(tmp_path / (k + '.npy'), a)
    src = tmp_path / 'job.py'
    src.write_text(textwrap.dedent(script))
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    r = subprocess.run([PY39, '-W', 'ignore', str(src)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        if 'ModuleNotFoundError' in r.stderr or 'ImportError' in r.stderr:
            pytest.skip('third-party module missing: ' + r.stderr.strip().splitlines()[-1])
        raise AssertionError(r.stderr[-2000:])
    print(r.stdout.strip())
    return tmp_path


def _textured(seed, h, w):
    """blocks, edges, isolated dots and noise: corners at every contrast from 1 to 200"""
    rng = np.random.RandomState(seed)
    img = np.full((h, w), 110, np.float64)
    for _ in range(60):
        y, x = rng.randint(0, h - 6), rng.randint(0, w - 6)
        hh, ww = rng.randint(2, 14), rng.randint(2, 14)
        img[y:y + hh, x:x + ww] += rng.randint(-100, 120)
    for _ in range(40):
        img[rng.randint(3, h - 3), rng.randint(3, w - 3)] += rng.randint(-110, 130)
    img += rng.randint(-6, 7, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_fast9_against_scikit_image(oracle, tmp_path):
    """cv::FAST(img, kps, th, nms) as ORBextractor.cc:829-837 calls it: (i) the set of pixels that pass the 9-of-16 segment test
    at threshold t, for EVERY t, and with it (ii) the score = the largest t a pixel still passes at -- against
    skimage.feature.corner_fast(n=9), which shares no code with OpenCV or with the oracle."""
    img = _textured(7, 72, 88)
    d = _run39('''
        import numpy as np, skimage
        from skimage.feature import corner_fast
        img = np.load('img.npy').astype(np.float64)
        score = np.zeros(img.shape, np.int32)
        for t in range(1, 255):
            m = corner_fast(img, n=9, threshold=float(t)) > 0
            if not m.any():
                break
            score[m] = t
        np.save('score.npy', score)
        print('scikit-image', skimage.__version__, 'corners at t=1:', int((score > 0).sum()))
        ''', tmp_path, img=img)
    want = np.load(d / 'score.npy')
    assert (want > 0).sum() > 300 and want.max() > 60
    for th in (1, 7, 20, 45):
        got = np.zeros_like(want)
        k = oracle.fast9(img, th, nms=False)
        got[k[:, 1], k[:, 0]] = k[:, 2]
        exp = np.where(want >= th, want, 0)
        assert np.array_equal(got > 0, exp > 0), 'segment test differs at threshold %d' % th
        assert np.array_equal(got, exp), 'score differs at threshold %d' % th
    # non-maximum suppression (FAST's 3x3 rule: strictly greater than all eight neighbours' scores) on the third-party score map
    for th in (7, 20):
        s = np.where(want >= th, want, 0)
        p = np.pad(s, 1)
        keep = s > 0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    keep &= s > p[1 + dy:1 + dy + s.shape[0], 1 + dx:1 + dx + s.shape[1]]
        ys, xs = np.nonzero(keep)
        exp = sorted(zip(xs.tolist(), ys.tolist(), s[ys, xs].tolist()))
        got = sorted(map(tuple, oracle.fast9(img, th, nms=True).tolist()))
        assert got == exp


def test_gauss7_against_scipy_ndimage(oracle, tmp_path):
    """GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in 8.8 fixed point (ORBextractor.cc:950): the 2-D integer convolution
    with the outer product of [18,34,48,56,48,34,18] and 'mirror' borders by scipy.ndimage (integer arithmetic, somebody else's
    border handling), rounded once: (sum + 32768) >> 16."""
    img = _textured(11, 61, 83)
    d = _run39('''
        import numpy as np, scipy
        from scipy import ndimage
        img = np.load('img.npy').astype(np.int64)
        k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
        s = ndimage.correlate(img, np.outer(k, k), mode='mirror')
        np.save('blur.npy', ((s + 32768) >> 16).astype(np.uint8))
        print('scipy', scipy.__version__)
        ''', tmp_path, img=img)
    assert np.array_equal(oracle.gauss7(img), np.load(d / 'blur.npy'))


def test_resize_geometry_against_scikit_image(oracle, tmp_path):
    """resize(..., INTER_LINEAR) (ORBextractor.cc:979-990): the sampling geometry (pixel centres: sx = (dx + 0.5) * scale - 0.5,
    clamped at the edges) against skimage.transform.resize(order=1, no anti-aliasing) in double precision.  OpenCV computes in
    11-bit fixed point, so the values may differ by one gray level -- a wrong geometry (corner-aligned, or off by half a pixel)
    differs by tens."""
    img = _textured(5, 120, 160)
    sizes = [(100, 133), (83, 111), (69, 93), (120, 80), (31, 160)]
    d = _run39('''
        import numpy as np, skimage
        from skimage.transform import resize
        img = np.load('img.npy').astype(np.float64)
        for i, (dh, dw) in enumerate(np.load('sizes.npy').tolist()):
            np.save('r%d.npy' % i, resize(img, (dh, dw), order=1, mode='edge', anti_aliasing=False, preserve_range=True, clip=False))
        print('scikit-image', skimage.__version__)
        ''', tmp_path, img=img, sizes=np.array(sizes))
    for i, (dh, dw) in enumerate(sizes):
        want = np.load(d / ('r%d.npy' % i))
        got = oracle.resize(img, dw, dh).astype(np.float64)
        err = np.abs(got - want)
        assert err.max() <= 1.0, (dh, dw, err.max())
        # (the vertical pass truncates twice at 1/4 gray level, (b * (S >> 4)) >> 16, before its final rounding: up to half a
        # level below the exact value, so 'within plain rounding' does not hold pixel by pixel -- the mean does)
        assert err.mean() < 0.4, (dh, dw, err.mean())


def test_brief_pattern_against_scikit_image():
    """The 256 learned rBRIEF test pairs (ORBextractor.cc:182-439, `bit_pattern_31_`): scikit-image ships the same table from
    the ORB authors with its own ORB (`skimage/feature/orb_descriptor_positions.txt`); ours, as compiled into the oracle and
    into the product, must equal it entry for entry."""
    import glob, re
    found = glob.glob('/opt/conda/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt')
    if not found:
        pytest.skip('no scikit-image data files in this image')
    want = np.loadtxt(found[0]).astype(np.int8)
    assert want.shape == (256, 4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ('oracle/brief_pattern.inc', 'os1_amd/csrc/brief_pattern.inc'):
        txt = re.sub(r'//.*', '', open(os.path.join(root, f)).read())
        got = np.array([int(x) for x in re.findall(r'-?\d+', txt)], np.int8).reshape(256, 4)
        assert np.array_equal(got, want), f


def test_ic_angle_against_scikit_image(oracle, tmp_path):
    """IC_Angle (ORBextractor.cc:86-113) + cv::fastAtan2: the keypoint orientations of a whole extraction, every level, against
    skimage.feature.corner_orientations with scikit-image's own copy of the circular 31-pixel patch (OFAST_MASK, the same `umax`
    table), evaluated on the oracle's UNBLURRED level images with an exact arctan2.  fastAtan2 is documented to 0.3 degrees."""
    from oracle.pyoracle import OracleExtractor
    img = np.kron(_textured(3, 96, 128), np.ones((4, 4), np.uint8))
    img = (img.astype(np.int32) + np.random.RandomState(1).randint(-8, 9, img.shape)).clip(0, 255).astype(np.uint8)
    ex = OracleExtractor(800, 1.2, 8, 20, 7, oracle)
    kps, _ = ex.extract(img)
    sf = ex.tables()['sf']
    assert len(kps) > 500 and len(set(kps['octave'].tolist())) == 8
    arrays = {}
    for l in range(8):
        k = kps[kps['octave'] == l]
        rc = np.stack([np.rint(k['y'] / sf[l]), np.rint(k['x'] / sf[l])], 1).astype(np.int64)
        arrays['lv%d' % l] = ex.level(l)
        arrays['rc%d' % l] = rc
    d = _run39('''
        import numpy as np, skimage
        from skimage.feature import corner_orientations
        from skimage.feature.orb import OFAST_MASK
        for l in range(8):
            img = np.load('lv%d.npy' % l).astype(np.float64)
            np.save('ang%d.npy' % l, np.rad2deg(corner_orientations(img, np.load('rc%d.npy' % l), OFAST_MASK)))
        print('scikit-image', skimage.__version__, 'patch pixels:', int(OFAST_MASK.sum()))
        ''', tmp_path, **arrays)
    worst = 0.0
    for l in range(8):
        want = np.load(d / ('ang%d.npy' % l)) % 360.0
        got = kps[kps['octave'] == l]['angle'].astype(np.float64)
        diff = np.abs((got - want + 180.0) % 360.0 - 180.0)
        worst = max(worst, float(diff.max()))
    assert worst <= 0.3, worst


def test_steered_brief_against_scikit_image(oracle, tmp_path):
    """computeOrbDescriptor (ORBextractor.cc:132-171) on the blurred levels: scikit-image's steered-BRIEF loop (`orb_cy._orb_loop`:
    the same table, row = sin*x + cos*y, col = cos*x - sin*y, bit = I(p0) < I(p1)) is given the oracle's blurred level images,
    keypoints and angles.  It rotates in double and rounds half away from zero where the reference rotates in float and rounds
    half to even, so a test location that lands within 1e-6 of a half-integer may pick the neighbouring pixel: a handful of bits
    per thousand descriptors.  What this pins: the use of the BLURRED image, the steering convention and its sign, the order
    of the 256 tests and the packing of bit j into byte j / 8, bit j % 8."""
    from oracle.pyoracle import OracleExtractor
    img = np.kron(_textured(9, 96, 128), np.ones((4, 4), np.uint8))
    img = (img.astype(np.int32) + np.random.RandomState(2).randint(-8, 9, img.shape)).clip(0, 255).astype(np.uint8)
    ex = OracleExtractor(800, 1.2, 8, 20, 7, oracle)
    kps, desc = ex.extract(img)
    sf = ex.tables()['sf']
    factor = np.float32(3.1415926535897932384626433832795 / 180.0)
    arrays = {}
    for l in range(8):
        k = kps[kps['octave'] == l]
        arrays['bl%d' % l] = ex.level(l, blurred=True)
        arrays['rc%d' % l] = np.stack([np.rint(k['y'] / sf[l]), np.rint(k['x'] / sf[l])], 1).astype(np.int64)
        arrays['an%d' % l] = (k['angle'].astype(np.float32) * factor).astype(np.float64)
    d = _run39('''
        import numpy as np, skimage
        from skimage.feature.orb_cy import _orb_loop
        for l in range(8):
            img = np.ascontiguousarray(np.load('bl%d.npy' % l).astype(np.float64))
            rc = np.ascontiguousarray(np.load('rc%d.npy' % l).astype(np.intp))
            bits = np.asarray(_orb_loop(img, rc, np.ascontiguousarray(np.load('an%d.npy' % l)))).astype(np.uint8)
            np.save('bits%d.npy' % l, bits.reshape(len(rc), 256))
        print('scikit-image', skimage.__version__)
        ''', tmp_path, **arrays)
    nbits = nbad = ndesc = nexact = 0
    for l in range(8):
        want = np.load(d / ('bits%d.npy' % l))
        got = np.unpackbits(desc[kps['octave'] == l], axis=1, bitorder='little')
        bad = (want != got).sum(1)
        nbits += want.size; nbad += int(bad.sum()); ndesc += len(bad); nexact += int((bad == 0).sum())
    print('descriptors %d, identical %d, differing bits %d of %d' % (ndesc, nexact, nbad, nbits))
    assert ndesc > 500
    assert nbad <= nbits * 2e-4 and nexact >= 0.97 * ndesc


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['textured', 'photograph'])
def test_product_against_third_parties_directly(tmp_path, which):
    """The HIP path with NO oracle in between: keypoints, angles and descriptors of a GPU extraction against scikit-image / scipy
    evaluated on the product's own pyramid levels -- every keypoint passes scikit-image's FAST-9 test with response = the
    largest passing threshold, its angle is corner_orientations' within fastAtan2's 0.3 degrees, and its descriptor is what
    scikit-image's steered-BRIEF loop reads from scipy's integer Gaussian of the level.  `photograph`: scikit-image's own `camera`
    sample image (tests/golden/natural_images.npz), where a fifth of the keypoints come from the minThFAST pass."""
    from os1_amd import api
    assert api.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    if which == 'textured':
        img = np.kron(_textured(21, 120, 160), np.ones((4, 4), np.uint8))
        img = (img.astype(np.int32) + np.random.RandomState(4).randint(-8, 9, img.shape)).clip(0, 255).astype(np.uint8)
    else:
        img = np.ascontiguousarray(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'natural_images.npz'))['camera'])
    ex = api.Extractor(1200, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    sf = ex.tables()['sf']
    assert len(kps) > 800 and len(set(kps['octave'].tolist())) == 8
    if which == 'photograph':
        assert (kps['response'] < 20).sum() > 50      # keypoints that only the minThFAST pass of their cell produced
    factor = np.float32(3.1415926535897932384626433832795 / 180.0)
    arrays = {}
    for l in range(8):
        k = kps[kps['octave'] == l]
        arrays['lv%d' % l] = ex.level(l)
        arrays['rc%d' % l] = np.stack([np.rint(k['y'] / sf[l]), np.rint(k['x'] / sf[l])], 1).astype(np.int64)
        arrays['an%d' % l] = (k['angle'].astype(np.float32) * factor).astype(np.float64)
    d = _run39('''
        import numpy as np, skimage
        from scipy import ndimage
        from skimage.feature import corner_fast, corner_orientations
        from skimage.feature.orb import OFAST_MASK
        from skimage.feature.orb_cy import _orb_loop
        k7 = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
        for l in range(8):
            lv = np.load('lv%d.npy' % l)
            rc = np.ascontiguousarray(np.load('rc%d.npy' % l).astype(np.intp))
            f = lv.astype(np.float64)
            # largest threshold each keypoint still passes at (monotone: bisect per keypoint over full-image runs)
            score = np.zeros(len(rc), np.int32)
            for t in range(1, 255):
                m = corner_fast(f, n=9, threshold=float(t))[rc[:, 0], rc[:, 1]] > 0
                if not m.any():
                    break
                score[m] = t
            np.save('score%d.npy' % l, score)
            np.save('ang%d.npy' % l, np.rad2deg(corner_orientations(f, rc, OFAST_MASK)))
            blur = ((ndimage.correlate(lv.astype(np.int64), np.outer(k7, k7), mode='mirror') + 32768) >> 16).astype(np.float64)
            bits = np.asarray(_orb_loop(np.ascontiguousarray(blur), rc, np.ascontiguousarray(np.load('an%d.npy' % l))))
            np.save('bits%d.npy' % l, bits.astype(np.uint8).reshape(len(rc), 256))
        print('scikit-image', skimage.__version__)
        ''', tmp_path, **arrays)
    ndesc = 0
    for l in range(8):
        sel = kps['octave'] == l
        k = kps[sel]
        score = np.load(d / ('score%d.npy' % l))
        assert (score >= 7).all(), 'level %d: a keypoint that is no FAST-9 corner at the minimum threshold' % l
        assert np.array_equal(score.astype(np.float32), k['response']), 'level %d: response is not the largest passing threshold' % l
        want = np.load(d / ('ang%d.npy' % l)) % 360.0
        diff = np.abs((k['angle'].astype(np.float64) - want + 180.0) % 360.0 - 180.0)
        assert diff.max() <= 0.3, (l, diff.max())
        assert np.array_equal(np.unpackbits(desc[sel], axis=1, bitorder='little'), np.load(d / ('bits%d.npy' % l))), 'level %d' % l
        ndesc += len(k)
    print('GPU keypoints checked against scikit-image / scipy:', ndesc)


def test_gray_conversion_against_pillow(oracle, tmp_path):
    """cvtColor(RGB2GRAY / BGR2GRAY) (Tracking.cc:96-109): which channel takes which of the ITU-R 601 weights, against Pillow's
    convert('L') (16-bit fixed point where OpenCV uses 14 or 15 bits: the values agree within one gray level; a swapped channel
    order differs by up to 47)."""
    rgb = np.random.RandomState(3).randint(0, 256, (48, 64, 3)).astype(np.uint8)
    rgb[:8] = [255, 0, 0]; rgb[8:16] = [0, 255, 0]; rgb[16:24] = [0, 0, 255]
    d = _run39('''
        import numpy as np, PIL
        from PIL import Image
        np.save('gray.npy', np.asarray(Image.fromarray(np.load('rgb.npy'), 'RGB').convert('L')))
        print('Pillow', PIL.__version__)
        ''', tmp_path, rgb=rgb)
    want = np.load(d / 'gray.npy').astype(np.int32)
    for variant in (0, 1):
        got = oracle.cvt_gray(rgb, rgb_order=True, variant=variant).astype(np.int32)
        assert np.abs(got - want).max() <= 1
        got = oracle.cvt_gray(np.ascontiguousarray(rgb[..., ::-1]), rgb_order=False, variant=variant).astype(np.int32)
        assert np.abs(got - want).max() <= 1
    assert np.abs(oracle.cvt_gray(rgb, rgb_order=False).astype(np.int32) - want).max() > 30   # the check has teeth


def test_committed_thirdparty_vectors(oracle):
    """The same third-party checks from COMMITTED vectors (tests/golden/thirdparty_vectors.npz <- tools/gen_thirdparty_golden.py, which
    runs scikit-image 0.18.3 / scipy 1.7.1 under the image's Anaconda python3.9): they hold wherever that interpreter is absent."""
    import hashlib
    from oracle.pyoracle import OracleExtractor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, 'tests', 'golden', 'thirdparty_vectors.npz'))
    img = g['img']
    assert np.array_equal(img, _textured(7, 72, 88))
    want = g['fast_score']
    for th in (1, 7, 20, 45):
        got = np.zeros_like(want)
        k = oracle.fast9(img, th, nms=False)
        got[k[:, 1], k[:, 0]] = k[:, 2]
        assert np.array_equal(got, np.where(want >= th, want, 0)), th
    assert np.array_equal(oracle.gauss7(img), g['blur'])
    big = np.kron(_textured(9, 96, 128), np.ones((4, 4), np.uint8))
    big = (big.astype(np.int32) + np.random.RandomState(2).randint(-8, 9, big.shape)).clip(0, 255).astype(np.uint8)
    assert hashlib.sha256(big.tobytes()).hexdigest() == str(g['big_sha256'][0])
    kps, desc = OracleExtractor(800, 1.2, 8, 20, 7, oracle).extract(big)
    n = 0
    for l in range(8):
        sel = kps['octave'] == l
        assert np.array_equal(desc[sel], g['bits%d' % l]), 'level %d: descriptors differ from scikit-image steered BRIEF' % l
        diff = np.abs((kps[sel]['angle'].astype(np.float64) - g['orient%d' % l] % 360.0 + 180.0) % 360.0 - 180.0)
        assert diff.max() <= 0.3, (l, diff.max())
        n += int(sel.sum())
    assert n == len(kps) > 500


@pytest.mark.gpu
def test_product_against_committed_thirdparty_vectors():
    """The HIP path against the COMMITTED scikit-image vectors (no oracle, no second interpreter): descriptors bit for bit, orientations
    within fastAtan2's 0.3 degrees, for every keypoint of an 8-level extraction."""
    from os1_amd import api
    assert api.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, 'tests', 'golden', 'thirdparty_vectors.npz'))
    big = np.kron(_textured(9, 96, 128), np.ones((4, 4), np.uint8))
    big = (big.astype(np.int32) + np.random.RandomState(2).randint(-8, 9, big.shape)).clip(0, 255).astype(np.uint8)
    kps, desc = api.Extractor(800, 1.2, 8, 20, 7)(big)
    n = 0
    for l in range(8):
        sel = kps['octave'] == l
        assert np.array_equal(desc[sel], g['bits%d' % l]), 'level %d' % l
        diff = np.abs((kps[sel]['angle'].astype(np.float64) - g['orient%d' % l] % 360.0 + 180.0) % 360.0 - 180.0)
        assert diff.max() <= 0.3, (l, diff.max())
        n += int(sel.sum())
    assert n == len(kps) > 500
