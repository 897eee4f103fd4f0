"""Photographs (tests/golden/natural_images.npz <- tools/gen_natural_golden.py: scikit-image's bundled sample images, licences in
that file's header) through the path.  Every other parity input is synthetic; these have smooth gradients, defocus and large
textureless regions, so that whole pyramid levels miss their quota (`mnFeaturesPerLevel`, /root/reference/src/ORBextractor.cc:467-478,
the `vToDistributeKeys` they get is smaller than N) and most FAST cells fall through to minThFAST (ORBextractor.cc:846-856).
`-m "not gpu"`: the fixture is what its generator wrote and the oracle shows those regimes on it.  `-m gpu`: the HIP path against the
oracle, stage by stage -- pyramid levels, per-level candidate lists, keypoints, descriptors -- byte-exact, at N = 1000 and 2000, on the
images as they are and upscaled 2.5x (>= 1280 px on the long side where the original has 512)."""
import hashlib
import os

import numpy as np
import pytest

from oracle.pyoracle import OracleExtractor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ['camera', 'coins', 'astronaut_gray', 'moon', 'dark_crop']
SHA16 = {'camera': '5cb24482a53416f9', 'coins': 'e080cc03805f1fa7', 'astronaut_gray': 'd4eb846291f30fd8',
         'astronaut_rgb_tl': '297abd13e1331e86', 'moon': 'a20362266d5b0102', 'dark_crop': '0d012fb8d49ca5f7'}


@pytest.fixture(scope='module')
def images():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'natural_images.npz'))
    return {k: np.ascontiguousarray(z[k]) for k in SHA16}


def _upscaled(oracle, img, f=2.5):
    """cv::resize(INTER_LINEAR) restatement of the oracle: a deterministic, smooth 2.5x enlargement (defocus-like content)."""
    return oracle.resize(img, int(img.shape[1] * f), int(img.shape[0] * f))


def _regimes(ox, kps):
    nf = ox.tables()['nfeat']
    per = [int((kps['octave'] == l).sum()) for l in range(8)]
    cands = [ox.candidates(l) for l in range(8)]
    below = [l for l in range(8) if per[l] < nf[l]]
    # a candidate with response < iniThFAST can only have come from the minThFAST pass of a cell that was empty at iniThFAST
    from_min = sum(int((c['response'] < 20).sum()) for c in cands)
    return below, from_min, sum(len(c) for c in cands)


def test_fixture_is_the_generators_output_and_reaches_the_regimes(images, oracle):
    for k, v in images.items():
        assert v.dtype == np.uint8 and hashlib.sha256(v.tobytes()).hexdigest()[:16] == SHA16[k], k
    ox = OracleExtractor(2000, 1.2, 8, 20, 7, oracle)
    k, _ = ox.extract(images['moon'])
    below, from_min, total = _regimes(ox, k)
    assert below == [3, 4, 5, 6, 7] and from_min > total // 2 and len(k) < 1700      # five levels short of their quota
    k, _ = ox.extract(images['dark_crop'])
    below, from_min, total = _regimes(ox, k)
    assert below == list(range(8)) and from_min >= 0.95 * total and 0 < len(k) < 900  # an under-exposed frame: nearly everything from minThFAST
    k, _ = ox.extract(images['camera'])
    below, from_min, total = _regimes(ox, k)
    assert below == [] and 0 < from_min < total // 4 and len(k) >= 2000               # a textured photograph saturates every level


@pytest.fixture(scope='module')
def api():
    from os1_amd import api as a
    assert a.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    return a


def _stages_equal(ex, ox, img):
    got, want = ex(img), ox.extract(img)
    for l in range(8):
        assert (ex.level(l) == ox.level(l)).all(), 'pyramid level %d' % l
        c, oc = ex.candidates(l), ox.candidates(l)
        assert len(c) == len(oc), 'level %d candidate count' % l
        assert (c[:, 0] == oc['x']).all() and (c[:, 1] == oc['y']).all() and (c[:, 2] == oc['response']).all(), 'level %d candidates' % l
    (gk, gd), (wk, wd) = got, want
    assert len(gk) == len(wk)
    for f in gk.dtype.names:
        assert (gk[f] == wk[f]).all(), f
    assert gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes()
    return want[0]


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_photographs_stage_by_stage(api, oracle, images, name):
    seen_below = seen_min = 0
    for img in (images[name], _upscaled(oracle, images[name])):
        for N in (1000, 2000):
            ex = api.Extractor(N, 1.2, 8, 20, 7)
            ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
            k = _stages_equal(ex, ox, img)
            below, from_min, _ = _regimes(ox, k)
            seen_below += len(below)
            seen_min += from_min
    assert seen_min > 0                                  # every photograph has cells that only answer at minThFAST
    if name in ('moon', 'dark_crop'):
        assert seen_below >= 5                           # ... and these two leave levels below their quota


@pytest.mark.gpu
def test_photograph_batch_and_colour_routes(api, oracle, images):
    """The same photographs as members of a batch (the throughput route: k_resize_fixed chain, global-memory quadtree) and the RGB
    crop through the colour-input route (Tracking.cc:96-109)."""
    ups = [_upscaled(oracle, images[n][:384, :384]) for n in ('camera', 'astronaut_gray', 'moon', 'dark_crop')]
    ex = api.Extractor(1500, 1.2, 8, 20, 7)
    ox = OracleExtractor(1500, 1.2, 8, 20, 7, oracle)
    for (gk, gd), im in zip(ex.extract_batch(ups), ups):
        wk, wd = ox.extract(im)
        assert gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes()
    rgb = images['astronaut_rgb_tl']
    for variant in (0, 1):
        gray = oracle.cvt_gray(rgb, True, variant)
        ex.set_input_format('rgb', variant)
        gk, gd = ex.extract_color(rgb)
        wk, wd = ox.extract(gray)
        assert gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes() and len(wk) > 300
