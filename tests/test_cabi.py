"""The C-ABI library loads and exports every symbol include/orbfe.h declares (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from os1_amd import api
    if not os.path.exists(api.lib_path()):
        api.build_library()
    return api.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'orbfe.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = sorted(set(re.findall(r'\b(orbfe_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'liborbfe.so does not export %s' % n


def test_keypoint_layout():
    from os1_amd.api import KP_DTYPE
    assert KP_DTYPE.itemsize == 28          # sizeof(cv::KeyPoint)
    assert [KP_DTYPE.fields[f][1] for f in ('x', 'y', 'size', 'angle', 'response', 'octave', 'class_id')] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_argument_validation_and_no_fallback(lib):
    from os1_amd import api
    h = ctypes.c_void_p()
    assert lib.orbfe_extractor_create(0, 1.2, 8, 20, 7, 0, ctypes.byref(h)) == -1      # nfeatures <= 0
    assert lib.orbfe_extractor_create(1000, 1.0, 8, 20, 7, 0, ctypes.byref(h)) == -1   # scale <= 1
    assert lib.orbfe_extractor_create(1000, 1.2, 99, 20, 7, 0, ctypes.byref(h)) == -1  # too many levels
    assert b'invalid' in lib.orbfe_last_error()
    if api.device_count() == 0:
        # CPU-only box: the product must fail loudly, never fall back
        with pytest.raises(api.OrbfeError) as e:
            api.Extractor(1000, 1.2, 8, 20, 7)
        assert e.value.code == -2
        with pytest.raises(api.OrbfeError):
            api.Matcher()


def test_hamming_host_helper():
    from os1_amd import api
    rng = np.random.default_rng(0)
    z = np.zeros(32, np.uint8)
    assert api.hamming(z, z) == 0 and api.hamming(z, ~z) == 256
    for _ in range(100):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert api.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
