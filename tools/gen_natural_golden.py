#!/usr/bin/env python3
"""tests/golden/natural_images.npz: photographs for the parity tests (every other parity input of this repository is synthetic:
value noise + rectangles + discs, or a degenerate pattern).  Real images have what those lack -- smooth gradients, defocus, large
textureless regions where whole pyramid levels miss their quota and most FAST cells fall through to minThFAST
(/root/reference/src/ORBextractor.cc:826-870, 571-795).

Source: scikit-image's bundled sample data (skimage.data, version printed below), read from the local package -- no download.
Licences as stated by scikit-image's data registry / README of those files:
  camera      -- 512x512 gray, "cameraman" replacement photo by Lav Varshney, CC0
  coins       -- 303x384 gray, Greek coins from Pompeii, Brooklyn Museum collection, "no known copyright restrictions"
  astronaut   -- 512x512 RGB, NASA photograph of Eileen Collins, public domain (NASA imagery); stored here as gray (0.299 R + 0.587 G +
                 0.114 B, rounded) and as the RGB original's top-left 256x256 (for the colour-input path)
  moon        -- 512x512 gray, low contrast, large dark area; public domain (NASA / USGS imagery)
  dark_crop   -- derived here: astronaut gray rows 0..383, cols 128..511, divided by 6 (an under-exposed frame, gray levels 0..42)
Only pixel data is stored (a fixture is data); the generator is this file.

Run with an interpreter that has scikit-image:   /opt/conda/bin/python3.9 tools/gen_natural_golden.py"""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import skimage
    from skimage import data
    cam = np.ascontiguousarray(data.camera(), np.uint8)
    coins = np.ascontiguousarray(data.coins(), np.uint8)
    astro = np.ascontiguousarray(data.astronaut(), np.uint8)
    moon = np.ascontiguousarray(data.moon(), np.uint8)
    a = astro.astype(np.float64)
    gray = np.clip(np.rint(0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]), 0, 255).astype(np.uint8)
    dark = np.ascontiguousarray(gray[:384, 128:] // 6)
    out = {'camera': cam, 'coins': coins, 'astronaut_gray': gray, 'astronaut_rgb_tl': np.ascontiguousarray(astro[:256, :256]),
           'moon': moon, 'dark_crop': dark}
    path = os.path.join(ROOT, 'tests', 'golden', 'natural_images.npz')
    np.savez_compressed(path, source=np.array('scikit-image %s skimage.data (local package files)' % skimage.__version__), **out)
    for k, v in out.items():
        print('%-18s %-14s mean %6.1f  sha256 %s' % (k, v.shape, v.mean(), hashlib.sha256(v.tobytes()).hexdigest()[:16]))
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
