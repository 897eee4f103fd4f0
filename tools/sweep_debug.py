"""Exploratory parity sweep (GPU box): random sizes / parameters, product vs oracle.  usage: sweep_debug.py seed count [blur_variant=0]
(blur_variant 1: the OpenCV 4.0-era GaussianBlur on both sides, orbfe_extractor_set_blur_variant / orc_extractor_set_gauss_variant)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from os1_amd import api
from os1_amd.synth import synth
from oracle.pyoracle import Oracle, OracleExtractor

seed, count = int(sys.argv[1]), int(sys.argv[2])
blur = int(sys.argv[3]) if len(sys.argv) > 3 else 0
oracle = Oracle()
rng = np.random.default_rng(seed)
bad = 0
for it in range(count):
    W = int(rng.integers(97, 1400))
    H = min(int(rng.integers(97, 1000)), int(1.7 * W))
    N = int(rng.integers(30, 3000))
    sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    ini, mn = [(20, 7), (7, 20), (12, 12), (40, 5), (9, 3), (100, 60), (1, 1)][it % 7]
    while min(W, H) / sf ** (nl - 1) < 66:
        nl -= 1
    # the exact level sizes (cvRound(cols * invScale), float32 as ORBextractor.cc:975-976) and the reference's nIni =
    # round(width / height) (ORBextractor.cc:574): nIni == 0 is a division by zero in the reference AND in its restatement
    isf = OracleExtractor(N, sf, nl, ini, mn, oracle).tables()['isf']
    def n_ini(l):
        w, h = int(np.rint(np.float32(W) * isf[l])), int(np.rint(np.float32(H) * isf[l]))
        return int(np.floor(np.float32(w - 32) / np.float32(h - 32) + np.float32(0.5))) if h > 32 else 0
    if any(n_ini(l) < 1 for l in range(nl)):
        print(it, 'skipped: a level has no quadtree root (the reference divides by zero)')
        continue
    img = synth(seed * 1000 + it, W, H)
    mode = int(rng.integers(0, 5))
    if mode == 1:
        img = (128 + (img.astype(np.int32) - 128) // 6).astype(np.uint8)
    if mode == 2:
        img[: H // 2, : W // 2] = 90
    if mode == 3:
        img = rng.integers(0, 256, img.shape, dtype=np.uint8)          # pure noise: a corner almost everywhere
    if mode == 4:
        img = np.clip(img.astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)   # saturated
    try:
        ex = api.Extractor(N, sf, nl, ini, mn)
        ex.set_blur_variant(blur)
    except api.OrbfeError as e:
        print(it, 'create refused:', e)
        continue
    ox = OracleExtractor(N, sf, nl, ini, mn, oracle)
    ox.set_gauss_variant(blur)
    wk, wd = ox.extract(img)
    gk, gd = ex(img)
    ok = gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes()
    # the BATCH route too (per-level resize launches with LDS-DMA staging, FAST launched per LDS class): four frames, from
    # host memory and from device memory with a row stride that is / is not a multiple of 4
    imgs = [img, np.ascontiguousarray(img[::-1]), np.ascontiguousarray(img[:, ::-1]), img]
    want = [(wk, wd), ox.extract(imgs[1]), ox.extract(imgs[2]), (wk, wd)]
    okb = True
    cap = max(ex.cap, ex.L.orbfe_extractor_max_keypoints_for_size(ex.h, H, W))   # (strips wider than 4.5 : 1 have more quadtree roots)
    kb, db = np.zeros((4, cap), api.KP_DTYPE), np.zeros((4, cap, 32), np.uint8)
    kps, desc, n = ex.extract_batch_ptrs([im.ctypes.data for im in imgs], H, W, W, False, kb, db)
    for i, w in enumerate(want):
        okb = okb and kps[i, :n[i]].tobytes() == w[0].tobytes() and desc[i, :n[i]].tobytes() == w[1].tobytes()
    stride = W + int(rng.choice([0, 3, 4, 61, 64]))
    dev = api.DeviceFrames(imgs, 0, stride=stride)
    kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, H, W, stride, True, kb, db)
    for i, w in enumerate(want):
        okb = okb and kps[i, :n[i]].tobytes() == w[0].tobytes() and desc[i, :n[i]].tobytes() == w[1].tobytes()
    dev.free()
    # one- and two-frame calls from page-locked memory (k_ingest where the alignment allows it, the copy engine otherwise) and from a
    # buffer of the caller's that was registered in place (orbfe_host_register), at a random byte offset
    pin = api.PinnedFrames(imgs[:2])
    kb1, db1, kb2, db2 = kb[:1].copy(), db[:1].copy(), kb[:2].copy(), db[:2].copy()
    k1, d1, n1 = ex.extract_batch_ptrs(pin.ptrs[:1], H, W, W, False, kb1, db1)
    okb = okb and k1[0, :n1[0]].tobytes() == want[0][0].tobytes() and d1[0, :n1[0]].tobytes() == want[0][1].tobytes()
    k2, d2, n2 = ex.extract_batch_ptrs(pin.ptrs, H, W, W, False, kb2, db2)
    for i in range(2):
        okb = okb and k2[i, :n2[i]].tobytes() == want[i][0].tobytes() and d2[i, :n2[i]].tobytes() == want[i][1].tobytes()
    pin.free()
    off = int(rng.choice([0, 4, 16, 1, 2]))
    ring = np.zeros(2 * W * H + 64, np.uint8)
    reg = api.RegisteredArray(ring)
    view = ring[off:off + 2 * W * H].reshape(2, H, W)
    view[:] = np.stack(imgs[:2])
    k2, d2, n2 = ex.extract_batch_ptrs([view[0].ctypes.data, view[1].ctypes.data], H, W, W, False, kb2, db2)
    for i in range(2):
        okb = okb and k2[i, :n2[i]].tobytes() == want[i][0].tobytes() and d2[i, :n2[i]].tobytes() == want[i][1].tobytes()
    k1, d1, n1 = ex.extract_batch_ptrs([view[1].ctypes.data], H, W, W, False, kb1, db1)
    okb = okb and k1[0, :n1[0]].tobytes() == want[1][0].tobytes() and d1[0, :n1[0]].tobytes() == want[1][1].tobytes()
    reg.close()
    print(it, W, H, N, sf, nl, ini, mn, mode, len(wk), 'OK' if ok else 'MISMATCH', 'batch', stride - W, 'OK' if okb else 'MISMATCH', flush=True)
    bad += (not ok) + (not okb)
print('mismatches', bad)
