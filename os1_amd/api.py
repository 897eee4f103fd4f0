"""ctypes binding of liborbfe.so (include/orbfe.h).  Mirrors the reference's operator interface:
Extractor(...)(image) ~ ORBextractor::operator(), Matcher.search_for_initialization(...) ~
ORBmatcher::SearchForInitialization, Matcher.search_by_projection(...) ~ SearchByProjection.
No compute happens in Python and there is no fallback: if the library or a GPU is missing the
constructors raise."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get('ORBFE_LIB') or os.path.join(_HERE, 'liborbfe.so')   # ORBFE_LIB: an A/B build of the library (developer aid)

KP_DTYPE = np.dtype([('x', 'f4'), ('y', 'f4'), ('size', 'f4'), ('angle', 'f4'), ('response', 'f4'),
                     ('octave', 'i4'), ('class_id', 'i4')])

MP_IN_VIEW, MP_BAD, MP_CANDIDATO, MP_OBSERVED = 1, 2, 4, 8


class OrbfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('orbfe error %d: %s' % (code, msg))
        self.code = code


def lib_path():
    return _SO


def build_library(force=False):
    """Compile liborbfe.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ['make', '-s', '-C', os.path.join(_HERE, 'csrc')]
    if force:
        subprocess.check_call(args + ['clean'])
    subprocess.check_call(args)
    return _SO


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise OrbfeError(-2, 'liborbfe.so is not built (run __graft_entry__.build() or make -C os1_amd/csrc)')
    L = C.CDLL(_SO)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_device_count.restype = ci
    L.orbfe_extractor_create.argtypes = [ci, cf, ci, ci, ci, ci, C.POINTER(vp)]
    L.orbfe_extractor_destroy.argtypes = [vp]
    L.orbfe_extractor_destroy.restype = None
    L.orbfe_extractor_levels.argtypes = [vp]
    L.orbfe_extractor_scale_factor.argtypes = [vp]
    L.orbfe_extractor_scale_factor.restype = cf
    L.orbfe_extractor_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbfe_extractor_features_per_level.argtypes = [vp, vp]
    L.orbfe_extractor_max_keypoints.argtypes = [vp]
    L.orbfe_extract.argtypes = [vp, vp, ci, ci, C.c_size_t, vp, vp, ci, C.POINTER(ci)]
    L.orbfe_extract_batch.argtypes = [vp, ci, vp, ci, ci, ci, C.c_size_t, vp, vp, ci, vp]
    L.orbfe_extract_batch_submit.argtypes = [vp, ci, vp, ci, ci, ci, C.c_size_t]
    L.orbfe_extract_batch_wait.argtypes = [vp]
    L.orbfe_extract_batch_collect.argtypes = [vp, vp, vp, ci, vp]
    L.orbfe_debug_level_size.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
    L.orbfe_debug_level_copy.argtypes = [vp, ci, ci, vp]
    L.orbfe_debug_candidates.argtypes = [vp, ci, ci, vp, ci, C.POINTER(ci)]
    L.orbfe_debug_stage_ms.argtypes = [vp, vp]
    L.orbfe_debug_sincos.argtypes = [vp, vp, ci, vp, vp]
    L.orbfe_hamming.argtypes = [vp, vp]
    L.orbfe_matcher_create.argtypes = [ci, C.POINTER(vp)]
    L.orbfe_matcher_destroy.argtypes = [vp]
    L.orbfe_matcher_destroy.restype = None
    L.orbfe_search_for_initialization.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, ci, cf, ci, C.POINTER(ci)]
    L.orbfe_search_for_initialization_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, cf, ci, vp]
    L.orbfe_frame_create.argtypes = [vp, vp, vp, ci, vp, C.POINTER(vp)]
    L.orbfe_frame_create_from_extract.argtypes = [vp, ci, vp, vp, C.POINTER(vp)]
    L.orbfe_frame_destroy.argtypes = [vp]
    L.orbfe_frame_destroy.restype = None
    L.orbfe_frame_size.argtypes = [vp]
    L.orbfe_frame_descriptors_device.argtypes = [vp]
    L.orbfe_frame_descriptors_device.restype = vp
    L.orbfe_frame_download.argtypes = [vp, vp, vp, vp, vp]
    L.orbfe_search_by_projection_frame.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, C.POINTER(ci)]
    L.orbfe_search_by_projection_frame_rows.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, cf, vp,
                                                        C.POINTER(ci)]
    L.orbfe_matcher_upload_async.argtypes = [vp, vp, vp, C.c_size_t]
    L.orbfe_matcher_synchronize.argtypes = [vp]
    L.orbfe_matcher_device.argtypes = [vp]
    L.orbfe_extractor_device.argtypes = [vp]
    L.orbfe_frame_device.argtypes = [vp]
    L.orbfe_resident_epoch.restype = C.c_ulonglong
    L.orbfe_resident_invalidate.restype = C.c_ulonglong
    L.orbfe_search_by_projection_uv_frame.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, ci, ci, ci, vp,
                                                      C.POINTER(ci)]
    L.orbfe_search_projected_frame.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, ci, C.c_double, ci, vp, vp,
                                               C.POINTER(ci)]
    L.orbfe_debug_resolve_rounds.argtypes = [vp]
    L.orbfe_debug_resolve_route.argtypes = [vp]
    L.orbfe_debug_resolve_phases.argtypes = [vp, vp]
    L.orbfe_search_by_projection.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp,
                                             C.POINTER(ci)]
    L.orbfe_search_by_projection_uv.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, cf, ci,
                                                ci, ci, vp, C.POINTER(ci)]
    L.orbfe_distinctive_descriptors.argtypes = [vp, ci, vp, vp, vp]
    L.orbfe_window_candidates.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
    L.orbfe_undistort_equidistant.argtypes = [vp, ci, cf, cf, cf, cf]
    L.orbfe_undistort_pinhole.argtypes = [vp, ci, cf, cf, cf, cf, vp, ci]
    L.orbfe_compute_image_bounds.argtypes = [ci, ci, ci, cf, cf, cf, cf, vp, ci, vp]
    L.orbfe_search_projected.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, ci, C.c_double, ci, vp, vp,
                                         C.POINTER(ci)]
    L.orbfe_search_for_triangulation.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, cf, cf,
                                                 vp, vp, ci, ci, vp, C.POINTER(ci)]
    L.orbfe_extractor_max_keypoints_for_size.argtypes = [vp, ci, ci]
    L.orbfe_extractor_set_vocabulary.argtypes = [vp, vp, ci]
    L.orbfe_extract_bow_raw.argtypes = [vp, ci, vp, vp, ci, C.POINTER(ci)]
    L.orbfe_bow_assemble.argtypes = [vp, vp, vp, ci, vp, vp, C.POINTER(ci), vp, vp, vp, C.POINTER(ci), vp]
    L.orbfe_stream_set_vocabulary.argtypes = [vp, vp, ci]
    L.orbfe_stream_bow_raw.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci)]
    L.orbfe_extract_bow.argtypes = [vp, ci, vp, vp, C.POINTER(ci), vp, vp, vp, C.POINTER(ci), vp, vp]
    L.orbfe_extractor_set_input_format.argtypes = [vp, ci, ci]
    L.orbfe_stream_set_input_format.argtypes = [vp, ci, ci]
    L.orbfe_extractor_set_blur_variant.argtypes = [vp, ci]
    L.orbfe_stream_set_blur_variant.argtypes = [vp, ci]
    L.orbfe_vocabulary_create.argtypes = [ci, ci, ci, ci, ci, vp, ci, C.POINTER(vp)]
    L.orbfe_vocabulary_create_from_image.argtypes = [ci, vp, C.c_size_t, C.POINTER(vp)]
    L.orbfe_vocabulary_destroy.argtypes = [vp]
    L.orbfe_vocabulary_destroy.restype = None
    L.orbfe_vocabulary_info.argtypes = [vp] + [C.POINTER(ci)] * 6
    L.orbfe_bow_transform.argtypes = [vp, vp, ci, ci, ci, vp, vp, C.POINTER(ci), vp, vp, vp, C.POINTER(ci), vp, vp]
    L.orbfe_search_by_bow.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, cf, ci, ci, vp,
                                      C.POINTER(ci)]
    L.orbfe_search_by_bow_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, cf, ci, ci, vp, vp]
    L.orbfe_debug_matcher_ms.argtypes = [vp, vp]
    L.orbfe_debug_features_in_area.argtypes = [vp, vp, ci, vp, cf, cf, cf, ci, ci, vp, ci, C.POINTER(ci)]
    L.orbfe_debug_kernel_ms.argtypes = [vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), ci]
    L.orbfe_debug_set_profiling.argtypes = [vp, ci]
    L.orbfe_device_malloc.argtypes = [ci, C.c_size_t, C.POINTER(vp)]
    L.orbfe_device_free.argtypes = [ci, vp]
    L.orbfe_device_upload.argtypes = [ci, vp, vp, C.c_size_t]
    L.orbfe_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.orbfe_host_free.argtypes = [vp]
    L.orbfe_host_register.argtypes = [vp, C.c_size_t]
    L.orbfe_host_unregister.argtypes = [vp]
    L.orbfe_device_synchronize.argtypes = [ci]
    L.orbfe_device_numa_node.argtypes = [ci]
    L.orbfe_bind_thread_to_device.argtypes = [ci]
    L.orbfe_debug_h2d_rate.argtypes = [ci, vp, C.c_size_t, ci, C.POINTER(C.c_double)]
    L.orbfe_stream_create.argtypes = [ci, cf, ci, ci, ci, ci, ci, ci, C.POINTER(vp)]
    L.orbfe_stream_destroy.argtypes = [vp]
    L.orbfe_stream_destroy.restype = None
    L.orbfe_stream_set_matching.argtypes = [vp, vp, ci, cf, ci]
    L.orbfe_stream_capacity.argtypes = [vp]
    L.orbfe_stream_set_queue_slots.argtypes = [vp, ci]
    L.orbfe_stream_queue_slots.argtypes = [vp]
    L.orbfe_stream_push.argtypes = [vp, vp, ci, ci, ci, C.c_size_t]
    L.orbfe_stream_pop.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.orbfe_stream_set_isolated_batches.argtypes = [vp, ci]
    L.orbfe_stream_batches_in_flight.argtypes = [vp]
    L.orbfe_stream_multi_create.argtypes = [ci, cf, ci, ci, ci, C.POINTER(ci), ci, ci, ci, C.POINTER(vp)]
    L.orbfe_stream_multi_destroy.argtypes = [vp]
    L.orbfe_stream_multi_destroy.restype = None
    L.orbfe_stream_multi_devices.argtypes = [vp]
    L.orbfe_stream_multi_capacity.argtypes = [vp]
    L.orbfe_stream_multi_device_of_next_push.argtypes = [vp]
    L.orbfe_stream_multi_set_matching.argtypes = [vp, vp, ci, cf, ci]
    L.orbfe_stream_multi_set_blur_variant.argtypes = [vp, ci]
    L.orbfe_stream_multi_push.argtypes = [vp, vp, ci, ci, ci, C.c_size_t]
    L.orbfe_stream_multi_pop.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.orbfe_stream_stats.argtypes = [vp, vp, ci]
    L.orbfe_stream_kernel_ms.argtypes = [vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), ci]
    L.orbfe_debug_quadtree.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci, C.POINTER(ci)]
    L.orbfe_debug_sincos_host_check.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_longlong)]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceRows:
    """Descriptor rows [n][32] in device memory (Frame.descriptors_device()); accepted wherever the searches on resident
    frames take query descriptor rows."""

    def __init__(self, ptr, n):
        self.ptr, self.n = int(ptr), int(n)

    def __len__(self):
        return self.n


def _rows(a):
    """(argument for the C call, object to keep alive) for query descriptor rows: numpy array or DeviceRows."""
    if isinstance(a, DeviceRows):
        return C.c_void_p(a.ptr), a
    a = np.ascontiguousarray(a, np.uint8)
    return _p(a), a


def _check(rc):
    if rc != 0:
        raise OrbfeError(rc, load_library().orbfe_last_error().decode('utf-8', 'replace'))


def device_count():
    return load_library().orbfe_device_count()


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return load_library().orbfe_hamming(_p(a), _p(b))


class Extractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on one GPU."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, device=0):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.orbfe_extractor_create(nfeatures, scale, nlevels, ini_th, min_th, device, C.byref(h)))
        self.h = h
        self.nlevels = nlevels
        self.cap = self.L.orbfe_extractor_max_keypoints(h)

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_extractor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        _check(self.L.orbfe_extractor_scale_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2)))
        _check(self.L.orbfe_extractor_features_per_level(self.h, _p(nf)))
        return dict(sf=sf, isf=isf, s2=s2, is2=is2, nfeat=nf)

    def __call__(self, image):
        """image: (H, W) uint8 host array -> (keypoints[KP_DTYPE], descriptors[n,32] uint8)."""
        image = np.asarray(image)
        if image.size == 0:
            n = C.c_int(0)
            _check(self.L.orbfe_extract(self.h, None, 0, 0, 0, None, None, 0, C.byref(n)))
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1
        cap = max(self.cap, self.L.orbfe_extractor_max_keypoints_for_size(self.h, image.shape[0], image.shape[1]))
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.orbfe_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0], _p(kps),
                                    _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    FORMATS = {'gray': 0, 'rgb': 1, 'bgr': 2, 'rgba': 3, 'bgra': 4}

    def set_blur_variant(self, variant):
        """0 = ORBFE_GAUSS_ED (GaussianBlur of OpenCV >= 4.1.1, the default), 1 = ORBFE_GAUSS_ROUNDED (OpenCV 4.0.0 - 4.1.0)."""
        _check(self.L.orbfe_extractor_set_blur_variant(self.h, int(variant)))

    def set_input_format(self, fmt='gray', variant=0):
        """Frames of later calls are interleaved 8-bit `fmt` pixels; variant 0 = 15-bit, 1 = 14-bit coefficients."""
        _check(self.L.orbfe_extractor_set_input_format(self.h, self.FORMATS[fmt], variant))

    def extract_color(self, image):
        """image: (H, W, 3|4) uint8 host array in the format given to set_input_format."""
        image = np.ascontiguousarray(image, np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.orbfe_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0], _p(kps),
                                    _p(desc), self.cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def set_vocabulary(self, voc, levelsup=4):
        """Fuse Frame::ComputeBoW behind the descriptor kernel (voc: api.Vocabulary or None)."""
        self._voc = voc                                  # keep it alive
        _check(self.L.orbfe_extractor_set_vocabulary(self.h, voc.h if voc is not None else None, levelsup))

    def bow(self, frame, n):
        """BoW of frame `frame` of the last collected batch (n = its keypoint count), layout of Vocabulary.transform."""
        ids = np.zeros(max(n, 1), np.uint32)
        vals = np.zeros(max(n, 1), np.float64)
        fvn = np.zeros(max(n, 1), np.uint32)
        fvo = np.zeros(n + 1, np.uint32)
        fvf = np.zeros(max(n, 1), np.uint32)
        wof = np.zeros(max(n, 1), np.uint32)
        nof = np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int(0), C.c_int(0)
        _check(self.L.orbfe_extract_bow(self.h, frame, _p(ids), _p(vals), C.byref(nw), _p(fvn), _p(fvo), _p(fvf), C.byref(nn),
                                        _p(wof), _p(nof)))
        nw, nn = nw.value, nn.value
        return ids[:nw], vals[:nw], (fvn[:nn], fvo[:nn + 1], fvf[:int(fvo[nn])]), wof[:n], nof[:n]

    def extract_batch_ptrs(self, ptrs, rows, cols, stride, on_device, kps=None, desc=None):
        """ptrs: sequence of raw addresses (host or device).  Returns (kps[B,cap], desc[B,cap,32], n[B])."""
        B = len(ptrs)
        arr = (C.c_void_p * B)(*ptrs)
        if kps is None:
            kps = np.zeros((B, self.cap), KP_DTYPE)
        if desc is None:
            desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        _check(self.L.orbfe_extract_batch(self.h, B, arr, int(on_device), rows, cols, stride, _p(kps), _p(desc),
                                          int(kps.shape[1]), _p(n)))   # cap = what the caller's buffers hold per frame
        return kps, desc, n

    def submit_ptrs(self, ptrs, rows, cols, stride, on_device):
        """Enqueue a batch (returns immediately); collect() waits for it."""
        B = len(ptrs)
        self._pending = (C.c_void_p * B)(*ptrs)          # keep the pointer array alive
        self._pendingB = B
        _check(self.L.orbfe_extract_batch_submit(self.h, B, self._pending, int(on_device), rows, cols, stride))

    def wait(self):
        """orbfe_extract_batch_wait: the GPU side of the submitted batch is done when this returns; collect() then only assembles."""
        _check(self.L.orbfe_extract_batch_wait(self.h))

    def collect(self, kps=None, desc=None):
        B = self._pendingB
        if kps is None:
            kps = np.zeros((B, self.cap), KP_DTYPE)
        if desc is None:
            desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        _check(self.L.orbfe_extract_batch_collect(self.h, _p(kps), _p(desc), self.cap, _p(n)))
        return kps, desc, n

    def extract_batch(self, images):
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        rows, cols = images[0].shape
        assert all(im.shape == (rows, cols) for im in images)
        kps, desc, n = self.extract_batch_ptrs([im.ctypes.data for im in images], rows, cols, cols, False)
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(len(images))]

    # ---- stage accessors (parity tests) -------------------------------------------------------
    def level(self, level, frame=0):
        w, h = C.c_int(), C.c_int()
        _check(self.L.orbfe_debug_level_size(self.h, level, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(self.L.orbfe_debug_level_copy(self.h, frame, level, _p(out)))
        return out

    def candidates(self, level, frame=0):
        n = C.c_int(0)
        _check(self.L.orbfe_debug_candidates(self.h, frame, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        _check(self.L.orbfe_debug_candidates(self.h, frame, level, _p(out), n.value, C.byref(n)))
        return out[:n.value].copy()

    def stage_ms(self):
        out = np.zeros(5, np.float32)
        _check(self.L.orbfe_debug_stage_ms(self.h, _p(out)))
        return out

    def kernel_ms(self, reset=False):
        """(ms[4] = pyramid, fast, compaction, describe; batches; frames) accumulated GPU time from HIP events."""
        ms = np.zeros(5, np.float64)
        b, f = C.c_longlong(0), C.c_longlong(0)
        _check(self.L.orbfe_debug_kernel_ms(self.h, _p(ms), C.byref(b), C.byref(f), int(reset)))
        return ms, b.value, f.value

    def set_profiling(self, enable=True):
        _check(self.L.orbfe_debug_set_profiling(self.h, int(enable)))

    def sincos(self, angle_deg):
        a = np.ascontiguousarray(angle_deg, np.float32)
        c = np.zeros_like(a)
        s = np.zeros_like(a)
        _check(self.L.orbfe_debug_sincos(self.h, _p(a), a.size, _p(c), _p(s)))
        return c, s


class Frame:
    """A Frame's features resident on the GPU (orbfe_frame): mvKeysUn, mDescriptors and the grid, built once.  Pass it
    as the `kps` argument of Matcher.search_by_projection / _uv / search_projected (desc and bounds are then ignored)."""

    def __init__(self, handle, n):
        self.L = load_library()
        self.h = handle
        self.n = n

    @classmethod
    def from_host(cls, matcher, kps_un, desc, bounds):
        L = load_library()
        kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        b = np.asarray(bounds, np.float32)
        h = C.c_void_p()
        _check(L.orbfe_frame_create(matcher.h, _p(kps_un), _p(desc), len(kps_un), _p(b), C.byref(h)))
        return cls(h, len(kps_un))

    @classmethod
    def from_extract(cls, extractor, frame_index, bounds, xy_un=None):
        """Frame `frame_index` of the extractor's last collected batch, taken where the kernels left it."""
        L = load_library()
        b = np.asarray(bounds, np.float32)
        xy = None if xy_un is None else np.ascontiguousarray(xy_un, np.float32)
        h = C.c_void_p()
        _check(L.orbfe_frame_create_from_extract(extractor.h, frame_index, _p(b), None if xy is None else _p(xy), C.byref(h)))
        return cls(h, L.orbfe_frame_size(h))

    def __len__(self):
        return self.n

    def descriptors_device(self):
        """The frame's descriptor rows in device memory, keypoint order (valid while the frame lives)."""
        return DeviceRows(self.L.orbfe_frame_descriptors_device(self.h) or 0, self.n)

    def download(self):
        """(kps [x, y, angle, octave filled], desc, grid_order, cell_start) as they lie on the device."""
        k = np.zeros(max(self.n, 1), KP_DTYPE)
        d = np.zeros((max(self.n, 1), 32), np.uint8)
        order = np.zeros(max(self.n, 1), np.int32)
        cs = np.zeros(64 * 48 + 1, np.int32)
        _check(self.L.orbfe_frame_download(self.h, _p(k), _p(d), _p(order), _p(cs)))
        return k[:self.n], d[:self.n], order[:int(cs[-1])], cs

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matcher:
    """Hot subset of ORBmatcher on one GPU."""

    def frame(self, kps_un, desc, bounds):
        return Frame.from_host(self, kps_un, desc, bounds)

    def resolve_rounds(self):
        return int(self.L.orbfe_debug_resolve_rounds(self.h))

    def resolve_route(self):
        return int(self.L.orbfe_debug_resolve_route(self.h))

    def resolve_phases(self):
        """Shader-clock ticks spent in k_resolve's set-up, fixed point and output phases."""
        o = np.zeros(4, np.int32)
        _check(self.L.orbfe_debug_resolve_phases(self.h, _p(o)))
        return np.diff(o.astype(np.int64)) & 0xffffffff

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.orbfe_matcher_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_for_initialization(self, kps1, desc1, kps2, desc2, bounds, prev_xy, window=100, nnratio=0.9,
                                  check_ori=True):
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.asarray(bounds, np.float32)
        prev = np.ascontiguousarray(prev_xy, np.float32).copy()
        m12 = np.full(max(len(kps1), 1), -1, np.int32)
        n = C.c_int(0)
        _check(self.L.orbfe_search_for_initialization(self.h, _p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2),
                                                      len(kps2), _p(b), _p(prev), _p(m12), window, nnratio,
                                                      int(check_ori), C.byref(n)))
        return n.value, m12[:len(kps1)], prev

    def search_for_initialization_batch(self, pairs, bounds, window=100, nnratio=0.9, check_ori=True):
        """pairs: list of (kps1, desc1, kps2, desc2, prev_xy) with C-contiguous arrays.  Returns
        [(nmatches, matches12, prev_xy_out)] per pair; one GPU submission for all pairs."""
        P = len(pairs)
        k1 = [np.ascontiguousarray(p[0], KP_DTYPE) for p in pairs]
        d1 = [np.ascontiguousarray(p[1], np.uint8) for p in pairs]
        k2 = [np.ascontiguousarray(p[2], KP_DTYPE) for p in pairs]
        d2 = [np.ascontiguousarray(p[3], np.uint8) for p in pairs]
        prev = [np.ascontiguousarray(p[4], np.float32).copy() for p in pairs]
        m12 = [np.full(max(len(k), 1), -1, np.int32) for k in k1]
        arr = lambda xs: (C.c_void_p * P)(*[x.ctypes.data for x in xs])
        n1 = np.array([len(k) for k in k1], np.int32)
        n2 = np.array([len(k) for k in k2], np.int32)
        b = np.asarray(bounds, np.float32)
        nm = np.zeros(max(P, 1), np.int32)
        _check(self.L.orbfe_search_for_initialization_batch(self.h, P, arr(k1), arr(d1), _p(n1), arr(k2), arr(d2),
                                                            _p(n2), _p(b), arr(prev), arr(m12), window, nnratio,
                                                            int(check_ori), _p(nm)))
        return [(int(nm[i]), m12[i][:len(k1[i])], prev[i]) for i in range(P)]

    def search_by_projection(self, kps, desc, bounds, scale_factors, kp_occupied, mp_xy, mp_level, mp_viewcos,
                             mp_flags, mp_desc, th, nnratio):
        if isinstance(kps, Frame):
            sf = np.ascontiguousarray(scale_factors, np.float32)
            occ = np.ascontiguousarray(kp_occupied, np.uint8)
            mp_xy = np.ascontiguousarray(mp_xy, np.float32)
            mp_level = np.ascontiguousarray(mp_level, np.int32)
            mp_viewcos = np.ascontiguousarray(mp_viewcos, np.float32)
            mp_flags = np.ascontiguousarray(mp_flags, np.uint8)
            mp_desc_p, _keep = _rows(mp_desc)
            assigned = np.full(max(len(kps), 1), -1, np.int32)
            n = C.c_int(0)
            _check(self.L.orbfe_search_by_projection_frame(self.h, kps.h, _p(sf), len(sf), _p(occ), _p(mp_xy), _p(mp_level),
                                                           _p(mp_viewcos), _p(mp_flags), mp_desc_p, len(mp_level), th,
                                                           nnratio, _p(assigned), C.byref(n)))
            return n.value, assigned[:len(kps)]
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
        n = C.c_int(0)
        _check(self.L.orbfe_search_by_projection(self.h, _p(kps), _p(desc), len(kps), _p(b), _p(sf), len(sf), _p(occ),
                                                 _p(mp_xy), _p(mp_level), _p(mp_viewcos), _p(mp_flags), _p(mp_desc),
                                                 len(mp_level), th, nnratio, _p(assigned), C.byref(n)))
        return n.value, assigned[:len(kps)]

    def search_by_projection_rows(self, frame, scale_factors, kp_occupied, mp_xy, mp_level, mp_viewcos, mp_flags, table, rows,
                                  th, nnratio):
        """orbfe_search_by_projection_frame_rows: the MapPoints' descriptors are rows of a DescTable (device copy + page-locked
        mirror); rows[i] = row number, bit 31 set = read that row from the mirror."""
        sf = np.ascontiguousarray(scale_factors, np.float32)
        occ = np.ascontiguousarray(kp_occupied, np.uint8)
        mp_xy = np.ascontiguousarray(mp_xy, np.float32)
        mp_level = np.ascontiguousarray(mp_level, np.int32)
        mp_viewcos = np.ascontiguousarray(mp_viewcos, np.float32)
        mp_flags = np.ascontiguousarray(mp_flags, np.uint8)
        rows = np.ascontiguousarray(rows, np.int32)
        assigned = np.full(max(len(frame), 1), -1, np.int32)
        n = C.c_int(0)
        _check(self.L.orbfe_search_by_projection_frame_rows(self.h, frame.h, _p(sf), len(sf), _p(occ), _p(mp_xy), _p(mp_level),
                                                            _p(mp_viewcos), _p(mp_flags), C.c_void_p(table.dev), C.c_void_p(table.host.base),
                                                            _p(rows), table.cap, len(mp_level), th, nnratio, _p(assigned), C.byref(n)))
        return n.value, assigned[:len(frame)]

    def upload_async(self, dst_device, src_host_ptr, nbytes):
        _check(self.L.orbfe_matcher_upload_async(self.h, C.c_void_p(dst_device), C.c_void_p(src_host_ptr), nbytes))

    def synchronize(self):
        _check(self.L.orbfe_matcher_synchronize(self.h))

    def search_by_projection_uv(self, kps, desc, bounds, scale_factors, kp_occupied, src_uv, src_level, src_angle,
                                src_flags, src_valid, src_desc, th, max_dist, skip_any_occupied, check_ori):
        if isinstance(kps, Frame):
            sf = np.ascontiguousarray(scale_factors, np.float32)
            occ = np.ascontiguousarray(kp_occupied, np.uint8)
            src_uv = np.ascontiguousarray(src_uv, np.float32)
            src_level = np.ascontiguousarray(src_level, np.int32)
            src_angle = np.ascontiguousarray(src_angle, np.float32)
            src_flags = np.ascontiguousarray(src_flags, np.uint8)
            src_valid = np.ascontiguousarray(src_valid, np.uint8)
            src_desc_p, _keep = _rows(src_desc)
            assigned = np.full(max(len(kps), 1), -1, np.int32)
            n = C.c_int(0)
            _check(self.L.orbfe_search_by_projection_uv_frame(self.h, kps.h, _p(sf), len(sf), _p(occ), _p(src_uv), _p(src_level),
                                                              _p(src_angle), _p(src_flags), _p(src_valid), src_desc_p,
                                                              len(src_level), th, max_dist, int(skip_any_occupied),
                                                              int(check_ori), _p(assigned), C.byref(n)))
            return n.value, assigned[:len(kps)]
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
        n = C.c_int(0)
        _check(self.L.orbfe_search_by_projection_uv(self.h, _p(kps), _p(desc), len(kps), _p(b), _p(sf), len(sf),
                                                    _p(occ), _p(src_uv), _p(src_level), _p(src_angle), _p(src_flags),
                                                    _p(src_valid), _p(src_desc), len(src_level), th, max_dist,
                                                    int(skip_any_occupied), int(check_ori), _p(assigned),
                                                    C.byref(n)))
        return n.value, assigned[:len(kps)]

    def window_candidates(self, kps, desc, bounds, qx, qy, qr, qmin, qmax, qdesc):
        """Ordered candidate lists (index, Hamming distance) per query -- the GPU part of every windowed search."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        b = np.asarray(bounds, np.float32)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        nq = len(qx)
        counts = np.zeros(max(nq, 1), np.uint32)
        offsets = np.zeros(max(nq, 1), np.uint32)
        cap = 1 << 16
        while True:
            pool = np.zeros(cap, np.uint32)
            used = C.c_size_t(0)
            rc = self.L.orbfe_window_candidates(self.h, _p(kps), _p(desc), len(kps), _p(b), nq, _p(qx), _p(qy), _p(qr),
                                                _p(qmin), _p(qmax), _p(qdesc), _p(counts), _p(offsets), _p(pool), cap,
                                                C.byref(used))
            if rc == -5:
                cap = int(used.value)
                continue
            _check(rc)
            break
        return [(pool[offsets[q]:offsets[q] + counts[q]] & 0xffff, pool[offsets[q]:offsets[q] + counts[q]] >> 16)
                for q in range(nq)]

    def distinctive_descriptors(self, desc_lists):
        """Batched MapPoint::ComputeDistinctiveDescriptors: list of (N_i, 32) arrays -> best index per list."""
        offs = np.zeros(len(desc_lists) + 1, np.int32)
        offs[1:] = np.cumsum([len(d) for d in desc_lists])
        allv = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in desc_lists])
                                     if len(desc_lists) and offs[-1] else np.zeros((1, 32), np.uint8))
        out = np.zeros(max(len(desc_lists), 1), np.int32)
        _check(self.L.orbfe_distinctive_descriptors(self.h, len(desc_lists), _p(offs), _p(allv), _p(out)))
        return out[:len(desc_lists)]

    def search_projected(self, kps, desc, bounds, uv, radius, level, valid, sdesc, kp_skip=None, claim=False,
                         inv_sigma2=None, chi2=5.99, max_dist=50):
        """Projected best-match loop of SearchByProjection(KF, Scw) / Fuse / SearchBySim3 -> (n, best_idx, best_dist)."""
        if isinstance(kps, Frame):
            uv = np.ascontiguousarray(uv, np.float32)
            radius = np.ascontiguousarray(radius, np.float32)
            level = np.ascontiguousarray(level, np.int32)
            valid = np.ascontiguousarray(valid, np.uint8)
            sdesc_p, _keep = _rows(sdesc)
            ns = len(radius)
            skip = None if kp_skip is None else np.ascontiguousarray(kp_skip, np.uint8)
            inv = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
            bi = np.full(max(ns, 1), -1, np.int32)
            bd = np.full(max(ns, 1), -1, np.int32)
            nm = C.c_int(0)
            _check(self.L.orbfe_search_projected_frame(self.h, kps.h, ns, _p(uv), _p(radius), _p(level), _p(valid), sdesc_p,
                                                       None if skip is None else _p(skip), int(claim),
                                                       None if inv is None else _p(inv), 0 if inv is None else len(inv), chi2,
                                                       max_dist, _p(bi), _p(bd), C.byref(nm)))
            return nm.value, bi[:ns], bd[:ns]
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
        nm = C.c_int(0)
        _check(self.L.orbfe_search_projected(self.h, _p(kps), _p(desc), len(kps), _p(b), ns, _p(uv), _p(radius), _p(level),
                                             _p(valid), _p(sdesc), None if skip is None else _p(skip), int(claim),
                                             None if inv is None else _p(inv), 0 if inv is None else len(inv), chi2,
                                             max_dist, _p(bi), _p(bd), C.byref(nm)))
        return nm.value, bi[:ns], bd[:ns]

    def search_for_triangulation(self, kps1, desc1, has_mp1, fv1, kps2, desc2, has_mp2, fv2, F12, ex, ey, scale2, sigma2,
                                 check_ori=True):
        """ORBmatcher::SearchForTriangulation from the epipole onwards -> (nmatches, pairs[n,2])."""
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
        nm = C.c_int(0)
        _check(self.L.orbfe_search_for_triangulation(self.h, _p(kps1), _p(desc1), _p(h1), len(kps1), _p(f1[0]), _p(f1[1]),
                                                     _p(f1[2]), len(f1[0]), _p(kps2), _p(desc2), _p(h2), len(kps2),
                                                     _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]), _p(F), ex, ey, _p(sc),
                                                     _p(sg), len(sc), int(check_ori), _p(pairs), C.byref(nm)))
        return nm.value, pairs[:nm.value]

    def search_by_bow(self, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio=0.7, check_ori=True,
                      strict=False):
        """ORBmatcher::SearchByBoW; fvX = (nodes, offsets, features) as returned by Vocabulary.transform.
        valid2=None is the (KeyFrame, Frame) overload, strict=True + valid2 the (KeyFrame, KeyFrame) one.
        Returns (nmatches, matches12)."""
        n1, n2 = len(desc1), len(desc2)                       # (numpy rows or DeviceRows: a resident frame's descriptors)
        d1p, _k1 = _rows(desc1)
        d2p, _k2 = _rows(desc2)
        angle1 = np.ascontiguousarray(angle1, np.float32)
        angle2 = np.ascontiguousarray(angle2, np.float32)
        valid1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        f1 = [np.ascontiguousarray(a, np.uint32) for a in fv1]
        f2 = [np.ascontiguousarray(a, np.uint32) for a in fv2]
        m12 = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(self.L.orbfe_search_by_bow(self.h, d1p, _p(angle1), _p(valid1), n1, _p(f1[0]), _p(f1[1]),
                                          _p(f1[2]), len(f1[0]), d2p, _p(angle2), None if v2 is None else _p(v2),
                                          n2, _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]), nnratio,
                                          int(check_ori), int(strict), _p(m12), C.byref(nm)))
        return nm.value, m12[:n1]

    def search_by_bow_batch(self, sides1, desc2, angle2, valid2, fv2, nnratio=0.7, check_ori=True, strict=False):
        """Tracking::Relocalization's SearchByBoW loop in one GPU submission: sides1 = [(desc1, angle1, valid1, fv1)] per
        candidate keyframe, all against one frame (side 2).  Returns [(nmatches, matches12)] per keyframe."""
        K = len(sides1)
        d1 = [np.ascontiguousarray(s[0], np.uint8) for s in sides1]
        a1 = [np.ascontiguousarray(s[1], np.float32) for s in sides1]
        v1 = [np.ascontiguousarray(s[2], np.uint8) for s in sides1]
        f1 = [[np.ascontiguousarray(a, np.uint32) for a in s[3]] for s in sides1]
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        angle2 = np.ascontiguousarray(angle2, np.float32)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        f2 = [np.ascontiguousarray(a, np.uint32) for a in fv2]
        m12 = [np.full(max(len(d), 1), -1, np.int32) for d in d1]
        arr = lambda xs: (C.c_void_p * max(K, 1))(*[x.ctypes.data for x in xs])
        n1 = np.array([len(d) for d in d1] or [0], np.int32)
        nf1 = np.array([len(f[0]) for f in f1] or [0], np.int32)
        nm = np.zeros(max(K, 1), np.int32)
        _check(self.L.orbfe_search_by_bow_batch(self.h, K, arr(d1), arr(a1), arr(v1), _p(n1), arr([f[0] for f in f1]),
                                                arr([f[1] for f in f1]), arr([f[2] for f in f1]), _p(nf1), _p(desc2), _p(angle2),
                                                None if v2 is None else _p(v2), len(desc2), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]),
                                                nnratio, int(check_ori), int(strict), arr(m12), _p(nm)))
        return [(int(nm[k]), m12[k][:len(d1[k])]) for k in range(K)]

    def stage_ms(self):
        out = np.zeros(3, np.float64)
        _check(self.L.orbfe_debug_matcher_ms(self.h, _p(out)))
        return out

    def get_features_in_area(self, kps, bounds, x, y, r, min_level, max_level):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        b = np.asarray(bounds, np.float32)
        out = np.zeros(max(len(kps), 1), np.int32)
        n = C.c_int(0)
        _check(self.L.orbfe_debug_features_in_area(self.h, _p(kps), len(kps), _p(b), x, y, r, min_level, max_level,
                                                   _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()


def quadtree(x, y, score, min_x, max_x, min_y, max_y, n_target):
    """Product host quadtree (no GPU needed): indices of the retained candidates, list order."""
    x = np.ascontiguousarray(x, np.int16)
    y = np.ascontiguousarray(y, np.int16)
    score = np.ascontiguousarray(score, np.uint8)
    out = np.zeros(max(len(x), 1), np.int32)
    n = C.c_int(0)
    _check(load_library().orbfe_debug_quadtree(_p(x), _p(y), _p(score), len(x), min_x, max_x, min_y, max_y, n_target,
                                               _p(out), len(out), C.byref(n)))
    return out[:n.value].copy()


def undistort_equidistant(xy, fx, fy, cx, cy):
    """Frame::antidistorsionarProyeccionEquidistante (host helper of the C ABI)."""
    xy = np.ascontiguousarray(xy, np.float32).copy()
    _check(load_library().orbfe_undistort_equidistant(_p(xy), len(xy), fx, fy, cx, cy))
    return xy


def undistort_pinhole(xy, fx, fy, cx, cy, dist):
    """cv::undistortPoints(pts, pts, K, dist, Mat(), K) as Frame::UndistortKeyPoints calls it (host helper of the C ABI)."""
    xy = np.ascontiguousarray(xy, np.float32).copy()
    d = np.ascontiguousarray(dist, np.float32)
    _check(load_library().orbfe_undistort_pinhole(_p(xy), len(xy), fx, fy, cx, cy, _p(d) if len(d) else None, len(d)))
    return xy


def compute_image_bounds(cols, rows, mode, fx, fy, cx, cy, dist=()):
    """Frame::ComputeImageBounds -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    d = np.ascontiguousarray(dist, np.float32)
    b = np.zeros(4, np.float32)
    _check(load_library().orbfe_compute_image_bounds(cols, rows, mode, fx, fy, cx, cy, _p(d) if len(d) else None, len(d), _p(b)))
    return b


def sincos_host_mismatches(lo_bits, hi_bits, step=1):
    bad = C.c_longlong(0)
    _check(load_library().orbfe_debug_sincos_host_check(lo_bits, hi_bits, step, C.byref(bad)))
    return bad.value


class Vocabulary:
    """DBoW2 ORB vocabulary resident in HBM; transform() = Frame::ComputeBoW."""

    def __init__(self, image, device=0):
        self.L = load_library()
        h = C.c_void_p()
        buf = np.frombuffer(image, np.uint8)
        _check(self.L.orbfe_vocabulary_create_from_image(device, _p(buf), buf.size, C.byref(h)))
        self.h = h

    def info(self):
        v = [C.c_int(0) for _ in range(6)]
        _check(self.L.orbfe_vocabulary_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(('k', 'L', 'scoring', 'weighting', 'n_nodes', 'n_words'), [x.value for x in v]))

    def assemble(self, leaf, node):
        """BowVector / FeatureVector from per-keypoint (leaf node, level node) pairs (Stream.bow_raw / orbfe_extract_bow_raw)."""
        leaf = np.ascontiguousarray(leaf, np.uint32)
        node = np.ascontiguousarray(node, np.uint32)
        n = len(leaf)
        ids = np.zeros(max(n, 1), np.uint32)
        vals = np.zeros(max(n, 1), np.float64)
        fvn = np.zeros(max(n, 1), np.uint32)
        fvo = np.zeros(n + 1, np.uint32)
        fvf = np.zeros(max(n, 1), np.uint32)
        wof = np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int(0), C.c_int(0)
        _check(self.L.orbfe_bow_assemble(self.h, _p(leaf), _p(node), n, _p(ids), _p(vals), C.byref(nw), _p(fvn), _p(fvo), _p(fvf),
                                         C.byref(nn), _p(wof)))
        nw, nn = nw.value, nn.value
        return ids[:nw], vals[:nw], (fvn[:nn], fvo[:nn + 1], fvf[:int(fvo[nn])]), wof[:n], node[:n]

    def transform(self, desc, levelsup=4):
        """-> (bow_ids, bow_values, (fv_nodes, fv_offsets, fv_features), word_of_feature, node_of_feature)"""
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
        _check(self.L.orbfe_bow_transform(self.h, _p(desc), n, 0, levelsup, _p(ids), _p(vals), C.byref(nw), _p(fvn),
                                          _p(fvo), _p(fvf), C.byref(nn), _p(wof), _p(nof)))
        nw, nn = nw.value, nn.value
        return ids[:nw], vals[:nw], (fvn[:nn], fvo[:nn + 1], fvf[:int(fvo[nn])]), wof[:n], nof[:n]

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_vocabulary_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArray:
    """A numpy array in page-locked HOST memory (orbfe_host_alloc): `.a` is the view.  Query descriptor rows kept in one
    are fetched by DMA straight from it by the `_frame` searches."""

    def __init__(self, shape, dtype=np.uint8):
        L = load_library()
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = C.c_void_p()
        _check(L.orbfe_host_alloc(max(n * dt.itemsize, 1), C.byref(p)))
        self.base = p.value
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n * dt.itemsize, 1),))
        self.a = raw[:n * dt.itemsize].view(dt).reshape(shape)

    def free(self):
        if getattr(self, 'base', None):
            self.a = None
            load_library().orbfe_host_free(C.c_void_p(self.base))
            self.base = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DescTable:
    """A caller-maintained descriptor table for orbfe_search_by_projection_frame_rows: `cap` 32-byte rows in device memory
    (`dev`) and their page-locked host mirror (`host.a`, a numpy view)."""

    def __init__(self, cap, device=0):
        self.L = load_library()
        self.device, self.cap = device, cap
        self.host = PinnedArray((cap, 32), np.uint8)
        self.host.a[:] = 0
        p = C.c_void_p()
        _check(self.L.orbfe_device_malloc(device, cap * 32, C.byref(p)))
        self.dev = p.value

    def upload(self, matcher, lo, hi):
        """rows [lo, hi) of the mirror -> device, asynchronously on the matcher's stream"""
        matcher.upload_async(self.dev + lo * 32, self.host.base + lo * 32, (hi - lo) * 32)

    def free(self):
        if self.dev:
            self.L.orbfe_device_free(self.device, C.c_void_p(self.dev))
            self.dev = None
            self.host.free()


class RegisteredArray:
    """A numpy array the caller owns, page-locked and mapped for the GPU in place (orbfe_host_register); unregistered by close()."""

    def __init__(self, a):
        assert a.flags['C_CONTIGUOUS']
        self.L = load_library()
        self.a = a
        _check(self.L.orbfe_host_register(C.c_void_p(a.ctypes.data), a.nbytes))
        self.live = True

    def close(self):
        if self.live:
            self.L.orbfe_host_unregister(C.c_void_p(self.a.ctypes.data))
            self.live = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedFrames:
    """A stack of equally sized u8 frames in page-locked HOST memory (orbfe_host_alloc): the fast way to hand
    host frames to the extractor (in_device_memory == 0)."""

    def __init__(self, frames):
        L = load_library()
        self.rows, self.cols = frames[0].shape
        self.stride = self.cols
        self.n = len(frames)
        self.frame_bytes = self.rows * self.stride
        p = C.c_void_p()
        _check(L.orbfe_host_alloc(self.frame_bytes * self.n, C.byref(p)))
        self.base = p.value
        view = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n, self.rows, self.cols))
        for i, f in enumerate(frames):
            view[i] = f
        self.ptrs = [self.base + i * self.frame_bytes for i in range(self.n)]

    def free(self):
        if getattr(self, 'base', None):
            load_library().orbfe_host_free(C.c_void_p(self.base))
            self.base = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceFrames:
    """A stack of equally sized u8 frames resident in HBM (hipMalloc'd through the C ABI)."""

    def __init__(self, frames, device=0, stride=None):
        L = load_library()
        self.device = device
        self.rows, self.cols = frames[0].shape
        self.stride = stride or self.cols
        self.n = len(frames)
        self.frame_bytes = self.rows * self.stride
        p = C.c_void_p()
        _check(L.orbfe_device_malloc(device, self.frame_bytes * self.n, C.byref(p)))
        self.base = p.value
        for i, f in enumerate(frames):
            buf = np.zeros((self.rows, self.stride), np.uint8)
            buf[:, :self.cols] = f
            _check(L.orbfe_device_upload(device, C.c_void_p(self.base + i * self.frame_bytes), _p(buf), buf.size))
        self.ptrs = [self.base + i * self.frame_bytes for i in range(self.n)]

    def free(self):
        if getattr(self, 'base', None):
            load_library().orbfe_device_free(self.device, C.c_void_p(self.base))
            self.base = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_synchronize(device=0):
    _check(load_library().orbfe_device_synchronize(device))


def device_numa_node(device=0):
    return load_library().orbfe_device_numa_node(device)


def bind_thread_to_device(device=0):
    """Restrict the calling thread (and threads it creates later) to the CPUs of the GPU's NUMA node; 0 = unchanged."""
    return load_library().orbfe_bind_thread_to_device(device)


def h2d_rate_gbs(device, host_ptr, nbytes, reps=8):
    g = C.c_double(0)
    _check(load_library().orbfe_debug_h2d_rate(device, C.c_void_p(host_ptr), nbytes, reps, C.byref(g)))
    return g.value


class Stream:
    """Native stream runner (orbfe_stream_*): push batches of frames, pop (keypoints, descriptors, matches
    against the predecessor frame) in order.  Orchestration (async extraction, matching thread) is C++."""

    def __init__(self, nfeatures, scale, nlevels, ini_th, min_th, device, batch, depth=2):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.orbfe_stream_create(nfeatures, scale, nlevels, ini_th, min_th, device, batch, depth, C.byref(h)))
        self.h = h
        self.batch = batch
        self.cap = self.L.orbfe_stream_capacity(h)

    def set_queue_slots(self, nslots):
        """Result slots = batches that may be pushed ahead of the pops + 2 (default depth + 4)."""
        _check(self.L.orbfe_stream_set_queue_slots(self.h, int(nslots)))

    def queue_slots(self):
        return int(self.L.orbfe_stream_queue_slots(self.h))

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_matching(self, bounds, window=100, nnratio=0.9, check_ori=True):
        b = np.asarray(bounds, np.float32)
        _check(self.L.orbfe_stream_set_matching(self.h, _p(b), window, nnratio, int(check_ori)))

    def set_blur_variant(self, variant):
        _check(self.L.orbfe_stream_set_blur_variant(self.h, int(variant)))

    def set_vocabulary(self, voc, levelsup=4):
        self._voc = voc
        _check(self.L.orbfe_stream_set_vocabulary(self.h, voc.h if voc is not None else None, levelsup))

    def bow_raw(self, frame):
        """(leaf, node) arrays of frame `frame` of the last popped batch (copies)."""
        pl, pn, n = C.c_void_p(), C.c_void_p(), C.c_int(0)
        _check(self.L.orbfe_stream_bow_raw(self.h, frame, C.byref(pl), C.byref(pn), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        leaf = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
        node = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
        return leaf, node

    def push_ptrs(self, ptrs, rows, cols, stride, on_device=True):
        assert len(ptrs) == self.batch
        arr = (C.c_void_p * self.batch)(*ptrs)
        _check(self.L.orbfe_stream_push(self.h, arr, int(on_device), rows, cols, stride))
        self.cap = self.L.orbfe_stream_capacity(self.h)      # grows with a geometry that returns more keypoints

    def pop(self, copy=False):
        """-> (kps[B,cap], desc[B,cap,32], n[B], matches12[B,cap], nmatches[B]) as views valid until the next pop."""
        pk, pd, pn, pm, pnm = (C.c_void_p() for _ in range(5))
        _check(self.L.orbfe_stream_pop(self.h, C.byref(pk), C.byref(pd), C.byref(pn), C.byref(pm), C.byref(pnm)))
        B, cap = self.batch, self.cap

        def view(ptr, dtype, shape):
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
            a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
            return a.copy() if copy else a
        return (view(pk, KP_DTYPE, (B, cap)), view(pd, np.uint8, (B, cap, 32)), view(pn, np.int32, (B,)),
                view(pm, np.int32, (B, cap)), view(pnm, np.int32, (B,)))

    def batches_in_flight(self):
        """Batches the runner keeps on the GPU at a time: `depth`, or fewer when the process has fewer hardware queues than that."""
        return int(self.L.orbfe_stream_batches_in_flight(self.h))

    def set_isolated_batches(self, on=True):
        """Frame 0 of every batch has no predecessor (orbfe_stream_set_isolated_batches)."""
        _check(self.L.orbfe_stream_set_isolated_batches(self.h, int(on)))

    def stats(self, reset=False):
        """{submit, collect, match} worker busy ms and batches done since the last reset."""
        out = np.zeros(4, np.float64)
        _check(self.L.orbfe_stream_stats(self.h, _p(out), int(reset)))
        return out

    def kernel_ms(self, reset=False):
        ms = np.zeros(5, np.float64)
        b, f = C.c_longlong(0), C.c_longlong(0)
        _check(self.L.orbfe_stream_kernel_ms(self.h, _p(ms), C.byref(b), C.byref(f), int(reset)))
        return ms, b.value, f.value


class MultiStream:
    """ONE stream over several devices (orbfe_stream_multi_*): batch k goes to devices[k % n], pop() returns the batches strictly in
    push order.  `devices` may name a device several times."""

    def __init__(self, nfeatures, scale, nlevels, ini_th, min_th, devices, batch, depth=2):
        self.L = load_library()
        h = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        _check(self.L.orbfe_stream_multi_create(nfeatures, scale, nlevels, ini_th, min_th, ids, len(devices), batch, depth, C.byref(h)))
        self.h = h
        self.batch = batch
        self.devices = list(devices)
        self.cap = self.L.orbfe_stream_multi_capacity(h)

    def close(self):
        if getattr(self, 'h', None):
            self.L.orbfe_stream_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_matching(self, bounds, window=100, nnratio=0.9, check_ori=True):
        b = np.asarray(bounds, np.float32)
        _check(self.L.orbfe_stream_multi_set_matching(self.h, _p(b), window, nnratio, int(check_ori)))

    def set_blur_variant(self, variant):
        _check(self.L.orbfe_stream_multi_set_blur_variant(self.h, int(variant)))

    def device_of_next_push(self):
        return int(self.L.orbfe_stream_multi_device_of_next_push(self.h))

    def push_ptrs(self, ptrs, rows, cols, stride, on_device=True):
        assert len(ptrs) == self.batch
        arr = (C.c_void_p * self.batch)(*ptrs)
        _check(self.L.orbfe_stream_multi_push(self.h, arr, int(on_device), rows, cols, stride))
        self.cap = self.L.orbfe_stream_multi_capacity(self.h)

    def pop(self, copy=False):
        """-> (kps[B,cap], desc[B,cap,32], n[B], matches12[B,cap], nmatches[B]) as views valid until the next pop."""
        pk, pd, pn, pm, pnm = (C.c_void_p() for _ in range(5))
        _check(self.L.orbfe_stream_multi_pop(self.h, C.byref(pk), C.byref(pd), C.byref(pn), C.byref(pm), C.byref(pnm)))
        B, cap = self.batch, self.cap

        def view(ptr, dtype, shape):
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
            a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
            return a.copy() if copy else a
        return (view(pk, KP_DTYPE, (B, cap)), view(pd, np.uint8, (B, cap, 32)), view(pn, np.int32, (B,)),
                view(pm, np.int32, (B, cap)), view(pnm, np.int32, (B,)))
