"""os1_amd -- MI355X-native ORB feature front end (extract + match) for AlejandroSilvestri/os1.

The product is the C-ABI shared library os1_amd/liborbfe.so (include/orbfe.h): hand-written HIP
kernels for gfx950 plus the C++ host engine.  This package only holds the thin ctypes binding the
tests / bench use (api.py), the synthetic frame generator (synth.py) and the build helper."""
from .api import (KP_DTYPE, OrbfeError, Extractor, Matcher, device_count, hamming, lib_path, load_library,  # noqa: F401
                  build_library, DeviceFrames, device_synchronize, Stream)
