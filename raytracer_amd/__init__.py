"""raytracer_amd: MI355X-native ray-casting and light-transport path of DHRRAM/RayTracer.

Python here is plumbing only (ctypes over the C ABI of include/rtx.h, scene description, synthetic scene
generator); all compute lives in librtx_amd.so (hand-written HIP for gfx950).
"""
from . import _ffi, scene, synth  # noqa: F401
from .api import Scene, Photons, Group, RtxError, device_count, set_device, film_clear, film_resolve  # noqa: F401
