"""vsdeoldify_amd — MI355X-native per-frame colorization inference behind the HAVC (vs-deoldify) API.

Only the hot path lives here: the DeOldify generators, the Zhang colorizers, DDColor and the per-pixel
merge / tweak filters, as hand-written HIP kernels for gfx950 behind a C ABI (include/havc_mi355.h, lib/libhavc_mi355.so),
plus the thin Python mirror of the reference's adapter classes.  No CPU fallback: see _native.py.
"""
__version__ = "0.2.0"

from ._native import HavcOutOfMemory, NativeLibraryError  # noqa: F401


def __getattr__(name):
    if name == "ModelImageRender":
        from .render import ModelImageRender
        return ModelImageRender
    if name == "ModelColorization":
        from .colorization import ModelColorization
        return ModelColorization
    if name == "DDColorRender":
        from .ddcolor import DDColorRender
        return DDColorRender
    if name in ("HAVC_colorizer", "HAVC_merge", "HAVC_ddeoldify", "ddeoldify", "HAVCFrameColorizer"):
        from . import havc
        return getattr(havc, name)
    if name == "DeviceImage":
        from .device import DeviceImage
        return DeviceImage
    if name in ("image_weighted_merge", "chroma_post_process", "chroma_stabilizer"):
        from . import imfilters
        return getattr(imfilters, name)
    raise AttributeError(name)
