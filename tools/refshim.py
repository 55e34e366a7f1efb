"""Import shims that let the read-only reference at /root/reference run on a CPU-only box.

Runs ONLY in the build container (it needs /root/reference).  Nothing here ships reference
code: it registers stand-in modules for packages the container lacks (vapoursynth,
fastprogress, torchvision, cv2) so that the reference's own hot-path modules
(vsdeoldify/deoldify, vsdeoldify/fastai, vsdeoldify/colorization/colorizers,
vsdeoldify/vsslib/imfilters.py ...) can be imported by path and executed to produce
golden vectors (see tools/gen_golden.py).

Stand-ins and what they pin:
  * torchvision.models.resnet*  -> oracle.resnet (our restatement of the standard ResNet v1.5);
    the encoder is therefore pinned only to that restatement.
  * cv2.cvtColor / Laplacian    -> oracle.cvcolor (our restatement of OpenCV's 8-bit fixed-point
    BT.601 YUV); parity with real OpenCV is UNPINNED at LSB level (cv2 absent).
  * fastprogress / vapoursynth  -> empty symbols, never executed on the path.
"""
import importlib.metadata
import os
import sys
import types

REF_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated in the build container")
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    if "vsdeoldify" in sys.modules and getattr(sys.modules["vsdeoldify"], "_havc_shim", False):
        return

    # --- fastprogress -------------------------------------------------------------------
    class _Bar:
        def __init__(self, *a, **k):
            pass

    fp = _mod("fastprogress")
    fpp = _mod("fastprogress.fastprogress", MasterBar=_Bar, ProgressBar=_Bar, master_bar=_Bar,
               progress_bar=_Bar, format_time=lambda t: str(t), IN_NOTEBOOK=False)
    fp.fastprogress = fpp
    _orig_version = importlib.metadata.version

    def _version(name):
        if name == "fastprogress":
            return "1.0.3"
        return _orig_version(name)

    importlib.metadata.version = _version

    # --- torchvision ----------------------------------------------------------------------
    from oracle import resnet as _rn

    def _placeholder(n):
        def f(*a, **k):
            raise NotImplementedError(n)
        f.__name__ = n
        return f

    tv = _mod("torchvision")
    tvm = _mod("torchvision.models", ResNet=_rn.ResNet, resnet18=_rn.resnet18, resnet34=_rn.resnet34,
               resnet50=_rn.resnet50, resnet101=_rn.resnet101, resnet152=_rn.resnet152,
               SqueezeNet=type("SqueezeNet", (), {}))
    for n in ("squeezenet1_0", "squeezenet1_1", "densenet121", "densenet169", "densenet201", "densenet161",
              "vgg16_bn", "vgg19_bn", "alexnet"):
        setattr(tvm, n, _placeholder(n))
    tv.models = tvm
    tv.transforms = _mod("torchvision.transforms")
    tv.utils = _mod("torchvision.utils")

    # --- cv2 ------------------------------------------------------------------------------
    from oracle import cvcolor as _cv

    _mod("cv2", cvtColor=_cv.cvtColor, COLOR_RGB2YUV=_cv.COLOR_RGB2YUV, COLOR_YUV2RGB=_cv.COLOR_YUV2RGB,
         COLOR_RGB2HSV=_cv.COLOR_RGB2HSV, COLOR_HSV2RGB=_cv.COLOR_HSV2RGB, Laplacian=_cv.Laplacian,
         CV_32F=_cv.CV_32F, CV_64F=_cv.CV_64F)

    # --- vapoursynth ----------------------------------------------------------------------
    _mod("vapoursynth", Error=type("Error", (Exception,), {}), VideoNode=object, VideoFrame=object,
         core=types.SimpleNamespace())

    # --- the reference package, WITHOUT running its __init__ (which needs VapourSynth) ---
    pkg = _mod("vsdeoldify")
    pkg.__path__ = [os.path.join(REF_ROOT, "vsdeoldify")]
    pkg._havc_shim = True
    vss = _mod("vsdeoldify.vsslib")
    vss.__path__ = [os.path.join(REF_ROOT, "vsdeoldify", "vsslib")]
    col = _mod("colorizers")
    col.__path__ = [os.path.join(REF_ROOT, "vsdeoldify", "colorization", "colorizers")]


def build_wide(nf_factor=2, arch="resnet101"):
    """Reference DynamicUnetWide, constructed exactly as unet_learner_wide does (generators.py:57-72)."""
    install()
    import torch
    from vsdeoldify.fastai.layers import NormType
    from vsdeoldify.fastai.vision.learner import create_body
    from vsdeoldify.fastai.vision import models
    from vsdeoldify.deoldify.unet import DynamicUnetWide
    body = create_body(getattr(models, arch), pretrained=False)
    m = DynamicUnetWide(body, n_classes=3, blur=True, blur_final=True, self_attention=True, y_range=(-3.0, 3.0),
                        norm_type=NormType.Spectral, last_cross=True, bottle=False, nf_factor=nf_factor)
    return m.eval()


def build_deep(nf_factor=1.5, arch="resnet34"):
    """Reference DynamicUnetDeep, as unet_learner_deep does (generators.py:133-147)."""
    install()
    from vsdeoldify.fastai.layers import NormType
    from vsdeoldify.fastai.vision.learner import create_body
    from vsdeoldify.fastai.vision import models
    from vsdeoldify.deoldify.unet import DynamicUnetDeep
    body = create_body(getattr(models, arch), pretrained=False)
    m = DynamicUnetDeep(body, n_classes=3, blur=True, blur_final=True, self_attention=True, y_range=(-3.0, 3.0),
                        norm_type=NormType.Spectral, last_cross=True, bottle=False, nf_factor=nf_factor)
    return m.eval()
