"""ctypes binding of libhavc_mi355.so (C ABI: include/havc_mi355.h).

The HIP library is the ONLY compute path of this package: there is no CPU / PyTorch fallback.
If the shared object is missing or no gfx950 device is visible, importing consumers fail loudly
(NativeLibraryError) instead of degrading silently.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAVC_MI355_LIB") or os.path.join(_HERE, "lib", "libhavc_mi355.so")   # override: A/B builds in tools/

HAVC_OK, HAVC_E_INVALID, HAVC_E_OOM, HAVC_E_HIP, HAVC_E_NODEVICE, HAVC_E_RANGE = 0, -1, -2, -3, -4, -5

# op types / flags (mirror include/havc_mi355.h)
OP_CONV, OP_MAXPOOL, OP_BLUR_RESIZE, OP_AFFINE, OP_ATTENTION, OP_PREP_RGB8, OP_COPY_CH = 1, 2, 3, 4, 5, 6, 7
OP_SUBSAMPLE2, OP_PROJ2, OP_BILINEAR2, OP_PREP_LAB_L = 8, 9, 10, 11
OP_DWCONV7, OP_LAYERNORM, OP_MHA, OP_PIXSHUF4_BLUR, OP_PREP_DDCOLOR, OP_DWCONV7_LN = 12, 13, 14, 15, 16, 17
OP_FOLD_QUERIES, OP_SHUF4_BLUR_AB = 18, 19
OP_EW, OP_DWCONV, OP_CHAN_ATTN, OP_MHA64, OP_CBAM, OP_GRU, OP_PLANAR_IN, OP_PLANAR_OUT, OP_CMN_DECODER_IN = 20, 21, 22, 23, 24, 25, 26, 27, 28
EW_SRC_BCAST, EW_RES, EW_RES_BCAST, EW_RELU, EW_DUAL = 1, 2, 4, 8, 16
F_RELU_PRE, F_AFFINE, F_RESIDUAL, F_RELU_POST = 0x1, 0x2, 0x4, 0x8
F_OUT_PIXSHUF, F_OUT_TRANSPOSED, F_OUT_RGB8, F_LEAKY, F_FUSE_RGB8, F_PS_BLUR = 0x10, 0x20, 0x40, 0x80, 0x100, 0x200
F_GELU, F_W_FROM_BUF = 0x400, 0x800
F_FUSE_PROJ = 0x2000
F_PRECISE = 0x4000      # hi / lo fp16 pairs, fp32-class arithmetic (include/havc_mi355.h HAVC_F_PRECISE)


def F_SPLITK(n):
    """HAVC_F_SPLITK(n): split-K count of a conv op (2..15), part of the plan"""
    assert 0 <= n <= 15
    return (n if n > 1 else 0) << 16

# numpy mirror of `struct havc_op` (natural C alignment; checked against sizeof in tests)
OP_DTYPE = np.dtype([
    ("type", "<i4"), ("flags", "<i4"),
    ("src", "<i4"), ("src2", "<i4"), ("dst", "<i4"),
    ("src_coff", "<i4"), ("src_cpitch", "<i4"),
    ("dst_coff", "<i4"), ("dst_cpitch", "<i4"),
    ("res_coff", "<i4"), ("res_cpitch", "<i4"),
    ("Hi", "<i4"), ("Wi", "<i4"), ("Ci", "<i4"),
    ("Ho", "<i4"), ("Wo", "<i4"), ("Co", "<i4"),
    ("kh", "<i4"), ("kw", "<i4"), ("stride", "<i4"), ("pad", "<i4"), ("dil", "<i4"),
    ("Kc", "<i4"), ("Npad", "<i4"),
    ("aux0", "<i4"), ("aux1", "<i4"),
    ("w_off", "<i8"), ("bias_off", "<i8"), ("scale_off", "<i8"), ("shift_off", "<i8"),
    ("f0", "<f4"), ("f1", "<f4"), ("f2", "<f4"), ("f3", "<f4"),
    ("flops", "<i8"), ("tag", "<i4"), ("reserved", "<i4"),
    ("pad_w_delta", "<i4"), ("out_step", "<i4"), ("out_oy", "<i4"), ("out_ox", "<i4"),
], align=True)
BUF_DTYPE = np.dtype([("elems_per_frame", "<i8"), ("elem_bytes", "<i4"), ("zero_init", "<i4")], align=True)


class Stats(C.Structure):
    _fields_ = [("last_ms", C.c_double), ("total_ms", C.c_double), ("total_flops", C.c_double),
                ("frames", C.c_int64), ("launches", C.c_int64), ("bytes_resident", C.c_int64)]


class NativeLibraryError(RuntimeError):
    pass


class HavcOutOfMemory(RuntimeError):
    """HAVC_E_OOM: the caller reproduces deoldify/filters.py:55-63 (return the input, warn)."""


class HavcRangeError(RuntimeError):
    """HAVC_E_RANGE: with the range check on, an activation left the fp16 range (inf / NaN in an activation buffer)."""


_lib = None

# every symbol include/havc_mi355.h declares: (name, restype, argtypes)
_P, _I, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
SYMBOLS = [
    ("havc_create", _I, [C.POINTER(_P), _I]),
    ("havc_destroy", None, [_P]),
    ("havc_last_error", C.c_char_p, [_P]),
    ("havc_device_count", _I, []),
    ("havc_synchronize", _I, [_P]),
    ("havc_get_stats", _I, [_P, C.POINTER(Stats)]),
    ("havc_reset_stats", _I, [_P]),
    ("havc_version", C.c_char_p, []),
    ("havc_weights_load", _I, [_P, _P, _SZ, C.POINTER(_P)]),
    ("havc_weights_free", None, [_P]),
    ("havc_net_create", _I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    ("havc_net_free", None, [_P]),
    ("havc_net_run_rgb8", _I, [_P, _P, _P, _I]),
    ("havc_net_upload", _I, [_P, _I, _P, _SZ]),
    ("havc_net_download", _I, [_P, _I, _P, _SZ]),
    ("havc_net_run_ops", _I, [_P, _I, _I, _I]),
    ("havc_net_enqueue_ops", _I, [_P, _I, _I, _I]),
    ("havc_net_bind", _I, [_P, _I, _P]),
    ("havc_range_check_enable", _I, [_P, _I]),
    ("havc_net_range_stats", _I, [_P, _P, _P, _I]),
    ("havc_get_stream", _P, [_P]),
    ("havc_colormnet_rgb_to_lab", _I, [_P, _P, _P, _I, _I]),
    ("havc_colormnet_lab_to_rgb", _I, [_P, _P, _P, _P, _I, _I]),
    ("havc_net_profile", _I, [_P, _I, _P, _I]),
    ("havc_net_autotune", _I, [_P, _I, C.POINTER(_I)]),
    ("havc_device_name", _I, [_P, C.c_char_p, _I]),
    ("havc_net_get_cfg", _I, [_P, _I]),
    ("havc_net_set_cfg", _I, [_P, _I, _I]),
    ("havc_deoldify_frames", _I, [_P, _P, _P, _F, _I, _P, _P, _I]),
    ("havc_zhang_frames", _I, [_P, _P, _P, _P, _I, _I, _I]),
    ("havc_ddcolor_frames", _I, [_P, _P, _P, _P, _I, _I, _I]),
    ("havc_pil_resize", _I, [_P, _P, _I, _I, _P, _I, _I, _I]),
    ("havc_blend", _I, [_P, _P, _P, _F, _P, _I, _I]),
    ("havc_chroma_post_process", _I, [_P, _P, _P, _P, _I, _I]),
    ("havc_chroma_stabilizer", _I, [_P, _P, _P, _D, _D, _P, _I, _I]),
    ("havc_chroma_stabilizer_adaptive", _I, [_P, _P, _P, _D, _D, _D, _P, _I, _I]),
    ("havc_chroma_temporal_limiter", _I, [_P, _P, _P, _D, _P, _I, _I]),
    ("havc_color_temporal_stabilizer", _I, [_P, _P, _P, _I, _P, _I, _I]),
    ("havc_image_luma_merge", _I, [_P, _P, _P, _I, _D, _D, _P, _I, _I]),
    ("havc_image_luma", _I, [_P, _P, _I, _I, C.POINTER(C.c_double)]),
    ("havc_image_tweak", _I, [_P, _P, _P, _I, _I, _I, _F, _F, _F, C.POINTER(C.c_double), _I]),
    ("havc_image_chroma_tweak", _I, [_P, _P, _P, _I, _I, _D, _D, _I, _I, C.POINTER(C.c_double), _I, _D, _I, _D]),
    ("havc_luma_lut", _I, [_P, _P, _P, _P, _I, _I]),
    ("havc_restore_color_gradient", _I, [_P, _P, _P, _P, _I, _I, _D, _I, _D, _D, _I, _I]),
    ("havc_colorize_clip", _I, [_P, _P, _P, _F, _P, _P, _I, _I, _I]),
    ("havc_spline64_resize", _I, [_P, _P, _I, _I, _P, _I, _I, _P]),
    ("havc_spline64_resize_n", _I, [_P, _P, _I, _I, _P, _I, _I, _P, _I]),
    ("havc_colorize_clip_host", _I, [_P, _P, _P, _F, _P, _P, _I, _I, _I]),
    ("havc_host_alloc", _I, [_P, _SZ, C.POINTER(_P)]),
    ("havc_host_free", _I, [_P, _P]),
    ("havc_ddcolor_frame_planar_f", _I, [_P, _P, C.POINTER(_P), _I, C.POINTER(_P), _I, _I, _I, _I]),
    ("havc_deoldify_frame_planar", _I, [_P, _P, _P, _F, _I, C.POINTER(_P), _I, C.POINTER(_P), _I]),
    ("havc_planar_to_rgb8", _I, [_P, C.POINTER(_P), _I, _P, _I, _I]),
    ("havc_rgb8_to_planar", _I, [_P, _P, C.POINTER(_P), _I, _I, _I]),
    ("havc_memory_read_topk", _I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    ("havc_memory_read_topk_usage", _I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    ("havc_memory_dense_readout", _I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    ("havc_memory_similarity", _I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    ("havc_local_correlation", _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F]),
    ("havc_local_attention", _I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    ("havc_dev_alloc", _I, [_P, _SZ, C.POINTER(_P)]),
    ("havc_dev_free", _I, [_P, _P]),
    ("havc_dev_upload", _I, [_P, _P, _P, _SZ]),
    ("havc_dev_download", _I, [_P, _P, _P, _SZ]),
    ("havc_dev_copy", _I, [_P, _P, _P, _SZ]),
    ("havc_batcher_create", _I, [_P, _P, _P, _F, _I, _I, _I, C.POINTER(_P)]),
    ("havc_batcher_create_frames", _I, [_P, _I, _P, _I, _I, _I, _I, C.POINTER(_P)]),
    ("havc_batcher_submit", _I, [_P, _P, _P]),
    ("havc_batcher_stats", _I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("havc_batcher_free", None, [_P]),
    ("havc_cmn_frame_in", _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    ("havc_cmn_frame_out", _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    ("havc_memory_read_banked", _I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_int64, _I, _I]),
    ("havc_memory_read_reserve", _I, [_P, _I, _I, _I]),
    ("havc_cmn_short_term", _I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    ("havc_cmn_join_add", _I, [_P, _P, _P, C.c_int64]),
    ("havc_cmn_side_mark", _I, [_P]),
    ("havc_cmn_side_begin", _I, [_P]),
    ("havc_ctx_set_stream_priority", _I, [_P, _I]),
    ("havc_ctx_set_stream_cus", _I, [_P, _I]),
    ("havc_cmn_side_end", _I, [_P]),
    ("havc_cmn_side_wait", _I, [_P, _I]),
    ("havc_cmn_value_in", _I, [_P, _P, _P, _P, C.c_int64]),
    ("havc_dev_copy_2d", _I, [_P, _P, _SZ, _P, _SZ, _SZ, _SZ]),
    ("havc_net_bind_many", _I, [_P, _I, _P, _P]),
    ("havc_net_enqueue_slices", _I, [_P, _I, _P, _P, _P]),
    ("havc_debug_stream_jitter", _I, [_I, _I]),
    ("havc_build_stamp", C.c_char_p, []),
    ("havc_tag_timing_enable", _I, [_P, _I, _I]),
    ("havc_tag_timing_read", _I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
]


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so (same soname as /opt/rocm's).  Whichever is loaded first serves BOTH torch and this
    library; torch does not find its GPU on the system runtime, so when torch is installed its copy is loaded first — without importing
    torch.  The two then share one HIP runtime (what lets libhavc use torch tensors through their device pointers) whatever the import
    order.  HAVC_NO_TORCH_HIP_PRELOAD=1 skips this."""
    if os.environ.get("HAVC_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """dlopen the HIP library (idempotent).  Raises NativeLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  vsdeoldify_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc == HAVC_OK:
        return
    msg = load().havc_last_error(ctx)
    msg = msg.decode() if msg else ""
    if rc == HAVC_E_OOM:
        raise HavcOutOfMemory(msg or "out of device memory")
    if rc == HAVC_E_INVALID:
        raise ValueError(f"havc: invalid argument: {msg}")
    if rc == HAVC_E_NODEVICE:
        raise NativeLibraryError(f"havc: {msg}")
    if rc == HAVC_E_RANGE:
        raise HavcRangeError(f"havc: {msg}")
    raise RuntimeError(f"havc: HIP failure ({rc}): {msg}")


def as_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU + one HIP stream (device.set(DeviceId(n)) analogue, deoldify/_device.py:21-30)."""

    def __init__(self, device_id=0):
        lib = load()
        h = C.c_void_p()
        check(lib.havc_create(C.byref(h), int(device_id)), None)
        self.h, self.lib, self.device_id = h, lib, device_id

    def close(self):
        if getattr(self, "h", None):
            for lst in self.__dict__.get("_pool", {}).values():          # DeviceImage free list (device.py): raw hipMallocs, freed with the ctx
                while lst:
                    self.lib.havc_dev_free(self.h, lst.pop())
            self.__dict__["_pool_bytes"] = 0
            self.lib.havc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        s = Stats()
        check(self.lib.havc_get_stats(self.h, C.byref(s)), self.h)
        return s

    def reset_stats(self):
        check(self.lib.havc_reset_stats(self.h), self.h)

    def synchronize(self):
        check(self.lib.havc_synchronize(self.h), self.h)

    def range_check(self, enable=True):
        """the fp16 contract's debug switch (include/havc_mi355.h): enable BEFORE creating nets"""
        check(self.lib.havc_range_check_enable(self.h, 1 if enable else 0), self.h)

    def stream_ptr(self):
        """the ctx's hipStream_t as an integer (torch.cuda.ExternalStream(ptr) puts torch's work on the same stream)"""
        return int(self.lib.havc_get_stream(self.h) or 0)

    def device_name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.havc_device_name(self.h, buf, 256), self.h)
        return buf.value.decode()

    # ---- device memory for resident clips ----
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.havc_dev_alloc(self.h, int(nbytes), C.byref(p)), self.h)
        return p

    def dev_free(self, p):
        check(self.lib.havc_dev_free(self.h, p), self.h)

    def dev_upload(self, d, host):
        host = np.ascontiguousarray(host)
        check(self.lib.havc_dev_upload(self.h, d, as_ptr(host), host.nbytes), self.h)

    def dev_copy(self, d_dst, d_src, nbytes):
        check(self.lib.havc_dev_copy(self.h, d_dst, d_src, int(nbytes)), self.h)

    def host_alloc(self, nbytes):
        """pinned host memory as a uint8 numpy array (freed with host_free(arr))"""
        p = C.c_void_p()
        check(self.lib.havc_host_alloc(self.h, int(nbytes), C.byref(p)), self.h)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            check(self.lib.havc_host_free(self.h, p), self.h)

    def dev_download(self, host, d):
        assert host.flags.c_contiguous
        check(self.lib.havc_dev_download(self.h, as_ptr(host), d, host.nbytes), self.h)


class Weights:
    def __init__(self, ctx, blob):
        self.ctx = ctx
        blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        h = C.c_void_p()
        check(ctx.lib.havc_weights_load(ctx.h, as_ptr(blob), blob.nbytes, C.byref(h)), ctx.h)
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.havc_weights_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Net:
    """weights + plan for one input size S and up to max_batch frames in flight."""

    def __init__(self, ctx, weights, ops, bufs, in_buf, out_buf, S, max_batch):
        self.ctx, self.weights = ctx, weights
        self.ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
        self.bufs = np.ascontiguousarray(bufs, dtype=BUF_DTYPE)
        self.S, self.max_batch, self.in_buf, self.out_buf = S, max_batch, in_buf, out_buf
        h = C.c_void_p()
        check(ctx.lib.havc_net_create(ctx.h, weights.h, as_ptr(self.ops), len(self.ops), as_ptr(self.bufs),
                                      len(self.bufs), in_buf, out_buf, S, max_batch, C.byref(h)), ctx.h)
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.havc_net_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, buf, arr):
        arr = np.ascontiguousarray(arr)
        check(self.ctx.lib.havc_net_upload(self.h, buf, as_ptr(arr), arr.nbytes), self.ctx.h)

    def download(self, buf, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        check(self.ctx.lib.havc_net_download(self.h, buf, as_ptr(out), out.nbytes), self.ctx.h)
        return out

    def run_ops(self, first, count, batch=1):
        check(self.ctx.lib.havc_net_run_ops(self.h, first, count, batch), self.ctx.h)

    def enqueue_ops(self, first, count, batch=1):
        """run_ops without the blocking timer: only enqueues on the ctx stream"""
        check(self.ctx.lib.havc_net_enqueue_ops(self.h, first, count, batch), self.ctx.h)

    def bind(self, buf, ptr):
        """point buffer `buf` at caller-owned device memory (int / c_void_p; None restores the net's own allocation)"""
        check(self.ctx.lib.havc_net_bind(self.h, int(buf), C.c_void_p(ptr) if isinstance(ptr, int) else ptr), self.ctx.h)

    def run_rgb8_dev(self, d_in, d_out, batch):
        check(self.ctx.lib.havc_net_run_rgb8(self.h, d_in, d_out, batch), self.ctx.h)

    def autotune(self, batch=None):
        """measure the conv tile configurations once and keep the fastest per op (same bytes, only speed changes).  The result is
        remembered across processes in ~/.cache/havc_mi355/tune (HAVC_TUNE_CACHE=0 disables, or names another directory): keyed by
        the plan, the batch, the device name and the library build, restored through havc_net_set_cfg (which refuses stale ids)."""
        if getattr(self, "_tuned", False):
            return 0
        batch = int(batch or self.max_batch)
        path = self._tune_cache_path(batch)
        if path and os.path.isfile(path):
            try:
                cfgs = np.fromfile(path, dtype=np.int32)
                if len(cfgs) == len(self.ops) and all(self.ctx.lib.havc_net_set_cfg(self.h, i, int(c)) == 0 for i, c in enumerate(cfgs)):
                    self._tuned = True
                    return int((cfgs != 0).sum())
            except OSError:
                pass
            for i in range(len(self.ops)):                    # a stale / foreign file: back to the heuristic, then measure
                self.ctx.lib.havc_net_set_cfg(self.h, i, 0)
        n = C.c_int(0)
        check(self.ctx.lib.havc_net_autotune(self.h, batch, C.byref(n)), self.ctx.h)
        self._tuned = True
        if path:
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = f"{path}.{os.getpid()}.tmp"
                np.asarray(self.cfgs(), dtype=np.int32).tofile(tmp)
                os.replace(tmp, path)
            except OSError:
                pass
        return n.value

    def _tune_cache_path(self, batch):
        import hashlib
        root = os.environ.get("HAVC_TUNE_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "havc_mi355", "tune"))
        if root in ("0", ""):
            return None
        try:
            st = os.stat(LIB_PATH)
        except OSError:
            return None
        ops = self.ops.copy()
        ops["reserved"] = 0
        for f in ("w_off", "bias_off", "scale_off", "shift_off", "tag"):
            ops[f] = 0                                        # weight offsets / labels do not change what is fastest
        h = hashlib.sha1(ops.tobytes() + self.bufs.tobytes())
        h.update(f"{batch}|{self.ctx.device_name()}|{st.st_size}|{st.st_mtime_ns}".encode())
        return os.path.join(root, h.hexdigest() + ".i32")

    def range_stats(self):
        """(abs_max [n_ops], non_finite [n_ops]) of the destination buffers after the last range-checked run"""
        a, b = np.zeros(len(self.ops), np.float32), np.zeros(len(self.ops), np.int64)
        check(self.ctx.lib.havc_net_range_stats(self.h, as_ptr(a), as_ptr(b), len(self.ops)), self.ctx.h)
        return a, b

    def cfgs(self):
        return [self.ctx.lib.havc_net_get_cfg(self.h, i) for i in range(len(self.ops))]

    def profile(self, batch=1):
        ms = np.zeros(len(self.ops), dtype=np.float32)
        check(self.ctx.lib.havc_net_profile(self.h, batch, as_ptr(ms), len(ms)), self.ctx.h)
        return ms


class Batcher:
    """havc_batcher: merges concurrent per-frame calls (one per VapourSynth worker thread, vsslib/vsmodels.py:201-230) into batches."""

    def __init__(self, ctx, video, second=None, video_weight=0.0, post_process=True, wait_us=200, callers=0, kind=0, frame_hw=None):
        """kind 0: DeOldify (video [+ second] nets, S x S frames); 1: DDColor, 2: Zhang (`video` = the net, frames of frame_hw)"""
        self.ctx = ctx
        h = C.c_void_p()
        if kind == 0:
            self.shape = (video.S, video.S, 3)
            check(ctx.lib.havc_batcher_create(ctx.h, video.h, second.h if second else None, float(video_weight), 1 if post_process else 0,
                                              int(wait_us), int(callers), C.byref(h)), ctx.h)
        else:
            self.shape = (int(frame_hw[0]), int(frame_hw[1]), 3)
            check(ctx.lib.havc_batcher_create_frames(ctx.h, int(kind), video.h, self.shape[1], self.shape[0], int(wait_us), int(callers),
                                                     C.byref(h)), ctx.h)
        self.h = h
        self._nets = (video, second)                           # keep the nets alive as long as the batcher

    def submit(self, frame):
        """uint8 [S, S, 3] -> uint8 [S, S, 3]; blocks until the batch this frame rode in is done (the GIL is released meanwhile)"""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == self.shape, (frame.shape, self.shape)
        out = np.empty_like(frame)
        check(self.ctx.lib.havc_batcher_submit(self.h, as_ptr(frame), as_ptr(out)), self.ctx.h)
        return out

    def stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx.lib.havc_batcher_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.havc_batcher_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
