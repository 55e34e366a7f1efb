"""ColorMNet exemplar path — the first two kernels behind the reference's call shapes.

  match_memory_readout   get_similarity + do_softmax(top_k) + readout, colormnet/model/memory_util.py:7-80, as
                         MemoryManager.match_memory chains them once per frame (colormnet/inference/memory_manager.py:58-150)
  get_similarity         the dense similarity alone (memory consolidation, memory_manager.py:264)
  local_correlation      the SpatialCorrelationSampler call of LocalGatedPropagation (colormnet/model/attention.py:827-835)
  local_attention        LocalGatedPropagation.forward (use_linear=False, one head) up to agg_value (attention.py:783-856)

Operands are fp32 in the reference's layouts: numpy arrays, CPU torch tensors (staged by the library) or CUDA / ROCm torch
tensors of the ctx's GPU (used in place through their device pointer: nothing is copied).  The memory bookkeeping on top of these is
colormnet_memory.py (MemoryManager), the torch-module adapter colormnet_torch.py; the encoders / decoder of ColorMNet are not built
(sequential in time, one clip per GPU: replicas only).  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _native as nat
from .render import get_context


class _Op:
    """one fp32 operand: keeps the array / tensor alive and exposes a pointer"""

    def __init__(self, x):
        self.torch = None
        if x is None:
            self.ptr, self.shape, self.keep = None, None, None
            return
        if hasattr(x, "data_ptr"):                               # torch tensor
            t = x.detach()
            if str(t.dtype) != "torch.float32":
                t = t.float()
            t = t.contiguous()
            self.keep, self.ptr, self.shape, self.torch = t, C.c_void_p(t.data_ptr()), tuple(t.shape), t
        else:
            a = np.ascontiguousarray(x, dtype=np.float32)
            self.keep, self.ptr, self.shape = a, nat.as_ptr(a), a.shape


def _out_like(ref, shape):
    """output buffer of the same kind as the first operand (torch tensor on its device, or ndarray)"""
    if ref.torch is not None:
        import torch
        t = torch.empty(shape, dtype=torch.float32, device=ref.torch.device)
        return t, C.c_void_p(t.data_ptr())
    a = np.empty(shape, np.float32)
    return a, nat.as_ptr(a)


class ordered:
    """Orders a libhavc call on device tensors against torch's work.  libhavc enqueues on the ctx stream (non-blocking), torch on its
    current stream: when the two differ, torch's stream is drained BEFORE the call (its producers — cat, float(), contiguous(), the convs
    that made q / k / v — are asynchronous) and the ctx stream AFTER it (torch reads the outputs on its own stream).  When torch already
    runs on the ctx stream (torch.cuda.ExternalStream(ctx.stream_ptr()), as colormnet_net.ColorMNetNetwork does) nothing is needed.
    Tensors of another GPU than the ctx's are refused."""

    def __init__(self, ctx, *tensors):
        self.ctx, self.sync = ctx, False
        dev = [t for t in tensors if t is not None and hasattr(t, "is_cuda") and t.is_cuda]
        if dev:
            import torch
            for t in dev:
                if t.device.index != ctx.device_id:
                    raise ValueError(f"tensor on cuda:{t.device.index} handed to the libhavc context of GPU {ctx.device_id}")
            cur = torch.cuda.current_stream(dev[0].device)
            if cur.cuda_stream != ctx.stream_ptr():
                cur.synchronize()
                self.sync = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.sync:
            self.ctx.synchronize()
        return False


def get_similarity(mk, ms, qk, qe, device_index=0):
    """memory_util.py:7-39.  mk [B,CK,...], ms [B,1,...] or [B,...] or None, qk [B,CK,...], qe like qk or None -> [B,N,HW]"""
    ctx = get_context(device_index)
    m, q = _Op(mk), _Op(qk)
    B, CK = m.shape[:2]
    N, HW = int(np.prod(m.shape[2:])), int(np.prod(q.shape[2:]))
    s, e = _Op(ms), _Op(qe)
    out, optr = _out_like(m, (B, N, HW))
    with ordered(ctx, m.torch, s.torch, q.torch, e.torch):
        nat.check(ctx.lib.havc_memory_similarity(ctx.h, m.ptr, s.ptr, q.ptr, e.ptr, optr, B, CK, N, HW), ctx.h)
    return out


def match_memory_readout(mk, ms, qk, qe, mv, top_k=30, device_index=0):
    """readout(do_softmax(get_similarity(mk, ms, qk, qe), top_k), mv): [B,CV,HW].  mv [B,CV,N] (memory_manager._readout: v @ affinity)."""
    ctx = get_context(device_index)
    m, q, v = _Op(mk), _Op(qk), _Op(mv)
    B, CK = m.shape[:2]
    N, HW, CV = int(np.prod(m.shape[2:])), int(np.prod(q.shape[2:])), v.shape[1]
    if int(np.prod(v.shape[2:])) != N:
        raise ValueError("memory values and memory keys disagree on the number of memory elements")
    s, e = _Op(ms), _Op(qe)
    out, optr = _out_like(m, (B, CV, HW))
    with ordered(ctx, m.torch, s.torch, q.torch, e.torch, v.torch):
        nat.check(ctx.lib.havc_memory_read_topk(ctx.h, m.ptr, s.ptr, q.ptr, e.ptr, v.ptr, optr, B, CK, CV, N, HW, int(top_k)), ctx.h)
    return out


def local_correlation(q, k, max_dis=7, dilation=1, q_scale=1.0, device_index=0):
    """attention.py:827-835: [n,C,h,w] x [n,C,h,w] -> [n, 1, (2 max_dis + 1)^2, h*w]"""
    ctx = get_context(device_index)
    a, b = _Op(q), _Op(k)
    n, c, h, w = a.shape
    ws = 2 * max_dis + 1
    out, optr = _out_like(a, (n, 1, ws * ws, h * w))
    with ordered(ctx, a.torch, b.torch):
        nat.check(ctx.lib.havc_local_correlation(ctx.h, a.ptr, b.ptr, optr, n, c, h, w, int(max_dis), int(dilation), float(q_scale)), ctx.h)
    return out


def local_attention(q, k, v, rel_w, rel_b, max_dis=7, dilation=1, device_index=0):
    """attention.py:783-856 (use_linear=False, num_head=1): returns (agg [h*w, n, Cv], local_attn [n, 1, ws*ws, h*w])"""
    ctx = get_context(device_index)
    a, b, vv, rw, rb = _Op(q), _Op(k), _Op(v), _Op(rel_w), _Op(rel_b)
    n, c, h, w = a.shape
    cv = vv.shape[1]
    ws = 2 * max_dis + 1
    agg, aptr = _out_like(a, (h * w, n, cv))
    attn, tptr = _out_like(a, (n, 1, ws * ws, h * w))
    with ordered(ctx, a.torch, b.torch, vv.torch, rw.torch, rb.torch):
        nat.check(ctx.lib.havc_local_attention(ctx.h, a.ptr, b.ptr, vv.ptr, rw.ptr, rb.ptr, aptr, tptr, n, c, cv, h, w, int(max_dis), int(dilation)), ctx.h)
    return agg, attn
