"""Images that stay in HBM between filters.

Every frame / filter entry point of libhavc_mi355 takes host OR device pointers per operand (include/havc_mi355.h,
"Pointers").  `DeviceImage` is the Python handle of a device operand: a uint8 [h, w, 3] frame or an [n, h, w, 3] stack in the
memory of one ctx's GPU.  The numpy-facing wrappers of imfilters.py / mcomb.py / render.py / ddcolor.py accept a DeviceImage
wherever they accept an ndarray and then return a DeviceImage: a whole HAVC merge graph (squash -> DeOldify -> DDColor ->
merge method -> up-sample + luma) composes without a single PCIe hop, and nothing blocks until `.numpy()` is called.

Allocation goes through a per-context free list (`hipMalloc` / `hipFree` synchronise the device): a buffer released by
`__del__` is handed to the next request of the same size.  All work of a ctx is ordered on its one stream, so recycling a
buffer behind work that is still queued is safe -- AS LONG AS the buffer comes from the pool of the context that runs the work.
That is the rule the wrappers follow: an output is allocated in the pool of the EXECUTING context (`DeviceImage(ctx, shape)`), never
in the pool of an operand that may belong to another context (two models side by side on two contexts, havc.py: a buffer taken from
the other context's pool could still be in use by work queued on that context's stream).  A buffer that crosses contexts (an operand
produced on one, consumed on the other) is handed over behind a `synchronize()` of the producer.
"""
import ctypes as C

import numpy as np

from . import _native as nat

import os

POOL_MAX_BYTES = int(os.environ.get("HAVC_POOL_MAX_MB", "8192")) << 20     # cached (idle) buffers per context; beyond it a release frees


def _pool(ctx):
    """the free list lives ON the context object: it dies with it (Context.close frees every cached buffer before havc_destroy) and can
    never be inherited by another context that happens to get the same id()"""
    p = ctx.__dict__.get("_pool")
    if p is None:
        p = ctx.__dict__["_pool"] = {}
        ctx.__dict__["_pool_bytes"] = 0
    return p


def pool_alloc(ctx, nbytes):
    free = _pool(ctx).get(nbytes)
    if free:
        ctx._pool_bytes -= nbytes
        return free.pop()
    try:
        return ctx.dev_alloc(nbytes)
    except nat.HavcOutOfMemory:
        pool_trim(ctx)                            # idle buffers of other sizes may be what is in the way: give them back and retry once
        return ctx.dev_alloc(nbytes)


def pool_release(ctx, ptr, nbytes):
    pool = _pool(ctx)
    if ctx._pool_bytes + nbytes > POOL_MAX_BYTES:
        ctx.dev_free(ptr)
        return
    pool.setdefault(nbytes, []).append(ptr)
    ctx._pool_bytes += nbytes


def pool_trim(ctx):
    """give every cached buffer of this ctx back to the driver"""
    for lst in _pool(ctx).values():
        while lst:
            ctx.dev_free(lst.pop())
    ctx._pool_bytes = 0


class DeviceImage:
    """uint8 image(s) in HBM: shape (h, w, 3) or (n, h, w, 3), C-contiguous, interleaved RGB."""

    def __init__(self, ctx, shape, ptr=None, owner=None):
        shape = tuple(int(s) for s in shape)
        if len(shape) not in (3, 4) or shape[-1] != 3:
            raise ValueError("DeviceImage: shape must be (h, w, 3) or (n, h, w, 3)")
        self.ctx, self.shape = ctx, shape
        self.nbytes = int(np.prod(shape))
        self._owner = owner                       # a view keeps its parent alive
        # complete: the contents were written by a BLOCKING call (from_numpy) and nothing has been enqueued into the buffer since: another context may
        # read it without ordering itself behind this one.  Views inherit it; everything an asynchronous op writes into stays False (the default).
        self._complete = False
        self._own = ptr is None
        self.ptr = pool_alloc(ctx, self.nbytes) if ptr is None else ptr

    @property
    def complete(self):
        """the owner of a view decides (a view of a buffer that has been written into asynchronously since is not complete either)"""
        return self._owner.complete if self._owner is not None else self._complete

    @complete.setter
    def complete(self, value):
        if self._owner is not None:
            self._owner.complete = value
        else:
            self._complete = bool(value)

    def __del__(self):
        try:
            if self._own and self.ptr is not None and self.ctx.h:
                pool_release(self.ctx, self.ptr, self.nbytes)
        except Exception:
            pass
        self.ptr = None

    # ---- host <-> device ----
    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        d = cls(ctx, arr.shape)
        ctx.dev_upload(d.ptr, arr)
        d.complete = True
        return d

    def numpy(self):
        out = np.empty(self.shape, np.uint8)
        self.ctx.dev_download(out, self.ptr)        # orders behind everything queued on the ctx stream
        return out

    # ---- geometry ----
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def n_frames(self):
        return self.shape[0] if len(self.shape) == 4 else 1

    @property
    def frame_shape(self):
        return self.shape[-3:]

    def frame(self, i):
        """view of frame i of a stack (no copy)"""
        if len(self.shape) != 4 or not 0 <= i < self.shape[0]:
            raise IndexError(i)
        fb = int(np.prod(self.shape[1:]))
        return DeviceImage(self.ctx, self.shape[1:], C.c_void_p(self.ptr.value + i * fb), owner=self)

    def frames(self, lo, hi):
        fb = int(np.prod(self.shape[1:]))
        return DeviceImage(self.ctx, (hi - lo,) + self.shape[1:], C.c_void_p(self.ptr.value + lo * fb), owner=self)

    def as_rows(self):
        """(n, h, w, 3) seen as ONE (n*h, w, 3) image: valid operand of every purely per-pixel filter"""
        if len(self.shape) == 3:
            return self
        n, h, w, _ = self.shape
        return DeviceImage(self.ctx, (n * h, w, 3), self.ptr, owner=self)

    def reshaped(self, shape):
        if int(np.prod(shape)) != self.nbytes:
            raise ValueError("reshape changes the size")
        return DeviceImage(self.ctx, shape, self.ptr, owner=self)

    def empty_like(self):
        return DeviceImage(self.ctx, self.shape)

    def copy_from(self, other):
        """device -> device copy (enqueued on the ctx stream)"""
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        self.ctx.dev_copy(self.ptr, other.ptr, self.nbytes)
        self.complete = False
        return self


def is_device(x):
    return isinstance(x, DeviceImage)


def operand_ptr(x):
    """ctypes pointer of an ndarray / DeviceImage operand"""
    return x.ptr if isinstance(x, DeviceImage) else nat.as_ptr(x)
