"""Weight packing and execution-plan emission for libhavc_mi355.so (host side, numpy only).

Offline-converter half of the drop-in: takes a reference state dict (fastai/basic_train.py:264-286
semantics: `{'model': sd}` or a bare dict), resolves spectral / weight norm with the STORED u, v
(SURVEY.md App. B), folds conv->BN where no activation sits between them, and lays every conv out as
the fp16 [Npad][tap][cin/8][8] matrix the implicit-GEMM kernel streams (csrc/conv_igemm.hip).
"""
from dataclasses import dataclass, field

import numpy as np

from . import _native as nat

EPS = 1e-5
TAG_TAIL_RES = 1          # the 259->259 (303->303) 3x3 res-block convs: the dominant kernel
TAG_FIRST_FREE = 16


def pad_to(n, m):
    return (n + m - 1) // m * m


def to_np(sd):
    """Accept torch tensors or numpy arrays; return {name: float32/64 ndarray}."""
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]                                    # Learner.load accepts {'model','opt'}
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def fold_spectral(sd, p):
    """W = weight_orig / (u . (W_mat v)) — eval-mode spectral_norm, stored u/v, no power iteration."""
    w = sd[p + ".weight_orig"].astype(np.float32)
    wm = w.reshape(w.shape[0], -1)
    sigma = np.dot(sd[p + ".weight_u"].astype(np.float32), wm @ sd[p + ".weight_v"].astype(np.float32))
    return w / np.float32(sigma)


def fold_weightnorm(sd, p):
    v = sd[p + ".weight_v"].astype(np.float32)
    g = sd[p + ".weight_g"].astype(np.float32)
    n = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape(-1, *([1] * (v.ndim - 1)))
    return g * v / n


def conv_weight(sd, p):
    if p + ".weight_orig" in sd:
        return fold_spectral(sd, p)
    if p + ".weight_g" in sd:
        return fold_weightnorm(sd, p)
    return sd[p + ".weight"].astype(np.float32)


def bn_scale_shift(sd, p):
    s = sd[p + ".weight"].astype(np.float32) / np.sqrt(sd[p + ".running_var"].astype(np.float32) + np.float32(EPS))
    return s, sd[p + ".bias"].astype(np.float32) - sd[p + ".running_mean"].astype(np.float32) * s


class WeightPack:
    """Append-only device blob; every tensor 256-byte aligned."""

    def __init__(self):
        self.parts, self.size = [], 0

    def add(self, arr):
        b = np.ascontiguousarray(arr).tobytes()
        off = self.size
        pad = (-len(b)) % 256
        self.parts.append(b + b"\0" * pad)
        self.size += len(b) + pad
        return off

    def blob(self):
        return b"".join(self.parts)


@dataclass
class View:
    """A logical C-channel NHWC tensor living in buffer `buf` at channel offset `coff`."""
    buf: int
    coff: int
    cpitch: int
    H: int
    W: int
    C: int                       # logical channels
    span: int                    # stored channels (multiple of 8) starting at coff
    cmap: np.ndarray = None      # position (relative to coff) of each logical channel

    def __post_init__(self):
        if self.cmap is None:
            self.cmap = np.arange(self.C)


@dataclass
class PackedConv:
    w_off: int
    bias_off: int
    scale_off: int
    shift_off: int
    Kc: int
    Npad: int
    Ci: int
    Cout: int
    Cin: int
    kh: int
    kw: int
    C8a: int = 0                 # chunks per tap in the main K segment (== Ci/8 when K is not split)
    pscale: float = 0.0          # precise packing: the factor between the MFMA accumulator and the convolution (op.f3); 0 = plain fp16 packing


def pitch_for(span):
    """channel pitch of a buffer holding `span` stored channels: 128-byte aligned rows (multiple of 64 channels) for
    everything wider than 64 channels — LDS-DMA line fetches halve in speed on unaligned pitches (dma_bench)."""
    return span if span <= 64 else pad_to(span, 64)


def split_weights(flat):
    """precise packing of a [Npad][K] fp32 weight matrix: three K segments for the unchanged fp16 MFMA main loop,
        [2^11 w_hi | 2^11 w_lo | w_hi]   with  w_hi = fp16(w / 2^s), w_lo = w / 2^s - w_hi,
    walked by the K table as x_hi, x_hi, x_lo' (x_lo' = 2^11 (x - x_hi), the activation's lo plane): the accumulator holds
    2^(11-s) x (x_hi w_hi + x_hi w_lo + x_lo w_hi) -- the convolution up to the 2^-22 x_lo w_lo term.  s >= 0 keeps 2^11 w_hi inside
    fp16 (|w| < 32 needs none; BN-folded encoder weights of a real checkpoint may).  Returns (fp16 [Npad][3K], accumulator scale 2^(s-11))."""
    w = np.array(flat, np.float32)          # private copy: the arithmetic below runs in place (these matrices are hundreds of MB)
    amax = float(np.abs(w).max()) if w.size else 0.0
    s = 0
    while amax / 2.0 ** s >= 31.0:
        s += 1
    if s:
        w /= np.float32(2.0 ** s)
    K = w.shape[1]
    out = np.empty((w.shape[0], 3 * K), np.float16)
    hi = w.astype(np.float16)
    out[:, 2 * K:] = hi
    hf = hi.astype(np.float32)
    w -= hf                                 # exact in fp32
    w *= np.float32(2048.0)
    out[:, K:2 * K] = w                     # rounds to fp16 on assignment
    hf *= np.float32(2048.0)
    out[:, :K] = hf                         # exact (|w| < 31)
    return out, float(2.0 ** (s - 11))


def _runs(idx):
    """maximal runs of consecutive destination positions: (first destination, first source index, length) for idx = destination of every source index"""
    idx = np.asarray(idx)
    out, s0 = [], 0
    for i in range(1, len(idx) + 1):
        if i == len(idx) or idx[i] != idx[i - 1] + 1:
            out.append((int(idx[s0]), s0, i - s0))
            s0 = i
    return out


def pack_conv(pack, W, cmap, Ci, bias=None, scale=None, shift=None, pixshuf=False, omap=None, ospan=None, precise=False):
    """W [Cout, Cin, KH, KW] fp32 -> fp16 [Npad][KH*KW][Ci/8][8] (+ fp32 bias/scale/shift [Npad]).
    cmap: position of every logical input channel inside the Ci-wide input span;
    omap/ospan: position of every logical output channel inside the ospan-wide output span.
    precise: the three-segment hi / lo packing of split_weights (HAVC_F_PRECISE convs; Kc is then 3 x the plain count)."""
    Cout, Cin, KH, KW = W.shape
    assert len(cmap) == Cin and Ci % 8 == 0
    if omap is None:
        omap, ospan = np.arange(Cout), Cout
    Npad = pad_to(ospan, 16)
    # scatter W into its padded row / channel positions run by run (round 6: the fancy-indexed double copy through a temporary was a third of a generator's packing
    # time -- 225 M parameters, 14 - 30 s per wide generator on the host; same values)
    Wt = np.zeros((Npad, KH, KW, Ci), np.float32)
    Wp = np.asarray(W, np.float32).transpose(0, 2, 3, 1)
    for od, os_, ol in _runs(omap):
        for cd, cs, cl in _runs(cmap):
            Wt[od:od + ol, :, :, cd:cd + cl] = Wp[os_:os_ + ol, :, :, cs:cs + cl]

    def vec(v):
        if v is None:
            return None
        o = np.zeros(Npad, np.float32)
        o[omap] = v
        return o
    bias, scale, shift = vec(bias), vec(scale), vec(shift)
    if pixshuf:
        assert Cout % 16 == 0 and Npad == Cout
        cps = Cout // 4
        # packed row q*cps + c  <-  original row c*4 + q   (PixelShuffle(2): q = dy*2 + dx)
        perm = (np.arange(cps)[None, :] * 4 + np.arange(4)[:, None]).reshape(-1)
        if pixshuf == "blur":
            # HAVC_F_PS_BLUR: packed row (c // 64) * 256 + q * 64 + c % 64: one 256-column tile = 4 sub-pixels x 64 channels.  A channel
            # count that is not a multiple of 64 (DynamicUnetDeep: 300, 336) is padded with zero rows to the next one; the epilogue
            # stores only the real channels (op.aux0).
            cpp = pad_to(cps, 64)
            cc = np.arange(cps)
            Npad = 4 * cpp
            W2 = np.zeros((Npad,) + Wt.shape[1:], np.float32)
            vs = [None if v is None else np.zeros(Npad, np.float32) for v in (bias, scale, shift)]
            for q in range(4):
                rows = (cc // 64) * 256 + q * 64 + cc % 64
                W2[rows] = Wt[cc * 4 + q]
                for dst, src in zip(vs, (bias, scale, shift)):
                    if dst is not None:
                        dst[rows] = src[cc * 4 + q]
            Wt, (bias, scale, shift) = W2, vs
        else:
            Wt = Wt[perm]
            bias = None if bias is None else bias[perm]
            scale = None if scale is None else scale[perm]
            shift = None if shift is None else shift[perm]
    # K order (must match havc_net_create's K table).  Main segment = chunks [0, C8a) of every tap, remainder segment =
    # chunks [C8a, C8); each segment padded to a multiple of 8 chunks (259 = 256 + 3: 32 chunks x 9 taps, then 9 single
    # chunks of x0).  When C8a is a multiple of 8 the main segment is ordered CHANNEL-GROUP MAJOR: for every group of 8
    # chunks (64 channels = one 128-byte line per pixel), all taps in turn.  A 64-deep stage of the pipelined kernel is
    # then one line of one tap, and the 9 taps of a group re-read the same lines back to back (L1/L2 hits) instead of
    # cycling through the whole 512-byte pixel between re-uses (tap-major order overflowed the 4 MiB L2 of an XCD).
    C8 = Ci // 8
    C8a = C8 - C8 % 8 if (C8 % 8 != 0 and C8 >= 16) else C8
    W5 = Wt.reshape(Npad, KH * KW, C8, 8)
    segs = []
    for lo, hi in ((0, C8a), (C8a, C8)):
        if hi > lo:
            seg = W5[:, :, lo:hi, :]
            if lo == 0 and C8a % 8 == 0:
                seg = seg.reshape(Npad, KH * KW, C8a // 8, 64).transpose(0, 2, 1, 3)      # [n][group][tap][64]
            klen = KH * KW * (hi - lo) * 8
            segs.append((seg, klen, klen + (-klen) % 64))
    ktot = sum(kp for _, _, kp in segs)
    pscale = 0.0
    if precise:
        flat = np.zeros((Npad, ktot), np.float32)
        off = 0
        for seg, klen, kp in segs:
            flat[:, off:off + klen] = seg.reshape(Npad, klen)
            off += kp
        flat, pscale = split_weights(flat)
        flat = flat.astype(np.float16)
    else:
        flat = np.zeros((Npad, ktot), np.float16)          # written segment by segment: the assignment rounds to fp16 like astype (no padded / concatenated fp32 copies)
        off = 0
        for seg, klen, kp in segs:
            flat[:, off:off + klen] = seg.reshape(Npad, klen)
            off += kp
    Kc = flat.shape[1] // 8
    return PackedConv(pack.add(flat), -1 if bias is None else pack.add(bias), -1 if scale is None else pack.add(scale),
                      -1 if shift is None else pack.add(shift), Kc, Npad, Ci, Cout, Cin, KH, KW, C8a, pscale)


# op types that exist in precise form (HAVC_F_PRECISE): the DeOldify generators (round 4: csrc/precise.hip), the Zhang colorizers and DDColor (round 5:
# csrc/precise2.hip, csrc/zhang.hip); BILINEAR2 works on fp32 maps in both modes
PRECISE_OPS = (nat.OP_CONV, nat.OP_MAXPOOL, nat.OP_BLUR_RESIZE, nat.OP_AFFINE, nat.OP_ATTENTION, nat.OP_PREP_RGB8, nat.OP_SUBSAMPLE2, nat.OP_PROJ2,
               nat.OP_BILINEAR2, nat.OP_PREP_LAB_L, nat.OP_DWCONV7, nat.OP_LAYERNORM, nat.OP_MHA, nat.OP_PREP_DDCOLOR, nat.OP_FOLD_QUERIES,
               nat.OP_SHUF4_BLUR_AB, nat.OP_DWCONV7_LN)


class PlanBuilder:
    def __init__(self, precise=False):
        """precise: every tensor is a hi / lo pair of fp16 planes in one buffer (pixel row = [hi: P | lo: P], View.cpitch = 2 P) and every
        op carries HAVC_F_PRECISE (include/havc_mi355.h); the op types of PRECISE_OPS exist in that form."""
        self.ops, self.bufs, self.names, self.precise = [], [], [], precise
        self.pm = 2 if precise else 1          # pitch multiplier

    def buf(self, elems_per_frame, elem_bytes=2, zero_init=False):
        self.bufs.append((int(elems_per_frame), elem_bytes, 1 if zero_init else 0))
        return len(self.bufs) - 1

    def tensor(self, H, W, C, zero_init=True):
        """fresh buffer holding one logical tensor; zero_init keeps pad channels 0 forever."""
        span = pad_to(C, 8)
        pitch = pitch_for(span) * self.pm
        return View(self.buf(H * W * pitch, 2, zero_init and span != C), 0, pitch, H, W, C, span)

    def _op(self, name, tag=None, **kw):
        op = np.zeros((), dtype=nat.OP_DTYPE)
        op["src2"] = -1
        for f in ("w_off", "bias_off", "scale_off", "shift_off"):
            op[f] = -1
        for k, v in kw.items():
            op[k] = v
        if self.precise:
            assert int(op["type"]) in PRECISE_OPS, name
            op["flags"] |= nat.F_PRECISE
        if tag is None:
            tag = TAG_FIRST_FREE + len(self.names)
        op["tag"] = tag
        self.names.append(name)
        self.ops.append(op)
        return len(self.ops) - 1

    def conv(self, name, pc, x, y, stride=1, pad=0, dil=1, flags=0, res=None, tag=None, f=(0, 0, 0, 0), Co=None,
             aux0=0, pad_w=None, out_hw=None, out_step=1, out_oy=0, out_ox=0, proj=None):
        """y may be a View (fp16 NHWC / pixel-shuffled target) or a raw buffer id (RGB8 / transposed).
        out_step=2 scatters output pixel (ho, wo) to (2*ho + out_oy, 2*wo + out_ox) of y (ConvTranspose parity convs,
        which also use dil=-1, asymmetric pad (pad_w) and an explicit out_hw)."""
        assert x.span == pc.Ci, (name, x.span, pc.Ci)
        assert bool(pc.pscale) == self.precise, name
        if self.precise:
            f = (f[0], f[1], f[2], pc.pscale)
        if out_hw is not None:
            Ho, Wo = out_hw
        else:
            Ho = (x.H + 2 * pad - dil * (pc.kh - 1) - 1) // stride + 1
            Wo = (x.W + 2 * pad - dil * (pc.kw - 1) - 1) // stride + 1
        kw = dict(type=nat.OP_CONV, flags=flags, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, Hi=x.H, Wi=x.W,
                  Ci=pc.Ci, Ho=Ho, Wo=Wo, kh=pc.kh, kw=pc.kw, stride=stride, pad=pad, dil=dil, Kc=pc.Kc,
                  Npad=pc.Npad, w_off=pc.w_off, bias_off=pc.bias_off, scale_off=pc.scale_off, shift_off=pc.shift_off,
                  f0=f[0], f1=f[1], f2=f[2], f3=f[3], aux0=aux0, aux1=pc.C8a,
                  pad_w_delta=(0 if pad_w is None else pad_w - pad), out_step=out_step, out_oy=out_oy, out_ox=out_ox,
                  flops=2 * Ho * Wo * pc.Cout * pc.Cin * pc.kh * pc.kw)
        if isinstance(y, View):
            kw.update(dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch)
            if flags & nat.F_OUT_PIXSHUF:
                assert y.H == 2 * Ho and y.W == 2 * Wo and pc.Cout == 4 * y.C and y.C % 4 == 0, name
                kw["Co"] = y.C
                if flags & nat.F_PS_BLUR:
                    assert pc.kh == 1 and stride == 1 and pad == 0 and pc.Npad == 4 * pad_to(y.C, 64), name
                    kw["Co"] = pc.Npad // 4                    # channels per sub-pixel in the packed rows (zero rows beyond y.C)
                    kw["aux0"] = y.span                        # channels actually stored
            else:
                assert y.H == Ho * out_step and y.W == Wo * out_step and y.C == pc.Cout, (name, y.H, Ho, y.C, pc.Cout)
                kw["Co"] = y.span
        else:
            kw.update(dst=y, Co=Co)
        if res is not None:
            assert flags & nat.F_RESIDUAL
            kw.update(src2=res.buf, res_coff=res.coff, res_cpitch=res.cpitch)
        if proj is not None:                                   # (matrix buffer, output buffer) of HAVC_F_FUSE_PROJ
            assert flags & nat.F_FUSE_PROJ and res is None and pc.Npad % 256 == 0 and (Ho * Wo) % 16 == 0
            kw.update(src2=proj[0], aux0=proj[1], dst=proj[1], Co=pc.Npad)
        return self._op(name, tag, **kw)

    def maxpool(self, name, x, y):
        return self._op(name, type=nat.OP_MAXPOOL, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf,
                        dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span)

    def blur_resize(self, name, x, y):
        assert x.span == y.span
        return self._op(name, type=nat.OP_BLUR_RESIZE, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf,
                        dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span)

    def affine(self, name, x, y, scale_off, shift_off, relu):
        assert x.span == y.span and x.H == y.H
        return self._op(name, type=nat.OP_AFFINE, flags=nat.F_RELU_POST if relu else 0, src=x.buf, src_coff=x.coff,
                        src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W,
                        Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span, scale_off=scale_off, shift_off=shift_off)

    def attention(self, name, x, qk, d, vT_buf, npitch, y, gamma, transposed=False):
        """fast plans: vT_buf = transposed value buffer [C][npitch]; precise plans: vT_buf = the NHWC value buffer, npitch = its pixel pitch -- or, with
        transposed=True (round 5), the value map as two transposed planes [2][C][npitch] (hi, lo): the P . H product then runs on MFMA"""
        N = x.H * x.W
        extra = dict(kh=self.buf(N * 2, 4)) if self.precise else {}          # precise: fp32 [N][2] softmax statistics per frame
        if self.precise and transposed:
            extra["flags"] = nat.F_OUT_TRANSPOSED
        return self._op(name, type=nat.OP_ATTENTION, **extra, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf,
                        dst_coff=y.coff, dst_cpitch=y.cpitch, src2=qk.buf, res_coff=qk.coff, res_cpitch=qk.cpitch,
                        Hi=x.H, Wi=x.W, Ci=x.C, Ho=x.H, Wo=x.W, Co=x.C, aux0=d, aux1=vT_buf, Kc=npitch, f0=gamma,
                        flops=2 * N * N * d + 2 * N * N * x.C)

    def subsample2(self, name, x, y):
        assert y.H == (x.H + 1) // 2 and y.W == (x.W + 1) // 2 and x.span == y.span
        return self._op(name, type=nat.OP_SUBSAMPLE2, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf,
                        dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span)

    def proj2(self, name, x, w_off, bias_off, mode, mul, out_buf):
        """per pixel C -> 2 projection in fp32 (mode 1: softmax first, mode 2: + bias, tanh); out_buf: fp32 [H*W*2]."""
        return self._op(name, type=nat.OP_PROJ2, flags=mode, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=out_buf,
                        Hi=x.H, Wi=x.W, Ci=x.C, Ho=x.H, Wo=x.W, Co=2, w_off=w_off, bias_off=bias_off, f0=mul,
                        flops=2 * x.H * x.W * x.C * 2)

    def bilinear2(self, name, src_buf, Hi, Wi, dst_buf, Ho, Wo, mul):
        return self._op(name, type=nat.OP_BILINEAR2, src=src_buf, dst=dst_buf, Hi=Hi, Wi=Wi, Ci=2, Ho=Ho, Wo=Wo, Co=2, f0=mul)

    def prep_lab_l(self, name, in_buf, S, y):
        return self._op(name, type=nat.OP_PREP_LAB_L, src=in_buf, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=S, Wi=S,
                        Ci=8, Ho=S, Wo=S, Co=8)

    def prep_rgb8(self, name, in_buf, S, y0, y1=None, y1_fill=0):
        """y1_fill: pad channels behind y1's 8-channel slot that no op reads and the kernel may zero as well (whole 64-byte stores)"""
        kw = dict(type=nat.OP_PREP_RGB8, src=in_buf, dst=y0.buf, dst_coff=y0.coff, dst_cpitch=y0.cpitch, Hi=S, Wi=S,
                  Ci=8, Ho=S, Wo=S, Co=8)
        if y1 is not None:
            kw.update(src2=y1.buf, res_coff=y1.coff, res_cpitch=y1.cpitch, aux0=0 if self.precise else y1_fill)
        return self._op(name, **kw)

    # ---- DDColor ops (csrc/ddcolor.hip) ----
    def dwconv7(self, name, x, y, w_off, bias_off, w_pitch):
        assert x.span == y.span and x.H == y.H and x.W == y.W
        return self._op(name, type=nat.OP_DWCONV7, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff,
                        dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span, w_off=w_off, bias_off=bias_off,
                        Kc=w_pitch, kh=7, kw=7, flops=2 * x.H * x.W * x.C * 49)

    def dwconv7_ln(self, name, x, y, w_off, bias_off, w_pitch, gamma_off, beta_off, eps):
        """depthwise 7x7 + LayerNorm over the channels in one kernel (ConvNeXt block head; instantiated for the ConvNeXt widths)."""
        assert x.span == y.span == x.C and x.C in (64, 192, 384, 768, 1536) and x.H == y.H and x.W == y.W
        return self._op(name, type=nat.OP_DWCONV7_LN, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff,
                        dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span, w_off=w_off, bias_off=bias_off,
                        Kc=w_pitch, kh=7, kw=7, scale_off=gamma_off, shift_off=beta_off, f0=eps, flops=2 * x.H * x.W * x.C * 49)

    def fold_queries(self, name, emb, n_queries, r_off, r_pitch, dst_buf):
        """M[o][c] = sum_q R[o][q] emb[q][c] (fp32 [2][C] per frame in dst_buf): einsum + refine conv folded, DDColor tail."""
        assert emb.H == 1 and emb.span == emb.C
        return self._op(name, type=nat.OP_FOLD_QUERIES, src=emb.buf, src_coff=emb.coff, src_cpitch=emb.cpitch, dst=dst_buf, Hi=1, Wi=emb.W,
                        Ci=emb.C, Ho=n_queries, Wo=1, Co=emb.C, w_off=r_off, Kc=r_pitch, flops=2 * 2 * n_queries * emb.C)

    def shuf4_blur_ab(self, name, proj_buf, Hi, Wi, img, rimg_off, bias_off, y, flops=0):
        """PixelShuffle(4) + blur of the projected 2-channel map + R_img . image + bias -> y channels 0-1."""
        assert y.H == 4 * Hi and y.W == 4 * Wi and img.H == y.H and img.W == y.W
        return self._op(name, type=nat.OP_SHUF4_BLUR_AB, src=proj_buf, src2=img.buf, res_coff=img.coff, res_cpitch=img.cpitch, dst=y.buf,
                        dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=Hi, Wi=Wi, Ci=2, Ho=y.H, Wo=y.W, Co=2, w_off=rimg_off, bias_off=bias_off, flops=flops)

    def shuf4_blur_proj(self, name, x, m_buf, img, rimg_off, bias_off, y, flops=0):
        """precise form of the DDColor tail (HAVC_OP_SHUF4_BLUR_AB with HAVC_F_PRECISE): x = the last_shuf conv's [Hi][Wi][16 * 256] pair tensor,
        m_buf = the folded einsum + refine projection (fp32 [2][256] per frame, fold_queries): PixelShuffle(4) + blur + projection + image term -> y channels 0-1."""
        assert self.precise and x.C == 16 * 256 and y.H == 4 * x.H and y.W == 4 * x.W and img.H == y.H and img.W == y.W
        return self._op(name, type=nat.OP_SHUF4_BLUR_AB, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, src2=img.buf, res_coff=img.coff, res_cpitch=img.cpitch,
                        aux0=m_buf, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.C, Ho=y.H, Wo=y.W, Co=2, w_off=rimg_off,
                        bias_off=bias_off, flops=flops)

    def layernorm(self, name, x, y, gamma_off, beta_off, eps, relu=False):
        assert x.C == y.C and x.H * x.W == y.H * y.W
        return self._op(name, type=nat.OP_LAYERNORM, flags=(nat.F_RELU_POST if relu else 0), src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff,
                        dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.C, Ho=y.H, Wo=y.W, Co=y.C, scale_off=gamma_off, shift_off=beta_off, f0=eps)

    def mha(self, name, q, kv, k_coff, v_coff, y, heads, n_q, n_k, scale):
        """q / y: token views [1, tokens_per_frame, E] of which the first n_q rows are queries; kv: view over the K/V buffer
        [*, tokens, >= E] with K at channel k_coff and V at v_coff, n_k keys."""
        assert q.H == 1 and y.H == 1 and q.W == y.W and q.C == heads * 32
        part = self.buf(heads * ((n_k + 255) // 256) * n_q * 34, 4)          # partial softmax states of the key-split kernel
        return self._op(name, type=nat.OP_MHA, aux1=part, src=q.buf, src_coff=q.coff, src_cpitch=q.cpitch, src2=kv.buf, res_cpitch=kv.cpitch,
                        res_coff=kv.coff + k_coff, aux0=kv.coff + v_coff, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=n_q, Wi=q.W,
                        Ci=q.C, Ho=n_k, Wo=kv.H * kv.W, Co=q.C, kh=heads, f0=scale, flops=4 * n_q * n_k * q.C)

    def pixshuf4_blur(self, name, x, y):
        assert y.H == 4 * x.H and y.W == 4 * x.W and x.C == 16 * y.C and y.C % 8 == 0
        return self._op(name, type=nat.OP_PIXSHUF4_BLUR, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff,
                        dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.C)

    def prep_ddcolor(self, name, in_buf, S, y0, y1=None):
        kw = dict(type=nat.OP_PREP_DDCOLOR, src=in_buf, dst=y0.buf, dst_coff=y0.coff, dst_cpitch=y0.cpitch, Hi=S, Wi=S, Ci=8, Ho=S, Wo=S, Co=8)
        if y1 is not None:
            kw.update(src2=y1.buf, res_coff=y1.coff, res_cpitch=y1.cpitch)
        return self._op(name, **kw)

    def conv_dyn(self, name, x, wview, y, n_rows):
        """1x1 conv whose weights are activations (HAVC_F_W_FROM_BUF): wview = token view [1, >= Npad rows, Cin] whose row pitch
        equals x.span (packed fp16 rows), n_rows real output channels."""
        Npad = pad_to(n_rows, 16)
        assert wview.cpitch == x.span and wview.coff == 0 and wview.H * wview.W >= Npad and x.span % 64 == 0
        Kc = x.span // 8
        return self._op(name, type=nat.OP_CONV, flags=nat.F_W_FROM_BUF, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, src2=wview.buf,
                        dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.span, Ho=x.H, Wo=x.W, Co=y.span, kh=1, kw=1,
                        stride=1, pad=0, dil=1, Kc=Kc, Npad=Npad, aux1=Kc, out_step=1, flops=2 * x.H * x.W * n_rows * x.C)

    # ---- ColorMNet ops (csrc/colormnet_net.hip) ----
    def ew(self, name, x, y, mode=0, ratio=(1.0, 1.0), factor=1, res=None, src_bcast=False, res_bcast=False, relu=False, dual=None):
        """copy / bilinear (align_corners False; ratio = source / destination as aten computes it) / area resample of view x into view y,
        optional + res, optional ReLU, optional second rectified output `dual` (a View)."""
        assert x.span == y.span and (res is None or res.span == x.span) and (dual is None or dual.span == x.span), name
        flags = (nat.EW_SRC_BCAST if src_bcast else 0) | (nat.EW_RES if res is not None else 0) | (nat.EW_RES_BCAST if res_bcast else 0) | \
                (nat.EW_RELU if relu else 0) | (nat.EW_DUAL if dual is not None else 0)
        kw = dict(type=nat.OP_EW, flags=flags, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch,
                  Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span, kh=mode, kw=factor, f0=ratio[0], f1=ratio[1])
        if res is not None:
            kw.update(src2=res.buf, res_coff=res.coff, res_cpitch=res.cpitch)
        if dual is not None:
            kw.update(aux0=dual.buf, aux1=dual.coff, Kc=dual.cpitch)
        return self._op(name, **kw)

    def dwconv(self, name, x, y, w_off, bias_off, w_pitch, k):
        assert x.span == y.span and x.H == y.H and x.W == y.W
        return self._op(name, type=nat.OP_DWCONV, src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch,
                        Hi=x.H, Wi=x.W, Ci=x.span, Ho=y.H, Wo=y.W, Co=y.span, w_off=w_off, bias_off=bias_off, Kc=w_pitch, kh=k, kw=k,
                        flops=2 * x.H * x.W * x.C * k * k)

    def chan_attn(self, name, q, k, heads, temp_off, w_buf, w_kc, part_g, part_n):
        assert q.span == k.span == q.C and q.H == k.H and q.W == k.W and q.C % heads == 0
        c = q.C // heads
        return self._op(name, type=nat.OP_CHAN_ATTN, src=q.buf, src_coff=q.coff, src_cpitch=q.cpitch, src2=k.buf, res_coff=k.coff, res_cpitch=k.cpitch,
                        dst=w_buf, Hi=q.H, Wi=q.W, Ci=q.C, Ho=q.H, Wo=q.W, Co=q.C, kh=heads, Kc=w_kc, scale_off=temp_off, aux0=part_g, aux1=part_n,
                        flops=2 * q.H * q.W * heads * c * c)

    def mha64(self, name, qkv, q_coff, k_coff, v_coff, y, heads, live, scale):
        assert qkv.H == 1 and y.H == 1 and y.W == qkv.W and y.C == heads * 64
        return self._op(name, type=nat.OP_MHA64, src=qkv.buf, src_coff=qkv.coff + q_coff, src_cpitch=qkv.cpitch, res_coff=qkv.coff + k_coff,
                        aux0=qkv.coff + v_coff, dst=y.buf, dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=1, Wi=qkv.W, Ci=heads * 64, Ho=live, Wo=qkv.W,
                        Co=heads * 64, kh=heads, f0=scale, flops=4 * live * live * heads * 64)

    def cbam(self, name, x, y, w_off, scale_buf, comp_buf, dual=None):
        assert x.span == y.span == x.C and x.H == y.H
        kw = dict(type=nat.OP_CBAM, flags=(nat.EW_DUAL if dual is not None else 0), src=x.buf, src_coff=x.coff, src_cpitch=x.cpitch, dst=y.buf,
                  dst_coff=y.coff, dst_cpitch=y.cpitch, Hi=x.H, Wi=x.W, Ci=x.C, Ho=y.H, Wo=y.W, Co=y.C, w_off=w_off, aux0=scale_buf, aux1=comp_buf)
        if dual is not None:
            kw.update(src2=dual.buf, res_coff=dual.coff, res_cpitch=dual.cpitch)
        return self._op(name, **kw)

    def gru(self, name, values, h_buf, out_buf, hd):
        return self._op(name, type=nat.OP_GRU, src=values.buf, src_coff=values.coff, src_cpitch=values.cpitch, src2=h_buf, dst=out_buf, Hi=values.H,
                        Wi=values.W, Ci=3 * hd, Ho=values.H, Wo=values.W, Co=hd)

    def planar_in(self, name, src_buf, C, y, pixel_major=False, bcast=False):
        return self._op(name, type=nat.OP_PLANAR_IN, flags=(1 if pixel_major else 0) | (2 if bcast else 0), src=src_buf, dst=y.buf, dst_coff=y.coff,
                        dst_cpitch=y.cpitch, Hi=y.H, Wi=y.W, Ci=C, Ho=y.H, Wo=y.W, Co=y.span)

    def cmn_decoder_in(self, name, g16, readout_buf, CV, hidden_buf, HD, dc, dcr_buf):
        """dc[:, 0:g16.span | +CV | +HD] = [g16 (one frame, every object) | readout | hidden], dcr = relu(dc): one launch (csrc/colormnet_net.hip)"""
        assert g16.span % 8 == 0 and CV % 8 == 0 and HD % 8 == 0 and dc.span >= g16.span + CV + HD and g16.H == dc.H and g16.W == dc.W, name
        return self._op(name, type=nat.OP_CMN_DECODER_IN, src=g16.buf, src_coff=g16.coff, src_cpitch=g16.cpitch, src2=readout_buf, aux0=hidden_buf, aux1=dcr_buf,
                        dst=dc.buf, dst_coff=dc.coff, dst_cpitch=dc.cpitch, Hi=dc.H, Wi=dc.W, Ci=g16.span, Ho=dc.H, Wo=dc.W, Co=g16.span + CV + HD, kh=CV, kw=HD)

    def planar_out(self, name, x, coff, C, dst_buf, act=0):
        return self._op(name, type=nat.OP_PLANAR_OUT, src=x.buf, src_coff=x.coff + coff, src_cpitch=x.cpitch, dst=dst_buf, Hi=x.H, Wi=x.W, Ci=C,
                        Ho=x.H, Wo=x.W, Co=C, kh=act)

    def finish(self):
        ops = np.array(self.ops, dtype=nat.OP_DTYPE)
        bufs = np.array(self.bufs, dtype=nat.BUF_DTYPE)
        return ops, bufs
