"""Zhang et al. colorizers (eccv16 / siggraph17) as a libhavc_mi355 weight blob + execution plan.

Topology restated from vsdeoldify/colorization/colorizers/eccv16.py:9-98 and siggraph17.py:7-161 (conv -> ReLU ... ->
BatchNorm per stack; dilated 3x3 stacks; ConvTranspose2d(k4, s2, p1); 313-way softmax + 1x1 (eccv16);
`[::2, ::2]` sub-sampling, three skip adds, LeakyReLU(0.2), tanh (siggraph17)).  The reference executes siggraph17's
model9up..model_out tail twice and keeps the second result (siggraph17.py:149-159): emitted once here.
ConvTranspose2d(4, 2, 1) = four 2x2 convolutions (one per output parity) with tap displacement -1 ("dil = -1"),
scattered to (2h+py, 2w+px): out[2h+py] takes kernel rows {1,3} (py=0) or {0,2} (py=1).
"""
import numpy as np

from . import _native as nat
from .plan import PlanBuilder, View, WeightPack, bn_scale_shift, pack_conv, pad_to, to_np

AB_NORM = 110.0                      # BaseColor.ab_norm, colorizers/base_color.py:11
KROWS = {0: (1, 3), 1: (0, 2)}       # ConvTranspose kernel rows feeding output parity 0 / 1 (tap a = 0, 1)


class ZhangGenerator:
    def __init__(self, state_dict, model="eccv16", precision="fast"):
        """precision: "fast" = fp16 activations, fp32 accumulate; "precise" = the reference's fp32 arithmetic (colorization/__init__.py:76-95 runs the nets
        in fp32) on hi / lo fp16 pairs: three-segment convs, fp32 softmax / tanh projection (HAVC_F_PRECISE, csrc/precise2.hip)."""
        assert model in ("eccv16", "siggraph17") and precision in ("fast", "precise")
        self.precise = precision == "precise"
        self.sd, self.model = to_np(state_dict), model
        self.pack, self._pc, self._vec = WeightPack(), {}, {}
        self._frozen = False
        self.plan(64)
        self.blob = self.pack.blob()
        self._frozen = True

    def _cached(self, store, key, fn):
        if key not in store:
            assert not self._frozen, key
            store[key] = fn()
        return store[key]

    def _conv(self, b, key, x, stride=1, pad=1, dil=1, relu=True, bn=None, leaky=None, res=None, relu_post=False, y=None):
        """nn.Conv2d(bias=True) [-> ReLU/LeakyReLU] [-> BatchNorm] (conv -> ReLU -> BN order, eccv16.py:12-16)."""
        sd = self.sd

        def make():
            kw = {}
            if bn:
                kw["scale"], kw["shift"] = bn_scale_shift(sd, bn)
            return pack_conv(self.pack, sd[key + ".weight"].astype(np.float32), x.cmap, x.span, bias=sd[key + ".bias"], precise=self.precise, **kw)
        pc = self._cached(self._pc, key, make)
        k = pc.kh
        Ho = (x.H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = y or b.tensor(Ho, Ho, pc.Cout)
        flags = (nat.F_RELU_PRE if relu else 0) | (nat.F_AFFINE if bn else 0)
        f = (0, 0, 0, 0)
        if leaky is not None:
            flags |= nat.F_RELU_PRE | nat.F_LEAKY
            f = (0, 0, leaky, 0)
        if res is not None:
            flags |= nat.F_RESIDUAL
        if relu_post:
            flags |= nat.F_RELU_POST
        b.conv(key, pc, x, y, stride=stride, pad=pad, dil=dil, flags=flags, res=res, f=f)
        return y

    def _convT(self, b, key, x, relu_pre=False, res=None, relu_post=False):
        """ConvTranspose2d(k=4, s=2, p=1, bias) as 4 parity convs; optional + res (skip) and ReLU after the add."""
        sd = self.sd
        WT = sd[key + ".weight"].astype(np.float32)              # [Cin, Cout, 4, 4]
        Cout = WT.shape[1]
        y = b.tensor(2 * x.H, 2 * x.W, Cout)
        for py in (0, 1):
            for px in (0, 1):
                def make(py=py, px=px):
                    Wsub = WT[:, :, KROWS[py], :][:, :, :, KROWS[px]].transpose(1, 0, 2, 3)    # [Cout, Cin, a, b]
                    return pack_conv(self.pack, np.ascontiguousarray(Wsub), x.cmap, x.span, bias=sd[key + ".bias"], precise=self.precise)
                pc = self._cached(self._pc, f"{key}.p{py}{px}", make)
                flags = (nat.F_RELU_PRE if relu_pre else 0) | (nat.F_RESIDUAL if res is not None else 0) | \
                        (nat.F_RELU_POST if relu_post else 0)
                # hi = ho + py - a  (stride 1, dil -1, pad -py); wi likewise with px
                b.conv(f"{key}.p{py}{px}", pc, x, y, stride=1, pad=-py, pad_w=-px, dil=-1, flags=flags, res=res,
                       out_hw=(x.H, x.W), out_step=2, out_oy=py, out_ox=px)
        return y

    def _stack(self, b, p, x, n, strides=None, pad=1, dil=1, first=0):
        """n x (conv -> ReLU) then BatchNorm at index first + 2n (fused into the last conv's epilogue)."""
        for k in range(n):
            last = k == n - 1
            x = self._conv(b, f"{p}.{first + 2 * k}", x, stride=(strides or [1] * n)[k], pad=pad, dil=dil,
                           bn=f"{p}.{first + 2 * n}" if last else None)
        return x

    def plan(self, S=256):
        assert S % 8 == 0
        sd = self.sd
        b = PlanBuilder(precise=self.precise)
        in_buf = b.buf(S * S * 3, 1)
        out_buf = b.buf(S * S * 2, 4)
        x = b.tensor(S, S, 4 if self.model == "siggraph17" else 1, zero_init=False)
        b.prep_lab_l("prep_lab_l", in_buf, S, x)
        if self.model == "eccv16":
            x = self._stack(b, "model1", x, 2, [1, 2])
            x = self._stack(b, "model2", x, 2, [1, 2])
            x = self._stack(b, "model3", x, 3, [1, 1, 2])
            x = self._stack(b, "model4", x, 3)
            x = self._stack(b, "model5", x, 3, pad=2, dil=2)
            x = self._stack(b, "model6", x, 3, pad=2, dil=2)
            x = self._stack(b, "model7", x, 3)
            x = self._convT(b, "model8.0", x, relu_pre=True)
            x = self._conv(b, "model8.2", x)
            x = self._conv(b, "model8.4", x)
            x = self._conv(b, "model8.6", x, pad=0, relu=False)                    # 256 -> 313 logits
            w_off = self._cached(self._vec, "model_out", lambda: self.pack.add(sd["model_out.weight"].reshape(2, -1).astype(np.float32)))
            q = b.buf(x.H * x.W * 2, 4)
            b.proj2("model_out", x, w_off, -1, 1, 1.0, q)                          # model_out(softmax(conv8_3))
            b.bilinear2("upsample4", q, x.H, x.W, out_buf, S, S, AB_NORM)          # unnormalize_ab(upsample4(.))
        else:
            c1 = self._stack(b, "model1", x, 2)
            c2 = self._stack(b, "model2", self._sub(b, "sub1", c1), 2)
            c3 = self._stack(b, "model3", self._sub(b, "sub2", c2), 3)
            x = self._stack(b, "model4", self._sub(b, "sub3", c3), 3)
            x = self._stack(b, "model5", x, 3, pad=2, dil=2)
            x = self._stack(b, "model6", x, 3, pad=2, dil=2)
            x = self._stack(b, "model7", x, 3)
            s8 = self._conv(b, "model3short8.0", c3, relu=False)
            x = self._convT(b, "model8up.0", x, res=s8, relu_post=True)            # model8[0] ReLU folded in
            x = self._conv(b, "model8.1", x)
            x = self._conv(b, "model8.3", x, bn="model8.5")
            s9 = self._conv(b, "model2short9.0", c2, relu=False)
            x = self._convT(b, "model9up.0", x, res=s9, relu_post=True)
            x = self._conv(b, "model9.1", x, bn="model9.3")
            s10 = self._conv(b, "model1short10.0", c1, relu=False)
            x = self._convT(b, "model10up.0", x, res=s10, relu_post=True)
            x = self._conv(b, "model10.1", x, relu=False, leaky=0.2)
            w_off = self._cached(self._vec, "model_out.w", lambda: self.pack.add(sd["model_out.0.weight"].reshape(2, -1).astype(np.float32)))
            b_off = self._cached(self._vec, "model_out.b", lambda: self.pack.add(sd["model_out.0.bias"].astype(np.float32)))
            b.proj2("model_out", x, w_off, b_off, 2, AB_NORM, out_buf)             # unnormalize_ab(tanh(conv))
        ops, bufs = b.finish()
        return ops, bufs, in_buf, out_buf, b.names

    def _sub(self, b, name, x):
        y = b.tensor((x.H + 1) // 2, (x.W + 1) // 2, x.C)
        b.subsample2(name, x, y)
        return y
