"""A tiny deterministic stand-in for the ColorMNet network (encode_key / encode_value / segment / short_term_attn with the reference's
argument lists and tensor shapes), used to drive the per-frame step logic: tools/gen_golden_colormnet_core.py records what the reference's
InferenceCore does with it, tests/test_colormnet_core.py replays vsdeoldify_amd.colormnet_core.InferenceCore on the same net.
Test infrastructure; none of this is the reference's network."""
import torch
import torch.nn.functional as F

CK, CV, HID = 8, 6, 4


class StubNet:
    def __init__(self, seed=5):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        self.wk, self.ws, self.we, self.wv, self.wh = r(CK, 3, 1, 1) * 0.8, r(1, 3, 1, 1), r(CK, 3, 1, 1), r(CV) * 0.7, r(HID) * 0.3
        self.pos = r(1, 3, 32, 32) * 0.3          # position term: no two cells alike (zero padding would otherwise create exact duplicates,
        self.calls = []                           # and which of two IDENTICAL memory elements a top-k keeps is implementation-defined)

    def encode_key(self, image, need_ek=True, need_sk=True):
        self.calls.append(("encode_key", bool(need_ek), bool(need_sk)))
        x = F.avg_pool2d(image, 16)
        x = x + self.pos[:, :, :x.shape[2], :x.shape[3]]
        key = F.conv2d(x, self.wk)
        shrinkage = F.conv2d(x, self.ws) ** 2 + 1 if need_sk else None
        selection = torch.sigmoid(F.conv2d(x, self.we)) if need_ek else None
        return key, shrinkage, selection, x, F.avg_pool2d(image, 8), F.avg_pool2d(image, 4)

    def encode_value(self, image, f16, hidden, masks, is_deep_update=True):
        self.calls.append(("encode_value", bool(is_deep_update)))
        m = F.avg_pool2d(masks, 16)                                            # [1, objects, h, w]
        value = (m.unsqueeze(2) + f16.mean(1, keepdim=True).unsqueeze(1)) * self.wv.view(1, 1, CV, 1, 1)
        new_hidden = hidden * 0.5 + m.unsqueeze(2) * self.wh.view(1, 1, HID, 1, 1) if is_deep_update else hidden
        return value, new_hidden

    def segment(self, feats, memory_readout, hidden, h_out=True, strip_bg=False):
        self.calls.append(("segment", bool(h_out)))
        low = memory_readout.mean(2) + feats[0].mean(1, keepdim=True) * 0.1      # [1, objects, h, w]
        logits = F.interpolate(low, scale_factor=16, mode="bilinear", align_corners=False)
        prob = torch.tanh(logits)
        new_hidden = hidden * 0.9 + memory_readout.mean(2, keepdim=True) * self.wh.view(1, 1, HID, 1, 1) if h_out else None
        return new_hidden, prob, prob

    def short_term_attn(self, q, k, v, u, size_2d):
        self.calls.append(("short_term_attn",))
        gate = torch.sigmoid((q * k).sum(1, keepdim=True))                     # [n, 1, h, w]
        out = (v * gate).flatten(start_dim=2).permute(2, 0, 1)                 # [h*w, n, C]
        return out, gate


def clip(seed=11, frames=9, h=112, w=112):
    """gray frames [3, h, w] in [-1, 1], an exemplar (its L planes and its ab planes).  112 x 112: no zero padding -- padded borders of
    two memory frames are IDENTICAL memory elements, and which of two identical elements a top-k keeps (hence which one collects the
    usage and may become a long-term prototype) is implementation-defined: torch.topk and the HIP kernel legitimately differ there."""
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.tanh(torch.randn(1, h, w, generator=g)).repeat(3, 1, 1) for _ in range(frames)]
    ex_l = torch.tanh(torch.randn(1, h, w, generator=g)).repeat(3, 1, 1)
    ex_ab = torch.tanh(torch.randn(2, h, w, generator=g))
    return imgs, ex_l, ex_ab
