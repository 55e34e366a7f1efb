"""ColorMNet — CPU restatement of the two memory kernels of the exemplar path.  ORACLE / TEST INFRASTRUCTURE ONLY.

  memory read     /root/reference/vsdeoldify/colormnet/model/memory_util.py:7-80 (get_similarity, do_softmax(top_k), readout),
                  called per frame from inference/memory_manager.py:58-150 (match_memory; top_k = 30, colormnet_render.py:122)
  local attention /root/reference/vsdeoldify/colormnet/model/attention.py:783-860 (LocalGatedPropagation.forward: the short-term
                  memory, model/network.py:37-45: d_qk = d_att = 64, one head, window 15 x 15, dilation 1, use_linear=False).
                  Lines 827-835 are the local correlation the reference takes from the CUDA-only `spatial_correlation_sampler`
                  wheel, with the pad_and_unfold + multiply-sum torch fallback restated here.

Pinned by tests/golden/colormnet_*.npz, produced by EXECUTING those reference functions (tools/gen_golden_colormnet.py).
"""
import math

import torch
import torch.nn.functional as F


def get_similarity(mk, ms, qk, qe):
    """memory_util.py:7-39.  mk [B,CK,N], ms [B,N] or None, qk [B,CK,HW], qe [B,CK,HW] or None -> [B,N,HW]."""
    CK = mk.shape[1]
    ms = ms.unsqueeze(2) if ms is not None else None
    if qe is not None:
        mkt = mk.transpose(1, 2)
        a_sq = mkt.pow(2) @ qe
        two_ab = 2 * (mkt @ (qk * qe))
        b_sq = (qe * qk.pow(2)).sum(1, keepdim=True)
        sim = -a_sq + two_ab - b_sq
    else:
        a_sq = mk.pow(2).sum(1).unsqueeze(2)
        two_ab = 2 * (mk.transpose(1, 2) @ qk)
        sim = -a_sq + two_ab
    return sim * ms / math.sqrt(CK) if ms is not None else sim / math.sqrt(CK)


def do_softmax(similarity, top_k=None):
    """memory_util.py:41-65: top-k softmax (exp WITHOUT max subtraction in the top-k branch, as the reference writes it)."""
    if top_k is not None:
        values, indices = torch.topk(similarity, k=top_k, dim=1)
        x_exp = values.exp()
        x_exp = x_exp / torch.sum(x_exp, dim=1, keepdim=True)
        return torch.zeros_like(similarity).scatter_(1, indices, x_exp)
    maxes = torch.max(similarity, dim=1, keepdim=True)[0]
    x_exp = torch.exp(similarity - maxes)
    return x_exp / torch.sum(x_exp, dim=1, keepdim=True)


def readout(affinity, mv):
    """memory_util.py:73-80 / memory_manager._readout: mv [B,CV,N] @ affinity [B,N,HW] -> [B,CV,HW]."""
    return torch.bmm(mv, affinity)


def memory_read(mk, ms, qk, qe, mv, top_k=30):
    return readout(do_softmax(get_similarity(mk, ms, qk, qe), top_k), mv)


def pad_and_unfold(x, max_dis=7, dilation=1):
    """attention.py:906-915."""
    p = max_dis * dilation
    ws = 2 * max_dis + 1
    return F.unfold(F.pad(x, (p, p, p, p), mode="constant", value=0), kernel_size=(ws, ws), stride=(1, 1), dilation=dilation)


def local_correlation(q, k, max_dis=7, dilation=1):
    """attention.py:827-835 (the torch fallback of SpatialCorrelationSampler): q, k [n,C,h,w] -> [n, 1, ws*ws, h*w] with
    out[n,0,(dy+R)*ws+(dx+R),y*w+x] = sum_c q[n,c,y,x] k[n,c,y+dy*dil,x+dx*dil] (zero outside the image)."""
    n, c, h, w = q.shape
    ws = 2 * max_dis + 1
    uk = pad_and_unfold(k, max_dis, dilation).view(n, c, ws * ws, h, w)
    return (q.unsqueeze(2) * uk).sum(dim=1).view(n, 1, ws * ws, h * w)


def local_attention(q, k, v, rel_w, rel_b, max_dis=7, dilation=1):
    """LocalGatedPropagation.forward with use_linear=False, one head, up to agg_value (attention.py:783-856):
    relative_emb = conv1x1(q); q /= sqrt(d_att); qk = local_correlation(q, k) + relative_emb; positions outside the image get
    -1e8; softmax over the window; agg[p] = sum_d attn[d, p] v[:, p + d].  Returns (agg [h*w, n, Cv], attn [n, 1, ws*ws, h*w])."""
    n, c, h, w = q.shape
    ws = 2 * max_dis + 1
    rel = F.conv2d(q, rel_w.view(ws * ws, c, 1, 1), rel_b).view(n, 1, ws * ws, h * w)
    qs = q / (c ** 0.5)
    qk = local_correlation(qs, k, max_dis, dilation) + rel
    mask = 1 - pad_and_unfold(torch.ones((1, 1, h, w)), max_dis, dilation).view(1, 1, ws * ws, h * w)
    qk = qk - mask * 1e8
    attn = torch.softmax(qk, dim=2)
    uv = pad_and_unfold(v, max_dis, dilation).view(n, v.shape[1], ws * ws, h * w)
    agg = (uv * attn).sum(dim=2)                                   # [n, Cv, hw]
    return agg.permute(2, 0, 1).contiguous(), attn
