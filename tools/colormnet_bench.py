"""Timing of the ColorMNet memory kernels at a realistic size (480 x 854 processing resolution -> 30 x 54 features at 1/16, key dim 64,
value dim 512 x 2 objects, top_k 30; working memory 10 frames + 10 000 long-term elements).   python tools/colormnet_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vsdeoldify_amd import colormnet as K
from vsdeoldify_amd.render import get_context

ctx = get_context(0)
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
h, w, CK, CV, OBJ = 30, 54, 64, 512, 2
HW = h * w
for N in (HW, 5 * HW, 10 * HW + 10000):
    mk = (torch.randn(1, CK, N, generator=g) * 0.5).to(dev)
    ms = (torch.rand(1, 1, N, generator=g) * 2 + 1).to(dev)
    qk = (torch.randn(1, CK, HW, generator=g) * 0.5).to(dev)
    qe = torch.rand(1, CK, HW, generator=g).to(dev)
    mv = torch.randn(1, OBJ * CV, N, generator=g).to(dev)
    for _ in range(3):
        K.match_memory_readout(mk, ms, qk, qe, mv, 30)
    t0 = time.perf_counter()
    for _ in range(10):
        out = K.match_memory_readout(mk, ms, qk, qe, mv, 30)
    dt = (time.perf_counter() - t0) / 10
    fl = 2.0 * N * HW * CK * 3
    print(f"match_memory N={N:6d} HW={HW}: {dt*1e3:7.3f} ms  (similarity {fl/dt/1e12:5.2f} TFLOP/s fp32)", flush=True)
    # what the reference does instead: dense affinity [N, HW] + bmm
    t0 = time.perf_counter()
    for _ in range(3):
        a_sq = (mk.transpose(1, 2).pow(2) @ qe); two_ab = 2 * (mk.transpose(1, 2) @ (qk * qe)); b_sq = (qe * qk.pow(2)).sum(1, keepdim=True)
        sim = (-a_sq + two_ab - b_sq) * ms.transpose(1, 2) / 8.0
        val, idx = torch.topk(sim, 30, dim=1); e = val.exp(); e = e / e.sum(1, keepdim=True)
        aff = torch.zeros_like(sim).scatter_(1, idx, e); ref = mv @ aff
    torch.cuda.synchronize()
    print(f"   torch (rocBLAS + topk + scatter + bmm) same op: {(time.perf_counter()-t0)/3*1e3:7.3f} ms;  max |diff| {float((ref-out).abs().max()):.2e}", flush=True)
q = torch.randn(1, 64, h, w, generator=g).to(dev); k = torch.randn(1, 64, h, w, generator=g).to(dev); v = torch.randn(1, 1024, h, w, generator=g).to(dev)
rw = (torch.randn(225, 64, generator=g) * 0.1).to(dev); rb = (torch.randn(225, generator=g) * 0.1).to(dev)
for _ in range(3):
    K.local_attention(q, k, v, rw, rb)
t0 = time.perf_counter()
for _ in range(10):
    K.local_attention(q, k, v, rw, rb)
print(f"local_attention {h}x{w}, C 64, Cv 1024: {(time.perf_counter()-t0)/10*1e3:7.3f} ms")
