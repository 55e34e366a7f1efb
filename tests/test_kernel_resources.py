"""CPU: register / scratch budget of the built gfx950 kernels, read from the code objects inside vsdeoldify_amd/lib/libhavc_mi355.so (AMDGPU metadata notes:
.vgpr_count, .private_segment_fixed_size, .vgpr_spill_count).  A spill in a hot loop or a kernel that no longer fits two waves per SIMD is a silent
performance regression that only a GPU run would show -- and GPU time is the scarce resource of this project; hipcc cross-compiles without a GPU, so the
budget can be checked wherever the library is built.  Rules:
  * the dominant kernels of the headline config -- the 256 x (256 + 16) tail conv tiles conv_pipe_kernel<2, 4, 8, 1, 0, EF> -- use no scratch and at most 256
    VGPRs (512 per SIMD lane / two waves per SIMD: 8 waves per CU on one 160 KB tile);
  * no kernel uses scratch except the ones listed below with their measured sizes (a new entry is a decision, not an accident)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vsdeoldify_amd", "lib", "libhavc_mi355.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernel-name pattern -> largest accepted private segment (bytes); everything else must be 0
SCRATCH_ALLOWED = [
    (r"^_Z19scratch_warm_kernel", 1024),                                   # touches scratch on purpose (first-use allocation off the hot path)
    (r"^_Z16conv_pipe_kernelILi2ELi4ELi8ELi1ELi30ELin1E", 40),            # the PRECISE 256 x 272 tile: 8 VGPRs spilled in its last stage / epilogue (round 5)
    (r"^_Z16conv_halo_kernel", 16),
    (r"^_Z17dwconv7_ln_kernelILi\d+ELb[01]ELi768E", 80),                  # DDColor depthwise 7 x 7 + LayerNorm, 768-thread variants (measured faster than 512)
    (r"^_Z22self_attention_kernel2ILi96E", 64),
    (r"^_Z17conv_igemm_kernelILi128ELi304E", 640),                        # a tile no plan of the four configs selects
]


def _kernels(tmp):
    so = os.path.join(tmp, "lib.so")
    shutil.copy(LIB, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
            out.setdefault(name.group(1), []).append(dict(vgpr=get("vgpr_count"), scratch=get("private_segment_fixed_size"), spill=get("vgpr_spill_count")))
    return out


def test_hot_kernels_fit_their_register_budget_and_nothing_spills_by_accident(tmp_path):
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("no built library or no LLVM binutils here")
    ks = _kernels(str(tmp_path))
    assert len(ks) > 300, len(ks)                                          # ~480 kernels: the extraction worked
    tail = {n: v for n, v in ks.items() if n.startswith("_Z16conv_pipe_kernelILi2ELi4ELi8ELi1ELi0E")}
    assert any(n.endswith("ELi1EEv8ConvArgs") for n in tail) and any("ELi261E" in n for n in tail), sorted(tail)     # conv + ReLU, conv + ReLU + residual + RGB8
    for n, vs in tail.items():
        for v in vs:
            assert v["scratch"] == 0 and v["spill"] == 0 and v["vgpr"] <= 256, (n, v)
    bad = []
    for n, vs in ks.items():
        for v in vs:
            lim = next((m for pat, m in SCRATCH_ALLOWED if re.search(pat, n)), 0)
            if v["scratch"] > lim:
                bad.append((n, v))
    assert not bad, bad
