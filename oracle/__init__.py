"""CPU oracle for the HAVC per-frame colorization hot path.  TEST INFRASTRUCTURE ONLY.

Everything under oracle/ is a CPU restatement of the reference algorithm (dan64/vs-deoldify
5.6.7) used as the *checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.  The product path (vsdeoldify_amd/) never imports this package and fails
loudly when the HIP library is missing.

Pinning status:
  * decoder / attention / shuffle / norm folding / pre+post tensor math / PIL resize, L, blend:
    PINNED by golden vectors produced by executing the reference itself in the build container
    (tools/gen_golden.py -> tests/golden/*.npz).
  * ResNet encoder: torchvision is absent from the container, so the fixtures were generated with oracle/resnet.py standing
    in for it; both that stand-in and oracle/unet.encoder are pinned INDEPENDENTLY against the `transformers` package's ResNet
    (full resnet101 / resnet34 depth, tests/test_oracle_pins.py).
  * cv2.cvtColor RGB<->YUV: PARITY UNPINNED at LSB level (cv2 absent) — restated from OpenCV's
    published 14-bit fixed-point BT.601 formulas in oracle/cvcolor.py.
  * skimage rgb2lab/lab2rgb: PARITY UNPINNED (skimage absent) — restated from the CIE standard.
  * DDColor (external vsddcolor wheel): PARITY UNPINNED (no source under /root/reference).
  * ColorMNet memory kernels (oracle/colormnet.py): PINNED bit-exactly by vectors from the executed reference
    (tools/gen_golden_colormnet.py -> tests/golden/colormnet_*.npz).
"""
