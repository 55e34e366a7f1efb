/*
 * havc_mi355.h — C ABI of libhavc_mi355.so: MI355X (gfx950) per-frame colorization inference for
 * the HAVC / vs-deoldify filter API.
 *
 * This is the drop-in boundary of the hot path.  The reference has no
 * FFI today: its boundary is three Python call shapes invoked once per frame from VapourSynth
 * ModifyFrame selectors.  Each entry point below cites the reference interface it replaces
 * (paths relative to the reference tree, vsdeoldify/...).  Plain pointers and sizes only; no
 * torch / numpy / PIL types.  All functions return 0 (HAVC_OK) or a negative HAVC_E_* code;
 * havc_last_error() returns a human-readable message for the calling thread's context.
 *
 * Pointers: every image operand of the frame / filter entry points may be a HOST pointer or a DEVICE pointer of the
 * ctx's GPU (havc_dev_alloc, or any hipMalloc of that device; unified addressing tells them apart per operand).  Host
 * operands are staged through the ctx workspace and the call returns when the result is back in host memory.  Device
 * operands are used in place: nothing is copied and a call whose OUTPUT is a device pointer only enqueues work on the ctx
 * stream (havc_synchronize, havc_dev_download or any later call with a host output orders against it).  A whole HAVC merge
 * graph therefore runs without leaving HBM (vsdeoldify_amd/device.py).  Outputs must not alias inputs unless stated.
 *
 * Threading: a havc_ctx owns one GPU, one HIP stream and one workspace arena.  Calls on the same
 * ctx are serialised by an internal mutex (ctypes releases the GIL, VapourSynth has several worker
 * threads); create one ctx per GPU (and per worker if concurrency on one GPU is wanted).
 * Buffers passed in are never retained after return.
 */
#ifndef HAVC_MI355_H
#define HAVC_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAVC_OK 0
#define HAVC_E_INVALID (-1)   /* bad argument / malformed plan            -> Python raises ValueError          */
#define HAVC_E_OOM (-2)       /* device allocation failed                 -> Python shim returns input + warns  */
                              /*   (deoldify/filters.py:55-63 semantics)                                       */
#define HAVC_E_HIP (-3)       /* any other HIP runtime failure            -> Python raises RuntimeError        */
#define HAVC_E_NODEVICE (-4)  /* no gfx950 device visible                 -> Python raises RuntimeError        */
#define HAVC_E_RANGE (-5)     /* range check on: an activation left the fp16 range (inf / NaN in a buffer)       */
                              /*                                          -> Python raises HavcRangeError      */

typedef struct havc_ctx havc_ctx;
typedef struct havc_weights havc_weights;
typedef struct havc_net havc_net;

/* ------------------------------------------------------------------------------------------------
 * Execution plan.  A network is a flat list of ops over numbered activation buffers (NHWC fp16,
 * channel count padded to a multiple of 8, pad channels always zero).  The plan is emitted by the
 * host-side model builder (vsdeoldify_amd/deoldify_net.py) from the reference topology
 * (deoldify/unet.py:94-285, Appendix B of SURVEY.md) for one input size S; weights are packed once
 * per model (vsdeoldify_amd/plan.py) with spectral/weight norm and conv->BN folded.
 * ---------------------------------------------------------------------------------------------- */
enum havc_op_type {
    HAVC_OP_CONV = 1,        /* implicit-GEMM convolution on MFMA with fused epilogue                     */
    HAVC_OP_MAXPOOL = 2,     /* 3x3 stride 2 pad 1 max pool (resnet stem)                                 */
    HAVC_OP_BLUR_RESIZE = 3, /* replicate-pad(1,0,1,0) + avgpool2x2 s1 (+ nearest resize) into dst@coff  */
    HAVC_OP_AFFINE = 4,      /* y = act(x*scale+shift) per channel, into dst@coff (skip BN+ReLU, layers.1) */
    HAVC_OP_ATTENTION = 5,   /* fastai SelfAttention, flash form: out = gamma * softmax_i(f_i.g_j) h + x   */
    HAVC_OP_PREP_RGB8 = 6,   /* u8 RGB -> PIL 'L' gray x3 -> /255 -> imagenet normalise -> fp16 C8 (dst) and, when src2 >= 0, the same 8
                                channels into src2 @ res_coff (the dense-merge slot of the tail tensor).  aux0 = number of pad channels BEHIND
                                that slot which no op reads and the kernel may zero as well (>= 24 and a 64-byte aligned slot: the second
                                destination is then written as whole 64-byte segments instead of 16-byte pieces: no read-modify-write) */
    HAVC_OP_COPY_CH = 7,     /* copy channel slice src@coff -> dst@coff (dense MergeLayer, layers.9)       */
    HAVC_OP_SUBSAMPLE2 = 8,  /* y[h][w] = x[2h][2w]  (siggraph17 `conv[:, :, ::2, ::2]`)                    */
    HAVC_OP_PROJ2 = 9,       /* per pixel C -> 2 projection in fp32: flags&1: softmax over C first (eccv16
                                model_out), flags&2: + bias then tanh (siggraph17 model_out); x f0; fp32 out */
    HAVC_OP_BILINEAR2 = 10,  /* 2-channel fp32 map, bilinear align_corners=False (nn.Upsample x4), x f0      */
    HAVC_OP_PREP_LAB_L = 11, /* u8 RGB -> CIELAB L (skimage rgb2lab, fp64) -> (L-50)/100 -> fp16 C8 ch 0      */
    /* ---- DDColor (ConvNeXt encoder, colour-query transformer) ---- */
    HAVC_OP_DWCONV7 = 12,    /* depthwise 7x7 pad 3 + bias; w_off: fp16 [49][Kc] (Kc = channel pitch), bias_off   */
    HAVC_OP_LAYERNORM = 13,  /* LayerNorm over the Ci channels of every pixel / token: scale_off gamma, shift_off beta, f0 eps */
    HAVC_OP_MHA = 14,        /* multi-head attention, head dim 32: Q = src view (Hi = queries, Wi = tokens per frame), K / V =
                                buffer src2 (pitch res_cpitch, K at res_coff, V at aux0; Ho = keys, Wo = tokens per frame),
                                kh = heads, f0 = softmax scale; dst view has the Q token stride; aux1 = fp32 buffer of
                                heads * ceil(keys / 256) * queries * 34 floats per frame for the key-split form (-1: one wave
                                per query)                                                                          */
    HAVC_OP_PIXSHUF4_BLUR = 15, /* PixelShuffle(4) of a [Hi][Wi][16 Co] tensor whose channels are ordered (dy*4+dx)*Co + c,
                                then the ICNR blur -> dst [4Hi][4Wi][Co]                                              */
    HAVC_OP_PREP_DDCOLOR = 16,  /* u8 RGB -> Lab L -> RGB of Lab(L,0,0) -> imagenet normalise -> fp16 C8 (dst) and a 3-channel
                                slice of src2 @ res_coff (the refine conv's image input)                             */
    HAVC_OP_FOLD_QUERIES = 18,  /* DDColor tail, step 1: M[o][c] = sum_q R[o][q] E[q][c] per frame, fp32.  src = token view of the colour
                                embeddings E (Hi = 1, Wi = tokens per frame, Ci = channels), w_off = fp32 R [2][Kc] (Kc = row pitch,
                                Ho = number of queries), dst = fp32 buffer [2][Ci] per frame.  Folds einsum(bqc,bchw->bqhw) and the
                                1x1 refine conv into one 2 x C matrix (both are linear per pixel)                          */
    HAVC_OP_SHUF4_BLUR_AB = 19, /* DDColor tail, step 3: PixelShuffle(4) + the ICNR blur of the 2-channel map that a FUSE_PROJ conv
                                left in src (fp32 [Hi*Wi][16][2] per frame), + R_img . image + bias: src2 = fp16 image view
                                (3 channels at res_coff, pitch res_cpitch), w_off = fp32 R_img [2][3], bias_off fp32 [2];
                                dst = fp16 view [4Hi][4Wi] channels 0-1                                                   */
    /* ---- ColorMNet network (colormnet/model/{modules,resnet,cbam,group_modules,basic}.py; csrc/colormnet_net.hip).  Ops of a ColorMNet plan
       run in slices with DIFFERENT batch counts (image features: 1 frame; per-object features: one frame per object), so every op below
       takes explicit broadcast flags: a broadcast operand is read from frame 0 whatever the batch index. ---- */
    HAVC_OP_EW = 20,         /* element-wise / resample: kh = mode (0 copy, 1 bilinear align_corners=False with ratios f0 (rows) / f1 (cols) =
                                source / destination as aten computes them, 2 area = mean over kw x kw blocks); flags HAVC_EW_*: SRC_BCAST,
                                RES (+ src2 view, Ho x Wo), RES_BCAST, RELU (on the result), DUAL (also write relu(result) to buffer aux0 at
                                channel offset aux1, pitch Kc).  F.interpolate / upsample_groups / downsample_groups / the `add` distributor /
                                the ReLU GroupResBlock applies to its input (group_modules.py:14-93)                                     */
    HAVC_OP_DWCONV = 21,     /* depthwise kh x kh (3 or 5), zero pad kh / 2, stride 1, bias_off optional; w_off: fp16 [kh*kh][Kc] (Kc = channel
                                pitch): CrossChannelAttention to_*_dw (resnet.py:296-303), DWConv2d (basic.py:75-94)                     */
    HAVC_OP_CHAN_ATTN = 22,  /* CrossChannelAttention core (resnet.py:310-331): q = src view, k = src2 view (res_coff / res_cpitch), kh = heads,
                                Ci = heads * c channels; scale_off = temperature fp32 [heads]; aux0 / aux1 = fp32 scratch buffers (partial
                                Gram [heads][S][c][c], partial norms [S][2][Ci]); dst = fp16 buffer [Ci][Kc * 8]: the block-diagonal softmax
                                matrix, applied to v by a 1 x 1 conv with HAVC_F_W_FROM_BUF                                              */
    HAVC_OP_MHA64 = 23,      /* multi-head self-attention, head dim 64: src = qkv token view (Hi = 1, Wi = tokens per frame, q at src_coff,
                                k at res_coff, v at aux0, all in buffer src with pitch src_cpitch), Ho = live tokens, kh = heads, f0 = scale */
    HAVC_OP_CBAM = 24,       /* CBAM (cbam.py:27-77) fused with the residual of FeatureFusionBlock (modules.py:35-39): dst = x (1 + channel
                                gate x spatial gate) = g + CBAM(g); w_off = fp32 {W1 [C/16][C], b1, W2 [C][C/16], b2, w7 [2][49], b7};
                                aux0 / aux1 = fp32 scratch (gate [3 C] = scale | avg | max, comp [Hi*Wi][2]); flags HAVC_EW_DUAL: relu(dst) -> buffer src2 at
                                res_coff, pitch res_cpitch                                                                               */
    HAVC_OP_GRU = 25,        /* HiddenReinforcer / HiddenUpdater gates (modules.py:66-76): src = values view (3 Co channels), src2 = fp32 planar
                                hidden [Co][Hi*Wi] per frame, dst = fp32 planar new hidden                                               */
    HAVC_OP_PLANAR_IN = 26,  /* fp32 planar [Ci][Hi*Wi] per frame (flags & 1: pixel-major [Hi*Wi][Ci]; flags & 2: broadcast frame 0) in buffer
                                src -> NHWC fp16 dst view (Co = stored channels, zeros above Ci)                                         */
    HAVC_OP_PLANAR_OUT = 27, /* NHWC fp16 src view, Ci channels -> fp32 planar [Ci][Hi*Wi] per frame in buffer dst; kh = activation: 0 none,
                                1 x^2 + 1, 2 sigmoid (KeyProjection, modules.py:226-229), 3 tanh (network.py:141)                        */
    HAVC_OP_CMN_DECODER_IN = 28, /* the input of ColorMNet's Decoder (modules.py:177-205: cat of the 1/16 image features, the memory readout and the hidden
                                state, per object) in ONE launch: src = fp16 view of the image features (Ci channels, ONE frame, broadcast to every
                                object), src2 = fp32 planar readout [kh][Hi*Wi] per object, aux0 = fp32 planar hidden [kw][Hi*Wi] per object ->
                                dst view channels [0, Ci + kh + kw) and the same rectified into buffer aux1 (same offset and pitch)       */
    HAVC_OP_DWCONV7_LN = 17,    /* DWCONV7 followed by LAYERNORM of its result, one kernel (ConvNeXt block head): fields of both
                                ops (w_off / bias_off / Kc; scale_off gamma, shift_off beta, f0 eps); Ci = 64, 192, 384, 768 or 1536.
                                The norm reads the fp32 conv result (the two-op form rounds it to fp16 in between)      */
};

/* conv epilogue flags: v = acc + bias; RELU_PRE; v = v*scale+shift; v += residual; RELU_POST */
#define HAVC_F_RELU_PRE 0x1
#define HAVC_F_AFFINE 0x2
#define HAVC_F_RESIDUAL 0x4
#define HAVC_F_RELU_POST 0x8
#define HAVC_F_OUT_PIXSHUF 0x10   /* weight rows ordered (dy,dx,c): store to (2h+dy, 2w+dx, c)             */
#define HAVC_F_OUT_TRANSPOSED 0x20 /* store as [b][n][pix_pitch] (V^T for attention)                       */
#define HAVC_F_OUT_RGB8 0x40      /* SigmoidRange(f0,f1) -> *std+mean -> clamp01 -> trunc(*255) -> u8 RGB   */
#define HAVC_F_LEAKY 0x80         /* RELU_* use LeakyReLU(f2)                                              */
#define HAVC_F_FUSE_PROJ 0x2000   /* the conv output is NOT stored: every 256-channel column tile j of a pixel (after bias / ReLU, rounded
                                     to fp16) is projected to 2 values with the per-frame fp32 matrix in buffer src2 ([2][256] per
                                     frame, HAVC_OP_FOLD_QUERIES) and stored as fp32 in buffer aux0: [pixel][Npad / 256][2].
                                     DDColor tail, step 2: einsum + refine conv applied BEFORE the shuffle / blur, which commute
                                     with them -- the 4096-channel tensor (2.1 GB per 16 frames) never exists.  Npad % 256 == 0,
                                     Ho * Wo % 16 == 0, no residual                                                         */
#define HAVC_F_PS_BLUR 0x200      /* with OUT_PIXSHUF: the always-on blur of CustomPixelShuffle_ICNR (ReplicationPad(1,0,1,0) +
                                     AvgPool2d(2, 1), deoldify/unet.py:46-52) runs in the conv epilogue; the shuffled tensor is
                                     never stored.  Needs 1x1 stride-1 conv, Co % 64 == 0, weight rows packed as
                                     (c / 64) * 256 + q * 64 + c % 64 (plan.py pack_conv pixshuf="blur"): a 256-column tile then
                                     holds all four sub-pixels of 64 channels, and GEMM rows are 16x16 pixel tiles that
                                     overlap by one row / column (the blur's top-left halo).  Co = channels per sub-pixel in
                                     the packed rows (a multiple of 64: zero rows pad other counts), aux0 = channels stored.  */
#define HAVC_F_GELU 0x400         /* exact (erf) GELU at the RELU_PRE position (ConvNeXt pwconv1)                            */
#define HAVC_F_W_FROM_BUF 0x800   /* conv weights are ACTIVATIONS: buffer src2 holds, per frame, fp16 [Npad][Kc * 8] rows (the colour
                                     embeddings of DDColor's einsum(bqc,bchw->bqhw)); launched once per frame; no residual  */
#define HAVC_F_NT_STORE 0x1000    /* the LDS-transposed epilogue stores its rows with non-temporal (streaming) stores: for outputs far
                                     larger than the 256 MiB Infinity Cache that are not re-read soon (set by the runtime, HAVC_NT_STORE_MB) */
#define HAVC_F_SPLITK(n) ((n) << 16) /* bits 16-19: split-K count n = 2..15 for convs with few output tiles and a long K (one frame of a small
                                     layer: 4 x 8 tiles on 256 CUs): the K range is cut into n parts (even stage boundaries), one block per
                                     (tile, part) writes fp32 partial sums to a ctx scratch buffer; a second launch (splitk_reduce_kernel) adds them
                                     IN A FIXED ORDER (0 .. n-1) and runs the epilogue.  (Round 5 also built the reduction into the conv kernel -- last block
                                     of a tile, agent-scope release / acquire -- measured it 2x slower and round 6 removed it: profiles/r5_splitk_fused_ab.txt.)
                                     The count is part of the PLAN (chosen by the emitter from the shape), not
                                     of the tile autotuner: every tile configuration produces the same bytes for a given count.  Plain convs
                                     only (no PS_BLUR / FUSE_* / W_FROM_BUF / extra-column tile)                                      */
#define HAVC_F_SPLITK_COUNT(flags) (((flags) >> 16) & 15)
#define HAVC_F_PRECISE 0x4000     /* fp32-class arithmetic on the fp16 MFMA path ("precise" mode; the reference computes in fp32 end to end,
                                     deoldify/filters.py:45-68, fastai/basic_train.py:352-363).  Every activation of a precise plan is a
                                     PAIR of fp16 tensors, hi = fp16(v) and lo = fp16((v - hi) * 2^11), in one buffer whose pixel row is
                                     [hi: P channels | lo: P channels] (cpitch = 2 P; a view's lo plane is cpitch / 2 elements behind its
                                     hi plane).  Conv weights are packed as three K segments [2^11 w_hi | 2^11 w_lo | w_hi] (Kc = 3 x the
                                     plain count) and the K table walks x_hi, x_hi, x_lo: the unchanged MFMA main loop accumulates
                                     2^11 (x_hi w_hi + x_hi w_lo + x_lo w_hi) in fp32 -- only the x_lo w_lo term (2^-22 relative) is
                                     dropped -- and the epilogue multiplies by f3 (2^-11 x the plan's per-conv weight pre-scale), does
                                     bias / ReLU / GELU (libm erff) / affine / residual in fp32 and stores a hi / lo pair.  Valid on CONV (no
                                     PS_BLUR, FUSE_*, OUT_TRANSPOSED, W_FROM_BUF, SPLITK), MAXPOOL, BLUR_RESIZE, AFFINE, PREP_RGB8 and ATTENTION
                                     (fp32 VALU kernels; aux1 = NHWC value buffer, Kc = its pixel pitch) and, since round 5 (the reference runs
                                     EVERY model in fp32: colorization/__init__.py:76-95, vsslib/vsmodels.py:353-363), on the ops of the Zhang
                                     colorizers and DDColor: SUBSAMPLE2, PROJ2, PREP_LAB_L, DWCONV7 (w_off then holds fp32 weights), LAYERNORM,
                                     MHA, PREP_DDCOLOR, FOLD_QUERIES and SHUF4_BLUR_AB -- the latter in a form of its own: src = the last_shuf
                                     conv's pair tensor [Hi][Wi][16 x 256], aux0 = the fp32 [2][256] projection of FOLD_QUERIES; the shuffle,
                                     the blur, the projection, the image term and the bias run in ONE fp32 kernel (csrc/precise2.hip).
                                     3x the MFMA work, 2x the activation bytes.                                                  */
#define HAVC_F_FUSE_RGB8 0x100    /* the conv output is NOT stored: a following 1x1 conv to 3 channels (fp32 weights at
                                     scale_off [3][Npad], bias at shift_off [3]) + OUT_RGB8 maths run in the epilogue and
                                     write u8 RGB to buffer aux0 (layers.10.1 + layers.11 + layers.12 of the generator);
                                     needs Npad == 272 (one 256+16 tile holds every channel of a pixel)              */

/* flags of HAVC_OP_EW / HAVC_OP_CBAM */
#define HAVC_EW_SRC_BCAST 1       /* the source view is read from frame 0 for every batch entry (image features shared by the objects) */
#define HAVC_EW_RES 2             /* + src2 view                                                                                      */
#define HAVC_EW_RES_BCAST 4       /* ... read from frame 0                                                                            */
#define HAVC_EW_RELU 8            /* ReLU on the result                                                                               */
#define HAVC_EW_DUAL 16           /* second, rectified copy of the result                                                             */

typedef struct havc_op {
    int32_t type, flags;
    int32_t src, src2, dst;              /* buffer ids; src2 = residual (conv) / qk buffer (attention)      */
    int32_t src_coff, src_cpitch;        /* channel offset / pitch (elements, multiples of 8)               */
    int32_t dst_coff, dst_cpitch;
    int32_t res_coff, res_cpitch;
    int32_t Hi, Wi, Ci;                  /* input spatial size, input channels used (multiple of 8)         */
    int32_t Ho, Wo, Co;                  /* output spatial size, output channels stored (multiple of 8)     */
    int32_t kh, kw, stride, pad, dil;
    int32_t Kc, Npad;                    /* packed weight matrix: Npad rows x Kc 16-byte chunks             */
    int32_t aux0, aux1;                  /* conv TRANSPOSED: aux0 = pixel pitch.  ATTENTION: x = src, out = dst,
                                            qk = src2 (pitch res_cpitch, f at res_coff, g at res_coff + d),
                                            aux0 = d, aux1 = V^T buffer id, Kc = V^T pixel pitch, Ci = dv,
                                            f0 = gamma.  PREP_RGB8: second destination = src2@res_coff      */
    int64_t w_off, bias_off, scale_off, shift_off; /* byte offsets into the weight blob, -1 = none          */
    float f0, f1, f2, f3;
    int64_t flops;                       /* algorithmic FLOPs per frame of this op (2*MAC), for stats       */
    int32_t tag;                         /* builder's label (index into its name table), for profiling      */
    int32_t reserved;                    /* forced conv tile configuration (0 = heuristic)                  */
    int32_t pad_w_delta;                 /* pad along W = pad + pad_w_delta (ConvTranspose parity sub-convs) */
    int32_t out_step, out_oy, out_ox;    /* out_step 2: output pixel (2*ho + out_oy, 2*wo + out_ox) of a 2Ho x 2Wo
                                            image (ConvTranspose2d(k4,s2,p1) = 4 parity convs with dil = -1) */
} havc_op;

typedef struct havc_buf {
    int64_t elems_per_frame;             /* elements (fp16 unless u8 flag) per frame                        */
    int32_t elem_bytes;                  /* 2 = fp16, 1 = u8                                                */
    int32_t zero_init;                   /* 1: memset once at allocation (pad channels nobody writes)       */
} havc_buf;

typedef struct havc_stats {
    double last_ms;          /* GPU time of the last havc_net_run / havc_*_frames call (HIP events)         */
    double total_ms;         /* accumulated GPU time since create / reset                                   */
    double total_flops;      /* accumulated algorithmic FLOPs                                               */
    int64_t frames;          /* frames processed                                                            */
    int64_t launches;        /* kernel launches issued                                                      */
    int64_t bytes_resident;  /* device bytes held (weights + activations + staging)                         */
} havc_stats;

/* ---- context (replaces: device.set(DeviceId(device_index)), deoldify/_device.py:21-30;
 *      vsdeoldify/__init__.py:2487).  device_id = HIP ordinal as seen by this process. ---- */
int havc_create(havc_ctx** out, int device_id);
/* level < 0: the ctx's two streams are re-created at the LOWEST priority of the device (hipDeviceGetStreamPriorityRange), > 0: at the highest, 0: default.
 * For a context whose work should fill the CUs another context leaves idle without delaying it: ColorMNet's batched look-ahead pass (16 frames of the key
 * encoder) next to the frame-by-frame memory step, whose small dependent launches are latency-bound.  Call it before any stream handle of the ctx is
 * handed out (havc_stream); both streams are drained first. */
int havc_ctx_set_stream_priority(havc_ctx* ctx, int level);
/* the ctx's two streams are re-created with a CU mask of n_cus compute units (hipExtStreamCreateWithCUMask; the first n bits = n / 8 CUs of every XCD):
 * the batched look-ahead pass of ColorMNet then cannot occupy the whole chip, and the memory step's small dependent launches (another context, all CUs)
 * always find free CUs.  Same calling rules as havc_ctx_set_stream_priority: both calls fail with HAVC_E_INVALID once havc_get_stream has handed a handle of the ctx
 * out (round 6: enforced, a wrapped handle would dangle).  hipExtStreamCreateWithCUMask takes no flags: masked streams are ordinary (blocking) streams with respect to
 * the NULL stream, which this library never uses.  Both are A/B switches of the ColorMNet look-ahead (measured slower, off by default: DESIGN.md section 9). */
int havc_ctx_set_stream_cus(havc_ctx* ctx, int n_cus);
void havc_destroy(havc_ctx* ctx);
const char* havc_last_error(const havc_ctx* ctx);      /* ctx may be NULL: last creation error            */
int havc_device_count(void);
int havc_synchronize(havc_ctx* ctx);
int havc_get_stats(havc_ctx* ctx, havc_stats* out);
int havc_reset_stats(havc_ctx* ctx);
const char* havc_version(void);

/* ---- weights (replaces: Learner.load -> torch.load(models/<name>.pth) + load_state_dict,
 *      fastai/basic_train.py:264-286; the blob is produced by vsdeoldify_amd/plan.py) ---- */
int havc_weights_load(havc_ctx* ctx, const void* blob, size_t nbytes, havc_weights** out);
void havc_weights_free(havc_weights* w);

/* ---- network = weights + plan for one input size (replaces: the nn.Module graph built by
 *      gen_inference_wide/deep, deoldify/generators.py:12-21,85-95).  max_batch frames of
 *      activations are allocated up front from the ctx arena (288 GB HBM: no per-frame
 *      allocation, no empty_cache() churn — deoldify/visualize.py:30-36). ---- */
int havc_net_create(havc_ctx* ctx, havc_weights* w, const havc_op* ops, int n_ops, const havc_buf* bufs,
                    int n_bufs, int in_buf, int out_buf, int S, int max_batch, havc_net** out);
void havc_net_free(havc_net* net);
/* run `batch` frames: d_in  = device u8 RGB interleaved [batch][S][S][3];
 *                     d_out = device u8 RGB interleaved [batch][S][S][3] (raw colour, trunc(x*255),
 *                     i.e. BaseFilter._model_process output, deoldify/filters.py:45-68). */
int havc_net_run_rgb8(havc_net* net, const uint8_t* d_in, uint8_t* d_out, int batch);
/* The fp16 contract's debug switch.  The reference computes in fp32 end to end (deoldify/filters.py:45-68, fastai/basic_train.py:352-363);
 * this library stores every activation in fp16 (fp32 accumulation).  With the range check ON (havc_range_check_enable, or HAVC_RANGE_CHECK=1
 * in the environment when the ctx is created) the destination buffer of EVERY op is scanned after the op: the largest finite magnitude and
 * the number of inf / NaN values are recorded per op, and a run that produced a non-finite value fails with HAVC_E_RANGE naming the first
 * offending op instead of colouring a frame from garbage.  Enable it BEFORE creating the nets (their buffers are then zero-filled so that
 * never-written padding cannot trip the scan).  Costs one small kernel + a synchronisation per op: a debug / validation mode for new
 * checkpoints, not a production setting.  havc_net_range_stats returns the figures of the net's last run (n_ops entries each). */
int havc_range_check_enable(havc_ctx* ctx, int enable);
int havc_net_range_stats(havc_net* net, float* abs_max, int64_t* non_finite, int n_ops);
/* Point activation buffer `buf` of the net at caller-owned device memory (>= batch * elems_per_frame * elem_bytes bytes; NULL restores the
 * net's own allocation).  The ColorMNet step hands its per-frame tensors (keys, values, hidden state, multi-scale features: the reference's
 * torch tensors, colormnet/inference/inference_core.py) to the plan and receives results this way, without a copy. */
int havc_net_bind(havc_net* net, int buf, void* device_ptr);
/* havc_net_run_ops without the blocking timer: the ops are only enqueued on the ctx stream */
int havc_net_enqueue_ops(havc_net* net, int first_op, int n_ops, int batch);
/* the ctx's HIP stream (hipStream_t), so that a caller can order its own device work with the library's: e.g.
 * torch.cuda.ExternalStream(ptr) makes torch enqueue on the SAME stream and no host synchronisation is needed between the two */
void* havc_get_stream(havc_ctx* ctx);
/* debug / unit-test access to activation buffers (host <-> device, blocking) */
int havc_net_upload(havc_net* net, int buf, const void* host, size_t nbytes);
int havc_net_download(havc_net* net, int buf, void* host, size_t nbytes);
int havc_net_run_ops(havc_net* net, int first_op, int n_ops, int batch);  /* run a slice of the plan */
/* per-op GPU time of one run (ms, n_ops entries), measured with HIP events around every op */
int havc_net_profile(havc_net* net, int batch, float* ms_per_op, int n_ops);

/* Measure the conv tile configurations of every conv op of the plan at `batch` frames and keep the fastest per op (stored in the
 * op's `reserved` field; identical shapes are measured once).  All configurations produce the same bytes, so this only changes
 * speed.  Costs about a second per net; *n_changed (may be NULL) = ops whose configuration differs from the heuristic's.
 * havc_net_get_cfg returns the configuration id an op will run with (0 = heuristic). */
int havc_net_autotune(havc_net* net, int batch, int* n_changed);
int havc_device_name(havc_ctx* ctx, char* buf, int nbuf);      /* "<marketing name>/<gcn arch>/<compute units>": keys the tuning cache */
int havc_net_get_cfg(havc_net* net, int op_index);
/* restore a configuration found by an earlier havc_net_autotune (a host-side cache); refuses ids that are not legal for the op */
int havc_net_set_cfg(havc_net* net, int op_index, int cfg);

/* ---- frame-in / frame-out entry points (host buffers, blocking) ---------------------------------
 * havc_deoldify_frames replaces ModelImageRender.get_transformed_image (deoldify/visualize.py:118-137)
 * for frames already at the render size S x S (the HAVC_colorizer flow, vsdeoldify/__init__.py:2502-2506):
 *   video pass ALWAYS; if `second` != NULL also the stable/artistic pass and
 *   Image.blend(second, video, video_weight) (visualize.py:129,135); post_process = ColorizerFilter.
 *   _post_process (deoldify/filters.py:100-110: cv2 YUV, keep Y of the source, UV of the colour).
 * rgb_in/rgb_out: host u8 interleaved RGB, n frames of S*S*3 bytes, tightly packed. */
int havc_deoldify_frames(havc_ctx* ctx, havc_net* video, havc_net* second, float video_weight, int post_process,
                         const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames);

/* ---- frame coalescer for the per-frame call shape --------------------------------------------------------------------------------
 * The reference calls ModelImageRender.get_transformed_image once per frame, from the std.ModifyFrame selectors of several VapourSynth
 * worker threads (vsslib/vsmodels.py:201-230).  havc_batcher_submit is that call for a frame already S x S (host u8 RGB in / out,
 * blocking, thread-safe): concurrent callers are merged into ONE havc_deoldify_frames over up to max_batch frames -- the first caller to
 * arrive leads, waits at most wait_us microseconds (or until `callers` requests are queued; 0 = until the batch is full), runs the
 * batch and hands every caller its frame.  The bytes are those of a call of its own (results do not depend on the batch size).
 * ctypes releases the GIL around the call, so Python threads coalesce too.  havc_batcher_free waits for callers in flight. */
typedef struct havc_batcher havc_batcher;
int havc_batcher_create(havc_ctx* ctx, havc_net* video, havc_net* second, float video_weight, int post_process, int wait_us, int callers,
                        havc_batcher** out);
/* the same for per-frame DDColor (kind 1: havc_ddcolor_frames) and Zhang (kind 2: havc_zhang_frames) calls on width x height frames */
int havc_batcher_create_frames(havc_ctx* ctx, int kind, havc_net* net, int width, int height, int wait_us, int callers, havc_batcher** out);
int havc_batcher_submit(havc_batcher* b, const uint8_t* rgb_in, uint8_t* rgb_out);
int havc_batcher_stats(havc_batcher* b, int64_t* calls, int64_t* batches);
void havc_batcher_free(havc_batcher* b);

/* havc_zhang_frames replaces ModelColorization.colorize_frame (colorization/__init__.py:76-95) with
 * preprocess_img / postprocess_tens (colorization/colorizers/util.py:25-55): PIL BICUBIC squash to 256x256, skimage
 * rgb2lab L of the original and of the squashed frame, eccv16 / siggraph17 forward at 256x256 (the plan of `net`),
 * bilinear ab -> frame size, lab2rgb, uint8(clip(x*255)).  rgb_in/rgb_out: host u8 interleaved RGB, n frames of w*h*3. */
int havc_zhang_frames(havc_ctx* ctx, havc_net* net, const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames, int width,
                      int height);
/* havc_ddcolor_frames stands where vsddcolor.ddcolor(clip, model, input_size, ...) is called (vsslib/vsmodels.py:353-360,
 * model 0 / 1): Lab L of the frame; the frame squashed to input_size x input_size (S of `net`; Pillow BILINEAR, skipped when it
 * already has that size -- the HAVC configurations with equal render factors) -> RGB of Lab(L, 0, 0) -> DDColor (ConvNeXt-L
 * encoder, pixel-shuffle decoder, colour-query transformer, refine conv; the plan of `net`) -> ab, bilinear (align_corners=False)
 * back to the frame size -> Lab(L_frame, ab) -> RGB u8.  PARITY UNPINNED: vsddcolor is an external wheel that is not part of the
 * reference tree (oracle/ddcolor.py).  rgb_in / rgb_out: host u8 interleaved RGB, n frames of width*height*3. */
int havc_ddcolor_frames(havc_ctx* ctx, havc_net* net, const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames, int width, int height);
/* The SHAPE the reference calls vsddcolor.ddcolor with (vsslib/vsmodels.py:353-363): one frame as three planar float32 (RGBS,
 * is_half = 0) or float16 (RGBH, is_half = 1) planes, full range [0, 1], in and out (host or device planes, row stride in bytes).
 * The planes are the RGB24 frame cast by zimg (k / 255): they are brought back to u8 with round(x * 255), run through the same
 * path as havc_ddcolor_frames, and the Lab -> RGB result is written as float / half planes WITHOUT the u8 quantisation (the
 * reference quantises afterwards with resize.Bicubic(format=RGB24), vsmodels.py:363).  PARITY UNPINNED like havc_ddcolor_frames. */
int havc_ddcolor_frame_planar_f(havc_ctx* ctx, havc_net* net, const void* const in_planes[3], int in_stride_bytes,
                                void* const out_planes[3], int out_stride_bytes, int is_half, int width, int height);
/* One VapourSynth RGB24 frame (three u8 planes with a row stride, S x S = the nets' render size) through ModelImageRender: the
 * selector body of vs_sc_deoldify (vsslib/vsmodels.py:214-230 = frame_to_image -> get_transformed_image -> image_to_frame,
 * vsslib/vsutils.py:60-110) with the planar <-> interleaved shuffles on the GPU.  Planes may be host or device memory. */
int havc_deoldify_frame_planar(havc_ctx* ctx, havc_net* video, havc_net* second, float video_weight, int post_process,
                               const uint8_t* const in_planes[3], int in_stride, uint8_t* const out_planes[3], int out_stride);
/* frame_to_image / frame_to_np_array and image_to_frame / np_array_to_frame (vsslib/vsutils.py:60-110): three u8 planes with a
 * row stride <-> interleaved RGB, w x h */
int havc_planar_to_rgb8(havc_ctx* ctx, const uint8_t* const planes[3], int stride, uint8_t* rgb, int width, int height);
int havc_rgb8_to_planar(havc_ctx* ctx, const uint8_t* rgb, uint8_t* const planes[3], int stride, int width, int height);
/* Pillow Image.resize (BILINEAR = 2, BICUBIC = 3), 8 bits per channel, bit-exact (libImaging/Resample.c) */
int havc_pil_resize(havc_ctx* ctx, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int resample);

/* ---- per-pixel filters on host buffers (vsslib/imfilters.py); u8 interleaved RGB, w*h pixels ---- */
/* Image.blend(a, b, w): trunc(a + w*(b-a)) in fp32 — image_weighted_merge, imfilters.py:113-124 */
int havc_blend(havc_ctx* ctx, const uint8_t* a, const uint8_t* b, float w, uint8_t* out, int width, int height);
/* chroma_post_process(img_m, orig): Y of orig + UV of img_m — imfilters.py:312-321 (cv2 BT.601 fixed point) */
int havc_chroma_post_process(havc_ctx* ctx, const uint8_t* color, const uint8_t* orig, uint8_t* out, int width,
                             int height);
/* chroma_stabilizer(img_stable, img_new, alpha, weight) — imfilters.py:160-200 */
int havc_chroma_stabilizer(havc_ctx* ctx, const uint8_t* img_stable, const uint8_t* img_new, double alpha,
                           double weight, uint8_t* out, int width, int height);

/* chroma_stabilizer_adaptive(img_stable, img_new, base_tol, max_extra, weight) — imfilters.py:202-269 (ChromaBoundAdaptiveMerge) */
int havc_chroma_stabilizer_adaptive(havc_ctx* ctx, const uint8_t* img_stable, const uint8_t* img_new, double base_tol,
                                    double max_extra, double weight, uint8_t* out, int width, int height);
/* _chroma_temporal_limiter(cur_img, prv_img, alpha) — imfilters.py:638-666 (vs_chroma_limiter, vsfilters.py:473-487) */
int havc_chroma_temporal_limiter(havc_ctx* ctx, const uint8_t* cur, const uint8_t* prv, double alpha, uint8_t* out, int width,
                                 int height);
/* _color_temporal_stabilizer(img_f, weight_list) — imfilters.py:680-705: n <= 9 frames, weights already / 100 */
int havc_color_temporal_stabilizer(havc_ctx* ctx, const uint8_t* const* frames, const double* weights, int n, uint8_t* out,
                                   int width, int height);
/* image_luma_merge (mode 0, tresh = round(luma*255)) / w_image_luma_merge (mode 1: tresh, grad as computed by
 * w_np_rgb_to_gray; mode 2: weight = luma/255; mode 3: weight = uint8(luma)/255 = image_luma_merge(luma=0)) — imfilters.py:66-100, nputils.py:101-253 (LumaMaskedMerge) */
int havc_image_luma_merge(havc_ctx* ctx, const uint8_t* img_dark, const uint8_t* img_white, int mode, double tresh, double grad,
                          uint8_t* out, int width, int height);
/* get_image_luma — imfilters.py:597-601: mean of the cv2 Y plane, in [0, 255] (caller divides / rounds) (AdaptiveLumaMerge) */
int havc_image_luma(havc_ctx* ctx, const uint8_t* img, int width, int height, double* mean_y);

/* image_tweak (vsslib/imfilters.py:463-504) without its gamma step (which raises in the reference, imfilters.py:517):
 * Pillow HSV hue shift by hue_offset (Pillow hue units, int(hue_deg / 360 * 255)), then ImageEnhance Brightness(brightness),
 * Contrast(contrast), Color(color) -- a factor of exactly 1 skips the step like the reference's `!= 1.0` tests -- then
 * np_adjust_chroma2 (vsslib/restcolor.py:353-376): only pixels whose cv2 hue in the ORIGINAL lies strictly inside one of
 * the n_ranges [lo, hi] degree ranges keep the tweaked value.  n_ranges <= 8. */
int havc_image_tweak(havc_ctx* ctx, const uint8_t* img, uint8_t* out, int width, int height, int hue_offset, float brightness,
                     float contrast, float color, const double* hue_ranges /* lo0, hi0, lo1, hi1, ... */, int n_ranges);
/* image_chroma_tweak (vsslib/imfilters.py:540-548 -> np_image_chroma_tweak, vsslib/restcolor.py:288-350), the body of the
 * HAVC_stabilizer filters vs_chroma_bright_tweak / vs_colormap (vsslib/vsfilters.py:525-590): cv2 HSV, H += hue/2 (wrapped to
 * [0,180]), S *= clamp(sat,0,10), V *= clamp(1+bright,0,10), back to RGB.  With has_adjust != 0 the parsed "hue_adjust" stage
 * follows: pixels whose tweaked hue lies strictly inside one of the n_ranges degree ranges take the colour re-tweaked by
 * (adj_sat, adj_hue), all others the ORIGINAL pixel; adj_weight > 0 merges towards the re-tweaked colour (adj_hue == 0) or the
 * original (adj_hue != 0), < 0 towards the original.
 * has_adjust == 2: the adjust stage ALONE, computed from the image itself = adjust_hue_range / adjust_chroma (vsslib/restcolor.py:221-286),
 * what vs_sc_ddcolor applies to every DDColor frame through vs_sc_adjust_clip_hue (vsslib/vsmodels.py:365-366, vsfilters.py:435-455) with
 * HAVC_colorizer's default ddtweak_p[1] = "300:360|0.8,0.1"; sat / bright / hue are ignored. */
int havc_image_chroma_tweak(havc_ctx* ctx, const uint8_t* img, uint8_t* out, int width, int height, double sat, double bright, int hue,
                            int has_adjust, const double* hue_ranges, int n_ranges, double adj_sat, int adj_hue, double adj_weight);
/* the per-pixel half of luma_adjusted_levels (vsslib/imfilters.py:335-372): cv2 RGB->YUV, Y' = lut[Y], YUV->RGB.  The caller
 * derives the 256-entry table from havc_image_luma exactly like the reference (vsdeoldify_amd/imfilters.py). */
int havc_luma_lut(havc_ctx* ctx, const uint8_t* img, const uint8_t* lut256, uint8_t* out, int width, int height);
/* restore_color_gradient (vsslib/restcolor.py:98-134), the per-frame body of ChromaRetentionMerge (vsslib/mcomb.py:450-516):
 * gray pixels of img_gray (low cv2 HSV saturation, mask algo 0/1/2 of restcolor.py:137-217) take the colours of img_color
 * (saturation scaled by sat); weight > 0 merges towards the colour image, < 0 towards the gray one; return_mask != 0
 * returns the mask replicated to RGB. */
int havc_restore_color_gradient(havc_ctx* ctx, const uint8_t* img_color, const uint8_t* img_gray, uint8_t* out, int width, int height,
                                double sat, int tht, double weight, double alpha, int algo, int return_mask);

/* ---- device-resident clip pipeline (bench + multi-GPU shard path) -----------------
 * One call colours n_frames 1080p-class frames that are ALREADY in HBM:
 *   d_src [n][h][w][3] u8 gray-as-RGB  -> Spline64 squash to S x S (harness stand-in for zimg,
 *   vsdeoldify/__init__.py:2504) -> DeOldify passes (video [+second], blend, YUV merge) -> Spline64 back to
 *   w x h (__init__.py:3547) -> chroma_post_process with the source luma (vsfilters.py:863-899) -> d_dst.
 * Frames are processed in batches of the nets' max_batch. */
int havc_colorize_clip(havc_ctx* ctx, havc_net* video, havc_net* second, float video_weight, const uint8_t* d_src,
                       uint8_t* d_dst, int n_frames, int width, int height);
/* The same flow for HOST frames, pipelined: batch i+1 is uploaded and batch i-1 downloaded on two copy streams while batch i
 * is on the compute stream (double-buffered device staging).  h_src / h_dst should be pinned (havc_host_alloc) for the copies
 * to overlap; pageable memory works but serialises.  Blocks until every result is in h_dst. */
int havc_colorize_clip_host(havc_ctx* ctx, havc_net* video, havc_net* second, float video_weight, const uint8_t* h_src,
                            uint8_t* h_dst, int n_frames, int width, int height);
int havc_host_alloc(havc_ctx* ctx, size_t nbytes, void** out);    /* pinned host memory (hipHostMalloc) */
int havc_host_free(havc_ctx* ctx, void* p);
/* device memory helpers for callers that keep clips resident (bench.py, sharded runner) */
/* The harness stand-in for the zimg `resize.Spline64` calls around the models (`__init__.py:2504`, `_clip_chroma_resize`
 * `__init__.py:3545-3554`): separable 8-tap Spline64 on u8 RGB; with luma_from != NULL (a dw x dh frame) the result keeps only its
 * chroma and takes the luma of that frame (vs_recover_clip_luma = chroma_post_process, vsslib/vsfilters.py:863-899), fused into the
 * vertical pass.  zimg itself is outside the parity contract: production keeps zimg. */
int havc_spline64_resize(havc_ctx* ctx, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, const uint8_t* luma_from);
/* n_frames tightly packed frames per operand in one call (device-resident clips) */
int havc_spline64_resize_n(havc_ctx* ctx, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, const uint8_t* luma_from,
                           int n_frames);
int havc_dev_alloc(havc_ctx* ctx, size_t nbytes, void** out);
int havc_dev_free(havc_ctx* ctx, void* p);
int havc_dev_upload(havc_ctx* ctx, void* d_dst, const void* h_src, size_t nbytes);
int havc_dev_download(havc_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes);
int havc_dev_copy(havc_ctx* ctx, void* d_dst, const void* d_src, size_t nbytes);      /* device -> device, enqueued on the ctx stream */

/* ---- ColorMNet exemplar path: the memory kernels (fp32, the reference's tensor layouts, host or device pointers) ----
 * havc_memory_read_topk replaces get_similarity + do_softmax(top_k) + readout (colormnet/model/memory_util.py:7-80) as
 * MemoryManager.match_memory calls them once per frame (colormnet/inference/memory_manager.py:58-150, top_k = 30):
 *   mk [B][CK][N] memory keys, ms [B][N] shrinkage or NULL, qk [B][CK][HW] query keys, qe [B][CK][HW] selection or NULL,
 *   mv [B][CV][N] memory values -> out [B][CV][HW].  The N x HW similarity lives only in the ctx workspace; the top-k softmax
 *   uses exp(v) / sum exp(v) without max subtraction, as the reference's top-k branch does.  top_k <= 64.
 * havc_memory_similarity returns the dense similarity [B][N][HW] (get_similarity alone; memory consolidation uses it). */
int havc_memory_read_topk(havc_ctx* ctx, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out, int B,
                          int CK, int CV, int N, int HW, int top_k);
/* the same with the row sums of the sparse affinity, usage [B][N] (do_softmax(..., return_usage=True), memory_util.py:63-64: what
 * MemoryManager.match_memory hands to KeyValueMemoryStore.update_usage, memory_manager.py:111-120,131-135).  Accumulated in 64-bit fixed
 * point, so the sums (and the long-term prototypes chosen from them) do not depend on the order of the atomics.  usage may be NULL. */
int havc_memory_read_topk_usage(havc_ctx* ctx, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out,
                                float* usage, int B, int CK, int CV, int N, int HW, int top_k);
/* memory consolidation (memory_manager.py:264-283): similarity of the N candidates against P prototype keys (qk / qe [B][CK][P]),
 * dense softmax over the candidates (no top-k), readout mv @ affinity -> out [B][CV][P].  CV <= 2048. */
int havc_memory_dense_readout(havc_ctx* ctx, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out, int B,
                              int CK, int CV, int N, int P);
int havc_memory_similarity(havc_ctx* ctx, const float* mk, const float* ms, const float* qk, const float* qe, float* sim, int B, int CK, int N, int HW);
/* havc_local_correlation replaces the SpatialCorrelationSampler call of LocalGatedPropagation (colormnet/model/attention.py:827-835;
 * kernel_size 1, patch_size 2 max_dis + 1, dilation_patch = dilation): q, k [n][C][H][W] ->
 * out [n][(2 max_dis + 1)^2][H * W], out[n][(dy+R) ws + (dx+R)][y W + x] = q_scale * sum_c q[n][c][y][x] k[n][c][y + dy dil][x + dx dil], zero outside.
 * havc_local_attention is LocalGatedPropagation.forward with use_linear=False, one head, up to agg_value (attention.py:783-856):
 * relative_emb = conv1x1(q; rel_w [ws*ws][C], rel_b), correlation of q / sqrt(C) with k, -1e8 on window positions outside the image,
 * softmax over the window, agg[p][n][cv] = sum_d attn[n][d][p] v[n][cv][p + d].  agg: [H*W][n][CV]; attn (may be NULL): [n][ws*ws][H*W].
 * max_dis <= 7.  (The depthwise 5x5 conv and the projection that follow are ordinary conv ops of the plan executor.) */
int havc_local_correlation(havc_ctx* ctx, const float* q, const float* k, float* out, int n, int C, int H, int W, int max_dis, int dilation,
                           float q_scale);
int havc_local_attention(havc_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_w, const float* rel_b, float* agg,
                         float* attn, int n, int C, int CV, int H, int W, int max_dis, int dilation);

/* ColorMNetRender's frame transforms (colormnet/colormnet_render.py:285-301,276-279; dataset/range_transform.py:24-47): u8 RGB [h][w][3] ->
 * normalised Lab, three fp32 planes [3][h][w] ((L - 50) / 50, a / 110, b / 110; skimage rgb2lab restated in fp64: PARITY UNPINNED), and
 * back: L plane [h][w] + ab planes [2][h][w] -> lab2rgb -> clip -> trunc(x * 255) u8 RGB.  Host or device pointers. */
int havc_colormnet_rgb_to_lab(havc_ctx* ctx, const uint8_t* rgb, float* lab, int width, int height);
int havc_colormnet_lab_to_rgb(havc_ctx* ctx, const float* l_plane, const float* ab, uint8_t* rgb, int width, int height);

/* ---- ColorMNet fast step (round 4): the frame loop without tensor bookkeeping between the kernels ------------------------------------------
 * The reference's InferenceCore / MemoryManager (colormnet/inference/inference_core.py:119-230, memory_manager.py:58-246, kv_memory_store.py:36-170)
 * glue their kernels with torch.cat / F.pad / repeat / stack / add / topk on every frame.  These entry points let the host-side drop-in
 * (vsdeoldify_amd/colormnet_fast.py) run a steady-state frame as a handful of enqueue-only calls on PRE-SIZED device buffers:
 *   havc_cmn_frame_in     u8 RGB (host or device) -> normalised Lab planes [3][h][w] (get_image, colormnet_render.py:285-301) and, with img != NULL,
 *                         the network input: the L plane three times, zero-padded to [3][padded_h][padded_w] (pad_divide_by 112)
 *   havc_cmn_frame_out    L plane [h][w] + the PADDED ab planes [2][padded_h][padded_w] the decoder wrote -> u8 RGB (host or device), the unpad folded in
 *   havc_memory_read_banked   havc_memory_read_topk_usage for B = 1 on memory BANKS: mk / mv rows have a pitch (elements) >= N, so the working and
 *                         long-term memories live side by side in one pre-sized buffer and are read without a concatenation; the usage update
 *                         (use_count += usage, life_count += 1 for elements [usage_from, N), kv_memory_store.py:93-101) happens in place
 *   havc_cmn_short_term   LocalGatedPropagation on the last memory frame (havc_local_attention) + the plan's `short` slice (depthwise 5x5 + Linear) on the
 *                         ctx's SECOND stream, forked behind the main stream: the memory read runs next to it.  agg [H*W][2 CV], attn [225][H*W],
 *                         short_out [2 CV][H*W]: caller-owned device scratch / output; agg_buf / short_buf: the plan's buffer ids they are bound to
 *   havc_cmn_join_add     main stream waits for that, then readout += short_out (inference_core.py `_read`)
 *   havc_cmn_side_begin / _end / _wait   (round 5) the READ of the next frame under the decoder of this one: between begin and end, havc_cmn_short_term,
 *                         havc_memory_read_banked and havc_cmn_join_add are enqueued on the ctx's second stream (begin orders that stream behind the main
 *                         stream's work so far -- or, after havc_cmn_side_mark, behind its work up to the mark: the host enqueues this frame's decoder between
 *                         mark and begin, so the main stream never idles while the host issues the section); havc_cmn_side_wait makes the main stream wait for the section and, with apply_usage != 0, launches the usage
 *                         update the section's memory read owes (a read that ran ahead leaves the counters alone: with apply_usage = 0 it is dropped
 *                         without a trace when the caller steps another frame than the announced one).  The read of frame t+1 depends on the memory
 *                         banks and on t+1's key, not on frame t's decoder (inference_core.py:119-230: memory and last_ti_key / last_ti_value change on
 *                         memory frames only), so on the frames between two memory frames it runs next to segment(t) (colormnet_fast.py)
 *   havc_cmn_value_in     encode_value's input [2][5][pixels] from the padded image [3][pixels] and the ab planes [2][pixels] (network.py:87-101)
 *   havc_dev_copy_2d      device -> device rows with pitches (appending a frame's key / value columns to the banks), enqueued on the ctx stream
 *   havc_net_bind_many / havc_net_enqueue_slices   several havc_net_bind / havc_net_enqueue_ops in one call */
int havc_cmn_frame_in(havc_ctx* ctx, const uint8_t* rgb, float* lab, float* img, int width, int height, int padded_w, int padded_h, int pad_left,
                      int pad_top);
int havc_cmn_frame_out(havc_ctx* ctx, const float* l_plane, const float* ab_padded, uint8_t* rgb, int width, int height, int padded_w, int padded_h,
                       int pad_left, int pad_top);
int havc_memory_read_banked(havc_ctx* ctx, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out,
                            float* use_count, float* life_count, int usage_from, int CK, int CV, int N, int64_t pitch, int HW, int top_k);
/* sizes the scratch of havc_memory_read_banked for every N <= N_max once (the banks' capacity): a memory that grows frame by frame would otherwise regrow
 * it a few times per clip, each time behind a device synchronisation (round 5: two such stalls of ~24 ms sat inside a 190 ms measurement) */
int havc_memory_read_reserve(havc_ctx* ctx, int N_max, int HW, int top_k);
int havc_cmn_short_term(havc_ctx* ctx, havc_net* net, int first_op, int n_ops, int agg_buf, int short_buf, const float* q, const float* k, const float* v,
                        const float* rel_w, const float* rel_b, float* agg, float* attn, float* short_out, int C, int CV, int H, int W, int max_dis);
int havc_cmn_join_add(havc_ctx* ctx, float* readout, const float* short_out, int64_t n);
int havc_cmn_side_mark(havc_ctx* ctx);
int havc_cmn_side_begin(havc_ctx* ctx);
int havc_cmn_side_end(havc_ctx* ctx);
int havc_cmn_side_wait(havc_ctx* ctx, int apply_usage);
int havc_cmn_value_in(havc_ctx* ctx, const float* image, const float* planes, float* value_in, int64_t pixels);
int havc_dev_copy_2d(havc_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows);
int havc_net_bind_many(havc_net* net, int count, const int32_t* bufs, void* const* device_ptrs);
int havc_net_enqueue_slices(havc_net* net, int count, const int32_t* first_op, const int32_t* n_ops, const int32_t* batch);
/* Race hunting (round 6; no reference counterpart -- the reference steps a frame on ONE stream, colormnet/inference/inference_core.py:119-230): with seed != 0
 * the multi-stream entry points (plan slices, havc_memory_read_banked, havc_cmn_short_term / _join_add / _side_begin / _side_wait, havc_dev_copy_2d) put a
 * do-nothing kernel of pseudo-random length (1 .. max_us microseconds, one call in three) in front of their work on the stream they use, which moves the
 * streams of a context and of the look-ahead context against each other; seed 0 switches it off.  Process-wide.  Results must not depend on it
 * (tools/cmn_race_stress.py, tests/test_gpu_colormnet_stress.py).  HAVC_STREAM_JITTER=<seed> / HAVC_STREAM_JITTER_US set the same state at load time. */
int havc_debug_stream_jitter(int seed, int max_us);
/* SHA-1 (hex) of the sources this library was built from (tools/build_stamp.py: every .hip / .cpp / .h / .inc file of vsdeoldify_amd/csrc, its Makefile, this header), written into csrc/build_stamp.h
 * by the Makefile.  The shared object is git-ignored and travels prebuilt to the GPU box: tests/test_host_logic.py compares this with the tree, so a stale binary fails a
 * CPU test wherever the suite runs (round 6; no reference counterpart). */
const char* havc_build_stamp(void);

/* timing of the dominant kernel for bench.py's roofline object: average duration (ms) and launch count
 * of HAVC_OP_CONV ops with the given tag over the launches since havc_reset_stats (HIP events on the
 * ctx stream are recorded around those launches only while enabled). */
int havc_tag_timing_enable(havc_ctx* ctx, int tag, int enable);
int havc_tag_timing_read(havc_ctx* ctx, double* avg_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* HAVC_MI355_H */
