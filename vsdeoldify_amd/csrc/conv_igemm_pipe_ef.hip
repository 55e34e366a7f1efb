// conv_pipe_kernel instantiations with the epilogue flags fixed at COMPILE TIME (template parameter EF of conv_pipe_kernel.inc) for the layer
// kinds the networks spend their time in.  Same main loop, same arithmetic and the same bytes as the run-time-flag kernels of conv_igemm_pipe.hip;
// the epilogue shrinks from ~24 000 instructions with ~390 scalar branches (every layer kind's path, flags tested inside the unrolled fragment
// loops) to 1 000 - 2 500 instructions: 10 - 20 % of a short-K GEMM (ConvNeXt pwconv1 768 -> 3072 + GELU: 0.115 -> 0.101 ms at 16 frames,
// 192 -> 768: 0.307 -> 0.245 ms; profiles/r4_epilogue_ablation.txt).  HAVC_EPI_SPECIAL=0 switches them off (A/B runs).
// Two translation units (parallel build): this file = the 256 x 256 / 256 x 272 / 128 x 128 / 256 x 128 / 128 x 256 tiles; conv_igemm_pipe_ef2.hip
// (this file again with HAVC_EF_PART 1) = the small tiles of the encoders and of ColorMNet's one-frame launches, conv_igemm_pipe_ef3.hip (PART 2) = the
// 96- and 192-row tiles the autotuner picks for ragged pixel counts.
#ifndef HAVC_EF_PART
#define HAVC_EF_PART 0
#endif
#include "conv_common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "conv_pipe_kernel.inc"

namespace {

template <int WM, int WN, int FM, int EXTRA, int EF>
int launch_ef(const ConvArgs& a, hipStream_t s) {
    using G = Geo<WM, WN, FM, EXTRA>;
    if ((a.Kc & 7) || !a.ktab || a.x_bytes == 0 || a.x_bytes >= OOB || a.w_bytes >= OOB || (EXTRA && a.Npad != G::BN + 16)) return (int)hipErrorInvalidValue;
    const int MT = (a.M + G::BM - 1) / G::BM, NT = (a.Npad - 16 * EXTRA + G::BN - 1) / G::BN;
    if (a.splitk > 1) return (int)hipErrorInvalidValue;   // split-K main loops store raw partial sums and never reach an epilogue: the run-time-flag kernel serves them
    constexpr int LDS = G::LDS_BYTES;
    ensure_lds_optin<conv_pipe_kernel<WM, WN, FM, EXTRA, 0, EF>>(LDS);
    ConvArgs ar = a;
    if (!(EF & HAVC_F_PS_BLUR)) conv_raster(ar, MT, NT, G::BN);
    hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, FM, EXTRA, 0, EF>), dim3(MT * NT), dim3(G::NW * 64), LDS, s, ar);
    return (int)hipGetLastError();
}

template <int WM, int WN, int FM, int EXTRA, int EF>
void optin_ef() { ensure_lds_optin<conv_pipe_kernel<WM, WN, FM, EXTRA, 0, EF>>(Geo<WM, WN, FM, EXTRA>::LDS_BYTES); }

constexpr int RELU = HAVC_F_RELU_PRE, AFF = HAVC_F_AFFINE, RES = HAVC_F_RESIDUAL, POST = HAVC_F_RELU_POST, PS = HAVC_F_OUT_PIXSHUF;

// layer kinds (epilogue flag sets) with a kernel of their own, per tile geometry:
//   0              bias only (K / V / Q projections, input projections)              RELU           conv + ReLU (BN folded: ResNet conv1 / conv2, tail res-block conv 1)
//   RELU | AFF     conv -> ReLU -> BatchNorm (the DeOldify decoder / middle convs)   RES            conv + residual
//   AFF | RES      ConvNeXt pwconv2: layer scale, + block input                      RES | POST     ResNet conv3: + identity, ReLU (POST alone: ColorMNet)
//   GELU           ConvNeXt pwconv1                                                  RELU | PS      1x1 conv + ReLU + PixelShuffle (without the fused blur)
// and on the 256 x 256 tile only: RELU | PS | PS_BLUR (shuffle + blur fused), RELU | FUSE_PROJ (DDColor last_shuf + einsum + refine);
// on the 256 x 272 tile: RELU (tail res-block conv 1) and RELU | RES | FUSE_RGB8 (conv 2 + layers.11 + SigmoidRange + u8).
#define HAVC_EF_COMMON(X, WM, WN, FM) \
    X(WM, WN, FM, 0, 0) X(WM, WN, FM, 0, RELU) X(WM, WN, FM, 0, RELU | AFF) X(WM, WN, FM, 0, RES) X(WM, WN, FM, 0, AFF | RES) \
    X(WM, WN, FM, 0, RES | POST) X(WM, WN, FM, 0, POST) X(WM, WN, FM, 0, HAVC_F_GELU) X(WM, WN, FM, 0, RELU | PS)
#if HAVC_EF_PART == 0
#define HAVC_EF_ALL(X) \
    HAVC_EF_COMMON(X, 2, 4, 8) X(2, 4, 8, 0, RELU | PS | HAVC_F_PS_BLUR) X(2, 4, 8, 0, RELU | HAVC_F_FUSE_PROJ) \
    X(2, 4, 8, 1, RELU) X(2, 4, 8, 1, RELU | RES | HAVC_F_FUSE_RGB8) \
    HAVC_EF_COMMON(X, 2, 2, 4) HAVC_EF_COMMON(X, 4, 2, 4) HAVC_EF_COMMON(X, 2, 4, 4)
#elif HAVC_EF_PART == 1
#define HAVC_EF_ALL(X) \
    HAVC_EF_COMMON(X, 1, 2, 4) HAVC_EF_COMMON(X, 1, 4, 4) HAVC_EF_COMMON(X, 2, 2, 6) HAVC_EF_COMMON(X, 1, 4, 8) HAVC_EF_COMMON(X, 4, 1, 4) \
    HAVC_EF_COMMON(X, 2, 1, 4)
#else
#define HAVC_EF_ALL(X) HAVC_EF_COMMON(X, 1, 4, 6) HAVC_EF_COMMON(X, 1, 2, 6) HAVC_EF_COMMON(X, 2, 4, 6)
#endif

constexpr int geo_cfg(int WM, int WN, int FM, int EX) {               // the configuration ids of launch_conv_pipe (conv_igemm_pipe.hip)
    return (WM == 2 && WN == 4 && FM == 8) ? 60 + EX : (WM == 2 && WN == 2 && FM == 4) ? 70 : (WM == 4 && WN == 2 && FM == 4) ? 98 :
           (WM == 2 && WN == 4 && FM == 4) ? 96 : (WM == 1 && WN == 2 && FM == 4) ? 72 : (WM == 1 && WN == 4 && FM == 4) ? 91 :
           (WM == 2 && WN == 2 && FM == 6) ? 93 : (WM == 1 && WN == 4 && FM == 8) ? 71 : (WM == 4 && WN == 1 && FM == 4) ? 99 :
           (WM == 2 && WN == 1 && FM == 4) ? 92 : (WM == 1 && WN == 4 && FM == 6) ? 90 : (WM == 1 && WN == 2 && FM == 6) ? 95 :
           (WM == 2 && WN == 4 && FM == 6) ? 97 : -1;
}

}  // namespace

#if HAVC_EF_PART == 0
int launch_conv_pipe_ef2(const ConvArgs& a, int cfg, int ef, hipStream_t s);
void preload_conv_pipe_ef2();
// -1: no specialised kernel for this (tile configuration, layer kind) -- the caller launches the run-time-flag kernel
int launch_conv_pipe_ef(const ConvArgs& a, int cfg, hipStream_t s) {
    static const bool on = [] { const char* e = getenv("HAVC_EPI_SPECIAL"); return !e || atoi(e) != 0; }();
    if (!on || (a.flags & HAVC_F_PRECISE) || a.splitk > 1 || a.oss != 1) return -1;
    const int ef = a.flags & HAVC_EPI_MASK;
#define X(WM, WN, FM, EX, EF) if (cfg == geo_cfg(WM, WN, FM, EX) && ef == (EF)) return launch_ef<WM, WN, FM, EX, (EF)>(a, s);
    HAVC_EF_ALL(X)
#undef X
    return launch_conv_pipe_ef2(a, cfg, ef, s);
}

void preload_conv_pipe_ef() {
#define X(WM, WN, FM, EX, EF) optin_ef<WM, WN, FM, EX, (EF)>();
    HAVC_EF_ALL(X)
#undef X
    preload_conv_pipe_ef2();
    (void)hipGetLastError();
}
#elif HAVC_EF_PART == 1
int launch_conv_pipe_ef3(const ConvArgs& a, int cfg, int ef, hipStream_t s);
void preload_conv_pipe_ef3();
int launch_conv_pipe_ef2(const ConvArgs& a, int cfg, int ef, hipStream_t s) {
#define X(WM, WN, FM, EX, EF) if (cfg == geo_cfg(WM, WN, FM, EX) && ef == (EF)) return launch_ef<WM, WN, FM, EX, (EF)>(a, s);
    HAVC_EF_ALL(X)
#undef X
    return launch_conv_pipe_ef3(a, cfg, ef, s);
}

void preload_conv_pipe_ef2() {
#define X(WM, WN, FM, EX, EF) optin_ef<WM, WN, FM, EX, (EF)>();
    HAVC_EF_ALL(X)
#undef X
    preload_conv_pipe_ef3();
}
#else
int launch_conv_pipe_ef3(const ConvArgs& a, int cfg, int ef, hipStream_t s) {
#define X(WM, WN, FM, EX, EF) if (cfg == geo_cfg(WM, WN, FM, EX) && ef == (EF)) return launch_ef<WM, WN, FM, EX, (EF)>(a, s);
    HAVC_EF_ALL(X)
#undef X
    return -1;
}

void preload_conv_pipe_ef3() {
#define X(WM, WN, FM, EX, EF) optin_ef<WM, WN, FM, EX, (EF)>();
    HAVC_EF_ALL(X)
#undef X
}
#endif
