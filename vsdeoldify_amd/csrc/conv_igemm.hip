// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC fp16 activations, fp16 packed weights, fp32 MFMA
// accumulation (v_mfma_f32_16x16x32_f16), fused epilogue.
//
// Replaces every nn.Conv2d / conv1d on the colorization hot path (reference: cuDNN/MIOpen via torch;
// shapes in SURVEY.md §8a-T1; graph in deoldify/unet.py:94-285, fastai/layers.py:81-220).
//
// GEMM view:  D[n][m] = sum_k W[n][k] * X[m][k]       (weights are the MFMA "A" operand, pixels the "B"
//   operand, so that a lane ends up holding 4 CONSECUTIVE output channels of one pixel -> 8-byte stores)
//   m = (b*Ho + ho)*Wo + wo,  n = output channel,  k = (tap, cin) with cin in 8-channel (16 B) chunks.
// A 16-byte chunk never straddles a tap, so the im2col gather is one predicated 16-B global load per
// chunk (zero for padding / M tail / K tail).  LDS tiles are [row][4 chunks] with the chunk index
// XOR-swizzled by g(row>>2) so that ds_read_b128 fragment reads and ds_write_b128 staging writes are
// bank-conflict free for the b128 lane groups of MI355X_MICROARCH.md §LDS.
#include "conv_common.h"

__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// (__launch_bounds__(256, 2): at most 256 registers per wave -- with the 512 a 4-wave block may have, the allocator parked the accumulators in AGPRs
//  and moved them through v_accvgpr_read / _write around every MFMA: 760 such moves for 16 MFMAs in the 128 x 128 tile's loop)
template <int BM, int BN, int WM, int WN, bool PRECISE = false>
__global__ void __launch_bounds__(256, (BN > 256 ? 1 : 2)) conv_igemm_kernel(const ConvArgs p) {
    constexpr int FM = BM / WM / 16;  // 16-pixel fragments per wave
    constexpr int FN = BN / WN / 16;  // 16-channel fragments per wave
    constexpr int A_IT = BM * 4 / 256;
    constexpr int B_IT = (BN * 4 + 255) / 256;
    constexpr int STAGE = (BM + BN) * 32;  // halfs per pipeline stage
    static_assert(BM % 64 == 0, "BM");
    __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;

    // XCD-aware bijective remap: each XCD (private L2) gets a contiguous run of tiles.
    const int nwg = gridDim.x;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int NT = (p.Npad + BN - 1) / BN;
    const int m0 = (pid / NT) * BM;
    const int n0 = (pid % NT) * BN;

    const int HoWo = p.Ho * p.Wo;
    const int j = tid & 3;
    // ---- per-thread im2col row state -----------------------------------------------------------
    int64_t a_base[A_IT];
    int a_hi0[A_IT], a_wi0[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (tid >> 2) + it * 64;
        const int m = m0 + row;
        const bool okm = m < p.M;
        const int mm = okm ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad_w;
        a_hi0[it] = okm ? hi0 : -(1 << 20);          // rows past M never pass the bounds test
        a_wi0[it] = wi0;
        a_base[it] = ((int64_t)(b * p.Hi * p.Wi) + (int64_t)hi0 * p.Wi + wi0) * p.x_cpitch + p.x_coff;
    }

    uint4 a_reg[A_IT], b_reg[B_IT];
    const int KT = p.Kc >> 2;

    auto load_tiles = [&](int kt) {
        const int2 e = p.ktab[kt * 4 + j];               // this lane's K chunk: (tap displacement, byte offset)
        const int dh = (short)(e.y & 0xffff), dw = e.y >> 16;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int hi = a_hi0[it] + dh;
            const int wi = a_wi0[it] + dw;
            const bool ok = (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(p.x + (a_base[it] + (e.x >> 1)));
            a_reg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int row = (tid >> 2) + it * 64;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BN && n0 + row < p.Npad) {
                const int64_t off = ((int64_t)(n0 + row) * p.Kc + kt * 4 + j) * 8;
                v = *reinterpret_cast<const uint4*>(p.w + off);
            }
            b_reg[it] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        half_t* As = smem + buf * STAGE;
        half_t* Bs = As + BM * 32;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (tid >> 2) + it * 64;
            *reinterpret_cast<uint4*>(As + (row * 4 + (j ^ swz(row))) * 8) = a_reg[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int row = (tid >> 2) + it * 64;
            if (row < BN) *reinterpret_cast<uint4*>(Bs + (row * 4 + (j ^ swz(row))) * 8) = b_reg[it];
        }
    };

    float4v acc[FN][FM];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load_tiles(kt + 1);
        const half_t* As = smem + buf * STAGE;
        const half_t* Bs = As + BM * 32;
        half8 xf[FM];
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int row = wm * (BM / WM) + mi * 16 + lr;
            xf[mi] = *reinterpret_cast<const half8*>(As + (row * 4 + (lg ^ swz(row))) * 8);
        }
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int row = wn * (BN / WN) + ni * 16 + lr;
            const half8 wf = *reinterpret_cast<const half8*>(Bs + (row * 4 + (lg ^ swz(row))) * 8);
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < KT) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- fused epilogue (conv_common.h) ----
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + lr;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) epilogue_frag<PRECISE>(p, acc[ni][mi], m, n0 + wn * (BN / WN) + ni * 16 + lg * 4, HoWo);
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
    const int MT = (a.M + BM - 1) / BM, NT = (a.Npad + BN - 1) / BN;
    if (a.flags & HAVC_F_PRECISE) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true>), dim3(MT * NT), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(MT * NT), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

enum { CFG_128x128, CFG_128x64, CFG_64x64, CFG_128x272, CFG_128x304, CFG_128x16, CFG_64x128 };

static int pick_config(const ConvArgs& a) {
    if (a.Npad <= 16) return CFG_128x16;
    if (a.Npad > 256 && a.Npad <= 272) return CFG_128x272;
    if (a.Npad > 272 && a.Npad <= 304) return CFG_128x304;
    const int64_t blocks128 = (int64_t)((a.M + 127) / 128) * ((a.Npad + 127) / 128);
    if (a.Npad <= 64) return a.M >= 128 * 256 ? CFG_128x64 : CFG_64x64;
    if (blocks128 >= 512) return CFG_128x128;     // >= 2 blocks per CU: big tile
    const int64_t blocks64x128 = (int64_t)((a.M + 63) / 64) * ((a.Npad + 127) / 128);
    if (blocks64x128 >= 512 || a.Npad % 128 == 0) return CFG_64x128;
    return CFG_64x64;
}

bool conv_pipe_supported(const ConvArgs& a, int extra);

// 0 = register-staged kernels below; 60 / 61 / 70 / 72 = software-pipelined LDS-DMA kernel (conv_igemm_pipe.hip) with
// 256x256(+16) / 128x128 / 64x128 tiles.  Thresholds from tools/conv_bench.py sweeps (profiles/r1_convbench_tiles.txt):
// the big tile needs >= 160 blocks to fill 256 CUs; the 18x18..70x70 encoder / bottleneck layers want many small tiles.
static int pick_pipe(const ConvArgs& a) {
    if (a.flags & HAVC_F_OUT_RGB8) return 0;
    const int extra = (a.Npad % 256 == 16) ? 1 : 0;
    if (!conv_pipe_supported(a, extra)) return 0;
    if (a.flags & HAVC_F_PS_BLUR) return extra ? 0 : 60;   // the epilogue needs the 16x16-pixel x (4 x 64)-column tile
    if (a.flags & HAVC_F_FUSE_RGB8) return 61;           // only the 256 x (256+16) tile sees every channel of a pixel
    if (a.flags & HAVC_F_FUSE_PROJ) return (extra || a.Npad % 256) ? 0 : 60;   // the projection epilogue lives in the 256 x 256 tile
    if (a.Npad >= 256 && (a.Npad % 256 == 0 || extra)) {
        const int64_t blocks = (int64_t)((a.M + 255) / 256) * ((a.Npad - 16 * extra) / 256);
        if (blocks >= 160) return 60 + extra;
    }
    if (a.Npad % 128 == 0) {
        const int64_t blocks128 = (int64_t)((a.M + 127) / 128) * (a.Npad / 128);
        return blocks128 >= 200 ? 70 : 72;
    }
    return 0;
}

const char* conv_config_name(const ConvArgs& a) {
    switch (pick_pipe(a)) {
        case 60: return "pipe256x256";
        case 61: return "pipe256x272";
        case 70: return "pipe128x128";
        case 72: return "pipe64x128";
    }
    static const char* names[] = {"128x128", "128x64", "64x64", "128x272", "128x304", "128x16", "64x128"};
    return names[pick_config(a)];
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.splitk > 1) {                                    // split-K runs on the plain pipelined tiles only (the count is the plan's, the tile the tuner's)
        int cfg = a.cfg;
        if (!conv_splitk_cfg_ok(cfg)) cfg = a.Npad % 256 == 0 ? 71 : (a.Npad % 128 == 0 ? 72 : (a.Npad <= 64 ? 92 : 72));
        if (!conv_pipe_supported(a, 0)) return (int)hipErrorInvalidValue;
        return launch_conv_pipe(a, cfg, s);
    }
    if ((a.flags & HAVC_F_FUSE_PROJ) && a.cfg != 0 && a.cfg != 60) return (int)hipErrorInvalidValue;
    if (a.cfg >= 60) return launch_conv_pipe(a, a.cfg, s);
    if (a.cfg == 0) {
        const int pc = pick_pipe(a);
        if (pc) return launch_conv_pipe(a, pc, s);
    }
    if (a.flags & (HAVC_F_FUSE_RGB8 | HAVC_F_PS_BLUR | HAVC_F_FUSE_PROJ)) return (int)hipErrorInvalidValue;
    switch (a.cfg > 0 ? a.cfg - 1 : pick_config(a)) {
        case CFG_128x128: return launch_cfg<128, 128, 2, 2>(a, s);
        case CFG_128x64: return launch_cfg<128, 64, 2, 2>(a, s);
        case CFG_64x64: return launch_cfg<64, 64, 2, 2>(a, s);
        case CFG_64x128: return launch_cfg<64, 128, 2, 2>(a, s);
        case CFG_128x272: return launch_cfg<128, 272, 4, 1>(a, s);
        case CFG_128x304: return launch_cfg<128, 304, 4, 1>(a, s);
        case CFG_128x16: return launch_cfg<128, 16, 4, 1>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

// Eager module load (havc_create, under the library's set-up mutex): the HIP runtime loads a translation unit's code object on the first use
// of one of its kernels; querying one here moves that -- and the big-LDS opt-ins below -- out of the first launch, which may come from
// several host threads at once ("set-up is serialised").
void preload_conv_igemm() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(conv_igemm_kernel<128, 128, 2, 2>)); (void)hipGetLastError(); }
