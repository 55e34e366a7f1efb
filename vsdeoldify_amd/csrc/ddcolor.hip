// HBM-bound / small kernels of the DDColor path (architecture: oracle/ddcolor.py).  NHWC fp16 activations,
// fp32 arithmetic.  Token tensors (the 100 colour queries) use the same layout with H = 1, W = tokens.
#include "kernels.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

static inline int grid_for_dd(int64_t work, int per_block = 256) {
    int64_t b = (work + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// ---- depthwise 7x7, pad 3, + bias (ConvNeXt block, convnext.py Block.dwconv).  w: fp16 [49][w_pitch], channel-contiguous ----
__global__ void dwconv7_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w, const float* __restrict__ bias, half_t* __restrict__ y,
                               int B, int H, int W, int C8, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch) {
    const int64_t total = (int64_t)B * H * W * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t pix = i / C8;
        const int wo = (int)(pix % W);
        pix /= W;
        const int ho = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[c8 * 8 + e] : 0.f;
        const half_t* xb = x + (int64_t)b * H * W * x_cpitch + x_coff + c8 * 8;
        for (int dy = 0; dy < 7; ++dy) {
            const int hi = ho + dy - 3;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const int wi = wo + dx - 3;
                if ((unsigned)wi >= (unsigned)W) continue;
                const half8 xv = *reinterpret_cast<const half8*>(xb + ((int64_t)hi * W + wi) * x_cpitch);
                const half8 wv = *reinterpret_cast<const half8*>(w + (dy * 7 + dx) * w_pitch + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)xv[e] * (float)wv[e];
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        *reinterpret_cast<half8*>(y + ((int64_t)(b * H + ho) * W + wo) * y_cpitch + y_coff + c8 * 8) = o;
    }
}
int launch_dwconv7(const half_t* x, const half_t* w, const float* bias, half_t* y, int B, int H, int W, int C, int x_cpitch, int x_coff,
                   int y_cpitch, int y_coff, int w_pitch, hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(dwconv7_kernel, dim3(grid_for_dd((int64_t)B * H * W * C8)), dim3(256), 0, s, x, w, bias, y, B, H, W, C8, x_cpitch,
                       x_coff, y_cpitch, y_coff, w_pitch);
    return (int)hipGetLastError();
}

// ---- LayerNorm over the C channels of every pixel / token (biased variance, two passes in registers) ----
// LP lanes share a pixel (up to 4 16-byte chunks per lane), 64 / LP pixels per wave: C = 192 -> 8 lanes x 3 chunks, 8 pixels per wave
// (one wave per pixel left 40 of 64 lanes idle there).  C <= 2048.  Channels C .. C8*8-1 (padding) are written as 0.
// Round 4: a lane's channels are the same for every pixel it visits, so its gamma / beta sit in registers (they were 64 scalar loads per lane per
// pixel), the grid is at most 1 024 blocks (four waves per SIMD, all resident) walking the pixels, and the LP-lane sums use DPP adds inside 16-lane rows instead of LDS-path shuffles.
__device__ __forceinline__ float ln_dpp_add(float v, int ctrl) {
    switch (ctrl) {            // the DPP control must be an immediate
        case 0xB1: return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
        case 0x4E: return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
        case 0x141: return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
        default: return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));     // row_mirror
    }
}
template <int LP>
__device__ __forceinline__ float ln_group_sum(float v) {       // sum over the LP-lane group of the lane, result in every lane of the group
    v = ln_dpp_add(v, 0xB1);
    v = ln_dpp_add(v, 0x4E);
    v = ln_dpp_add(v, 0x141);                                  // 8 lanes
    if (LP >= 16) v = ln_dpp_add(v, 0x140);                    // 16 lanes (one DPP row)
    if (LP >= 32) v += __shfl_xor(v, 16);
    if (LP >= 64) v += __shfl_xor(v, 32);
    return v;
}
template <int LP>
__global__ void __launch_bounds__(256) layernorm_c_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, int64_t npix, int C, int x_cpitch, int x_coff, int y_cpitch,
                                   int y_coff, int relu) {
    constexpr int PPW = 64 / LP;                               // pixels per wave
    const int lane = threadIdx.x & 63, l = lane % LP, sub = lane / LP;
    const int C8 = (C + 7) / 8;
    const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float g[4][8], bt[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c0 = (l + k * LP) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[k][e] = 0.f; bt[k][e] = 0.f; }
        if (c0 + 8 <= C) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
            g[k][0] = g0.x; g[k][1] = g0.y; g[k][2] = g0.z; g[k][3] = g0.w; g[k][4] = g1.x; g[k][5] = g1.y; g[k][6] = g1.z; g[k][7] = g1.w;
            bt[k][0] = b0.x; bt[k][1] = b0.y; bt[k][2] = b0.z; bt[k][3] = b0.w; bt[k][4] = b1.x; bt[k][5] = b1.y; bt[k][6] = b1.z; bt[k][7] = b1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < C) { g[k][e] = gamma[c0 + e]; bt[k][e] = beta[c0 + e]; }
        }
    }
    for (int64_t p0 = wave0 * PPW; p0 < npix; p0 += nwaves * PPW) {
        const int64_t p = p0 + sub;
        const bool live = p < npix;
        float v[4][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = l + k * LP;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
            if (live && c8 < C8) {
                const half8 h = *reinterpret_cast<const half8*>(x + p * x_cpitch + x_coff + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[k][e] = (c8 * 8 + e < C) ? (float)h[e] : 0.f; sum += v[k][e]; }
            }
        }
        sum = ln_group_sum<LP>(sum);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = l + k * LP;
            if (live && c8 < C8)
#pragma unroll
                for (int e = 0; e < 8; ++e) if (c8 * 8 + e < C) { const float d = v[k][e] - mean; sq += d * d; }
        }
        sq = ln_group_sum<LP>(sq);
        const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = l + k * LP;
            if (live && c8 < C8) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c8 * 8 + e;
                    float t = c < C ? (v[k][e] - mean) * rstd * g[k][e] + bt[k][e] : 0.f;
                    if (relu) t = fmaxf(t, 0.f);                       // ColorMNet Fuse: relu(norm3(x)) (colormnet/model/resnet.py:395-396)
                    o[e] = (half_t)t;
                }
                *reinterpret_cast<half8*>(y + p * y_cpitch + y_coff + c8 * 8) = o;
            }
        }
    }
}
int launch_layernorm_c(const half_t* x, half_t* y, const float* gamma, const float* beta, float eps, int64_t npix, int C, int x_cpitch,
                       int x_coff, int y_cpitch, int y_coff, hipStream_t s, int relu) {
    if (C > 2048) return (int)hipErrorInvalidValue;
    const int need = ((C + 7) / 8 + 3) / 4;                    // lanes per pixel at 4 chunks per lane
#define LN_LAUNCH(LP)                                                                                                                          \
    hipLaunchKernelGGL(layernorm_c_kernel<LP>, dim3(std::min(grid_for_dd((npix + 64 / LP - 1) / (64 / LP), 4), 1024)), dim3(256), 0, s, x, y, gamma, beta, eps, npix, \
                       C, x_cpitch, x_coff, y_cpitch, y_coff, relu)
    if (need <= 8) LN_LAUNCH(8);
    else if (need <= 16) LN_LAUNCH(16);
    else if (need <= 32) LN_LAUNCH(32);
    else LN_LAUNCH(64);
#undef LN_LAUNCH
    return (int)hipGetLastError();
}

// ---- ConvNeXt block head in ONE kernel: depthwise 7x7 (+ bias) followed by the channel LayerNorm (convnext.py Block: dwconv, norm) ----
// The two-kernel form above is load bound (98 16-byte loads per 392 MACs, every tap fetched again for every output) and writes /
// re-reads the conv result.  Here a thread owns 4 channels of a 4 x 4 pixel patch: an input row segment (10 pixels) is loaded ONCE,
// converted to fp32 once and feeds up to 4 x 7 x 4 MACs per element; the 49 x C weights sit in LDS (fp32 up to 768 channels, fp16
// above), the MACs are v_pk_fma_f32 on channel pairs, and the next row segment is in flight while the current one is multiplied.
// A block = all C/4 channel groups of S = 768 / (C/4) patches, so LayerNorm's per-pixel sums stay inside the block: 8-lane DPP
// adds, then C/32 partials per pixel through LDS, summed by every thread in the same order (deterministic).  The norm works on the
// fp32 accumulators (the unfused pair rounds the conv result to fp16 first).
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4e __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_sum8(float v) {          // sum over the 8-lane group of the lane, result in every lane
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false);          // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false);         // row_half_mirror
    return v + __builtin_bit_cast(float, t);
}

template <bool WAVE>
__device__ __forceinline__ float dw_group_sum(float v) {      // 8-lane sums, or (WAVE) the whole wave's sum in every lane
    v = dpp_sum8(v);
    if (WAVE) {
        int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false);     // row_mirror: 16 lanes
        v += __builtin_bit_cast(float, t);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
    }
    return v;
}

__device__ __forceinline__ void split_hl_dd(float v, half_t& hi, half_t& lo) {      // conv_common.h split_hl (this unit does not include it)
    hi = (half_t)v;
    lo = (half_t)((v - (float)hi) * 2048.f);
}

constexpr int DWLN_T = 4, DWLN_R = 4;
constexpr unsigned DWLN_OOB = 0xF0000000u;        // voffset beyond every descriptor range: the buffer load returns zeros
static inline size_t dwln_lds_bytes(int C, bool wf32, int threads) {
    const int C4 = C / 4, S = threads / C4;
    return (size_t)49 * C * (wf32 ? 4 : 2) + (size_t)S * (C4 / 8) * (DWLN_T * DWLN_R) * 4;
}
typedef unsigned int uint2e __attribute__((ext_vector_type(2)));

// C4 (channel groups of 4) is a template parameter so that the 49 LDS weight offsets are immediates, not 49 live registers.
// THREADS / PAIR: 768 threads (3 waves per SIMD, <= 168 VGPRs) with one output row per step, or 512 threads (2 per SIMD, 256 VGPRs)
// with two output rows per step (a step then covers the LDS latency by itself) -- picked per channel count by measurement.
// PREC (round 6): the precise form (HAVC_F_PRECISE; the wheel's torch modules run in fp32, vsslib/vsmodels.py:353-363): x and y are hi / lo pair tensors (pixel row =
// [hi: P | lo: P], cpitch = 2 P), the weights are fp32 [49][w_pitch] in the blob and in LDS; the arithmetic between load and store is the fp32 of the fast form.
template <int C4, bool WF32, int THREADS, bool PAIR, bool PREC = false>
__global__ void __launch_bounds__(THREADS) dwconv7_ln_kernel(const half_t* __restrict__ x, const void* __restrict__ w_,
                                                                   const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, half_t* __restrict__ y, int B,
                                                                   int H, int W, unsigned x_bytes, int x_cpitch, int x_coff, int y_cpitch,
                                                                   int y_coff, int w_pitch) {
    constexpr int T = DWLN_T, R = DWLN_R, NP = T * R, XR = R + 6;
    constexpr bool WAVE_SUM = (C4 % 64 == 0);                  // a patch's threads are whole waves: LayerNorm partials per wave (DPP + 2 shuffles), not per 8 lanes
    constexpr int C = C4 * 4, S = THREADS / C4, G = WAVE_SUM ? C4 / 64 : C4 / 8;
    constexpr int WB = WF32 ? 16 : 8;                         // bytes of one thread's weights per tap
    static_assert(!PREC || WF32, "precise: fp32 weights in LDS");
    const half_t* w = reinterpret_cast<const half_t*>(w_);
    extern __shared__ __attribute__((aligned(16))) char dw_smem[];
    const int tid = threadIdx.x;
    const int c4 = tid % C4, strip = tid / C4;
    const bool active = strip < S;
    char* wbase = dw_smem + c4 * WB;                          // tap t at wbase + t * C4 * WB
    float* red = reinterpret_cast<float*>(dw_smem + (size_t)49 * C4 * WB);
    {   // all of a thread's weight loads in flight at once (the block has a single pass over them: a serial loop costs ~1 us per trip)
        constexpr int NW_IT = (49 * C4 + THREADS - 1) / THREADS;
        if constexpr (PREC) {
            const float* wf = reinterpret_cast<const float*>(w_);
            float4e fv[NW_IT];
#pragma unroll
            for (int k = 0; k < NW_IT; ++k) {
                const int i = tid + k * THREADS, tap = i / C4, cc = i - tap * C4;
                fv[k] = float4e{0.f, 0.f, 0.f, 0.f};
                if (i < 49 * C4) fv[k] = *reinterpret_cast<const float4e*>(wf + (size_t)tap * w_pitch + cc * 4);
            }
#pragma unroll
            for (int k = 0; k < NW_IT; ++k) {
                const int i = tid + k * THREADS;
                if (i < 49 * C4) *reinterpret_cast<float4e*>(dw_smem + (size_t)i * 16) = fv[k];
            }
        } else {
        half4 hv[NW_IT];
#pragma unroll
        for (int k = 0; k < NW_IT; ++k) {
            const int i = tid + k * THREADS, tap = i / C4, cc = i - tap * C4;
            hv[k] = half4{0, 0, 0, 0};
            if (i < 49 * C4) hv[k] = *reinterpret_cast<const half4*>(w + (size_t)tap * w_pitch + cc * 4);
        }
#pragma unroll
        for (int k = 0; k < NW_IT; ++k) {
            const int i = tid + k * THREADS;
            if (i < 49 * C4) {
                if (WF32) *reinterpret_cast<float4e*>(dw_smem + (size_t)i * 16) = __builtin_convertvector(hv[k], float4e);
                else *reinterpret_cast<half4*>(dw_smem + (size_t)i * 8) = hv[k];
            }
        }
        }
    }
    __syncthreads();
    float2v bias2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (bias && active) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4);
        bias2[0][0] = bv.x; bias2[0][1] = bv.y; bias2[1][0] = bv.z; bias2[1][1] = bv.w;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(x), 0, x_bytes, 0x00020000);
    const int n_px = (W + R - 1) / R, n_py = (H + T - 1) / T;
    const int64_t total = (int64_t)B * n_py * n_px;
    const float inv_c = 1.0f / (float)C;
    float* my_red = red + (size_t)(active ? strip : 0) * G * NP;
    const unsigned pix_bytes = (unsigned)x_cpitch * 2u;

    // Block -> patches: vertically adjacent patches share 6 of their 10 input rows, so every XCD (blockIdx & 7, own L2) gets ONE
    // contiguous range of patches (whole frames at the DDColor sizes) and each of its blocks a contiguous sub-range: the 6.25x
    // re-read of the input is then served by that XCD's L2 instead of HBM.
    int64_t it_begin, it_end;
    {
        const int nb = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nb >> 3, rem = nb & 7;
        const int pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
        const int64_t iters = (total + S - 1) / S, per = (iters + nb - 1) / nb;
        it_begin = (int64_t)pid * per;
        it_end = it_begin + per < iters ? it_begin + per : iters;
    }
    for (int64_t it = it_begin; it < it_end; ++it) {
        const int64_t patch = it * S + strip;
        const bool valid = active && patch < total;
        int b = 0, ho0 = 0, wo0 = 0;
        if (valid) {
            const int px = (int)(patch % n_px);
            const int64_t q = patch / n_px;
            const int py = (int)(q % n_py);
            b = (int)(q / n_py);
            ho0 = py * T;
            wo0 = px * R;
        }
        // byte offset of (frame b, row 0, pixel wo0 - 3, this thread's channels), modulo 2^32: lanes whose column is outside the image
        // are masked, for the others adding the column offset brings it back in range (the range check is on the final voffset)
        const unsigned col0 = (unsigned)(((int64_t)b * H * W + (wo0 - 3)) * x_cpitch + x_coff + c4 * 4) * 2u;
        unsigned colmask = 0;                                  // bit j: input column wo0 - 3 + j is inside the image
#pragma unroll
        for (int j = 0; j < XR; ++j) colmask |= ((unsigned)(wo0 + j - 3) < (unsigned)W ? 1u : 0u) << j;
        if (!valid) colmask = 0;
        constexpr int NPL = PREC ? 2 : 1;                        // planes per pixel: hi (+ lo, x_cpitch BYTES = half a pixel row behind it)
        auto load_row = [&](int r, uint2e (&dst)[NPL][XR]) {
            const int hi = ho0 + r - 3;
            const bool rok = (unsigned)hi < (unsigned)H;
            const unsigned rowoff = col0 + (unsigned)hi * (unsigned)W * pix_bytes;
#pragma unroll
            for (int j = 0; j < XR; ++j) {
                const unsigned vo = (rok && ((colmask >> j) & 1)) ? rowoff + (unsigned)j * pix_bytes : DWLN_OOB;
                dst[0][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, vo, 0, 0);
                if constexpr (PREC) dst[1][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, vo, (unsigned)x_cpitch, 0);      // (soffset: an out-of-range voffset stays out of range)
            }
        };
        float2v acc[T][R][2];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < R; ++j) { acc[t][j][0] = bias2[0]; acc[t][j][1] = bias2[1]; }
        auto wread = [&](const char* p) -> float4e {
            if (WF32) return *reinterpret_cast<const float4e*>(p);
            return __builtin_convertvector(*reinterpret_cast<const half4*>(p), float4e);
        };
        uint2e nxt[NPL][XR];
        load_row(0, nxt);
        // Per input row: convert it, start the loads of the next one, then for every output row t it feeds (tap row dy = r - t) seven
        // steps of one LDS weight read (issued one step ahead) + 2R packed FMAs.  The sched_barriers pin that order: left alone,
        // the scheduler hoists every weight read of the row and spills.
#pragma unroll 1
        for (int r = 0; r < T + 6; ++r) {
            float2v xr[XR][2];
#pragma unroll
            for (int j = 0; j < XR; ++j) {
                float4e f = __builtin_convertvector(__builtin_bit_cast(half4, nxt[0][j]), float4e);
                if constexpr (PREC) f += __builtin_convertvector(__builtin_bit_cast(half4, nxt[1][j]), float4e) * (1.f / 2048.f);      // join_hl: exact product, one rounding
                xr[j][0][0] = f[0]; xr[j][0][1] = f[1]; xr[j][1][0] = f[2]; xr[j][1][1] = f[3];
            }
            if (r + 1 < T + 6) load_row(r + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            // output rows in pairs: a step = two weight reads (tap rows dy and dy - 1) + 4R packed FMAs on the same pixel operands, long
            // enough to cover the LDS latency of the reads issued one step ahead; single rows at the top / bottom of the window
            auto one_row = [&](int t, int dy) {
                const char* wrow = wbase + dy * (7 * C4 * WB);
                float4e wq[2];
                wq[0] = wread(wrow);
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    if (dx < 6) wq[(dx + 1) & 1] = wread(wrow + (dx + 1) * C4 * WB);
                    const float4e f = wq[dx & 1];
                    const float2v w0 = {f[0], f[1]}, w1 = {f[2], f[3]};
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        acc[t][j][0] = __builtin_elementwise_fma(xr[j + dx][0], w0, acc[t][j][0]);
                        acc[t][j][1] = __builtin_elementwise_fma(xr[j + dx][1], w1, acc[t][j][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto two_rows = [&](int t, int dy) {               // rows t (tap row dy) and t + 1 (tap row dy - 1)
                const char* wrow = wbase + dy * (7 * C4 * WB);
                float4e wa[2], wb[2];
                wa[0] = wread(wrow);
                wb[0] = wread(wrow - 7 * C4 * WB);
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    if (dx < 6) {
                        wa[(dx + 1) & 1] = wread(wrow + (dx + 1) * C4 * WB);
                        wb[(dx + 1) & 1] = wread(wrow + (dx + 1 - 7) * C4 * WB);
                    }
                    const float4e fa = wa[dx & 1], fb = wb[dx & 1];
                    const float2v a0 = {fa[0], fa[1]}, a1 = {fa[2], fa[3]}, b0 = {fb[0], fb[1]}, b1 = {fb[2], fb[3]};
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        acc[t][j][0] = __builtin_elementwise_fma(xr[j + dx][0], a0, acc[t][j][0]);
                        acc[t][j][1] = __builtin_elementwise_fma(xr[j + dx][1], a1, acc[t][j][1]);
                        acc[t + 1][j][0] = __builtin_elementwise_fma(xr[j + dx][0], b0, acc[t + 1][j][0]);
                        acc[t + 1][j][1] = __builtin_elementwise_fma(xr[j + dx][1], b1, acc[t + 1][j][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (PAIR) {
#pragma unroll
                for (int t = 0; t < T; t += 2) {
                    const int dy = r - t;                      // tap row for output row t; t + 1 uses dy - 1
                    const bool va = dy >= 0 && dy <= 6, vb = dy - 1 >= 0 && dy - 1 <= 6;
                    if (va && vb) two_rows(t, dy);
                    else if (va) one_row(t, dy);
                    else if (vb) one_row(t + 1, dy - 1);
                }
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int dy = r - t;
                    if (dy >= 0 && dy <= 6) one_row(t, dy);
                }
            }
        }
        // ---- LayerNorm over the C channels of each of the NP pixels (biased variance of the centred values, as F.layer_norm) ----
        float part[NP];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < R; ++j)
                part[t * R + j] = dw_group_sum<WAVE_SUM>((acc[t][j][0][0] + acc[t][j][0][1]) + (acc[t][j][1][0] + acc[t][j][1][1]));
        auto block_sum = [&](float (&v)[NP]) {                  // v: 8-lane sums in, per-pixel sums over all C4 groups out
            if (active && (c4 & (WAVE_SUM ? 63 : 7)) == 0) {
#pragma unroll
                for (int p = 0; p < NP; p += 4) *reinterpret_cast<float4*>(my_red + (c4 >> (WAVE_SUM ? 6 : 3)) * NP + p) = float4{v[p], v[p + 1], v[p + 2], v[p + 3]};
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < NP; ++p) v[p] = 0.f;
#pragma unroll 4
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int p = 0; p < NP; p += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(my_red + g * NP + p);
                    v[p] += q.x; v[p + 1] += q.y; v[p + 2] += q.z; v[p + 3] += q.w;
                }
            }
            __syncthreads();
        };
        block_sum(part);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float mean = part[t * R + j] * inv_c;
                float sq = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    acc[t][j][k][0] -= mean;
                    acc[t][j][k][1] -= mean;
                    sq += acc[t][j][k][0] * acc[t][j][k][0] + acc[t][j][k][1] * acc[t][j][k][1];
                }
                part[t * R + j] = dw_group_sum<WAVE_SUM>(sq);
            }
        block_sum(part);
        if (valid) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c4 * 4), b4 = *reinterpret_cast<const float4*>(beta + c4 * 4);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int ho = ho0 + t, wo = wo0 + j;
                    if (ho >= H || wo >= W) continue;
                    const float rstd = __builtin_amdgcn_rsqf(part[t * R + j] * inv_c + eps);      // v_rsq_f32 (1 ulp); 1 / sqrtf was a 20-instruction division sequence, 16 times per thread
                    const float of[4] = {acc[t][j][0][0] * rstd * g4.x + b4.x, acc[t][j][0][1] * rstd * g4.y + b4.y,
                                         acc[t][j][1][0] * rstd * g4.z + b4.z, acc[t][j][1][1] * rstd * g4.w + b4.w};
                    half_t* yp = y + ((int64_t)(b * H + ho) * W + wo) * y_cpitch + y_coff + c4 * 4;
                    half4 o;
                    if constexpr (PREC) {
                        half4 ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { half_t a_, b_; split_hl_dd(of[e], a_, b_); o[e] = a_; ol[e] = b_; }
                        *reinterpret_cast<half4*>(yp + (y_cpitch >> 1)) = ol;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (half_t)of[e];
                    }
                    *reinterpret_cast<half4*>(yp) = o;
                }
        }
    }
}
bool dwconv7_ln_supported(int C) { return C == 64 || C == 192 || C == 384 || C == 768 || C == 1536; }

template <int C4, bool WF32, int THREADS, bool PAIR, bool PREC = false>
static void dwln_optin() {                                  // > 64 KiB of dynamic LDS: once per kernel instantiation and device
    static std::atomic<uint64_t> optin{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (optin.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv7_ln_kernel<C4, WF32, THREADS, PAIR, PREC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    optin.fetch_or(bit, std::memory_order_release);
}
// every instantiation launch_dwconv7_ln may pick: opted in eagerly from havc_create (see preload_elementwise)
void preload_ddcolor() {
    dwln_optin<16, true, 768, false>();
    dwln_optin<48, true, 512, true>(); dwln_optin<48, true, 768, false>();
    dwln_optin<96, true, 512, true>(); dwln_optin<96, true, 768, false>();
    dwln_optin<192, true, 512, true>(); dwln_optin<192, true, 768, false>();
    dwln_optin<384, false, 768, false>(); dwln_optin<384, false, 512, true>();
    dwln_optin<48, true, 512, false, true>(); dwln_optin<96, true, 512, false, true>(); dwln_optin<192, true, 512, false, true>();      // precise
    dwln_optin<48, true, 512, false>(); dwln_optin<96, true, 512, false>(); dwln_optin<192, true, 512, false>();                        // HAVC_DWLN_VARIANT=3
    (void)hipGetLastError();
}

template <int C4, bool WF32, int THREADS, bool PAIR, bool PREC = false>
static int launch_dwln(const half_t* x, const void* w, const float* bias, const float* gamma, const float* beta, float eps, half_t* y, int B, int H,
                       int W, unsigned x_bytes, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch, hipStream_t s) {
    constexpr int S = THREADS / C4;
    const size_t lds = dwln_lds_bytes(C4 * 4, WF32, THREADS);
    const int64_t total = (int64_t)B * ((H + DWLN_T - 1) / DWLN_T) * ((W + DWLN_R - 1) / DWLN_R);
    const int64_t need = (total + S - 1) / S;
    const int grid = (int)(need < 256 ? need : 256);
    dwln_optin<C4, WF32, THREADS, PAIR, PREC>();
    hipLaunchKernelGGL((dwconv7_ln_kernel<C4, WF32, THREADS, PAIR, PREC>), dim3(grid), dim3(THREADS), lds, s, x, w, bias, gamma, beta, eps, y, B, H, W, x_bytes,
                       x_cpitch, x_coff, y_cpitch, y_coff, w_pitch);
    return (int)hipGetLastError();
}
int launch_dwconv7_ln(const half_t* x, const half_t* w, const float* bias, const float* gamma, const float* beta, float eps, half_t* y, int B,
                      int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch, hipStream_t s) {
    const size_t xb = ((size_t)B * H * W * x_cpitch + x_coff) * 2;
    if (!dwconv7_ln_supported(C) || xb >= DWLN_OOB) return (int)hipErrorInvalidValue;
    static const int variant = [] { const char* e = getenv("HAVC_DWLN_VARIANT"); return e ? atoi(e) : 0; }();      // A/B switch (profiling)
#define DWLN_ARGS x, w, bias, gamma, beta, eps, y, B, H, W, (unsigned)xb, x_cpitch, x_coff, y_cpitch, y_coff, w_pitch, s
    switch (C) {
        case 64: return launch_dwln<16, true, 768, false>(DWLN_ARGS);
        // variant 3 (round 6 A/B): 512 threads, one output row per step -- the geometry of the precise form (no spills, 2 waves per SIMD)
        case 192: return variant == 3 ? launch_dwln<48, true, 512, false>(DWLN_ARGS) : variant == 1 ? launch_dwln<48, true, 512, true>(DWLN_ARGS) : launch_dwln<48, true, 768, false>(DWLN_ARGS);
        case 384: return variant == 3 ? launch_dwln<96, true, 512, false>(DWLN_ARGS) : variant == 1 ? launch_dwln<96, true, 512, true>(DWLN_ARGS) : launch_dwln<96, true, 768, false>(DWLN_ARGS);
        case 768: return variant == 3 ? launch_dwln<192, true, 512, false>(DWLN_ARGS) : variant == 1 ? launch_dwln<192, true, 512, true>(DWLN_ARGS) : launch_dwln<192, true, 768, false>(DWLN_ARGS);
        case 1536: return variant == 2 ? launch_dwln<384, false, 768, false>(DWLN_ARGS) : launch_dwln<384, false, 512, true>(DWLN_ARGS);
    }
#undef DWLN_ARGS
    return (int)hipErrorInvalidValue;
}

// precise form: x / y pair tensors (cpitch = 2 P), fp32 weights [49][w_pitch]; the ConvNeXt widths whose fp32 weights fit the LDS (192 / 384 / 768 channels: 150 KB at 768;
// the 1536-channel stage -- 16 x 16 pixels per frame -- keeps dwconv7_p + layernorm_p)
bool dwconv7_ln_p_supported(int C) { return C == 192 || C == 384 || C == 768; }
int launch_dwconv7_ln_p(const half_t* x, const float* w, const float* bias, const float* gamma, const float* beta, float eps, half_t* y, int B,
                        int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch, hipStream_t s) {
    const size_t xb = ((size_t)B * H * W * x_cpitch + x_coff) * 2;
    if (!dwconv7_ln_p_supported(C) || xb >= DWLN_OOB || (x_cpitch & 1) || (y_cpitch & 1)) return (int)hipErrorInvalidValue;
#define DWLN_ARGS x, w, bias, gamma, beta, eps, y, B, H, W, (unsigned)xb, x_cpitch, x_coff, y_cpitch, y_coff, w_pitch, s
    switch (C) {
        case 192: return launch_dwln<48, true, 512, false, true>(DWLN_ARGS);
        case 384: return launch_dwln<96, true, 512, false, true>(DWLN_ARGS);
        case 768: return launch_dwln<192, true, 512, false, true>(DWLN_ARGS);
    }
#undef DWLN_ARGS
    return (int)hipErrorInvalidValue;
}

// ---- DDColor tail: einsum(bqc,bchw->bqhw) and the 1x1 refine conv are both linear per pixel, and so are the shuffle and the blur in
// front of them -- so they are applied in the other order.  Step 1 folds the colour embeddings E [queries][C] and the refine rows
// R [2][queries] of a frame into ONE 2 x C matrix (fp32); step 2 is the FUSE_PROJ epilogue of the last_shuf conv (conv_pipe_epilogue
// .inc); step 3 shuffles + blurs the resulting 2-channel map and adds the image term of the refine conv. ----
__global__ void fold_queries_kernel(const half_t* __restrict__ e, int e_cpitch, int e_coff, int tok, const float* __restrict__ r, int r_pitch,
                                    int nq, float* __restrict__ out, int C) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const half_t* eb = e + (int64_t)b * tok * e_cpitch + e_coff + c;
    float a0 = 0.f, a1 = 0.f;
    for (int q = 0; q < nq; ++q) {
        const float v = (float)eb[(int64_t)q * e_cpitch];
        a0 = fmaf(r[q], v, a0);
        a1 = fmaf(r[r_pitch + q], v, a1);
    }
    out[((int64_t)b * 2 + 0) * C + c] = a0;
    out[((int64_t)b * 2 + 1) * C + c] = a1;
}
int launch_fold_queries(const half_t* e, int e_cpitch, int e_coff, int tok, const float* r, int r_pitch, int nq, float* out, int B, int C,
                        hipStream_t s) {
    hipLaunchKernelGGL(fold_queries_kernel, dim3((C + 63) / 64, B), dim3(64), 0, s, e, e_cpitch, e_coff, tok, r, r_pitch, nq, out, C);
    return (int)hipGetLastError();
}

// proj: fp32 [B][Hi*Wi][16][2] (sub-pixel dy*4+dx of the shuffle) -> y[b][Y][X][0..1] = 0.25 * (the 2x2 window ending at (Y, X), replicate-
// padded on the top / left: ReplicationPad2d((1,0,1,0)) + AvgPool2d(2, 1)) + R_img . img[Y][X] + bias
// PREC (round 6, the precise plan's HAVC_F_FUSE_PROJ tail): the image is a pair view and the ab map is written as pairs; the arithmetic is the same fp32
template <bool PREC>
__global__ void shuf4_blur_ab_kernel(const float* __restrict__ proj, const half_t* __restrict__ img, int img_cpitch, int img_coff,
                                     const float* __restrict__ rimg, const float* __restrict__ bias, half_t* __restrict__ y, int y_cpitch,
                                     int y_coff, int B, int Hi, int Wi) {
    const int Ho = Hi * 4, Wo = Wi * 4;
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % Wo);
        const int64_t t = i / Wo;
        const int Y = (int)(t % Ho), b = (int)(t / Ho);
        const int y0 = max(Y - 1, 0), x0 = max(X - 1, 0);
        auto ld = [&](int ay, int ax) -> float2 {
            return *reinterpret_cast<const float2*>(proj + ((((int64_t)b * Hi + (ay >> 2)) * Wi + (ax >> 2)) * 16 + (ay & 3) * 4 + (ax & 3)) * 2);
        };
        const float2 v00 = ld(y0, x0), v01 = ld(y0, X), v10 = ld(Y, x0), v11 = ld(Y, X);
        const half_t* ip = img + i * img_cpitch + img_coff;
        float i0 = (float)ip[0], i1 = (float)ip[1], i2 = (float)ip[2];
        if (PREC) {
            const int ilo = img_cpitch >> 1;
            i0 += (float)ip[ilo] * (1.f / 2048.f); i1 += (float)ip[ilo + 1] * (1.f / 2048.f); i2 += (float)ip[ilo + 2] * (1.f / 2048.f);
        }
        const float a = (v00.x + v01.x + v10.x + v11.x) * 0.25f + (rimg[0] * i0 + rimg[1] * i1 + rimg[2] * i2) + bias[0];
        const float bb = (v00.y + v01.y + v10.y + v11.y) * 0.25f + (rimg[3] * i0 + rimg[4] * i1 + rimg[5] * i2) + bias[1];
        half_t* yp = y + i * y_cpitch + y_coff;
        if (PREC) {
            const int ylo = y_cpitch >> 1;
            half_t hh, ll;
            split_hl_dd(a, hh, ll); yp[0] = hh; yp[ylo] = ll;
            split_hl_dd(bb, hh, ll); yp[1] = hh; yp[ylo + 1] = ll;
            continue;
        }
        if ((y_coff & 7) == 0) {                            // the ab map is an 8-channel chunk with six zero pads: one 16-byte store (no partial-line writes)
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)0.f;
            o[0] = (half_t)a; o[1] = (half_t)bb;
            *reinterpret_cast<half8*>(yp) = o;
        } else {
            yp[0] = (half_t)a;
            yp[1] = (half_t)bb;
        }
    }
}
int launch_shuf4_blur_ab(const float* proj, const half_t* img, int img_cpitch, int img_coff, const float* rimg, const float* bias, half_t* y,
                         int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s) {
    hipLaunchKernelGGL(shuf4_blur_ab_kernel<false>, dim3(grid_for_dd((int64_t)B * Hi * 4 * Wi * 4)), dim3(256), 0, s, proj, img, img_cpitch, img_coff, rimg,
                       bias, y, y_cpitch, y_coff, B, Hi, Wi);
    return (int)hipGetLastError();
}
int launch_shuf4_blur_ab_p(const float* proj, const half_t* img, int img_cpitch, int img_coff, const float* rimg, const float* bias, half_t* y,
                           int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s) {
    hipLaunchKernelGGL(shuf4_blur_ab_kernel<true>, dim3(grid_for_dd((int64_t)B * Hi * 4 * Wi * 4)), dim3(256), 0, s, proj, img, img_cpitch, img_coff, rimg,
                       bias, y, y_cpitch, y_coff, B, Hi, Wi);
    return (int)hipGetLastError();
}

// ---- multi-head attention on token / pixel buffers: O[b][q][h*D + :] = softmax_k(Q.K * scale) V, D = 32 ----
// One wave per (frame, head, query); lanes stride over the keys with a private online softmax, merged at the end.
// Q rows live at q + (b * q_stride_tok + i) * q_cpitch + q_coff + h * 32; K / V likewise with their own offsets.
__global__ void mha32_kernel(const half_t* __restrict__ q, int q_cpitch, int q_coff, int q_tok, const half_t* __restrict__ kv, int kv_cpitch,
                             int k_coff, int v_coff, int kv_tok, half_t* __restrict__ o, int o_cpitch, int o_coff, int o_tok, int B, int heads,
                             int Lq, int Lk, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t total = (int64_t)B * heads * Lq;
    if (wave >= total) return;
    const int iq = (int)(wave % Lq);
    const int h = (int)((wave / Lq) % heads);
    const int b = (int)(wave / ((int64_t)Lq * heads));
    float qv[32];
    {
        const half_t* qp = q + ((int64_t)b * q_tok + iq) * q_cpitch + q_coff + h * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const half8 t = *reinterpret_cast<const half8*>(qp + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[c * 8 + e] = (float)t[e] * scale;
        }
    }
    float m = -INFINITY, l = 0.f, acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    const half_t* kb = kv + (int64_t)b * kv_tok * kv_cpitch + h * 32;
    for (int j = lane; j < Lk; j += 64) {
        const half_t* kp = kb + (int64_t)j * kv_cpitch;
        float sdot = 0.f;
        half8 vv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const half8 t = *reinterpret_cast<const half8*>(kp + k_coff + c * 8);
            vv[c] = *reinterpret_cast<const half8*>(kp + v_coff + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot += qv[c * 8 + e] * (float)t[e];
        }
        const float mn = fmaxf(m, sdot), corr = __expf(m - mn), pj = __expf(sdot - mn);
        l = l * corr + pj;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c * 8 + e] = acc[c * 8 + e] * corr + pj * (float)vv[c][e];
        m = mn;
    }
    // merge the 64 partial softmaxes
    float mall = m;
    for (int o2 = 32; o2 > 0; o2 >>= 1) mall = fmaxf(mall, __shfl_xor(mall, o2));
    const float w = (m == -INFINITY) ? 0.f : __expf(m - mall);
    l *= w;
    for (int o2 = 32; o2 > 0; o2 >>= 1) l += __shfl_xor(l, o2);
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        float a = acc[e] * w;
        for (int o2 = 32; o2 > 0; o2 >>= 1) a += __shfl_xor(a, o2);
        acc[e] = a;
    }
    if (lane == 0) {
        half_t* op = o + ((int64_t)b * o_tok + iq) * o_cpitch + o_coff + h * 32;
        const float inv = 1.f / l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            half8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (half_t)(acc[c * 8 + e] * inv);
            *reinterpret_cast<half8*>(op + c * 8) = t;
        }
    }
}
int launch_mha32(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                 half_t* o, int o_cpitch, int o_coff, int o_tok, int B, int heads, int Lq, int Lk, float scale, hipStream_t s) {
    const int64_t waves = (int64_t)B * heads * Lq;
    hipLaunchKernelGGL(mha32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff,
                       kv_tok, o, o_cpitch, o_coff, o_tok, B, heads, Lq, Lk, scale);
    return (int)hipGetLastError();
}

// ---- the same attention, split over the keys (flash-decoding form): what DDColor's cross-attention needs (100 queries against up
// to 16 384 keys: one wave per query re-reads every K / V row 100 times).  Block (split c, head h, frame b), one thread per query:
// the block stages KC keys of K and V in LDS once (every thread then reads the SAME row: LDS broadcast), each thread runs its
// query over them with an online softmax (rescaled once per 8 keys) and writes {m, l, acc[32]}; mha32_merge_kernel folds the splits.
constexpr int MHA_KC = 256;
__global__ void __launch_bounds__(128) mha32_split_kernel(const half_t* __restrict__ q, int q_cpitch, int q_coff, int q_tok,
                                                          const half_t* __restrict__ kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                                                          float* __restrict__ part, int heads, int Lq, int Lk, float scale) {
    __shared__ __attribute__((aligned(16))) half_t Ks[MHA_KC][32];
    __shared__ __attribute__((aligned(16))) half_t Vs[MHA_KC][32];
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nsplit = gridDim.x;
    const int k0 = c * MHA_KC, nk = min(MHA_KC, Lk - k0);
    const half_t* kb = kv + ((int64_t)b * kv_tok + k0) * kv_cpitch + h * 32;
    for (int i = threadIdx.x; i < nk * 8; i += blockDim.x) {                // 8 16-byte chunks per key: 4 of K, 4 of V
        const int key = i >> 3, ch = i & 7;
        const half8 t = *reinterpret_cast<const half8*>(kb + (int64_t)key * kv_cpitch + (ch < 4 ? k_coff + ch * 8 : v_coff + (ch - 4) * 8));
        *reinterpret_cast<half8*>(ch < 4 ? &Ks[key][ch * 8] : &Vs[key][(ch - 4) * 8]) = t;
    }
    __syncthreads();
    for (int iq = threadIdx.x; iq < Lq; iq += blockDim.x) {
        float qv[32], acc[32];
        const half_t* qp = q + ((int64_t)b * q_tok + iq) * q_cpitch + q_coff + h * 32;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const half8 t = *reinterpret_cast<const half8*>(qp + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { qv[cc * 8 + e] = (float)t[e] * scale; acc[cc * 8 + e] = 0.f; }
        }
        float m = -INFINITY, l = 0.f;
        for (int j0 = 0; j0 < nk; j0 += 8) {
            float sc[8];
            float gmax = m;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = -INFINITY;
                if (j0 + j < nk) {
                    d = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const half8 t = *reinterpret_cast<const half8*>(&Ks[j0 + j][cc * 8]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) d += qv[cc * 8 + e] * (float)t[e];
                    }
                }
                sc[j] = d;
                gmax = fmaxf(gmax, d);
            }
            const float corr = __expf(m - gmax);                              // m = -inf on the first group: corr = 0, acc is 0 anyway
            l *= corr;
#pragma unroll
            for (int e = 0; e < 32; ++e) acc[e] *= corr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j0 + j < nk) {
                    const float pj = __expf(sc[j] - gmax);
                    l += pj;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const half8 t = *reinterpret_cast<const half8*>(&Vs[j0 + j][cc * 8]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[cc * 8 + e] += pj * (float)t[e];
                    }
                }
            }
            m = gmax;
        }
        float* o = part + ((((int64_t)b * heads + h) * nsplit + c) * Lq + iq) * 34;
        o[0] = m; o[1] = l;
#pragma unroll
        for (int e = 0; e < 32; ++e) o[2 + e] = acc[e];
    }
}
// ---- the key-split attention on MFMA (round 2).  Same splitting and the same {m, l, acc[32]} partial states as mha32_split_kernel
// (mha32_merge_kernel folds them); what changes is who does the arithmetic: a block (split of 256 keys, head, frame) = 4 waves, a
// wave owns up to two 16-query fragments.  Head dim 32 is exactly the K of v_mfma_f32_16x16x32_f16:
//   S^T[key][query] = K[key][:] . Q[query][:]      A = K rows straight from global (16 B per lane: row = key, chunk = lane >> 4)
//   O^T[dv][query]  = V^T[dv][key] . P^T[key][query] A = V^T from an LDS image transposed while staging, B = P from registers
// S fragment f covers keys 32 (f >> 1) + (i >> 2) 8 + (f & 1) 4 + (i & 3) (i = MFMA row), so that a lane's eight P values of two
// neighbouring fragments are keys lg * 8 .. + 7 of a 32-key step in natural order: the second MFMA's B operand needs no shuffle.
typedef float float4v_dd __attribute__((ext_vector_type(4)));
constexpr int MHA_VP = MHA_KC + 8;                 // V^T row pitch (halfs): 528 B rows spread the 16 rows of a fragment read over the banks
// Round 4: a block walks `nch` consecutive 256-key chunks of its (head, frame) with a running (max, sum, acc) per query fragment before it writes
// its partial state: at 16 384 keys x 64 frames the {m, l, acc[32]} partials of 64 splits per head were 0.45 GB written and read again by the merge
// kernel next to 1.07 GB of K / V; with four chunks per block they are a quarter of that.  nch is chosen by the launcher (>= 1 024 blocks stay).
__global__ void __launch_bounds__(256, 2) mha32_split_mfma_kernel(const half_t* __restrict__ q, int q_cpitch, int q_coff, int q_tok,
                                                               const half_t* __restrict__ kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                                                               float* __restrict__ part, int heads, int Lq, int Lk, float scale, int nch) {
    __shared__ __attribute__((aligned(16))) half_t VsT[32 * MHA_VP];
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nsplit = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int nqf = (Lq + 15) / 16;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    float4v_dd o0_run[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, o1_run[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int chk = 0; chk < nch; ++chk) {
        const int k0 = (c * nch + chk) * MHA_KC, nk = min(MHA_KC, Lk - k0);
        if (nk <= 0) break;                                    // block-uniform
        const half_t* kb = kv + ((int64_t)b * kv_tok + k0) * kv_cpitch + h * 32;
        if (chk) __syncthreads();                              // every wave is done with the previous chunk's V image
        // V tile -> LDS, transposed: thread (key, chunk) scatters its 8 channels into 8 rows
        for (int i = tid; i < MHA_KC * 4; i += 256) {
            const int key = i >> 2, ch = i & 3;
            half8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (half_t)0.f;
            if (key < nk) t = *reinterpret_cast<const half8*>(kb + (int64_t)key * kv_cpitch + v_coff + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) VsT[(ch * 8 + e) * MHA_VP + key] = t[e];
        }
        // K fragments of the whole chunk in registers: 16 fragments x 16 B per lane
        half8 kf[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int key = 32 * (f >> 1) + (lr >> 2) * 8 + (f & 1) * 4 + (lr & 3);
            kf[f] = *reinterpret_cast<const half8*>(kb + (int64_t)min(key, nk - 1) * kv_cpitch + k_coff + lg * 8);     // masked below when key >= nk
        }
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int qfi = wave + slot * 4;
            if (qfi >= nqf) continue;                          // wave-uniform
            const int iq = qfi * 16 + lr;
            const half8 qv = *reinterpret_cast<const half8*>(q + ((int64_t)b * q_tok + min(iq, Lq - 1)) * q_cpitch + q_coff + h * 32 + lg * 8);
            float4v_dd sacc[16];
            float mx = -INFINITY;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                sacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[f], qv, float4v_dd{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 32 * (f >> 1) + lg * 8 + (f & 1) * 4 + r;
                    sacc[f][r] = key < nk ? sacc[f][r] * scale : -INFINITY;
                    mx = fmaxf(mx, sacc[f][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[slot], mx);        // finite: a chunk has at least one key
            const float alpha = __expf(m_run[slot] - m_new);   // exp(-inf) = 0 on the first chunk
            float l = 0.f;
            float4v_dd o0 = o0_run[slot], o1 = o1_run[slot];
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __expf(sacc[2 * s2 + (j >> 2)][j & 3] - m_new);
                    l += pv;
                    pf[j] = (half_t)pv;
                }
                const half8 v0 = *reinterpret_cast<const half8*>(&VsT[lr * MHA_VP + s2 * 32 + lg * 8]);
                const half8 v1 = *reinterpret_cast<const half8*>(&VsT[(16 + lr) * MHA_VP + s2 * 32 + lg * 8]);
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, pf, o1, 0, 0, 0);
            }
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            m_run[slot] = m_new;
            l_run[slot] = l_run[slot] * alpha + l;
            o0_run[slot] = o0;
            o1_run[slot] = o1;
        }
    }
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int qfi = wave + slot * 4, iq = qfi * 16 + lr;
        if (qfi >= nqf || iq >= Lq) continue;
        float* op = part + ((((int64_t)b * heads + h) * nsplit + c) * Lq + iq) * 34;
        if (lg == 0) { op[0] = m_run[slot]; op[1] = l_run[slot]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { op[2 + lg * 4 + r] = o0_run[slot][r]; op[2 + 16 + lg * 4 + r] = o1_run[slot][r]; }
    }
}

__global__ void mha32_merge_kernel(const float* __restrict__ part, half_t* __restrict__ o, int o_cpitch, int o_coff, int o_tok, int B, int heads,
                                   int Lq, int nsplit) {
    const int64_t total = (int64_t)B * heads * Lq * 4;                       // 4 threads per (frame, head, query): 8 channels each
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i & 3);
        int64_t r = i >> 2;
        const int iq = (int)(r % Lq);
        r /= Lq;
        const int h = (int)(r % heads), b = (int)(r / heads);
        const float* p0 = part + (((int64_t)b * heads + h) * nsplit * Lq + iq) * 34;
        float M = -INFINITY;
        for (int c = 0; c < nsplit; ++c) M = fmaxf(M, p0[(int64_t)c * Lq * 34]);
        float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nsplit; ++c) {
            const float* pc = p0 + (int64_t)c * Lq * 34;
            const float w = __expf(pc[0] - M);
            L += pc[1] * w;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pc[2 + cc * 8 + e] * w;
        }
        half8 t;
        const float inv = 1.f / L;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (half_t)(acc[e] * inv);
        *reinterpret_cast<half8*>(o + ((int64_t)b * o_tok + iq) * o_cpitch + o_coff + h * 32 + cc * 8) = t;
    }
}
int mha32_nsplit(int Lk) { return (Lk + MHA_KC - 1) / MHA_KC; }
int launch_mha32_split(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                       half_t* o, int o_cpitch, int o_coff, int o_tok, float* part, int B, int heads, int Lq, int Lk, float scale, hipStream_t s) {
    const int nchunks = mha32_nsplit(Lk);                                     // 256-key chunks; the partial-state buffer is sized for one split per chunk
    static const bool v1 = getenv("HAVC_MHA_V1") != nullptr;                  // A/B switch (profiling): the one-thread-per-query kernel
    static const int nch_env = [] { const char* e = getenv("HAVC_MHA_NCH"); return e ? atoi(e) : 0; }();       // A/B: chunks per block (0 = automatic)
    // chunks per block: as many (<= 4) as keep >= 1 024 blocks in flight (one frame of the coarse level stays at one chunk per block)
    int nch = 1;
    if (!v1) {
        const int64_t blocks1 = (int64_t)nchunks * heads * B;
        nch = nch_env > 0 ? nch_env : (int)std::min<int64_t>(4, std::max<int64_t>(1, blocks1 / 1024));
        nch = std::max(1, std::min(nch, nchunks));
    }
    const int nsplit = (nchunks + nch - 1) / nch;
    if (v1)
        hipLaunchKernelGGL(mha32_split_kernel, dim3(nsplit, heads, B), dim3(128), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff, kv_tok,
                           part, heads, Lq, Lk, scale);
    else
        hipLaunchKernelGGL(mha32_split_mfma_kernel, dim3(nsplit, heads, B), dim3(256), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff,
                           kv_tok, part, heads, Lq, Lk, scale, nch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mha32_merge_kernel, dim3(grid_for_dd((int64_t)B * heads * Lq * 4)), dim3(256), 0, s, part, o, o_cpitch, o_coff, o_tok, B, heads,
                       Lq, nsplit);
    return (int)hipGetLastError();
}

// ---- PixelShuffle(4) + ReplicationPad(1,0,1,0) + AvgPool2d(2,1) (decoder.last_shuf).  Input channel order (dy*4+dx)*C + c ----
__global__ void pixshuf4_blur_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Hi, int Wi, int C8, int x_cpitch, int x_coff,
                                     int y_cpitch, int y_coff) {
    const int Ho = Hi * 4, Wo = Wi * 4, C = C8 * 8;
    const int64_t total = (int64_t)B * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t pix = i / C8;
        const int X = (int)(pix % Wo);
        pix /= Wo;
        const int Y = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        const int y0 = max(Y - 1, 0), x0 = max(X - 1, 0);
        auto ld = [&](int ay, int ax) -> half8 {
            return *reinterpret_cast<const half8*>(x + ((int64_t)(b * Hi + (ay >> 2)) * Wi + (ax >> 2)) * x_cpitch + x_coff +
                                                   ((ay & 3) * 4 + (ax & 3)) * C + c8 * 8);
        };
        const half8 v00 = ld(y0, x0), v01 = ld(y0, X), v10 = ld(Y, x0), v11 = ld(Y, X);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v00[e] + (float)v01[e] + (float)v10[e] + (float)v11[e]) * 0.25f);
        *reinterpret_cast<half8*>(y + ((int64_t)(b * Ho + Y) * Wo + X) * y_cpitch + y_coff + c8 * 8) = o;
    }
}
int launch_pixshuf4_blur(const half_t* x, half_t* y, int B, int Hi, int Wi, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                         hipStream_t s) {
    hipLaunchKernelGGL(pixshuf4_blur_kernel, dim3(grid_for_dd((int64_t)B * Hi * 4 * Wi * 4 * (C / 8))), dim3(256), 0, s, x, y, B, Hi, Wi, C / 8,
                       x_cpitch, x_coff, y_cpitch, y_coff);
    return (int)hipGetLastError();
}
