// HBM-bound / small kernels of the DDColor path (architecture: oracle/ddcolor.py).  NHWC fp16 activations,
// fp32 arithmetic.  Token tensors (the 100 colour queries) use the same layout with H = 1, W = tokens.
#include "kernels.h"
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

// ---- LayerNorm over the C channels of every pixel / token (biased variance, two passes in registers), one wave per pixel ----
// C <= 2048 (4 x 16-byte chunks per lane).  Channels C .. C8*8-1 (padding) are written as 0.
__global__ void layernorm_c_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, int64_t npix, int C, int x_cpitch, int x_coff, int y_cpitch,
                                   int y_coff) {
    const int lane = threadIdx.x & 63;
    const int C8 = (C + 7) / 8;
    const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave0; p < npix; p += nwaves) {
        float v[4][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = lane + k * 64;
            if (c8 < C8) {
                const half8 h = *reinterpret_cast<const half8*>(x + p * x_cpitch + x_coff + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[k][e] = (c8 * 8 + e < C) ? (float)h[e] : 0.f; sum += v[k][e]; }
            }
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = lane + k * 64;
            if (c8 < C8)
#pragma unroll
                for (int e = 0; e < 8; ++e) if (c8 * 8 + e < C) { const float d = v[k][e] - mean; sq += d * d; }
        }
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c8 = lane + k * 64;
            if (c8 < C8) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c8 * 8 + e;
                    o[e] = c < C ? (half_t)((v[k][e] - mean) * rstd * gamma[c] + beta[c]) : (half_t)0.f;
                }
                *reinterpret_cast<half8*>(y + p * y_cpitch + y_coff + c8 * 8) = o;
            }
        }
    }
}
int launch_layernorm_c(const half_t* x, half_t* y, const float* gamma, const float* beta, float eps, int64_t npix, int C, int x_cpitch,
                       int x_coff, int y_cpitch, int y_coff, hipStream_t s) {
    if (C > 2048) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_c_kernel, dim3(grid_for_dd(npix, 4)), dim3(256), 0, s, x, y, gamma, beta, eps, npix, C, x_cpitch, x_coff,
                       y_cpitch, y_coff);
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
__global__ void __launch_bounds__(256) mha32_split_mfma_kernel(const half_t* __restrict__ q, int q_cpitch, int q_coff, int q_tok,
                                                               const half_t* __restrict__ kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                                                               float* __restrict__ part, int heads, int Lq, int Lk, float scale) {
    __shared__ __attribute__((aligned(16))) half_t VsT[32 * MHA_VP];
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nsplit = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int k0 = c * MHA_KC, nk = min(MHA_KC, Lk - k0);
    const half_t* kb = kv + ((int64_t)b * kv_tok + k0) * kv_cpitch + h * 32;
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
    // K fragments of the whole split in registers: 16 fragments x 16 B per lane
    half8 kf[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        const int key = 32 * (f >> 1) + (lr >> 2) * 8 + (f & 1) * 4 + (lr & 3);
        kf[f] = *reinterpret_cast<const half8*>(kb + (int64_t)min(key, nk - 1) * kv_cpitch + k_coff + lg * 8);     // masked below when key >= nk
    }
    __syncthreads();
    const int nqf = (Lq + 15) / 16;
    for (int qfi = wave; qfi < nqf; qfi += 4) {
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
        float l = 0.f;
        float4v_dd o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            half8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = __expf(sacc[2 * s2 + (j >> 2)][j & 3] - mx);
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
        if (iq < Lq) {
            float* op = part + ((((int64_t)b * heads + h) * nsplit + c) * Lq + iq) * 34;
            if (lg == 0) { op[0] = mx; op[1] = l; }
#pragma unroll
            for (int r = 0; r < 4; ++r) { op[2 + lg * 4 + r] = o0[r]; op[2 + 16 + lg * 4 + r] = o1[r]; }
        }
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
    const int nsplit = mha32_nsplit(Lk);
    static const bool v1 = getenv("HAVC_MHA_V1") != nullptr;                  // A/B switch (profiling): the one-thread-per-query kernel
    if (v1)
        hipLaunchKernelGGL(mha32_split_kernel, dim3(nsplit, heads, B), dim3(128), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff, kv_tok,
                           part, heads, Lq, Lk, scale);
    else
        hipLaunchKernelGGL(mha32_split_mfma_kernel, dim3(nsplit, heads, B), dim3(256), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff,
                           kv_tok, part, heads, Lq, Lk, scale);
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
