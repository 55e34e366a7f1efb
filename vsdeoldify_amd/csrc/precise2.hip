// "Precise" mode (HAVC_F_PRECISE), second part: the non-conv ops of DDColor and of the Zhang colorizers on hi / lo fp16 pairs (round 5).
//
// Round 4 built the mode for the DeOldify generators (precise.hip); the reference is fp32 on every model
// (vsdeoldify/colorization/__init__.py:76-95, vsdeoldify/vsslib/vsmodels.py:353-363), so the same contract -- CIEDE2000 < 1.0 per pixel against the
// fp32 graph -- needs the same arithmetic here: convolutions on the three-segment fp16 MFMA walk (conv_common.h, plan.py split_weights) and everything
// else below reading a pair as ONE fp32 value, computing in fp32 like torch, storing a pair.  Layout as in precise.hip: pixel row = [hi: P | lo: P].
// These kernels are written for correctness first (one wave per pixel / token, gathers through L1 / L2); the fast path keeps its own tuned kernels.
#include "conv_common.h"

namespace {

inline int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

__device__ __forceinline__ void load8(const half_t* p, int lo, float v[8]) {
    const half8 h = *reinterpret_cast<const half8*>(p), l = *reinterpret_cast<const half8*>(p + lo);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = join_hl(h[e], l[e]);
}
__device__ __forceinline__ void store8(half_t* p, int lo, const float v[8]) {
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { half_t a, b; split_hl(v[e], a, b); h[e] = a; l[e] = b; }
    *reinterpret_cast<half8*>(p) = h;
    *reinterpret_cast<half8*>(p + lo) = l;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- per-pixel C -> 2 projection (proj2_kernel of elementwise.hip on pairs; eccv16.py:95, siggraph17.py:113-114) ----
__global__ void proj2_p_kernel(const half_t* __restrict__ x, int x_cp, int x_co, int C, const float* __restrict__ w, const float* __restrict__ bias, int mode,
                               float mul, float* __restrict__ out, int64_t npix) {
    const int lane = threadIdx.x & 63, lo = x_cp >> 1;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pix = wave; pix < npix; pix += nwaves) {
        const half_t* xp = x + pix * x_cp + x_co;
        float v[8];
        int n = 0;
        float mx = -3.0e38f;
        for (int c = lane; c < C; c += 64) { v[n] = join_hl(xp[c], xp[c + lo]); mx = fmaxf(mx, v[n]); ++n; }
        float den = 1.f;
        if (mode & 1) {
            mx = wave_max(mx);
            float s = 0.f;
            n = 0;
            for (int c = lane; c < C; c += 64) { v[n] = expf(v[n] - mx); s += v[n]; ++n; }
            den = wave_sum(s);
        }
        float a0 = 0.f, a1 = 0.f;
        n = 0;
        for (int c = lane; c < C; c += 64) { a0 += v[n] * w[c]; a1 += v[n] * w[C + c]; ++n; }
        a0 = wave_sum(a0) / den;
        a1 = wave_sum(a1) / den;
        if (mode & 2) { a0 = tanhf(a0 + bias[0]); a1 = tanhf(a1 + bias[1]); }
        if (lane == 0) { out[pix * 2] = a0 * mul; out[pix * 2 + 1] = a1 * mul; }
    }
}

// ---- LayerNorm over the C channels of a pixel / token (nn.LayerNorm, ConvNeXt's channels-last LayerNorm): one wave per pixel, two passes ----
template <int NCH>
__global__ void layernorm_p_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, int64_t npix, int C8, int x_cp, int x_co, int y_cp, int y_co, int relu) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float inv_c = 1.f / (float)(C8 * 8);
    for (int64_t pix = wave; pix < npix; pix += nwaves) {
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
            if (ch < C8) {
                load8(x + pix * x_cp + x_co + ch * 8, x_cp >> 1, v[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[k][e];
            }
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (lane + k * 64 < C8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
            }
        const float rstd = 1.f / sqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch >= C8) continue;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (v[k][e] - mean) * rstd * gamma[ch * 8 + e] + beta[ch * 8 + e];
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
            store8(y + pix * y_cp + y_co + ch * 8, y_cp >> 1, o);
        }
    }
}

// ---- depthwise 7x7, pad 3, + bias (ConvNeXt block head): fp32 weights [49][w_pitch], one thread per (pixel, 8 channels) ----
__global__ void dwconv7_p_kernel(const half_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, half_t* __restrict__ y, int B, int H,
                                 int W, int C8, int x_cp, int x_co, int y_cp, int y_co, int w_pitch) {
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
        for (int dy = 0; dy < 7; ++dy) {
            const int hi = ho - 3 + dy;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int dx = 0; dx < 7; ++dx) {
                const int wi = wo - 3 + dx;
                if ((unsigned)wi >= (unsigned)W) continue;
                float v[8];
                load8(x + ((int64_t)(b * H + hi) * W + wi) * x_cp + x_co + c8 * 8, x_cp >> 1, v);
                const float* wp = w + (dy * 7 + dx) * w_pitch + c8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], wp[e], acc[e]);
            }
        }
        store8(y + ((int64_t)(b * H + ho) * W + wo) * y_cp + y_co + c8 * 8, y_cp >> 1, acc);
    }
}

// ---- multi-head attention, head dim 32, fp32 (nn.MultiheadAttention of the colour decoder): block = (8 queries, head, frame) ----
// K / V tiles of 64 keys staged in LDS as fp32 (pitch 36, see MP_P); thread (query t >> 5, key lane t & 31)
// walks keys kl, kl + 32 of every tile with a private online softmax; the 32 key lanes of a query merge their states by shuffles at the end.
constexpr int MP_QG = 8, MP_TK = 64, MP_P = 36;       // row pitch 36 floats (round 6): 16-byte aligned rows -> the K / V rows are read with ds_read_b128 (8 instead of 32 LDS
                                                        // instructions per row); 32 key lanes x 16 B at a 144-byte pitch touch every bank exactly 4 times: no excess conflicts
__global__ void __launch_bounds__(256) mha32_p_kernel(const half_t* __restrict__ q, int q_cp, int q_co, int q_tok, const half_t* __restrict__ kv, int kv_cp,
                                                      int k_co, int v_co, int kv_tok, half_t* __restrict__ o, int o_cp, int o_co, int o_tok, int heads, int Lq,
                                                      int Lk, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[MP_TK * MP_P], Vs[MP_TK * MP_P];
    const int t = threadIdx.x, qi = t >> 5, kl = t & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int iq = blockIdx.x * MP_QG + qi;
    const bool valid = iq < Lq;
    float qv[32];
    {
        const half_t* qp = q + ((int64_t)b * q_tok + (valid ? iq : 0)) * q_cp + q_co + h * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float tv[8];
            load8(qp + c * 8, q_cp >> 1, tv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[c * 8 + e] = tv[e] * scale;        // q * scaling before Q K^T, as torch does
        }
    }
    float m = -INFINITY, l = 0.f, acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    const half_t* kb = kv + (int64_t)b * kv_tok * kv_cp + h * 32;
    const int lr = t >> 2, lc = t & 3;                                         // loader role: key row, 8-channel chunk
    for (int k0 = 0; k0 < Lk; k0 += MP_TK) {
        __syncthreads();
        {
            float kx[8], vx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) kx[e] = vx[e] = 0.f;
            if (k0 + lr < Lk) {
                const half_t* kp = kb + (int64_t)(k0 + lr) * kv_cp;
                load8(kp + k_co + lc * 8, kv_cp >> 1, kx);
                load8(kp + v_co + lc * 8, kv_cp >> 1, vx);
            }
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                *reinterpret_cast<float4*>(&Ks[lr * MP_P + lc * 8 + e]) = make_float4(kx[e], kx[e + 1], kx[e + 2], kx[e + 3]);
                *reinterpret_cast<float4*>(&Vs[lr * MP_P + lc * 8 + e]) = make_float4(vx[e], vx[e + 1], vx[e + 2], vx[e + 3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kk = kl + half * 32;
            if (k0 + kk >= Lk) continue;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {                  // (same order of the 32 additions as before)
                const float4 k4 = *reinterpret_cast<const float4*>(&Ks[kk * MP_P + e]);
                s += qv[e] * k4.x; s += qv[e + 1] * k4.y; s += qv[e + 2] * k4.z; s += qv[e + 3] * k4.w;
            }
            const float mn = fmaxf(m, s);
            const float corr = (m == -INFINITY) ? 0.f : expf(m - mn), pj = expf(s - mn);
            l = l * corr + pj;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                const float4 v4 = *reinterpret_cast<const float4*>(&Vs[kk * MP_P + e]);
                acc[e] = acc[e] * corr + pj * v4.x; acc[e + 1] = acc[e + 1] * corr + pj * v4.y;
                acc[e + 2] = acc[e + 2] * corr + pj * v4.z; acc[e + 3] = acc[e + 3] * corr + pj * v4.w;
            }
            m = mn;
        }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {                                   // merge the 32 key lanes of a query (xor < 32 stays inside the half wave)
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        l = l * c1 + l2 * c2;
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = acc[e] * c1 + __shfl_xor(acc[e], off) * c2;
        m = mn;
    }
    if (kl == 0 && valid) {
        half_t* op = o + ((int64_t)b * o_tok + iq) * o_cp + o_co + h * 32;
        const float inv = 1.f / l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ov[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = acc[c * 8 + e] * inv;
            store8(op + c * 8, o_cp >> 1, ov);
        }
    }
}

// ---- DDColor tail on pairs ----
// fold_queries: M[o][c] = sum_q R[o][q] E[q][c] (fp32 [2][C] per frame), E = the colour embeddings as a pair token view
__global__ void fold_queries_p_kernel(const half_t* __restrict__ e, int e_cp, int e_co, int tok, const float* __restrict__ r, int r_pitch, int nq,
                                      float* __restrict__ out, int C) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const half_t* eb = e + (int64_t)b * tok * e_cp + e_co + c;
    const int lo = e_cp >> 1;
    float a0 = 0.f, a1 = 0.f;
    for (int q = 0; q < nq; ++q) {
        const float v = join_hl(eb[(int64_t)q * e_cp], eb[(int64_t)q * e_cp + lo]);
        a0 = fmaf(r[q], v, a0);
        a1 = fmaf(r[r_pitch + q], v, a1);
    }
    out[((int64_t)b * 2 + 0) * C + c] = a0;
    out[((int64_t)b * 2 + 1) * C + c] = a1;
}

// PixelShuffle(4) + ReplicationPad2d((1,0,1,0)) + AvgPool2d(2, 1) of the last_shuf conv's [Hi][Wi][16 x 256] pair tensor (channels ordered
// (dy*4+dx)*256 + c), the folded einsum + refine projection M[b] (fp32 [2][256]), the image term of the refine conv and its bias:
//   ab[Y][X][o] = sum_c M[b][o][c] * blur(shuffle(x))[Y][X][c] + R_img[o] . img[Y][X] + bias[o]
// One wave per output pixel, lane = 4 channels; window sum row by row like torch's avg_pool2d.
__global__ void shuf4_blur_proj_p_kernel(const half_t* __restrict__ x, int x_cp, int x_co, const float* __restrict__ M, const half_t* __restrict__ img,
                                         int img_cp, int img_co, const float* __restrict__ rimg, const float* __restrict__ bias, half_t* __restrict__ y,
                                         int y_cp, int y_co, int B, int Hi, int Wi) {
    constexpr int C = 256;
    const int lane = threadIdx.x & 63, Ho = Hi * 4, Wo = Wi * 4, xlo = x_cp >> 1;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t i = wave; i < total; i += nwaves) {
        const int X = (int)(i % Wo);
        const int64_t tt = i / Wo;
        const int Y = (int)(tt % Ho), b = (int)(tt / Ho);
        const int y0 = max(Y - 1, 0), x0 = max(X - 1, 0);
        auto ld = [&](int ay, int ax, float v[4]) {
            const half_t* p = x + ((int64_t)(b * Hi + (ay >> 2)) * Wi + (ax >> 2)) * x_cp + x_co + ((ay & 3) * 4 + (ax & 3)) * C + lane * 4;
            const half4 h = *reinterpret_cast<const half4*>(p), l = *reinterpret_cast<const half4*>(p + xlo);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = join_hl(h[r], l[r]);
        };
        float v00[4], v01[4], v10[4], v11[4];
        ld(y0, x0, v00); ld(y0, X, v01); ld(Y, x0, v10); ld(Y, X, v11);
        const float4 m0 = *reinterpret_cast<const float4*>(M + ((int64_t)b * 2 + 0) * C + lane * 4);
        const float4 m1 = *reinterpret_cast<const float4*>(M + ((int64_t)b * 2 + 1) * C + lane * 4);
        const float mm0[4] = {m0.x, m0.y, m0.z, m0.w}, mm1[4] = {m1.x, m1.y, m1.z, m1.w};
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float f = (v00[r] + v01[r] + v10[r] + v11[r]) * 0.25f;
            a0 += mm0[r] * f;
            a1 += mm1[r] * f;
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            const half_t* ip = img + i * img_cp + img_co;
            const int ilo = img_cp >> 1;
            const float i0 = join_hl(ip[0], ip[ilo]), i1 = join_hl(ip[1], ip[ilo + 1]), i2 = join_hl(ip[2], ip[ilo + 2]);
            const float a = a0 + (rimg[0] * i0 + rimg[1] * i1 + rimg[2] * i2) + bias[0];
            const float bb = a1 + (rimg[3] * i0 + rimg[4] * i1 + rimg[5] * i2) + bias[1];
            half_t* yp = y + i * y_cp + y_co;
            const int ylo = y_cp >> 1;
            half_t hh, ll;
            split_hl(a, hh, ll); yp[0] = hh; yp[ylo] = ll;
            split_hl(bb, hh, ll); yp[1] = hh; yp[ylo + 1] = ll;
        }
    }
}

// siggraph17 `x[:, :, ::2, ::2]` on pairs: both planes of the 8-channel chunk
__global__ void subsample2_p_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Ho, int Wo, int Hi, int Wi, int C8, int x_cp, int x_co,
                                    int y_cp, int y_co) {
    const int64_t total = (int64_t)B * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t pix = i / C8;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        const half_t* s = x + ((int64_t)(b * Hi + 2 * ho) * Wi + 2 * wo) * x_cp + x_co + c8 * 8;
        half_t* d = y + ((int64_t)(b * Ho + ho) * Wo + wo) * y_cp + y_co + c8 * 8;
        *reinterpret_cast<half8*>(d) = *reinterpret_cast<const half8*>(s);
        *reinterpret_cast<half8*>(d + (y_cp >> 1)) = *reinterpret_cast<const half8*>(s + (x_cp >> 1));
    }
}

}  // namespace

int launch_proj2_p(const half_t* x, int x_cpitch, int x_coff, int C, const float* w, const float* bias, int mode, float mul, float* out, int64_t npix,
                   hipStream_t s) {
    if (C > 512) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(proj2_p_kernel, dim3(grid_for(npix * 64)), dim3(256), 0, s, x, x_cpitch, x_coff, C, w, bias, mode, mul, out, npix);
    return (int)hipGetLastError();
}

int launch_layernorm_p(const half_t* x, half_t* y, const float* gamma, const float* beta, float eps, int64_t npix, int C, int x_cpitch, int x_coff,
                       int y_cpitch, int y_coff, int relu, hipStream_t s) {
    if ((C & 7) || C > 2048) return (int)hipErrorInvalidValue;
    const int C8 = C / 8;
    const dim3 grid(grid_for(npix * 64)), block(256);
    if (C8 <= 64) hipLaunchKernelGGL(layernorm_p_kernel<1>, grid, block, 0, s, x, y, gamma, beta, eps, npix, C8, x_cpitch, x_coff, y_cpitch, y_coff, relu);
    else if (C8 <= 128) hipLaunchKernelGGL(layernorm_p_kernel<2>, grid, block, 0, s, x, y, gamma, beta, eps, npix, C8, x_cpitch, x_coff, y_cpitch, y_coff, relu);
    else hipLaunchKernelGGL(layernorm_p_kernel<4>, grid, block, 0, s, x, y, gamma, beta, eps, npix, C8, x_cpitch, x_coff, y_cpitch, y_coff, relu);
    return (int)hipGetLastError();
}

int launch_dwconv7_p(const half_t* x, const float* w, const float* bias, half_t* y, int B, int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch,
                     int y_coff, int w_pitch, hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(dwconv7_p_kernel, dim3(grid_for((int64_t)B * H * W * C8)), dim3(256), 0, s, x, w, bias, y, B, H, W, C8, x_cpitch, x_coff, y_cpitch,
                       y_coff, w_pitch);
    return (int)hipGetLastError();
}

int launch_mha32_p(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok, half_t* o,
                   int o_cpitch, int o_coff, int o_tok, int B, int heads, int Lq, int Lk, float scale, hipStream_t s) {
    if (Lq < 1 || Lk < 1 || B < 1 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mha32_p_kernel, dim3((Lq + MP_QG - 1) / MP_QG, heads, B), dim3(256), 0, s, q, q_cpitch, q_coff, q_tok, kv, kv_cpitch, k_coff, v_coff,
                       kv_tok, o, o_cpitch, o_coff, o_tok, heads, Lq, Lk, scale);
    return (int)hipGetLastError();
}

int launch_fold_queries_p(const half_t* e, int e_cpitch, int e_coff, int tok, const float* r, int r_pitch, int nq, float* out, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(fold_queries_p_kernel, dim3((C + 63) / 64, B), dim3(64), 0, s, e, e_cpitch, e_coff, tok, r, r_pitch, nq, out, C);
    return (int)hipGetLastError();
}

int launch_shuf4_blur_proj_p(const half_t* x, int x_cpitch, int x_coff, const float* M, const half_t* img, int img_cpitch, int img_coff, const float* rimg,
                             const float* bias, half_t* y, int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s) {
    hipLaunchKernelGGL(shuf4_blur_proj_p_kernel, dim3(grid_for((int64_t)B * Hi * 4 * Wi * 4 * 64)), dim3(256), 0, s, x, x_cpitch, x_coff, M, img, img_cpitch,
                       img_coff, rimg, bias, y, y_cpitch, y_coff, B, Hi, Wi);
    return (int)hipGetLastError();
}

int launch_subsample2_p(const half_t* x, half_t* y, int B, int Ho, int Wo, int Hi, int Wi, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                        hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(subsample2_p_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C8)), dim3(256), 0, s, x, y, B, Ho, Wo, Hi, Wi, C8, x_cpitch, x_coff, y_cpitch,
                       y_coff);
    return (int)hipGetLastError();
}

void preload_precise2() {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(mha32_p_kernel));
    (void)hipGetLastError();
}
