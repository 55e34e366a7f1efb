// "Precise" mode (HAVC_F_PRECISE, include/havc_mi355.h): the non-conv ops of the DeOldify generators on hi / lo fp16 pairs.
//
// The reference computes in fp32 end to end (deoldify/filters.py:45-68, fastai/basic_train.py:352-363).  The fast path of this library
// rounds every activation to fp16; this path keeps 22 significand bits per value as two fp16 numbers (hi = fp16(v),
// lo = fp16((v - hi) * 2^11)) so that the convolutions still run on the fp16 MFMA main loop (three K segments, conv_common.h) while
// everything else -- the kernels below -- reads a pair as ONE fp32 value, computes in fp32 like torch and stores a pair again.
// Layout: a precise buffer's pixel row is [hi: P channels | lo: P channels]; a view's lo plane is cpitch / 2 elements behind its hi plane.
// All kernels are HBM-bound except the attention, which is fp32 VALU (the N x N map of fastai's SelfAttention never leaves the CU).
#include "conv_common.h"
#include <atomic>
#include <cstdlib>

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

// ---- model input (prep_rgb8_kernel of elementwise.hip in fp32): u8 RGB -> PIL 'L' -> / 255 -> imagenet normalise ----
__global__ void prep_rgb8_p_kernel(const uint8_t* __restrict__ rgb, half_t* __restrict__ y0, int y0_cp, int y0_co, half_t* __restrict__ y1, int y1_cp,
                                   int y1_co, int64_t npix) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        const unsigned L = (19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16;
        const float f = (float)L / 255.f;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (f - mean[c]) / stdv[c];
        store8(y0 + i * y0_cp + y0_co, y0_cp >> 1, v);
        if (y1) store8(y1 + i * y1_cp + y1_co, y1_cp >> 1, v);
    }
}

// ---- MaxPool2d(3, stride 2, pad 1) ----
__global__ void maxpool_p_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo, int C8, int x_cp, int x_co,
                                 int y_cp, int y_co) {
    const int64_t total = (int64_t)B * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t pix = i / C8;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        for (int dy = 0; dy < 3; ++dy) {
            const int hi = ho * 2 - 1 + dy;
            if ((unsigned)hi >= (unsigned)Hi) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int wi = wo * 2 - 1 + dx;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                float v[8];
                load8(x + ((int64_t)(b * Hi + hi) * Wi + wi) * x_cp + x_co + c8 * 8, x_cp >> 1, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        store8(y + ((int64_t)(b * Ho + ho) * Wo + wo) * y_cp + y_co + c8 * 8, y_cp >> 1, m);
    }
}

// ---- ReplicationPad2d((1,0,1,0)) + AvgPool2d(2, stride 1) [+ nearest resize]: blur_resize_kernel of elementwise.hip in fp32 ----
// torch's avg_pool2d sums the window row by row and divides by the window size: ((v00 + v01) + v10) + v11, then / 4.
__global__ void blur_resize_p_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo, int C8, int x_cp, int x_co,
                                     int y_cp, int y_co, float sh, float sw) {
    const int64_t total = (int64_t)B * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        int64_t pix = i / C8;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        int sy = ho, sx = wo;
        if (Ho != Hi) sy = min((int)floorf(ho * sh), Hi - 1);
        if (Wo != Wi) sx = min((int)floorf(wo * sw), Wi - 1);
        const int y0 = max(sy - 1, 0), x0 = max(sx - 1, 0);
        const half_t* base = x + (int64_t)b * Hi * Wi * x_cp + x_co + c8 * 8;
        const int lo = x_cp >> 1;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        load8(base + ((int64_t)y0 * Wi + x0) * x_cp, lo, v00);
        load8(base + ((int64_t)y0 * Wi + sx) * x_cp, lo, v01);
        load8(base + ((int64_t)sy * Wi + x0) * x_cp, lo, v10);
        load8(base + ((int64_t)sy * Wi + sx) * x_cp, lo, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v00[e] + v01[e] + v10[e] + v11[e]) * 0.25f;
        store8(y + ((int64_t)(b * Ho + ho) * Wo + wo) * y_cp + y_co + c8 * 8, y_cp >> 1, o);
    }
}

// ---- y = [relu](x * scale + shift) ----
__global__ void affine_p_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                int64_t npix, int C8, int x_cp, int x_co, int y_cp, int y_co) {
    const int64_t total = npix * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const int64_t pix = i / C8;
        float v[8];
        load8(x + pix * x_cp + x_co + c8 * 8, x_cp >> 1, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (scale) v[e] = v[e] * scale[c8 * 8 + e] + shift[c8 * 8 + e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        store8(y + pix * y_cp + y_co + c8 * 8, y_cp >> 1, v);
    }
}

// ---- fastai SelfAttention in fp32 (fastai/layers.py:81-96): beta = softmax_i(f_i . g_j), o_j = gamma * sum_i beta_ij h_i + x_j ----
// Two passes over the keys, both with the N x N map kept on the CU:
//   pattn_stats_kernel: per query j the row maximum m_j and l_j = sum_i exp(s_ij - m_j)  (torch's softmax: exp(x - max) / sum)
//   pattn_apply_kernel: s_ij again, p = exp(s_ij - m_j) / l_j, o_j += p h_i for a chunk of CC channels, epilogue gamma o + x.
// Block = 64 queries x 256 threads; key tiles of 32.  S role: thread (query t & 63, key group t >> 6) = 8 keys; PV role: thread
// (channel lane t & 31, query group t >> 5) = 8 queries x NV float4 of channels (cg * 4 + k * 128: consecutive lanes read consecutive
// 16-byte LDS slots, the p values are wave-uniform broadcasts).  Sums run in a fixed order (d, then keys ascending): deterministic.
struct PAttnArgs {
    const half_t* qk; const half_t* h; const half_t* x; half_t* out; float* stats;
    int qk_cp, f_co, g_co, d, h_cp, h_co, x_cp, x_co, o_cp, o_co, B, N, C;
    int64_t qk_fs, h_fs, x_fs, o_fs;     // frame strides (elements)
    float gamma;
};
constexpr int PA_TJ = 64, PA_TI = 32;

__device__ __forceinline__ void pa_load_rows(const half_t* base, int cp, int co, int row0, int rows, int N, int nch8, float* dst, int dpitch, int t) {
    // rows x nch8 chunks of 8 channels (hi + lo) -> fp32 LDS rows of pitch dpitch; rows beyond N are zeros
    for (int q = t; q < rows * nch8; q += 256) {
        const int r = q / nch8, ch = q - r * nch8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (row0 + r < N) load8(base + (int64_t)(row0 + r) * cp + co + ch * 8, cp >> 1, v);
        float4* o = reinterpret_cast<float4*>(dst + r * dpitch + ch * 8);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

__device__ __forceinline__ void pa_scores(const float* Gs, const float* Fs, int DP, int d, int sj, int ig, float s[8]) {
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) s[ii] = 0.f;
    for (int dd = 0; dd < d; dd += 4) {
        const float4 g = *reinterpret_cast<const float4*>(Gs + sj * DP + dd);
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const float4 f = *reinterpret_cast<const float4*>(Fs + (ig * 8 + ii) * DP + dd);
            s[ii] += f.x * g.x; s[ii] += f.y * g.y; s[ii] += f.z * g.z; s[ii] += f.w * g.w;
        }
    }
}

__global__ void __launch_bounds__(256) pattn_stats_kernel(const PAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DP = a.d + 4;
    float* Gs = sm;
    float* Fs = Gs + PA_TJ * DP;
    float* red = Fs + PA_TI * DP;                 // [4][64][2]
    const int j0 = blockIdx.x * PA_TJ, b = blockIdx.y, t = threadIdx.x, sj = t & 63, ig = t >> 6;
    const half_t* qk = a.qk + (int64_t)b * a.qk_fs;
    pa_load_rows(qk, a.qk_cp, a.g_co, j0, PA_TJ, a.N, a.d / 8, Gs, DP, t);
    float m = -3.0e38f, l = 0.f;
    for (int i0 = 0; i0 < a.N; i0 += PA_TI) {
        __syncthreads();
        pa_load_rows(qk, a.qk_cp, a.f_co, i0, PA_TI, a.N, a.d / 8, Fs, DP, t);
        __syncthreads();
        float s[8];
        pa_scores(Gs, Fs, DP, a.d, sj, ig, s);
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            if (i0 + ig * 8 + ii >= a.N) continue;
            if (s[ii] > m) { l = l * expf(m - s[ii]) + 1.f; m = s[ii]; }
            else l += expf(s[ii] - m);
        }
    }
    red[(ig * 64 + sj) * 2] = m;
    red[(ig * 64 + sj) * 2 + 1] = l;
    __syncthreads();
    if (t < 64 && j0 + t < a.N) {
        float M = red[t * 2];
#pragma unroll
        for (int k = 1; k < 4; ++k) M = fmaxf(M, red[(k * 64 + t) * 2]);
        float L = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) L += red[(k * 64 + t) * 2 + 1] * expf(red[(k * 64 + t) * 2] - M);
        float* st = a.stats + ((int64_t)b * a.N + j0 + t) * 2;
        st[0] = M;
        st[1] = L;
    }
}

template <int NV>
__global__ void __launch_bounds__(256) pattn_apply_kernel(const PAttnArgs a) {
    constexpr int CC = NV * 128;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DP = a.d + 4;
    float* Gs = sm;
    float* Fs = Gs + PA_TJ * DP;
    float* Ps = Fs + PA_TI * DP;                  // [TI][TJ]
    float* Hs = Ps + PA_TI * PA_TJ;               // [TI][CC]
    const int j0 = blockIdx.x * PA_TJ, b = blockIdx.y, c0 = blockIdx.z * CC, t = threadIdx.x;
    const int sj = t & 63, ig = t >> 6, cg = t & 31, jg = t >> 5;
    const half_t* qk = a.qk + (int64_t)b * a.qk_fs;
    const half_t* hb = a.h + (int64_t)b * a.h_fs;
    pa_load_rows(qk, a.qk_cp, a.g_co, j0, PA_TJ, a.N, a.d / 8, Gs, DP, t);
    float m = 0.f, l = 1.f;
    if (j0 + sj < a.N) {
        const float* st = a.stats + ((int64_t)b * a.N + j0 + sj) * 2;
        m = st[0];
        l = st[1];
    }
    float4 acc[8][NV];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[jj][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0; i0 < a.N; i0 += PA_TI) {
        __syncthreads();
        pa_load_rows(qk, a.qk_cp, a.f_co, i0, PA_TI, a.N, a.d / 8, Fs, DP, t);
        pa_load_rows(hb, a.h_cp, a.h_co + c0, i0, PA_TI, a.N, CC / 8, Hs, CC, t);
        __syncthreads();
        float s[8];
        pa_scores(Gs, Fs, DP, a.d, sj, ig, s);
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) Ps[(ig * 8 + ii) * PA_TJ + sj] = (i0 + ig * 8 + ii < a.N) ? expf(s[ii] - m) / l : 0.f;
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < PA_TI; ++i) {
            const float4 p0 = *reinterpret_cast<const float4*>(Ps + i * PA_TJ + jg * 8), p1 = *reinterpret_cast<const float4*>(Ps + i * PA_TJ + jg * 8 + 4);
            const float pj[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 hv = *reinterpret_cast<const float4*>(Hs + i * CC + k * 128 + cg * 4);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    acc[jj][k].x += pj[jj] * hv.x; acc[jj][k].y += pj[jj] * hv.y;
                    acc[jj][k].z += pj[jj] * hv.z; acc[jj][k].w += pj[jj] * hv.w;
                }
            }
        }
    }
    // epilogue: out = gamma * o + x, as a hi / lo pair (4 consecutive channels per store)
    const half_t* xb = a.x + (int64_t)b * a.x_fs;
    half_t* ob = a.out + (int64_t)b * a.o_fs;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int j = j0 + jg * 8 + jj;
        if (j >= a.N) continue;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = c0 + k * 128 + cg * 4;
            const half_t* xp = xb + (int64_t)j * a.x_cp + a.x_co + c;
            const half4 xh = *reinterpret_cast<const half4*>(xp), xl = *reinterpret_cast<const half4*>(xp + (a.x_cp >> 1));
            const float o4[4] = {acc[jj][k].x, acc[jj][k].y, acc[jj][k].z, acc[jj][k].w};
            half4 oh, ol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = a.gamma * o4[r] + join_hl(xh[r], xl[r]);
                half_t hh, ll;
                split_hl(v, hh, ll);
                oh[r] = hh; ol[r] = ll;
            }
            half_t* op = ob + (int64_t)j * a.o_cp + a.o_co + c;
            *reinterpret_cast<half4*>(op) = oh;
            *reinterpret_cast<half4*>(op + (a.o_cp >> 1)) = ol;
        }
    }
}

// ---- the apply pass on the matrix pipe (round 5) ----
// pattn_apply_kernel spends 11.3 ms per 16 frames of the 560 x 560 generator on fp32 FMAs (39 TFLOP/s: 13 % of a precise pass).  The P . H product
// (94 % of the attention's work) runs here on v_mfma_f32_16x16x32_f16 with the SAME three-term splitting the precise convolutions use, so it stays
// fp32-class:   p' = 512 p = hi' + lo'' / 2048 (fp16 pair),  h = h_hi + h_lo' / 2048 (the value map's pair),
//   acc = (64 hi') (32 h_hi) + hi' h_lo' + lo'' h_hi = 2048 (hi' h_hi + (hi' h_lo + lo h_hi))  =  2048 * 512 * p h  up to the lo x lo term (2^-22),
// every product of two fp16 numbers exact in fp32, fp32 accumulation; the factors 64 / 32 / 512 are powers of two chosen so that nothing overflows
// (64 * 512 p <= 32768, 32 |h| < 65504 for |h| < 2047) and hi' is a normal fp16 number down to p = 1.2e-7.  S = f . g and the softmax stay fp32
// VALU exactly as in the kernels above (same statistics pass).  Operands: O^T[c][j] = sum_i H^T[c][i] P[i][j] -- A = the value map TRANSPOSED
// ([C][npitch] planes, hi then lo, written by the value conv's transposed precise epilogue), B = P[query][key]; a lane of the accumulator owns 4
// consecutive channels of one query.  Block = 64 queries x 256 channels x key tiles of 32, 4 waves (wave w: channels 64 w .. +63), ~70 KiB of LDS:
// two blocks per CU cover each other's load latency.
constexpr int PM_CC = 256;
struct PAttnMArgs {
    const half_t* qk; const half_t* vT; const half_t* x; half_t* out; const float* stats;
    int qk_cp, f_co, g_co, d, npitch, x_cp, x_co, o_cp, o_co, N, C;
    int64_t qk_fs, v_fs, x_fs, o_fs;     // frame strides (elements); v_fs covers both planes of a frame: [2][C][npitch]
    float gamma;
};

__global__ void __launch_bounds__(256, 2) pattn_apply_mfma_kernel(const PAttnMArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int DP = a.d + 4;
    float* Gs = reinterpret_cast<float*>(smc);
    float* Fs = Gs + PA_TJ * DP;
    char* P3 = reinterpret_cast<char*>(Fs + PA_TI * DP);                 // [3][64 queries][32 keys] fp16: 2048-scaled hi', hi', lo''
    char* Hh = P3 + 3 * PA_TJ * 64;                                         // [256 channels][32 keys] fp16
    char* Hl = Hh + PM_CC * 64;
    const int j0 = blockIdx.x * PA_TJ, b = blockIdx.y, c0 = blockIdx.z * PM_CC, t = threadIdx.x;
    const int sj = t & 63, ig = t >> 6, lane = t & 63, wave = t >> 6, lr = lane & 15, lg = lane >> 4;
    const half_t* qk = a.qk + (int64_t)b * a.qk_fs;
    const half_t* vh = a.vT + (int64_t)b * a.v_fs + (int64_t)c0 * a.npitch;
    const half_t* vl = vh + (int64_t)a.C * a.npitch;
    pa_load_rows(qk, a.qk_cp, a.g_co, j0, PA_TJ, a.N, a.d / 8, Gs, DP, t);
    float m = 0.f, l = 1.f;
    if (j0 + sj < a.N) {
        const float* st = a.stats + ((int64_t)b * a.N + j0 + sj) * 2;
        m = st[0];
        l = st[1];
    }
    const float pl = 512.f / l;
    float4v acc[4][4];
#pragma unroll
    for (int cf = 0; cf < 4; ++cf)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[cf][q] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < a.N; i0 += PA_TI) {
        __syncthreads();
        pa_load_rows(qk, a.qk_cp, a.f_co, i0, PA_TI, a.N, a.d / 8, Fs, DP, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                        // 256 channel rows x 4 chunks of 8 keys per plane
            const int q = t + k * 256, row = q >> 2, ch = q & 3;
            *reinterpret_cast<half8*>(Hh + row * 64 + ch * 16) = *reinterpret_cast<const half8*>(vh + (int64_t)row * a.npitch + i0 + ch * 8);
            *reinterpret_cast<half8*>(Hl + row * 64 + ch * 16) = *reinterpret_cast<const half8*>(vl + (int64_t)row * a.npitch + i0 + ch * 8);
        }
        __syncthreads();
        {
            float s[8];
            pa_scores(Gs, Fs, DP, a.d, sj, ig, s);
            half8 s0, s1, s2;
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const float pv = (i0 + ig * 8 + ii < a.N) ? expf(s[ii] - m) * pl : 0.f;       // 512 p
                const half_t hi = (half_t)pv;
                s1[ii] = hi;
                s0[ii] = (half_t)((float)hi * 64.f);
                s2[ii] = (half_t)((pv - (float)hi) * 2048.f);
            }
            *reinterpret_cast<half8*>(P3 + (0 * PA_TJ + sj) * 64 + ig * 16) = s0;
            *reinterpret_cast<half8*>(P3 + (1 * PA_TJ + sj) * 64 + ig * 16) = s1;
            *reinterpret_cast<half8*>(P3 + (2 * PA_TJ + sj) * 64 + ig * 16) = s2;
        }
        __syncthreads();
#pragma unroll
        for (int seg = 0; seg < 3; ++seg) {
            half8 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const half8*>(P3 + (seg * PA_TJ + q * 16 + lr) * 64 + lg * 16);
            const char* hp = (seg == 1) ? Hl : Hh;
#pragma unroll
            for (int cf = 0; cf < 4; ++cf) {
                half8 av = *reinterpret_cast<const half8*>(hp + (wave * 64 + cf * 16 + lr) * 64 + lg * 16);
                if (seg == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = av[e] * (half_t)32.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[cf][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq[q], acc[cf][q], 0, 0, 0);
            }
        }
    }
    // epilogue: out = gamma * o + x as a hi / lo pair; lane = query q * 16 + lr, channels c0 + wave * 64 + cf * 16 + lg * 4 .. +3
    const half_t* xb = a.x + (int64_t)b * a.x_fs;
    half_t* ob = a.out + (int64_t)b * a.o_fs;
    const float sc = a.gamma * (1.f / (2048.f * 512.f));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q * 16 + lr;
        if (j >= a.N) continue;
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
            const int c = c0 + wave * 64 + cf * 16 + lg * 4;
            const half_t* xp = xb + (int64_t)j * a.x_cp + a.x_co + c;
            const half4 xh = *reinterpret_cast<const half4*>(xp), xl = *reinterpret_cast<const half4*>(xp + (a.x_cp >> 1));
            half4 oh, ol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = sc * acc[cf][q][r] + join_hl(xh[r], xl[r]);
                half_t hh, ll;
                split_hl(v, hh, ll);
                oh[r] = hh; ol[r] = ll;
            }
            half_t* op = ob + (int64_t)j * a.o_cp + a.o_co + c;
            *reinterpret_cast<half4*>(op) = oh;
            *reinterpret_cast<half4*>(op + (a.o_cp >> 1)) = ol;
        }
    }
}

// ---- S = f . g on the matrix pipe as well (round 5, second step) ----
// With P . H on MFMA the two fp32-VALU evaluations of S (statistics pass + apply pass: 2 x 2 N^2 d FLOP) were half of what was left of the attention.
// f and g are fp16 pairs of the same NHWC buffer, channels contiguous: a pixel's 8-channel chunk IS a 16x16x32 operand fragment (row = pixel, K = channel),
// no transposition.  Three-term splitting with the scales on both sides:  acc = (32 f_hi)(64 g_hi) + f_hi g_lo' + f_lo' g_hi = 2048 (f . g)  (|f| < 2047,
// |g| < 1023).  D[key][query]: a lane owns 4 consecutive keys of one query -- its online-softmax state in the statistics kernel, an 8-byte piece of the
// P operand in the apply kernel.  Wave w of a block owns queries 16 w .. 16 w + 15; its g fragments stay in registers for the whole block.
template <int D32>
struct PGFrag { half8 hi64[D32], lo[D32], hi[D32]; };

template <int D32>
__device__ __forceinline__ void pm_load_g(const half_t* qk, int qk_cp, int g_co, int row, PGFrag<D32>& g, int lg) {
    const half_t* base = qk + (int64_t)row * qk_cp + g_co + lg * 8;
#pragma unroll
    for (int kk = 0; kk < D32; ++kk) {
        g.hi[kk] = *reinterpret_cast<const half8*>(base + kk * 32);
        g.lo[kk] = *reinterpret_cast<const half8*>(base + (qk_cp >> 1) + kk * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) g.hi64[kk][e] = g.hi[kk][e] * (half_t)64.f;
    }
}

// s[kf][r] = f(key i0 + kf * 16 + lg * 4 + r) . g(query of this lane's column); keys >= N give 0 (the caller masks them)
template <int D32>
__device__ __forceinline__ void pm_scores(const half_t* qk, int qk_cp, int f_co, int i0, int N, const PGFrag<D32>& g, int lr, int lg, float s[2][4]) {
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        const int row = i0 + kf * 16 + lr;
        const bool ok = row < N;
        const half_t* base = qk + (int64_t)(ok ? row : 0) * qk_cp + f_co + lg * 8;
        float4v acc = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < D32; ++kk) {
            half8 fh = *reinterpret_cast<const half8*>(base + kk * 32), fl = *reinterpret_cast<const half8*>(base + (qk_cp >> 1) + kk * 32);
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) fh[e] = fl[e] = (half_t)0.f;
            }
            half8 f32;
#pragma unroll
            for (int e = 0; e < 8; ++e) f32[e] = fh[e] * (half_t)32.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f32, g.hi64[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, g.lo[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, g.hi[kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kf][r] = acc[r] * (1.f / 2048.f);
    }
}

template <int D32>
__global__ void __launch_bounds__(256) pattn_stats_mfma_kernel(const PAttnArgs a) {
    const int j0 = blockIdx.x * PA_TJ, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lg = lane >> 4;
    const half_t* qk = a.qk + (int64_t)b * a.qk_fs;
    const int jq = j0 + wave * 16 + lr;
    PGFrag<D32> g;
    pm_load_g<D32>(qk, a.qk_cp, a.g_co, jq < a.N ? jq : a.N - 1, g, lg);
    float m = -3.0e38f, l = 0.f;
    for (int i0 = 0; i0 < a.N; i0 += PA_TI) {
        float s[2][4];
        pm_scores<D32>(qk, a.qk_cp, a.f_co, i0, a.N, g, lr, lg, s);
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (i0 + kf * 16 + lg * 4 + r >= a.N) continue;
                const float v = s[kf][r];
                if (v > m) { l = l * expf(m - v) + 1.f; m = v; }
                else l += expf(v - m);
            }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {               // the four key groups (lg) of a query
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float M = fmaxf(m, m2);
        l = l * expf(m - M) + l2 * expf(m2 - M);
        m = M;
    }
    if (lg == 0 && jq < a.N) {
        float* st = a.stats + ((int64_t)b * a.N + jq) * 2;
        st[0] = m;
        st[1] = l;
    }
}

template <int D32>
__global__ void __launch_bounds__(256, 2) pattn_apply_mfma2_kernel(const PAttnMArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    char* P3 = smc;                                                        // [3][64 queries][32 keys] fp16: 64-scaled hi', hi', lo''
    char* Hh = P3 + 3 * PA_TJ * 64;                                         // [256 channels][32 keys] fp16
    char* Hl = Hh + PM_CC * 64;
    const int j0 = blockIdx.x * PA_TJ, b = blockIdx.y, c0 = blockIdx.z * PM_CC, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, lr = lane & 15, lg = lane >> 4;
    const half_t* qk = a.qk + (int64_t)b * a.qk_fs;
    const half_t* vh = a.vT + (int64_t)b * a.v_fs + (int64_t)c0 * a.npitch;
    const half_t* vl = vh + (int64_t)a.C * a.npitch;
    const int jq = j0 + wave * 16 + lr;
    PGFrag<D32> g;
    pm_load_g<D32>(qk, a.qk_cp, a.g_co, jq < a.N ? jq : a.N - 1, g, lg);
    float m = 0.f, l = 1.f;
    if (jq < a.N) {
        const float* st = a.stats + ((int64_t)b * a.N + jq) * 2;
        m = st[0];
        l = st[1];
    }
    const float pl = 512.f / l;
    float4v acc[4][4];
#pragma unroll
    for (int cf = 0; cf < 4; ++cf)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[cf][q] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < a.N; i0 += PA_TI) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                        // 256 channel rows x 4 chunks of 8 keys per plane
            const int q = t + k * 256, row = q >> 2, ch = q & 3;
            *reinterpret_cast<half8*>(Hh + row * 64 + ch * 16) = *reinterpret_cast<const half8*>(vh + (int64_t)row * a.npitch + i0 + ch * 8);
            *reinterpret_cast<half8*>(Hl + row * 64 + ch * 16) = *reinterpret_cast<const half8*>(vl + (int64_t)row * a.npitch + i0 + ch * 8);
        }
        {
            float s[2][4];
            pm_scores<D32>(qk, a.qk_cp, a.f_co, i0, a.N, g, lr, lg, s);
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                half4 s0, s1, s2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = (i0 + kf * 16 + lg * 4 + r < a.N) ? expf(s[kf][r] - m) * pl : 0.f;       // 512 p
                    const half_t hi = (half_t)pv;
                    s1[r] = hi;
                    s0[r] = (half_t)((float)hi * 64.f);
                    s2[r] = (half_t)((pv - (float)hi) * 2048.f);
                }
                const int off = (wave * 16 + lr) * 64 + (kf * 16 + lg * 4) * 2;
                *reinterpret_cast<half4*>(P3 + 0 * PA_TJ * 64 + off) = s0;
                *reinterpret_cast<half4*>(P3 + 1 * PA_TJ * 64 + off) = s1;
                *reinterpret_cast<half4*>(P3 + 2 * PA_TJ * 64 + off) = s2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int seg = 0; seg < 3; ++seg) {
            half8 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const half8*>(P3 + (seg * PA_TJ + q * 16 + lr) * 64 + lg * 16);
            const char* hp = (seg == 1) ? Hl : Hh;
#pragma unroll
            for (int cf = 0; cf < 4; ++cf) {
                half8 av = *reinterpret_cast<const half8*>(hp + (wave * 64 + cf * 16 + lr) * 64 + lg * 16);
                if (seg == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = av[e] * (half_t)32.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[cf][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq[q], acc[cf][q], 0, 0, 0);
            }
        }
    }
    const half_t* xb = a.x + (int64_t)b * a.x_fs;
    half_t* ob = a.out + (int64_t)b * a.o_fs;
    const float sc = a.gamma * (1.f / (2048.f * 512.f));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q * 16 + lr;
        if (j >= a.N) continue;
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
            const int c = c0 + wave * 64 + cf * 16 + lg * 4;
            const half_t* xp = xb + (int64_t)j * a.x_cp + a.x_co + c;
            const half4 xh = *reinterpret_cast<const half4*>(xp), xl = *reinterpret_cast<const half4*>(xp + (a.x_cp >> 1));
            half4 oh, ol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = sc * acc[cf][q][r] + join_hl(xh[r], xl[r]);
                half_t hh, ll;
                split_hl(v, hh, ll);
                oh[r] = hh; ol[r] = ll;
            }
            half_t* op = ob + (int64_t)j * a.o_cp + a.o_co + c;
            *reinterpret_cast<half4*>(op) = oh;
            *reinterpret_cast<half4*>(op + (a.o_cp >> 1)) = ol;
        }
    }
}

template <auto Kernel>
void lds_optin() {                                         // > 64 KiB of dynamic LDS: once per kernel and device (eagerly: preload_precise)
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.fetch_or(bit, std::memory_order_release);
}

}  // namespace

void preload_precise() {
    lds_optin<pattn_stats_kernel>(); lds_optin<pattn_apply_kernel<4>>(); lds_optin<pattn_apply_kernel<2>>(); lds_optin<pattn_apply_kernel<1>>();
    lds_optin<pattn_apply_mfma_kernel>();
    (void)hipGetLastError();
}

int launch_prep_rgb8_p(const uint8_t* rgb, half_t* y0, int y0_cpitch, int y0_coff, half_t* y1, int y1_cpitch, int y1_coff, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(prep_rgb8_p_kernel, dim3(grid_for(npix)), dim3(256), 0, s, rgb, y0, y0_cpitch, y0_coff, y1, y1_cpitch, y1_coff, npix);
    return (int)hipGetLastError();
}

int launch_maxpool3x3s2_p(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                          hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(maxpool_p_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C8)), dim3(256), 0, s, x, y, B, Hi, Wi, Ho, Wo, C8, x_cpitch, x_coff, y_cpitch, y_coff);
    return (int)hipGetLastError();
}

int launch_blur_resize_p(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                         hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(blur_resize_p_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C8)), dim3(256), 0, s, x, y, B, Hi, Wi, Ho, Wo, C8, x_cpitch, x_coff, y_cpitch,
                       y_coff, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return (int)hipGetLastError();
}

int launch_affine_p(const half_t* x, half_t* y, const float* scale, const float* shift, int relu, int64_t npix, int C, int x_cpitch, int x_coff, int y_cpitch,
                    int y_coff, hipStream_t s) {
    const int C8 = C / 8;
    hipLaunchKernelGGL(affine_p_kernel, dim3(grid_for(npix * C8)), dim3(256), 0, s, x, y, scale, shift, relu, npix, C8, x_cpitch, x_coff, y_cpitch, y_coff);
    return (int)hipGetLastError();
}

bool attention_p_supported(int d, int C) { return d >= 8 && d <= 128 && (d & 7) == 0 && C >= 128 && (C & 127) == 0; }

// qk: view holding f at f_coff and g at g_coff (d channels each); h: value view (C channels); x / out: C channels; stats: fp32 [B][N][2] scratch
int launch_attention_p(const half_t* qk, int qk_cpitch, int f_coff, int g_coff, int d, int64_t qk_fs, const half_t* h, int h_cpitch, int h_coff, int64_t h_fs,
                       const half_t* x, int x_cpitch, int x_coff, int64_t x_fs, half_t* out, int o_cpitch, int o_coff, int64_t o_fs, float* stats, int B, int N,
                       int C, float gamma, hipStream_t s) {
    if (!attention_p_supported(d, C)) return (int)hipErrorInvalidValue;
    PAttnArgs a{};
    a.qk = qk; a.h = h; a.x = x; a.out = out; a.stats = stats;
    a.qk_cp = qk_cpitch; a.f_co = f_coff; a.g_co = g_coff; a.d = d; a.h_cp = h_cpitch; a.h_co = h_coff; a.x_cp = x_cpitch; a.x_co = x_coff;
    a.o_cp = o_cpitch; a.o_co = o_coff; a.B = B; a.N = N; a.C = C; a.qk_fs = qk_fs; a.h_fs = h_fs; a.x_fs = x_fs; a.o_fs = o_fs; a.gamma = gamma;
    const int DP = d + 4, NJ = (N + PA_TJ - 1) / PA_TJ;
    const int lds_a = (PA_TJ * DP + PA_TI * DP + 4 * 64 * 2) * 4;
    lds_optin<pattn_stats_kernel>();
    hipLaunchKernelGGL(pattn_stats_kernel, dim3(NJ, B), dim3(256), lds_a, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int CC = (C % 512 == 0) ? 512 : ((C % 256 == 0) ? 256 : 128);
    const int lds_b = (PA_TJ * DP + PA_TI * DP + PA_TI * PA_TJ + PA_TI * CC) * 4;
    const dim3 grid(NJ, B, C / CC);
    if (CC == 512) { lds_optin<pattn_apply_kernel<4>>(); hipLaunchKernelGGL(pattn_apply_kernel<4>, grid, dim3(256), lds_b, s, a); }
    else if (CC == 256) { lds_optin<pattn_apply_kernel<2>>(); hipLaunchKernelGGL(pattn_apply_kernel<2>, grid, dim3(256), lds_b, s, a); }
    else { lds_optin<pattn_apply_kernel<1>>(); hipLaunchKernelGGL(pattn_apply_kernel<1>, grid, dim3(256), lds_b, s, a); }
    return (int)hipGetLastError();
}

bool attention_pm_supported(int d, int C, int npitch) { return d >= 8 && d <= 128 && (d & 7) == 0 && C >= PM_CC && (C % PM_CC) == 0 && npitch >= 32 && (npitch & 31) == 0; }

// The same attention with the value map given TRANSPOSED as two planes per frame, [2][C][npitch] fp16 (hi plane, lo plane; npitch >= N rounded up to a
// multiple of 32, columns beyond N zero): statistics pass as above, apply pass on MFMA (pattn_apply_mfma_kernel).
int launch_attention_pm(const half_t* qk, int qk_cpitch, int f_coff, int g_coff, int d, int64_t qk_fs, const half_t* vT, int npitch, int64_t v_fs,
                        const half_t* x, int x_cpitch, int x_coff, int64_t x_fs, half_t* out, int o_cpitch, int o_coff, int64_t o_fs, float* stats, int B, int N,
                        int C, float gamma, hipStream_t s) {
    if (!attention_pm_supported(d, C, npitch) || npitch < ((N + 31) & ~31)) return (int)hipErrorInvalidValue;
    PAttnArgs a{};
    a.qk = qk; a.stats = stats; a.qk_cp = qk_cpitch; a.f_co = f_coff; a.g_co = g_coff; a.d = d; a.B = B; a.N = N; a.C = C; a.qk_fs = qk_fs;
    const int DP = d + 4, NJ = (N + PA_TJ - 1) / PA_TJ;
    static const bool s_mfma = [] { const char* e = getenv("HAVC_PRECISE_ATTN_S_MFMA"); return e ? atoi(e) != 0 : true; }();      // A/B: 0 = S on the fp32 VALU
    const bool smf = s_mfma && (d == 64 || d == 96) && (qk_cpitch & 15) == 0 && (f_coff & 7) == 0 && (g_coff & 7) == 0;
    if (smf) {
        if (d == 64) hipLaunchKernelGGL(pattn_stats_mfma_kernel<2>, dim3(NJ, B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(pattn_stats_mfma_kernel<3>, dim3(NJ, B), dim3(256), 0, s, a);
    } else {
        const int lds_a = (PA_TJ * DP + PA_TI * DP + 4 * 64 * 2) * 4;
        lds_optin<pattn_stats_kernel>();
        hipLaunchKernelGGL(pattn_stats_kernel, dim3(NJ, B), dim3(256), lds_a, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    PAttnMArgs m{};
    m.qk = qk; m.vT = vT; m.x = x; m.out = out; m.stats = stats;
    m.qk_cp = qk_cpitch; m.f_co = f_coff; m.g_co = g_coff; m.d = d; m.npitch = npitch; m.x_cp = x_cpitch; m.x_co = x_coff; m.o_cp = o_cpitch; m.o_co = o_coff;
    m.N = N; m.C = C; m.qk_fs = qk_fs; m.v_fs = v_fs; m.x_fs = x_fs; m.o_fs = o_fs; m.gamma = gamma;
    if (smf) {
        const int lds_c = 3 * PA_TJ * 64 + 2 * PM_CC * 64;              // 44 KiB: no opt-in needed
        if (d == 64) hipLaunchKernelGGL(pattn_apply_mfma2_kernel<2>, dim3(NJ, B, C / PM_CC), dim3(256), lds_c, s, m);
        else hipLaunchKernelGGL(pattn_apply_mfma2_kernel<3>, dim3(NJ, B, C / PM_CC), dim3(256), lds_c, s, m);
        return (int)hipGetLastError();
    }
    const int lds_b = (PA_TJ * DP + PA_TI * DP) * 4 + 3 * PA_TJ * 64 + 2 * PM_CC * 64;
    lds_optin<pattn_apply_mfma_kernel>();
    hipLaunchKernelGGL(pattn_apply_mfma_kernel, dim3(NJ, B, C / PM_CC), dim3(256), lds_b, s, m);
    return (int)hipGetLastError();
}
