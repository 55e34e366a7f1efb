// fastai SelfAttention (fastai/layers.py:81-96) in flash form for gfx950: the N x N map (4900 x 4900 at
// render_factor 35) is never written to HBM.
//   f = Wq x, g = Wk x, h = Wv x;  beta = softmax_i(f_i . g_j)  (dim=1: over i, NO 1/sqrt(d));
//   out_j = gamma * sum_i h_i beta[i][j] + x_j
// => "keys" K_i = f_i, "queries" Q_j = g_j, "values" V_i = h_i.
// Layout: qk [B][N][qk_pitch] fp16 (f at f_coff, g at g_coff, D channels each) from a 1x1 conv;
//         vT [B][DV][npitch] fp16 (values, TRANSPOSED store of the 1x1 conv, zero beyond N);
//         x / out NHWC fp16.
// Block = 4 waves = 64 queries x one 128-wide slice of DV; S^T = K Q^T and O^T = V^T P^T on
// v_mfma_f32_16x16x32_f16 so that every lane owns ONE query column: the online-softmax row max / sum are
// in-lane reductions plus two wavefront shuffles (xor 16, xor 32).
#include "kernels.h"
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swzr(int row) { return (4 - ((row >> 2) & 3)) & 3; }   // 16 consecutive rows
__device__ __forceinline__ int swzk(int row) { return (4 - ((row >> 3) & 3)) & 3; }   // permuted key rows

template <int D>
__global__ void __launch_bounds__(256, 2) self_attention_kernel(const half_t* __restrict__ qk, int qk_pitch, int f_coff,
                                                             int g_coff, const half_t* __restrict__ vT, int DV,
                                                             int npitch, const half_t* __restrict__ x, int x_cpitch,
                                                             int x_coff, half_t* __restrict__ out, int o_cpitch,
                                                             int o_coff, int N, float gamma) {
    constexpr int KS = D / 32;            // k-steps over the feature dim
    constexpr int KCH = D / 8;            // 16-B chunks per key row
    constexpr int K_IT = 64 * KCH / 256;  // K-tile chunks per thread
    __shared__ __attribute__((aligned(16))) half_t smem[KS * 64 * 32 + 2 * 128 * 32];
    half_t* Ks = smem;
    half_t* Vs = smem + KS * 64 * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.z, dv0 = blockIdx.y * 128, q0 = blockIdx.x * 64;
    const int q = q0 + wave * 16 + lr;
    const bool q_ok = q < N;
    const half_t* qk_b = qk + (int64_t)b * N * qk_pitch;
    const half_t* vT_b = vT + ((int64_t)b * DV + dv0) * npitch;

    half8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
        if (q_ok) v = *reinterpret_cast<const half8*>(qk_b + (int64_t)q * qk_pitch + g_coff + ks * 32 + lg * 8);
        qf[ks] = v;
    }

    float4v o[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = float4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    for (int kv0 = 0; kv0 < N; kv0 += 64) {
        __syncthreads();
        // ---- stage K tile [KS][64 keys][32] and V^T tile [2][128 dv][32 keys] ----
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int i = tid + it * 256;
            const int key = i / KCH, c = i % KCH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kv0 + key < N) v = *reinterpret_cast<const uint4*>(qk_b + (int64_t)(kv0 + key) * qk_pitch + f_coff + c * 8);
            *reinterpret_cast<uint4*>(Ks + (((c >> 2) * 64 + key) * 4 + ((c & 3) ^ swzk(key))) * 8) = v;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * 256;
            const int row = i >> 3, kc = i & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(vT_b + (int64_t)row * npitch + kv0 + kc * 8);
            *reinterpret_cast<uint4*>(Vs + (((kc >> 2) * 128 + row) * 4 + ((kc & 3) ^ swzr(row))) * 8) = v;
        }
        __syncthreads();

        // ---- S^T[key][query]: fragment f = 2s+h covers keys 32s + (i>>2)*8 + h*4 + (i&3), i = row index ----
        float4v sacc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            sacc[f] = float4v{0.f, 0.f, 0.f, 0.f};
            const int key = 32 * (f >> 1) + (lr >> 2) * 8 + (f & 1) * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(Ks + ((ks * 64 + key) * 4 + (lg ^ swzk(key))) * 8);
                sacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], sacc[f], 0, 0, 0);
            }
        }
        // lane (lr, lg), fragment f, reg r  <->  key kv0 + 32*(f>>1) + lg*8 + (f&1)*4 + r, query lr
        float mx = -1e30f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 32 * (f >> 1) + lg * 8 + (f & 1) * 4 + r;
                if (key >= N) sacc[f][r] = -1e30f;
                mx = fmaxf(mx, sacc[f][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        half8 pf[2];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(sacc[f][r] - m_new);
                psum += pv;
                pf[f >> 1][(f & 1) * 4 + r] = (half_t)pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][r] *= alpha;
        // ---- O^T[dv][query] += V^T[dv][key] P^T[key][query] ----
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 16 + lr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 vf = *reinterpret_cast<const half8*>(Vs + ((s * 128 + row) * 4 + (lg ^ swzr(row))) * 8);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s], o[t], 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    if (!q_ok) return;
    const float inv = gamma / l_run;
    const half_t* xr = x + ((int64_t)b * N + q) * x_cpitch + x_coff + dv0;
    half_t* orow = out + ((int64_t)b * N + q) * o_cpitch + o_coff + dv0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = t * 16 + lg * 4;
        const half4 xv = *reinterpret_cast<const half4*>(xr + c);
        half4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[t][r] * inv + (float)xv[r]);
        *reinterpret_cast<half4*>(orow + c) = ov;
    }
}

// ---- version 2 (round 2): 128 queries x 256 output channels per block ------------------------------------------------
// Same arithmetic and fragment conventions as above; what changes is the blocking.  Version 1 recomputed S = K Q^T for each of
// the DV / 128 output slices and fed every MFMA from its own ds_read_b128 (16 queries per wave: no operand reuse): 1 LDS read per
// MFMA, S computed 4x.  Here a wave owns TWO query fragments (32 queries), so every K / V^T fragment read from LDS feeds two
// MFMAs, the block covers 256 output channels (S computed DV / 256 = 2x at C = 512), and the next key tile is in flight while
// the current one is multiplied: the V^T tile (32 KiB) goes straight to a second LDS buffer by LDS-DMA (buffer_load ... lds; the
// bank swizzle is applied to the SOURCE chunk each lane fetches, the LDS image is lane-linear), the small K tile through
// registers.  2 x 32 KiB V^T + K = 72-76 KiB: two blocks per CU.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int D>
__global__ void __launch_bounds__(256, 2) self_attention_kernel2(const half_t* __restrict__ qk, int qk_pitch, int f_coff, int g_coff,
                                                                 const half_t* __restrict__ vT, int DV, int npitch, const half_t* __restrict__ x,
                                                                 int x_cpitch, int x_coff, half_t* __restrict__ out, int o_cpitch, int o_coff, int N,
                                                                 float gamma) {
    constexpr int KS = D / 32;
    constexpr int KCH = D / 8;
    constexpr int K_IT = 64 * KCH / 256;
    constexpr int DVB = 256, TF = DVB / 16, VB = 2 * DVB * 32;         // halfs per V^T buffer
    constexpr int V_PIECES = VB * 2 / 1024 / 4;                        // 1-KiB DMA pieces per wave per tile (8)
    __shared__ __attribute__((aligned(16))) half_t smem[KS * 64 * 32 + 2 * VB];
    half_t* Ks = smem;
    half_t* Vs = smem + KS * 64 * 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.z, dv0 = blockIdx.y * DVB, q0 = blockIdx.x * 128 + wave * 32;
    const half_t* qk_b = qk + (int64_t)b * N * qk_pitch;
    const half_t* vT_b = vT + ((int64_t)b * DV + dv0) * npitch;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vT_b), 0, (unsigned)(DVB * npitch * 2), 0x00020000);

    // V^T DMA role: piece p = wave * V_PIECES + j covers the 64 16-byte slots L = p * 64 + lane of a buffer;
    // slot L = (s * DVB + row) * 4 + pos holds source chunk kc = s * 4 + (pos ^ swzr(row)) of row `row`
    unsigned v_voff[V_PIECES];
#pragma unroll
    for (int j = 0; j < V_PIECES; ++j) {
        const int L = (wave * V_PIECES + j) * 64 + lane;
        const int pos = L & 3, row = (L >> 2) & (DVB - 1), sidx = L >> 10;
        v_voff[j] = (unsigned)((row * npitch + (sidx * 4 + (pos ^ swzr(row))) * 8) * 2);
    }
    auto dma_v = [&](int kv0, int buf) {
        char* base = reinterpret_cast<char*>(Vs + buf * VB) + wave * V_PIECES * 1024;
#pragma unroll
        for (int j = 0; j < V_PIECES; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(base + j * 1024), 16, v_voff[j], (unsigned)(kv0 * 2), 0, 0);
    };

    half8 qf[2][KS];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = q0 + qi * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
            if (q < N) v = *reinterpret_cast<const half8*>(qk_b + (int64_t)q * qk_pitch + g_coff + ks * 32 + lg * 8);
            qf[qi][ks] = v;
        }
    }
    float4v o[2][TF];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int t = 0; t < TF; ++t) o[qi][t] = float4v{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    uint4 kreg[K_IT];
    auto fetch_k = [&](int kv0) {                      // keys past N are masked below: any finite-or-not value will do, clamp the row
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int i = tid + it * 256;
            const int key = min(kv0 + i / KCH, N - 1), c = i % KCH;
            kreg[it] = *reinterpret_cast<const uint4*>(qk_b + (int64_t)key * qk_pitch + f_coff + c * 8);
        }
    };
    auto stash_k = [&]() {
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int i = tid + it * 256;
            const int key = i / KCH, c = i % KCH;
            *reinterpret_cast<uint4*>(Ks + (((c >> 2) * 64 + key) * 4 + ((c & 3) ^ swzk(key))) * 8) = kreg[it];
        }
    };

    fetch_k(0);
    dma_v(0, 0);
    int buf = 0;
    for (int kv0 = 0; kv0 < N; kv0 += 64, buf ^= 1) {
        __syncthreads();                               // every wave is done with the previous tile (K buffer, V^T buffer buf ^ 1)
        stash_k();                                     // (the compiler's wait for kreg also covers this tile's V^T DMA: issued earlier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kv0 + 64 < N) {                            // next tile: in flight under the MFMAs below
            fetch_k(kv0 + 64);
            dma_v(kv0 + 64, buf ^ 1);
        }
        const half_t* Vc = Vs + buf * VB;

        // ---- S^T[key][query] for both query fragments: one K fragment read feeds two MFMAs ----
        float4v sacc[2][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            sacc[0][f] = sacc[1][f] = float4v{0.f, 0.f, 0.f, 0.f};
            const int key = 32 * (f >> 1) + (lr >> 2) * 8 + (f & 1) * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(Ks + ((ks * 64 + key) * 4 + (lg ^ swzk(key))) * 8);
                sacc[0][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][ks], sacc[0][f], 0, 0, 0);
                sacc[1][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][ks], sacc[1][f], 0, 0, 0);
            }
        }
        half8 pf[2][2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float mx = -1e30f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + 32 * (f >> 1) + lg * 8 + (f & 1) * 4 + r;
                    if (key >= N) sacc[qi][f][r] = -1e30f;
                    mx = fmaxf(mx, sacc[qi][f][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qi], mx);
            const float alpha = __expf(m_run[qi] - m_new);
            m_run[qi] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(sacc[qi][f][r] - m_new);
                    psum += pv;
                    pf[qi][f >> 1][(f & 1) * 4 + r] = (half_t)pv;
                }
            l_run[qi] = l_run[qi] * alpha + psum;
#pragma unroll
            for (int t = 0; t < TF; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qi][t][r] *= alpha;
        }
        // ---- O^T[dv][query] += V^T[dv][key] P^T[key][query]: one V^T fragment read feeds two MFMAs ----
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            const int row = t * 16 + lr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 vf = *reinterpret_cast<const half8*>(Vc + ((s * DVB + row) * 4 + (lg ^ swzr(row))) * 8);
                o[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[0][s], o[0][t], 0, 0, 0);
                o[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[1][s], o[1][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int q = q0 + qi * 16 + lr;
        if (q >= N) continue;
        const float inv = gamma / l;
        const half_t* xr = x + ((int64_t)b * N + q) * x_cpitch + x_coff + dv0;
        half_t* orow = out + ((int64_t)b * N + q) * o_cpitch + o_coff + dv0;
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            const int c = t * 16 + lg * 4;
            const half4 xv = *reinterpret_cast<const half4*>(xr + c);
            half4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[qi][t][r] * inv + (float)xv[r]);
            *reinterpret_cast<half4*>(orow + c) = ov;
        }
    }
}

int launch_attention(const half_t* qk, int qk_pitch, int f_coff, int g_coff, int d, const half_t* vT, int dv,
                     int npitch, const half_t* x, int x_cpitch, int x_coff, half_t* out, int o_cpitch, int o_coff,
                     int B, int N, float gamma, hipStream_t s) {
    if (dv % 128 != 0 || npitch % 64 != 0 || npitch < N) return (int)hipErrorInvalidValue;
    static const bool v1 = getenv("HAVC_ATTENTION_V1") != nullptr;            // A/B switch (profiling)
    if (dv % 256 == 0 && !v1) {
        dim3 grid2((N + 127) / 128, dv / 256, B);
        if (d == 64)
            hipLaunchKernelGGL(self_attention_kernel2<64>, grid2, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x, x_cpitch, x_coff,
                               out, o_cpitch, o_coff, N, gamma);
        else if (d == 96)
            hipLaunchKernelGGL(self_attention_kernel2<96>, grid2, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x, x_cpitch, x_coff,
                               out, o_cpitch, o_coff, N, gamma);
        else if (d == 32)
            hipLaunchKernelGGL(self_attention_kernel2<32>, grid2, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x, x_cpitch, x_coff,
                               out, o_cpitch, o_coff, N, gamma);
        else
            return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
    dim3 grid((N + 63) / 64, dv / 128, B);
    if (d == 64)
        hipLaunchKernelGGL(self_attention_kernel<64>, grid, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x,
                           x_cpitch, x_coff, out, o_cpitch, o_coff, N, gamma);
    else if (d == 96)
        hipLaunchKernelGGL(self_attention_kernel<96>, grid, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x,
                           x_cpitch, x_coff, out, o_cpitch, o_coff, N, gamma);
    else if (d == 32)
        hipLaunchKernelGGL(self_attention_kernel<32>, grid, dim3(256), 0, s, qk, qk_pitch, f_coff, g_coff, vT, dv, npitch, x,
                           x_cpitch, x_coff, out, o_cpitch, o_coff, N, gamma);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// Eager module load (havc_create, under the library's set-up mutex): the HIP runtime loads a translation unit's code object on the first use
// of one of its kernels; querying one here moves that -- and the big-LDS opt-ins below -- out of the first launch, which may come from
// several host threads at once ("set-up is serialised").
void preload_attention() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(self_attention_kernel2<64>)); (void)hipGetLastError(); }
