// Shared device helpers of the convolution kernels (conv_igemm.hip, conv_igemm_pipe.hip): the fused epilogue.
#pragma once
#include "kernels.h"
#include "../../include/havc_mi355.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

// GELU v Phi(v) of the ConvNeXt block / the ViT MLP (nn.GELU(), exact form) for results that are ROUNDED TO fp16: Phi(v) as a logistic function of an odd
// polynomial, Phi(v) = 1 / (1 + exp(-u)), u = v P(min(v^2, 36)) with a degree-3 P fitted (minimax over |v| <= 12, tools/fit_gelu.py) to
// logit Phi: |v Phi(v) - approximation| < 1.2e-5 everywhere -- a fifth of half an fp16 ulp at |result| >= 0.06, exact 0.5 v at v -> 0, v / 0 beyond the clamp
// (exp2 overflows to inf, v_rcp(inf) = 0).  10 VALU instructions with two transcendentals (v_exp, v_rcp) per value against 16 of the Abramowitz-Stegun
// erf it replaces (round 4): the epilogue of DDColor's pwconv1 evaluates 0.4 G of these per frame and no other wave's MFMAs hide them (one block per CU).
// One definition for every conv kernel, so all tile configurations keep producing the same bytes.  The precise mode uses libm's erff (epilogue_frag_precise).
__device__ __forceinline__ float gelu_erf(float v) {
    const float L2E = 1.44269504088896341f;
    const float v2 = fminf(v * v, 36.0f);
    float p = fmaf(-1.72152684e-05f * -L2E, v2, -5.10199108e-04f * -L2E);
    p = fmaf(p, v2, 7.34616312e-02f * -L2E);
    p = fmaf(p, v2, 1.59538024f * -L2E);
    const float e = __builtin_amdgcn_exp2f(p * v);               // exp(-u)
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// output pixel index of GEMM row m (identity unless the conv scatters with an output step: ConvTranspose parity convs)
__device__ __forceinline__ int64_t out_pixel(const ConvArgs& p, int m, int HoWo) {
    if (p.oss == 1) return m;
    const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
    return ((int64_t)(b * p.Ho * p.oss + ho * p.oss + p.ooy)) * (p.Wo * p.oss) + wo * p.oss + p.oox;
}

// ---- precise mode (HAVC_F_PRECISE): an fp32-class value travels as TWO fp16 numbers, hi = fp16(v) and lo = fp16((v - hi) * 2^11) ----
// (v - hi) is exact in fp32 and so is the power-of-two scale: hi + lo * 2^-11 keeps 22 significand bits.  The lo plane of a tensor sits
// half a pixel pitch behind the hi plane (a precise buffer's pixel row is [hi: P channels | lo: P channels], cpitch = 2 P).
__device__ __forceinline__ void split_hl(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)((v - (float)hi) * 2048.f);
}
__device__ __forceinline__ float join_hl(half_t hi, half_t lo) { return (float)hi + (float)lo * (1.f / 2048.f); }

// Precise epilogue of one accumulator fragment.  The accumulator holds 2^11 / wscale x the convolution (conv kernels: the three K
// segments x_hi * (2^11 w_hi), x_hi * (2^11 w_lo), (2^11 x_lo) * w_hi summed by the unchanged MFMA main loop; plan.py pack_conv):
// p.pscale brings it back, everything after that is fp32 like the reference (deoldify/filters.py:45-68), no fp16 rounding points;
// the result is stored as a hi / lo pair.
__device__ __forceinline__ void epilogue_frag_precise(const ConvArgs& p, const float4v acc, int m, int n, int HoWo) {
    const bool leaky = p.flags & HAVC_F_LEAKY;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] * p.pscale;
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (p.flags & HAVC_F_OUT_RGB8) {
        if (n == 0) {
            uint8_t* y = reinterpret_cast<uint8_t*>(p.y) + (int64_t)m * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float s = 1.f / (1.f + expf(-v[r]));
                s = s * (p.f1 - p.f0) + p.f0;
                s = s * p.istd[r] + p.mean[r];
                s = fminf(fmaxf(s, 0.f), 1.f);
                y[r] = (uint8_t)(int)(s * 255.f);
            }
        }
        return;
    }
    if (p.flags & HAVC_F_RELU_PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : (leaky ? v[r] * p.f2 : 0.f);
    }
    if (p.flags & HAVC_F_GELU) {                               // nn.GELU() in its exact form, libm erff (the fast path's 1.5e-7 polynomial is for fp16 results)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));
    }
    if (p.flags & HAVC_F_AFFINE) {
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + n);
        const float4 sh = *reinterpret_cast<const float4*>(p.shift + n);
        v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
        v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
    }
    half_t* y = reinterpret_cast<half_t*>(p.y);
    const int ylo = p.y_cpitch >> 1;
    half4 oh, ol;
    if (p.flags & HAVC_F_OUT_PIXSHUF) {
        const int q = n / p.Co, c = n - q * p.Co;
        if (q >= 4) return;
        const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int64_t pix = (int64_t)(b * 2 * p.Ho + 2 * ho + (q >> 1)) * (2 * p.Wo) + 2 * wo + (q & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { half_t a, b; split_hl(v[r], a, b); oh[r] = a; ol[r] = b; }
        half_t* d = y + pix * p.y_cpitch + p.y_coff + c;
        *reinterpret_cast<half4*>(d) = oh;
        *reinterpret_cast<half4*>(d + ylo) = ol;
        return;
    }
    if (n >= p.Co) return;
    const int64_t mo = out_pixel(p, m, HoWo);
    if (p.flags & HAVC_F_RESIDUAL) {
        const half_t* rp = p.res + mo * p.res_cpitch + p.res_coff + n;
        const half4 rh = *reinterpret_cast<const half4*>(rp), rl = *reinterpret_cast<const half4*>(rp + (p.res_cpitch >> 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += join_hl(rh[r], rl[r]);
    }
    if (p.flags & HAVC_F_RELU_POST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : (leaky ? v[r] * p.f2 : 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { half_t a, b; split_hl(v[r], a, b); oh[r] = a; ol[r] = b; }
    if (p.flags & HAVC_F_OUT_TRANSPOSED) {                      // [2][Co][pix_pitch] per frame: hi plane, then lo plane (the precise attention's value operand)
        const int b = m / HoWo;
        const int64_t base = (int64_t)b * 2 * p.Co * p.pix_pitch + (m - b * HoWo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[base + (int64_t)(n + r) * p.pix_pitch] = oh[r];
            y[base + (int64_t)(p.Co + n + r) * p.pix_pitch] = ol[r];
        }
        return;
    }
    half_t* d = y + mo * p.y_cpitch + p.y_coff + n;
    *reinterpret_cast<half4*>(d) = oh;
    *reinterpret_cast<half4*>(d + ylo) = ol;
}

// one 16x16 accumulator fragment -> fused epilogue -> store.  lane owns pixel m, channels n..n+3.
// PRECISE is a COMPILE-TIME switch: the precise epilogue lives in kernel instantiations of its own, the fast kernels carry none of its
// code (a run-time branch here cost the dominant fast kernel 60 % -- register allocation of the whole kernel changed).
// UNIT_STEP: the caller knows the op's out step is 1 (the compile-time-flag kernels are only launched then): the output pixel IS the GEMM row, and the
// reciprocals of out_pixel's divisions need not stay live across the main loop (round 6: one of them was the tail conv's only spilled register)
template <bool PRECISE = false, bool UNIT_STEP = false>
__device__ __forceinline__ void epilogue_frag(const ConvArgs& p, const float4v acc, int m, int n, int HoWo, const int flags) {
    if (m >= p.M || n >= p.Npad) return;
    if (PRECISE) { epilogue_frag_precise(p, acc, m, n, HoWo); return; }
    const bool leaky = flags & HAVC_F_LEAKY;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r];
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (flags & HAVC_F_OUT_RGB8) {
        if (n == 0) {
            uint8_t* y = reinterpret_cast<uint8_t*>(p.y) + (int64_t)m * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float s = 1.f / (1.f + __expf(-v[r]));
                s = s * (p.f1 - p.f0) + p.f0;
                s = s * p.istd[r] + p.mean[r];
                s = fminf(fmaxf(s, 0.f), 1.f);
                y[r] = (uint8_t)(int)(s * 255.f);
            }
        }
        return;
    }
    if (flags & HAVC_F_RELU_PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : (leaky ? v[r] * p.f2 : 0.f);
    }
    if (flags & HAVC_F_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (flags & HAVC_F_AFFINE) {
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + n);
        const float4 sh = *reinterpret_cast<const float4*>(p.shift + n);
        v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
        v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
    }
    if (flags & HAVC_F_OUT_PIXSHUF) {
        const int q = n / p.Co, c = n - q * p.Co;
        if (q >= 4) return;
        const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int64_t pix = (int64_t)(b * 2 * p.Ho + 2 * ho + (q >> 1)) * (2 * p.Wo) + 2 * wo + (q & 1);
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.y) + pix * p.y_cpitch + p.y_coff + c) = o;
        return;
    }
    if (n >= p.Co) return;
    const int64_t mo = UNIT_STEP ? (int64_t)m : out_pixel(p, m, HoWo);
    if (flags & HAVC_F_RESIDUAL) {
        // Same rounding points as the LDS epilogue of the pipelined kernels (conv_pipe_epilogue.inc): the conv result is rounded
        // to fp16 BEFORE the residual is added (and once more after), so every tile configuration produces the same bytes and a
        // frame colours identically whatever configuration the heuristic / autotuner picks for a batch size.
        const half4 rv = *reinterpret_cast<const half4*>(p.res + mo * p.res_cpitch + p.res_coff + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)(half_t)v[r] + (float)rv[r];
    }
    if (flags & HAVC_F_RELU_POST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : (leaky ? v[r] * p.f2 : 0.f);
    }
    half_t* y = reinterpret_cast<half_t*>(p.y);
    if (flags & HAVC_F_OUT_TRANSPOSED) {
        const int b = m / HoWo;
        const int64_t base = (int64_t)b * p.Co * p.pix_pitch + (m - b * HoWo);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[base + (int64_t)(n + r) * p.pix_pitch] = (half_t)v[r];
    } else {
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(y + mo * p.y_cpitch + p.y_coff + n) = o;
    }
}


template <bool PRECISE = false>
__device__ __forceinline__ void epilogue_frag(const ConvArgs& p, const float4v acc, int m, int n, int HoWo) {
    epilogue_frag<PRECISE>(p, acc, m, n, HoWo, p.flags);
}
