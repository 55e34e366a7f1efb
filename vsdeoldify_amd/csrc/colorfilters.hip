// HBM-bound u8 per-pixel filters of the HAVC post path (vsslib/imfilters.py), interleaved RGB in HBM.
// Integer arithmetic follows OpenCV's 8-bit BT.601 "YUV" fixed point (yuv_shift = 14) and Pillow's
// ImagingBlend exactly (bit-exact targets; see oracle/cvcolor.py, oracle/imaging.py).
#include "kernels.h"

static inline int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- OpenCV RGB2YUV / YUV2RGB, 8-bit (color_yuv.simd.hpp RGB2YCrCb_i / YCrCb2RGB_i, isCrCb=false) ----
__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void rgb2yuv(int r, int g, int b, int& y, int& u, int& v) {
    y = descale14(r * 4899 + g * 9617 + b * 1868);
    u = sat8(descale14((b - y) * 8061 + (128 << 14)));
    v = sat8(descale14((r - y) * 14369 + (128 << 14)));
    y = sat8(y);
}
__device__ __forceinline__ void yuv2rgb(int y, int u, int v, int& r, int& g, int& b) {
    u -= 128;
    v -= 128;
    b = sat8(y + descale14(u * 33292));
    g = sat8(y + descale14(u * -6472 + v * -9519));
    r = sat8(y + descale14(v * 18678));
}

// ---- PIL Image.blend(a, b, w): (UINT8)((int)a + w * ((int)b - (int)a)) in float32 with the product
// ROUNDED before the add (Pillow's x86-64 build has no FMA).  HIP's __fmul_rn/__fadd_rn are plain operators
// and hipcc contracts a + w*d into v_fma_f32 by default, so contraction is switched off explicitly here
// (and the whole file is built with -ffp-contract=off). ----
__device__ __forceinline__ uint8_t blend1(uint8_t a, uint8_t b, float w) {
#pragma clang fp contract(off)
    const float prod = w * (float)((int)b - (int)a);
    const float sum = (float)(int)a + prod;
    return (uint8_t)(int)sum;
}

__global__ void blend_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, float w,
                                uint8_t* __restrict__ out, int64_t n) {
    // 4 bytes per thread when aligned
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uchar4 va = reinterpret_cast<const uchar4*>(a)[i], vb = reinterpret_cast<const uchar4*>(b)[i];
        uchar4 o;
        o.x = blend1(va.x, vb.x, w); o.y = blend1(va.y, vb.y, w);
        o.z = blend1(va.z, vb.z, w); o.w = blend1(va.w, vb.w, w);
        reinterpret_cast<uchar4*>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = blend1(a[i], b[i], w);
    }
}

int launch_blend_u8(const uint8_t* a, const uint8_t* b, float w, uint8_t* out, int64_t nbytes, hipStream_t s) {
    hipLaunchKernelGGL(blend_u8_kernel, dim3(grid_for(nbytes / 4 + 1)), dim3(256), 0, s, a, b, w, out, nbytes);
    return (int)hipGetLastError();
}

// ---- chroma_post_process / ColorizerFilter._post_process: Y from orig, U,V from colour ----
__global__ void yuv_merge_kernel(const uint8_t* __restrict__ color, const uint8_t* __restrict__ orig,
                                 uint8_t* __restrict__ out, int64_t npix) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int y, u, v, y2, u2, v2, r, g, b;
        rgb2yuv(color[i * 3], color[i * 3 + 1], color[i * 3 + 2], y, u, v);
        rgb2yuv(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2], y2, u2, v2);
        yuv2rgb(y2, u, v, r, g, b);
        out[i * 3] = (uint8_t)r; out[i * 3 + 1] = (uint8_t)g; out[i * 3 + 2] = (uint8_t)b;
    }
}

int launch_yuv_merge(const uint8_t* color, const uint8_t* orig, uint8_t* out, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(yuv_merge_kernel, dim3(grid_for(npix)), dim3(256), 0, s, color, orig, out, npix);
    return (int)hipGetLastError();
}

// ---- chroma_stabilizer (imfilters.py:160-200): clip U,V of img_new into [u8(U1(1-a)), u8(U1(1+a))] of
// img_stable (float64 products, truncating casts, as numpy does), Y from img_stable, optional blend ----
__global__ void chroma_stabilizer_kernel(const uint8_t* __restrict__ st, const uint8_t* __restrict__ nw, double alpha,
                                         float weight, int do_blend, uint8_t* __restrict__ out, int64_t npix) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int y1, u1, v1, y2, u2, v2, r, g, b;
        const int sr = st[i * 3], sg = st[i * 3 + 1], sb = st[i * 3 + 2];
        rgb2yuv(sr, sg, sb, y1, u1, v1);
        rgb2yuv(nw[i * 3], nw[i * 3 + 1], nw[i * 3 + 2], y2, u2, v2);
        const int u_up = (int)fmin(fmax((double)u1 * (1.0 + alpha), 0.0), 255.0);
        const int v_up = (int)fmin(fmax((double)v1 * (1.0 + alpha), 0.0), 255.0);
        const int u_dn = (int)fmin(fmax((double)u1 * (1.0 - alpha), 0.0), 255.0);
        const int v_dn = (int)fmin(fmax((double)v1 * (1.0 - alpha), 0.0), 255.0);
        u2 = u2 > u_up ? u_up : u2;  u2 = u2 < u_dn ? u_dn : u2;   // array_clip: cap to max, then floor to min
        v2 = v2 > v_up ? v_up : v2;  v2 = v2 < v_dn ? v_dn : v2;
        yuv2rgb(y1, u2, v2, r, g, b);
        if (do_blend) {
            r = blend1((uint8_t)sr, (uint8_t)r, weight);
            g = blend1((uint8_t)sg, (uint8_t)g, weight);
            b = blend1((uint8_t)sb, (uint8_t)b, weight);
        }
        out[i * 3] = (uint8_t)r; out[i * 3 + 1] = (uint8_t)g; out[i * 3 + 2] = (uint8_t)b;
    }
}

int launch_chroma_stabilizer(const uint8_t* stable, const uint8_t* inew, double alpha, float weight, uint8_t* out,
                             int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(chroma_stabilizer_kernel, dim3(grid_for(npix)), dim3(256), 0, s, stable, inew, alpha, weight,
                       weight < 1.0f ? 1 : 0, out, npix);
    return (int)hipGetLastError();
}

// ---- separable polyphase resample (Spline64 taps computed on the host; harness stand-in for zimg) ----
// pass 1 (horizontal): u8 [n][sh][sw][3] -> float [n][sh][dw][3]
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, float* __restrict__ tmp, const int* __restrict__ start,
                                const float* __restrict__ wts, int taps, int sw, int dw, int64_t rows) {
    const int64_t total = rows * dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw);
        const int64_t row = i / dw;
        const uint8_t* sp = src + row * sw * 3;
        const int s0 = start[x];
        const float* w = wts + (int64_t)x * taps;
        float r = 0.f, g = 0.f, b = 0.f;
        for (int t = 0; t < taps; ++t) {
            int sx = s0 + t;
            sx = sx < 0 ? 0 : (sx >= sw ? sw - 1 : sx);
            const float wt = w[t];
            r += wt * sp[sx * 3]; g += wt * sp[sx * 3 + 1]; b += wt * sp[sx * 3 + 2];
        }
        float* o = tmp + i * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}

// pass 2 (vertical): float [n][sh][dw][3] -> u8 [n][dh][dw][3]; if orig != null, fuse chroma_post_process
// (vsfilters.py:863-899 -> imfilters.py:312-321): keep luma of orig, take U,V of the resampled colour.
__global__ void resize_v_kernel(const float* __restrict__ tmp, uint8_t* __restrict__ dst, const uint8_t* __restrict__ orig,
                                const int* __restrict__ start, const float* __restrict__ wts, int taps, int sh, int dh,
                                int dw, int n_frames) {
    const int64_t total = (int64_t)n_frames * dh * dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw);
        const int y = (int)((i / dw) % dh);
        const int f = (int)(i / ((int64_t)dw * dh));
        const int s0 = start[y];
        const float* w = wts + (int64_t)y * taps;
        const float* base = tmp + (int64_t)f * sh * dw * 3 + (int64_t)x * 3;
        float r = 0.f, g = 0.f, b = 0.f;
        for (int t = 0; t < taps; ++t) {
            int sy = s0 + t;
            sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            const float wt = w[t];
            const float* p = base + (int64_t)sy * dw * 3;
            r += wt * p[0]; g += wt * p[1]; b += wt * p[2];
        }
        int ri = sat8((int)floorf(r + 0.5f)), gi = sat8((int)floorf(g + 0.5f)), bi = sat8((int)floorf(b + 0.5f));
        if (orig) {
            int yy, u, v, y2, u2, v2;
            rgb2yuv(ri, gi, bi, yy, u, v);
            rgb2yuv(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2], y2, u2, v2);
            yuv2rgb(y2, u, v, ri, gi, bi);
        }
        dst[i * 3] = (uint8_t)ri; dst[i * 3 + 1] = (uint8_t)gi; dst[i * 3 + 2] = (uint8_t)bi;
    }
}

int launch_resize_passes(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int n_frames, float* tmp,
                         const int* h_start, const float* h_w, int h_taps, const int* v_start, const float* v_w,
                         int v_taps, const uint8_t* orig, hipStream_t s) {
    const int64_t rows = (int64_t)n_frames * sh;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(rows * dw)), dim3(256), 0, s, src, tmp, h_start, h_w, h_taps, sw, dw,
                       rows);
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((int64_t)n_frames * dh * dw)), dim3(256), 0, s, tmp, dst, orig,
                       v_start, v_w, v_taps, sh, dh, dw, n_frames);
    return (int)hipGetLastError();
}
