// Zhang et al. colorizers: colour-space pre/post kernels and Pillow's 8-bit resampler.
//   rgb -> CIELAB L  and  Lab -> rgb follow scikit-image's float64 formulas (skimage/color/colorconv.py: rgb2xyz,
//   xyz2lab, lab2xyz, xyz2rgb; D65 / 2 deg) as used by preprocess_img / postprocess_tens
//   (vsdeoldify/colorization/colorizers/util.py:25-55).  fp64 on the device: HBM-bound kernels, and the reference
//   truncates x*255, so fp32 colour math would flip LSBs for no speed gain.
//   pil_resize_*: Pillow ImagingResample 8bpc (libImaging/Resample.c), integer coefficients from the host: bit-exact.
#include "kernels.h"
#include <mutex>

static inline int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
static int ensure_gray_lut(hipStream_t s);      // (defined next to gray_lut_kernel)

__device__ __forceinline__ double srgb_to_linear(double c) { return c > 0.04045 ? pow((c + 0.055) / 1.055, 2.4) : c / 12.92; }

__device__ __forceinline__ double lab_f(double t) { return t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0; }

__device__ __forceinline__ double rgb_to_L_full(int r, int g, int b) {
    const double R = srgb_to_linear(r / 255.0), G = srgb_to_linear(g / 255.0), B = srgb_to_linear(b / 255.0);
    const double y = 0.212671 * R + 0.715160 * G + 0.072169 * B;        // / 1.0 (D65 Yn)
    return 116.0 * lab_f(y) - 16.0;
}

// Gray pixels (R = G = B: every frame of the HAVC flows after convert_format_RGB24 of a B&W source, and the bench clips) take L from a 256-entry table that
// gray_lut_kernel fills ONCE per device with this very function: the same bits as the per-pixel path, without its three fp64 pow and one cbrt per pixel
// (round 5: DDColor's prep kernel was 1.7 ms per 128 frames at 512 x 512).  g_gray_dd: prep_ddcolor's three normalised values of Lab(L, 0, 0) -> RGB.
__device__ double g_gray_L[256];
__device__ float g_gray_dd[256][4];
__device__ __forceinline__ double rgb_to_L(int r, int g, int b) {
    if (r == g && g == b) return g_gray_L[r];
    return rgb_to_L_full(r, g, b);
}

// model input: u8 RGB -> (float32(L) - 50) / 100 in channel 0 of an 8-channel fp16 pixel (channels 1..7 zero:
// siggraph17's ab hints and mask are zeros, siggraph17.py:129-132)
// y_lo > 0 (precise mode, HAVC_F_PRECISE): the value is stored as a hi / lo fp16 pair, the lo plane y_lo elements behind the hi plane
__device__ __forceinline__ void split_pair(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)((v - (float)hi) * 2048.f);
}
__global__ void prep_lab_l_kernel(const uint8_t* __restrict__ rgb, half_t* __restrict__ y, int y_cpitch, int y_coff, int64_t npix, int y_lo) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const float L = (float)rgb_to_L(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
        half8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ol[e] = (half_t)0.f;
        const float v = (L - 50.f) / 100.f;
        o[0] = (half_t)v;
        if (y_lo) {
            half_t a, b;
            split_pair(v, a, b);
            o[0] = a; ol[0] = b;
            *reinterpret_cast<half8*>(y + i * y_cpitch + y_coff + y_lo) = ol;
        }
        *reinterpret_cast<half8*>(y + i * y_cpitch + y_coff) = o;
    }
}

int launch_prep_lab_l(const uint8_t* rgb, half_t* y, int y_cpitch, int y_coff, int64_t npix, hipStream_t s, int y_lo) {
    if (int le = ensure_gray_lut(s)) return le;
    hipLaunchKernelGGL(prep_lab_l_kernel, dim3(grid_for(npix)), dim3(256), 0, s, rgb, y, y_cpitch, y_coff, npix, y_lo);
    return (int)hipGetLastError();
}

// postprocess_tens + the uint8 cast of colorize_frame: L of the ORIGINAL frame, ab = bilinear(ab map -> frame size)
// (identity when sizes match), lab2rgb in float64, uint8(clip(x*255, 0, 255)).
__device__ __forceinline__ void bilin_src_f(int dst, float scale, int in, int& i0, int& i1, float& l) {
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 > in - 1 ? in - 1 : i0;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = src - (float)i0;
}

__device__ __forceinline__ double lab_finv(double t) { return t > 0.2068966 ? t * t * t : (t - 16.0 / 116.0) / 7.787; }
__device__ __forceinline__ double linear_to_srgb(double c) { return c > 0.0031308 ? 1.055 * pow(c, 1.0 / 2.4) - 0.055 : c * 12.92; }

__global__ void zhang_post_kernel(const uint8_t* __restrict__ orig, const float2* __restrict__ ab, int abH, int abW,
                                  uint8_t* __restrict__ out, int n_frames, int w, int h, float sh, float sw,
                                  double m00, double m01, double m02, double m10, double m11, double m12, double m20, double m21,
                                  double m22) {
    const int64_t total = (int64_t)n_frames * w * h;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h), f = (int)(i / ((int64_t)w * h));
        const double L = (double)(float)rgb_to_L(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]);
        const float2* base = ab + (int64_t)f * abH * abW;
        float a, bb;
        if (abH == h && abW == w) {
            const float2 v = base[y * abW + x];
            a = v.x; bb = v.y;
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bilin_src_f(y, sh, abH, y0, y1, ly);
            bilin_src_f(x, sw, abW, x0, x1, lx);
            const float2 p00 = base[y0 * abW + x0], p01 = base[y0 * abW + x1], p10 = base[y1 * abW + x0], p11 = base[y1 * abW + x1];
            const float hy = 1.f - ly, hx = 1.f - lx;
            a = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
            bb = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
        }
        const double fy = (L + 16.0) / 116.0;
        const double fx = (double)a / 500.0 + fy;
        double fz = fy - (double)bb / 200.0;
        fz = fz < 0.0 ? 0.0 : fz;
        const double X = lab_finv(fx) * 0.95047, Y = lab_finv(fy), Z = lab_finv(fz) * 1.08883;
        const double r = linear_to_srgb(m00 * X + m01 * Y + m02 * Z);
        const double g = linear_to_srgb(m10 * X + m11 * Y + m12 * Z);
        const double b = linear_to_srgb(m20 * X + m21 * Y + m22 * Z);
        const double rc = fmin(fmax(r, 0.0), 1.0) * 255.0, gc = fmin(fmax(g, 0.0), 1.0) * 255.0, bc = fmin(fmax(b, 0.0), 1.0) * 255.0;
        out[i * 3] = (uint8_t)(int)rc; out[i * 3 + 1] = (uint8_t)(int)gc; out[i * 3 + 2] = (uint8_t)(int)bc;
    }
}

int launch_zhang_post(const uint8_t* orig, const float* ab, int abH, int abW, uint8_t* out, int n_frames, int w, int h,
                      hipStream_t s) {
    if (int le = ensure_gray_lut(s)) return le;
    // rgb_from_xyz = inv(xyz_from_rgb) exactly as skimage computes it (scipy.linalg.inv of the 3x3 sRGB matrix)
    static const double M[9] = {3.240481343200526, -1.5371515162713185, -0.4985363261688878,
                                -0.9692549499965682, 1.8759900014898907, 0.04155592655829284,
                                0.05564663913517716, -0.20404133836651123, 1.0573110696453443};
    hipLaunchKernelGGL(zhang_post_kernel, dim3(grid_for((int64_t)n_frames * w * h)), dim3(256), 0, s, orig, (const float2*)ab, abH, abW,
                       out, n_frames, w, h, (float)abH / (float)h, (float)abW / (float)w, M[0], M[1], M[2], M[3], M[4], M[5], M[6],
                       M[7], M[8]);
    return (int)hipGetLastError();
}

// ---- DDColor wrapper (oracle/ddcolor.py colorize_frame): the network sees the RGB rendering of Lab(L, 0, 0), imagenet-normalised ----
__constant__ double kRgbFromXyz[9] = {3.240481343200526, -1.5371515162713185, -0.4985363261688878,
                                      -0.9692549499965682, 1.8759900014898907, 0.04155592655829284,
                                      0.05564663913517716, -0.20404133836651123, 1.0573110696453443};
__device__ __forceinline__ void lab_to_rgb01(double L, double a, double bb, double& r, double& g, double& b) {
    const double fy = (L + 16.0) / 116.0;
    const double fx = a / 500.0 + fy;
    double fz = fy - bb / 200.0;
    fz = fz < 0.0 ? 0.0 : fz;
    const double X = lab_finv(fx) * 0.95047, Y = lab_finv(fy), Z = lab_finv(fz) * 1.08883;
    r = fmin(fmax(linear_to_srgb(kRgbFromXyz[0] * X + kRgbFromXyz[1] * Y + kRgbFromXyz[2] * Z), 0.0), 1.0);
    g = fmin(fmax(linear_to_srgb(kRgbFromXyz[3] * X + kRgbFromXyz[4] * Y + kRgbFromXyz[5] * Z), 0.0), 1.0);
    b = fmin(fmax(linear_to_srgb(kRgbFromXyz[6] * X + kRgbFromXyz[7] * Y + kRgbFromXyz[8] * Z), 0.0), 1.0);
}
__global__ void gray_lut_kernel() {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int k = threadIdx.x;
    const double L = rgb_to_L_full(k, k, k);
    g_gray_L[k] = L;
    double c[3];
    lab_to_rgb01(L, 0.0, 0.0, c[0], c[1], c[2]);
#pragma unroll
    for (int e = 0; e < 3; ++e) g_gray_dd[k][e] = ((float)c[e] - mean[e]) / stdv[e];
    g_gray_dd[k][3] = 0.f;
}
// filled once per device, synchronously, before the first kernel that reads it (under a mutex: contexts of one device may get here from several threads)
static int ensure_gray_lut(hipStream_t s) {
    static std::mutex mu;
    static uint64_t done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    if (done & bit) return 0;
    hipLaunchKernelGGL(gray_lut_kernel, dim3(1), dim3(256), 0, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    done |= bit;
    return 0;
}

__global__ void prep_ddcolor_kernel(const uint8_t* __restrict__ rgb, half_t* __restrict__ y, int y_cpitch, int y_coff, half_t* __restrict__ y2,
                                    int y2_cpitch, int y2_coff, int64_t npix, int precise) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int pr = rgb[i * 3], pg = rgb[i * 3 + 1], pb = rgb[i * 3 + 2];
        float nv[3];
        if (pr == pg && pg == pb) {
            nv[0] = g_gray_dd[pr][0]; nv[1] = g_gray_dd[pr][1]; nv[2] = g_gray_dd[pr][2];
        } else {
            double c[3];
            lab_to_rgb01(rgb_to_L_full(pr, pg, pb), 0.0, 0.0, c[0], c[1], c[2]);
#pragma unroll
            for (int e = 0; e < 3; ++e) nv[e] = ((float)c[e] - mean[e]) / stdv[e];
        }
        half8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ol[e] = (half_t)0.f;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float v = nv[e];
            o[e] = (half_t)v;
            if (precise) { half_t a, b; split_pair(v, a, b); o[e] = a; ol[e] = b; }
        }
        *reinterpret_cast<half8*>(y + i * y_cpitch + y_coff) = o;
        if (precise) *reinterpret_cast<half8*>(y + i * y_cpitch + y_coff + (y_cpitch >> 1)) = ol;        // precise: pixel row = [hi: P | lo: P]
        if (y2) {                                           // the refine conv's image slice
            half_t* q = y2 + i * y2_cpitch + y2_coff;
            if ((y2_coff & 7) == 0) {
                // an 8-channel chunk whose channels 3 .. 7 are zero pads in every plan: ONE 16-byte store (three 2-byte stores into an otherwise
                // untouched line cost the memory system a read-modify-write per pixel: this kernel was 1.7 ms per 128 frames at 512 x 512)
                *reinterpret_cast<half8*>(q) = o;
                if (precise) *reinterpret_cast<half8*>(q + (y2_cpitch >> 1)) = ol;
            } else {
                q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
                if (precise) { q += y2_cpitch >> 1; q[0] = ol[0]; q[1] = ol[1]; q[2] = ol[2]; }
            }
        }
    }
}
int launch_prep_ddcolor(const uint8_t* rgb, half_t* y, int y_cpitch, int y_coff, half_t* y2, int y2_cpitch, int y2_coff, int64_t npix,
                        hipStream_t s, int precise) {
    if (int le = ensure_gray_lut(s)) return le;
    hipLaunchKernelGGL(prep_ddcolor_kernel, dim3(grid_for(npix)), dim3(256), 0, s, rgb, y, y_cpitch, y_coff, y2, y2_cpitch, y2_coff, npix, precise);
    return (int)hipGetLastError();
}
// Lab(L of the original frame, ab from the network: fp16 NHWC channels 0, 1 at abH x abW, bilinear align_corners=False to the
// frame size when that differs) -> RGB u8, truncating cast of clip(x, 0, 1) * 255
// out_planes != NULL: the frame is written as three float (planes_half = 0) or half planes [3][h][w] in [0, 1] -- the RGBS / RGBH
// shape vs-deoldify hands to / takes from vsddcolor.ddcolor (vsslib/vsmodels.py:353-363) -- instead of interleaved u8.
__global__ void ddcolor_post_kernel(const uint8_t* __restrict__ orig, const half_t* __restrict__ ab, int ab_cpitch, int ab_coff, int abH,
                                    int abW, uint8_t* __restrict__ out, void* __restrict__ out_planes, int planes_half, int n_frames, int w, int h,
                                    float sh, float sw, int ab_lo) {
    // ab_lo > 0 (precise nets): the ab map is a hi / lo pair tensor, lo plane ab_lo elements behind the hi plane
    auto rd = [&](const half_t* p) -> float { return ab_lo ? (float)p[0] + (float)p[ab_lo] * (1.f / 2048.f) : (float)p[0]; };
    const int64_t total = (int64_t)n_frames * w * h;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h), f = (int)(i / ((int64_t)w * h));
        const double L = rgb_to_L(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]);
        const half_t* base = ab + (int64_t)f * abH * abW * ab_cpitch + ab_coff;
        float a, bb;
        if (abH == h && abW == w) {
            a = rd(base + ((int64_t)y * abW + x) * ab_cpitch);
            bb = rd(base + ((int64_t)y * abW + x) * ab_cpitch + 1);
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bilin_src_f(y, sh, abH, y0, y1, ly);
            bilin_src_f(x, sw, abW, x0, x1, lx);
            const half_t *p00 = base + ((int64_t)y0 * abW + x0) * ab_cpitch, *p01 = base + ((int64_t)y0 * abW + x1) * ab_cpitch,
                         *p10 = base + ((int64_t)y1 * abW + x0) * ab_cpitch, *p11 = base + ((int64_t)y1 * abW + x1) * ab_cpitch;
            const float hy = 1.f - ly, hx = 1.f - lx;
            a = hy * (hx * rd(p00) + lx * rd(p01)) + ly * (hx * rd(p10) + lx * rd(p11));
            bb = hy * (hx * rd(p00 + 1) + lx * rd(p01 + 1)) + ly * (hx * rd(p10 + 1) + lx * rd(p11 + 1));
        }
        double r, g, b;
        lab_to_rgb01(L, (double)a, (double)bb, r, g, b);
        if (out_planes) {
            const int64_t plane = (int64_t)w * h, o = (int64_t)f * 3 * plane + (int64_t)y * w + x;
            if (planes_half) {
                half_t* q = reinterpret_cast<half_t*>(out_planes);
                q[o] = (half_t)(float)r; q[o + plane] = (half_t)(float)g; q[o + 2 * plane] = (half_t)(float)b;
            } else {
                float* q = reinterpret_cast<float*>(out_planes);
                q[o] = (float)r; q[o + plane] = (float)g; q[o + 2 * plane] = (float)b;
            }
        } else {
            out[i * 3] = (uint8_t)(int)(r * 255.0); out[i * 3 + 1] = (uint8_t)(int)(g * 255.0); out[i * 3 + 2] = (uint8_t)(int)(b * 255.0);
        }
    }
}
int launch_ddcolor_post(const uint8_t* orig, const half_t* ab, int ab_cpitch, int ab_coff, int abH, int abW, uint8_t* out_u8, void* out_planes,
                        int planes_half, int n_frames, int w, int h, hipStream_t s, int ab_lo) {
    if (int le = ensure_gray_lut(s)) return le;
    hipLaunchKernelGGL(ddcolor_post_kernel, dim3(grid_for((int64_t)n_frames * w * h)), dim3(256), 0, s, orig, ab, ab_cpitch, ab_coff, abH, abW, out_u8,
                       out_planes, planes_half, n_frames, w, h, (float)abH / (float)h, (float)abW / (float)w, ab_lo);
    return (int)hipGetLastError();
}

// RGBS / RGBH planes in [0, 1] (full range, cast from RGB24 by zimg: exactly k / 255) -> interleaved u8: round to nearest, clamped
__global__ void planar_f_to_rgb8_kernel(const void* __restrict__ planes, int is_half, uint8_t* __restrict__ rgb, int64_t npix) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float v = is_half ? (float)reinterpret_cast<const half_t*>(planes)[p * npix + i] : reinterpret_cast<const float*>(planes)[p * npix + i];
            const float q = floorf(fminf(fmaxf(v, 0.f), 1.f) * 255.f + 0.5f);
            rgb[i * 3 + p] = (uint8_t)(int)q;
        }
    }
}
int launch_planar_f_to_rgb8(const void* planes, int is_half, uint8_t* rgb, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(planar_f_to_rgb8_kernel, dim3(grid_for(npix)), dim3(256), 0, s, planes, is_half, rgb, npix);
    return (int)hipGetLastError();
}

// ---- Pillow 8bpc resample: out = clip8((2^21 + sum_k px[xmin+k] * coef[k]) >> 22) ----
__global__ void pil_resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                    const int* __restrict__ kk, int ksize, int sw, int dw, int64_t rows) {
    const int64_t total = rows * dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw);
        const int64_t row = i / dw;
        const uint8_t* sp = src + row * sw * 3;
        const int xmin = bounds[x * 2], xmax = bounds[x * 2 + 1];
        const int* k = kk + (int64_t)x * ksize;
        int r = 1 << 21, g = 1 << 21, b = 1 << 21;
        for (int t = 0; t < xmax; ++t) {
            const int c = k[t];
            const uint8_t* p = sp + (xmin + t) * 3;
            r += p[0] * c; g += p[1] * c; b += p[2] * c;
        }
        r >>= 22; g >>= 22; b >>= 22;
        uint8_t* o = dst + i * 3;
        o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
        o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
}

__global__ void pil_resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                    const int* __restrict__ kk, int ksize, int sh, int dh, int w, int n_frames) {
    const int64_t total = (int64_t)n_frames * dh * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % dh), f = (int)(i / ((int64_t)w * dh));
        const int ymin = bounds[y * 2], ymax = bounds[y * 2 + 1];
        const int* k = kk + (int64_t)y * ksize;
        const uint8_t* base = src + ((int64_t)f * sh * w + x) * 3;
        int r = 1 << 21, g = 1 << 21, b = 1 << 21;
        for (int t = 0; t < ymax; ++t) {
            const int c = k[t];
            const uint8_t* p = base + (int64_t)(ymin + t) * w * 3;
            r += p[0] * c; g += p[1] * c; b += p[2] * c;
        }
        r >>= 22; g >>= 22; b >>= 22;
        uint8_t* o = dst + i * 3;
        o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
        o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
}

// horizontal pass (skipped when sw == dw) into tmp [n][sh][dw][3], vertical pass (skipped when sh == dh) into dst
int launch_pil_resize_passes(const uint8_t* src, int sw, int sh, uint8_t* tmp, uint8_t* dst, int dw, int dh, int n_frames,
                             const int* hb, const int* hk, int hks, const int* vb, const int* vk, int vks, hipStream_t s) {
    const uint8_t* mid = src;
    if (sw != dw) {
        uint8_t* o = (sh != dh) ? tmp : dst;
        hipLaunchKernelGGL(pil_resize_h_kernel, dim3(grid_for((int64_t)n_frames * sh * dw)), dim3(256), 0, s, src, o, hb, hk, hks, sw, dw,
                           (int64_t)n_frames * sh);
        mid = o;
    }
    if (sh != dh) {
        hipLaunchKernelGGL(pil_resize_v_kernel, dim3(grid_for((int64_t)n_frames * dh * dw)), dim3(256), 0, s, mid, dst, vb, vk, vks, sh, dh,
                           dw, n_frames);
    } else if (sw == dw) {
        (void)hipMemcpyAsync(dst, src, (size_t)n_frames * sh * sw * 3, hipMemcpyDeviceToDevice, s);
    }
    return (int)hipGetLastError();
}


// ---- ColorMNet frame wrapper (colormnet/colormnet_render.py:285-301 get_image, :276-279; dataset/range_transform.py:24-47): ----
// RGB2Lab = float32(skimage.color.rgb2lab(u8 image)) -> Normalize(mean [50,0,0], std [50,110,110]) as three fp32 planes, and back:
// inv_lll2rgb_trans ((x - [-1,0,0]) / [1/50, 1/110, 1/110] in fp32), skimage lab2rgb, clip(0, 1), * 255, truncate to u8.
__global__ void cmn_rgb_to_lab_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ lab, int64_t npix) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const double R = srgb_to_linear(rgb[i * 3] / 255.0), G = srgb_to_linear(rgb[i * 3 + 1] / 255.0), B = srgb_to_linear(rgb[i * 3 + 2] / 255.0);
        const double fx = lab_f((0.412453 * R + 0.357580 * G + 0.180423 * B) / 0.95047);
        const double fy = lab_f(0.212671 * R + 0.715160 * G + 0.072169 * B);
        const double fz = lab_f((0.019334 * R + 0.119193 * G + 0.950227 * B) / 1.08883);
        const float L = (float)(116.0 * fy - 16.0), a = (float)(500.0 * (fx - fy)), b = (float)(200.0 * (fy - fz));
        lab[i] = (L - 50.f) / 50.f;
        lab[npix + i] = a / 110.f;
        lab[2 * npix + i] = b / 110.f;
    }
}
int launch_cmn_rgb_to_lab(const uint8_t* rgb, float* lab, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(cmn_rgb_to_lab_kernel, dim3(grid_for(npix)), dim3(256), 0, s, rgb, lab, npix);
    return (int)hipGetLastError();
}
__global__ void cmn_lab_to_rgb_kernel(const float* __restrict__ lp, const float* __restrict__ ab, uint8_t* __restrict__ rgb, int64_t npix) {
    const float s50 = (float)(1 / 50.), s110 = (float)(1 / 110.);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const float L = (lp[i] - (-1.f)) / s50, a = ab[i] / s110, b = ab[npix + i] / s110;
        double r, g, bl;
        lab_to_rgb01((double)L, (double)a, (double)b, r, g, bl);
        rgb[i * 3] = (uint8_t)(r * 255.0);
        rgb[i * 3 + 1] = (uint8_t)(g * 255.0);
        rgb[i * 3 + 2] = (uint8_t)(bl * 255.0);
    }
}
int launch_cmn_lab_to_rgb(const float* l_plane, const float* ab, uint8_t* rgb, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(cmn_lab_to_rgb_kernel, dim3(grid_for(npix)), dim3(256), 0, s, l_plane, ab, rgb, npix);
    return (int)hipGetLastError();
}

// The frame transforms of ColorMNetRender WITH the padding InferenceCore puts around them (pad_divide_by 112, inference_core.py:49,123), so that
// no tensor op sits between the frame and the network: rgb -> normalised Lab planes [3][h][w] AND the network input, the L plane three times,
// zero-padded to [3][Hp][Wp] (offset pad_t / pad_l); and back: the L plane + the PADDED ab planes the decoder wrote -> u8 RGB of the frame.
__global__ void cmn_frame_in_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ lab, float* __restrict__ img, int w, int h, int Wp, int Hp,
                                    int pad_l, int pad_t) {
    const int64_t npix = (int64_t)w * h, nimg = (int64_t)Wp * Hp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nimg; i += (int64_t)gridDim.x * blockDim.x) {
        const int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
        const int y = yp - pad_t, x = xp - pad_l;
        float Ln = 0.f;
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) {
            const int64_t p = (int64_t)y * w + x;
            const double R = srgb_to_linear(rgb[p * 3] / 255.0), G = srgb_to_linear(rgb[p * 3 + 1] / 255.0), B = srgb_to_linear(rgb[p * 3 + 2] / 255.0);
            const double fx = lab_f((0.412453 * R + 0.357580 * G + 0.180423 * B) / 0.95047);
            const double fy = lab_f(0.212671 * R + 0.715160 * G + 0.072169 * B);
            const double fz = lab_f((0.019334 * R + 0.119193 * G + 0.950227 * B) / 1.08883);
            const float L = (float)(116.0 * fy - 16.0), a = (float)(500.0 * (fx - fy)), b = (float)(200.0 * (fy - fz));
            Ln = (L - 50.f) / 50.f;
            lab[p] = Ln;
            lab[npix + p] = a / 110.f;
            lab[2 * npix + p] = b / 110.f;
        }
        if (img) { img[i] = Ln; img[nimg + i] = Ln; img[2 * nimg + i] = Ln; }
    }
}
int launch_cmn_frame_in(const uint8_t* rgb, float* lab, float* img, int w, int h, int Wp, int Hp, int pad_l, int pad_t, hipStream_t s) {
    hipLaunchKernelGGL(cmn_frame_in_kernel, dim3(grid_for((int64_t)Wp * Hp)), dim3(256), 0, s, rgb, lab, img, w, h, Wp, Hp, pad_l, pad_t);
    return (int)hipGetLastError();
}
__global__ void cmn_frame_out_kernel(const float* __restrict__ lp, const float* __restrict__ ab, uint8_t* __restrict__ rgb, int w, int h, int Wp, int Hp,
                                     int pad_l, int pad_t) {
    const float s50 = (float)(1 / 50.), s110 = (float)(1 / 110.);
    const int64_t npix = (int64_t)w * h, nimg = (int64_t)Wp * Hp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        const int64_t q = (int64_t)(y + pad_t) * Wp + x + pad_l;
        const float L = (lp[i] - (-1.f)) / s50, a = ab[q] / s110, b = ab[nimg + q] / s110;
        double r, g, bl;
        lab_to_rgb01((double)L, (double)a, (double)b, r, g, bl);
        rgb[i * 3] = (uint8_t)(r * 255.0);
        rgb[i * 3 + 1] = (uint8_t)(g * 255.0);
        rgb[i * 3 + 2] = (uint8_t)(bl * 255.0);
    }
}
int launch_cmn_frame_out(const float* l_plane, const float* ab, uint8_t* rgb, int w, int h, int Wp, int Hp, int pad_l, int pad_t, hipStream_t s) {
    hipLaunchKernelGGL(cmn_frame_out_kernel, dim3(grid_for((int64_t)w * h)), dim3(256), 0, s, l_plane, ab, rgb, w, h, Wp, Hp, pad_l, pad_t);
    return (int)hipGetLastError();
}

// Eager module load (havc_create, under the library's set-up mutex): the HIP runtime loads a translation unit's code object on the first use
// of one of its kernels; querying one here moves that -- and the big-LDS opt-ins below -- out of the first launch, which may come from
// several host threads at once ("set-up is serialised").
void preload_zhang() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(cmn_lab_to_rgb_kernel)); (void)hipGetLastError(); }
