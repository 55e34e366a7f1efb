// HBM-bound u8 per-pixel filters of SURVEY.md §8 a17 / a19: image_tweak (Pillow ImageEnhance chain + hue shift + hue-range
// mask), the Y look-up table of luma_adjusted_levels, restore_color_gradient (ChromaRetentionMerge).  Interleaved RGB in
// HBM, one thread per pixel, everything in registers.  Arithmetic restates Pillow's C (libImaging Convert.c / Blend.c:
// float locals, double intermediates, truncating casts) and OpenCV's 8-bit HSV integer path; see oracle/tweaks.py,
// oracle/cvcolor.py.  Built with -ffp-contract=off: Pillow's / numpy's products are rounded before the add.
#include "kernels.h"

#pragma clang fp contract(off)

static inline int grid_for_px(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

__device__ __forceinline__ int clip8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---- Pillow rgb2hsv_row / hsv2rgb (Convert.c, "following colorsys.py") ----
__device__ __forceinline__ void pil_rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = clip8i((int)((double)h * 255.0));
    us = clip8i((int)((double)s * 255.0));
}
__device__ __forceinline__ void pil_hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) { r = g = b = v; return; }
    const double hd = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hd);
    const float f = (float)(hd - (double)(float)i);
    const float fs = (float)((double)(float)s / 255.0);
    const double vd = (double)(float)v;
    const int p = clip8i((int)round(vd * (1.0 - (double)fs)));
    const int q = clip8i((int)round(vd * (1.0 - (double)fs * (double)f)));
    const int t = clip8i((int)round(vd * (1.0 - (double)fs * (1.0 - (double)f))));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}
// Pillow ImagingBlend(im1 = degenerate a, im2 = image b, alpha) incl. the extrapolating branch (ImageEnhance factors > 1)
__device__ __forceinline__ int pil_blendx(int a, int b, float alpha) {
    const float prod = alpha * (float)(b - a);
    const float t = (float)a + prod;
    if (alpha >= 0.f && alpha <= 1.f) return (int)(uint8_t)(int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int pil_L(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16; }

// ---- OpenCV RGB2HSV_b (hrange 180): 12-bit reciprocal tables built with cvRound ----
__device__ __forceinline__ void cv_rgb2hsv(int r, int g, int b, int& h, int& s, int& v) {
    v = max(r, max(g, b));
    const int vmin = min(r, min(g, b)), diff = v - vmin;
    const int sdiv = v ? __double2int_rn((double)(255 << 12) / (double)v) : 0;
    const int hdiv = diff ? __double2int_rn((double)(180 << 12) / (6.0 * (double)diff)) : 0;
    s = (diff * sdiv + (1 << 11)) >> 12;
    int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    hh = (hh * hdiv + (1 << 11)) >> 12;
    h = hh < 0 ? hh + 180 : hh;
}
// OpenCV HSV2RGB_b -> HSV2RGB_f (float32 throughout, reciprocal multiplies), saturate_cast<uchar>(cvRound(x * 255))
__device__ __forceinline__ void cv_hsv2rgb(int h8, int s8, int v8, int& r, int& g, int& b) {
    const float hscale = 6.0f / 180.0f, inv255 = 1.0f / 255.0f;
    float h = (float)h8 * hscale;
    const float s = (float)s8 * inv255, v = (float)v8 * inv255;
    if (s8 == 0) { r = g = b = clip8i((int)rintf(v * 255.0f)); return; }
    if (h >= 6.0f) h -= 6.0f;
    const int i = (int)floorf(h);
    const float f = h - (float)i;
    const float p = v * (1.f - s), q = v * (1.f - s * f), t = v * (1.f - s * (1.f - f));
    float rf, gf, bf;
    switch (i) {
        case 0: rf = v; gf = t; bf = p; break;
        case 1: rf = q; gf = v; bf = p; break;
        case 2: rf = p; gf = v; bf = t; break;
        case 3: rf = p; gf = q; bf = v; break;
        case 4: rf = t; gf = p; bf = v; break;
        default: rf = v; gf = p; bf = q; break;
    }
    r = clip8i((int)rintf(rf * 255.0f)); g = clip8i((int)rintf(gf * 255.0f)); b = clip8i((int)rintf(bf * 255.0f));
}

// ---- image_tweak (imfilters.py:463-504) ----
// SUM_L: stop in front of the contrast step and accumulate the Pillow-L sum of the intermediate image (ImageEnhance.Contrast
// needs its mean); otherwise the whole chain.
template <bool SUM_L>
__global__ void image_tweak_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, int64_t npix, TweakArgs a,
                                   unsigned long long* __restrict__ sum) {
    unsigned long long local = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int r0 = img[i * 3], g0 = img[i * 3 + 1], b0 = img[i * 3 + 2];
        int r = r0, g = g0, b = b0;
        if (a.hue_offset != 0) {
            int h, s, v;
            pil_rgb2hsv(r, g, b, h, s, v);
            h = ((h + a.hue_offset) % 256 + 256) % 256;           // numpy int16 %: non-negative
            pil_hsv2rgb(h, s, v, r, g, b);
        }
        if (a.brightness != 1.f) { r = pil_blendx(0, r, a.brightness); g = pil_blendx(0, g, a.brightness); b = pil_blendx(0, b, a.brightness); }
        if (SUM_L) { local += (unsigned)pil_L(r, g, b); continue; }
        if (a.contrast != 1.f) { r = pil_blendx(a.mean_l, r, a.contrast); g = pil_blendx(a.mean_l, g, a.contrast); b = pil_blendx(a.mean_l, b, a.contrast); }
        if (a.color != 1.f) {
            const int L = pil_L(r, g, b);
            r = pil_blendx(L, r, a.color); g = pil_blendx(L, g, a.color); b = pil_blendx(L, b, a.color);
        }
        if (a.n_ranges > 0) {                                     // np_adjust_chroma2: tweaked pixel only inside the hue ranges of the ORIGINAL
            int h, s, v;
            cv_rgb2hsv(r0, g0, b0, h, s, v);
            bool cond = false;
            for (int k = 0; k < a.n_ranges; ++k) cond |= ((double)h > a.range_lo[k] * 0.5) && ((double)h < a.range_hi[k] * 0.5);
            if (!cond) { r = r0; g = g0; b = b0; }
        }
        out[i * 3] = (uint8_t)r; out[i * 3 + 1] = (uint8_t)g; out[i * 3 + 2] = (uint8_t)b;
    }
    if (SUM_L) {
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
        if ((threadIdx.x & 63) == 0 && local) atomicAdd(sum, local);
    }
}

int launch_image_tweak(const uint8_t* img, uint8_t* out, int64_t npix, const TweakArgs& a, unsigned long long* d_sum, bool sum_only,
                       hipStream_t s) {
    if (sum_only) {
        hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((image_tweak_kernel<true>), dim3(grid_for_px(npix)), dim3(256), 0, s, img, out, npix, a, d_sum);
    } else {
        hipLaunchKernelGGL((image_tweak_kernel<false>), dim3(grid_for_px(npix)), dim3(256), 0, s, img, out, npix, a, d_sum);
    }
    return (int)hipGetLastError();
}

// ---- luma_adjusted_levels (imfilters.py:335-372): cv2 YUV, Y through a 256-entry table, back ----
__device__ __forceinline__ int descale14t(int x) { return (x + (1 << 13)) >> 14; }
__global__ void luma_lut_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lut, uint8_t* __restrict__ out, int64_t npix) {
    __shared__ uint8_t sl[256];
    sl[threadIdx.x & 255] = lut[threadIdx.x & 255];
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
        int y = descale14t(r * 4899 + g * 9617 + b * 1868);
        const int u = clip8i(descale14t((b - y) * 8061 + (128 << 14))) - 128;
        const int v = clip8i(descale14t((r - y) * 14369 + (128 << 14))) - 128;
        y = sl[clip8i(y)];
        out[i * 3] = (uint8_t)clip8i(y + descale14t(v * 18678));
        out[i * 3 + 1] = (uint8_t)clip8i(y + descale14t(u * -6472 + v * -9519));
        out[i * 3 + 2] = (uint8_t)clip8i(y + descale14t(u * 33292));
    }
}
int launch_luma_lut(const uint8_t* img, const uint8_t* d_lut, uint8_t* out, int64_t npix, hipStream_t s) {
    hipLaunchKernelGGL(luma_lut_kernel, dim3(grid_for_px(npix)), dim3(256), 0, s, img, d_lut, out, npix);
    return (int)hipGetLastError();
}

// ---- restore_color_gradient (restcolor.py:98-217) ----
__global__ void restore_color_gradient_kernel(const uint8_t* __restrict__ color, const uint8_t* __restrict__ gray, uint8_t* __restrict__ out,
                                              int64_t npix, double sat, int tht, double alpha, double weight, int algo, int return_mask) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int gr = gray[i * 3], gg = gray[i * 3 + 1], gb = gray[i * 3 + 2];
        int h, s, v;
        cv_rgb2hsv(gr, gg, gb, h, s, v);
        // gradient mask from the saturation of the "gray" image: white (255) where it is gray
        int mask;
        if (algo == 0) {
            const double sd = (double)s;
            const double grad = s < tht ? 2.0 * sd / alpha - (double)tht : 2.0 * (sd - (double)tht) * alpha;
            double m = 255.0 - (double)tht - grad;
            m = m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m);
            mask = (int)m;
        } else {
            const int t = tht < 0 ? 0 : (tht > 255 ? 255 : tht);
            if (t == 0) mask = 0;
            else {
                const float sf = (float)s;
                double mn;
                if (algo == 1) {
                    const float max_s = (float)min(2 * t, 200);
                    const float sc = fminf(fmaxf(sf, 0.f), max_s);
                    mn = (double)powf(1.0f - sc / max_s, (float)alpha);
                } else {
                    const float s_rel = fminf(fmaxf(sf / (float)t, 0.f), 2.f);
                    mn = exp((double)((float)(-alpha) * s_rel) * 0.6931471805599453);
                    if (sf >= (float)(2 * t)) mn = 0.0;
                }
                double m = mn * 255.0;
                m = m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m);
                mask = (int)m;
            }
        }
        if (return_mask) { out[i * 3] = out[i * 3 + 1] = out[i * 3 + 2] = (uint8_t)mask; continue; }
        int cr = color[i * 3], cg = color[i * 3 + 1], cb = color[i * 3 + 2];
        {
            int ch, cs, cv;
            cv_rgb2hsv(cr, cg, cb, ch, cs, cv);
            if (sat != 1.0) {
                const double sc = sat < 0.0 ? 0.0 : (sat > 10.0 ? 10.0 : sat);
                cs = (int)(uint8_t)(long long)((double)cs * sc);          // numpy float64 -> uint8 assignment (wraps above 255)
            }
            cv_hsv2rgb(ch, cs, cv, cr, cg, cb);
        }
        const double mw = (double)mask / 255.0, mb = 1.0 - mw;
        const int src_g[3] = {gr, gg, gb}, src_c[3] = {cr, cg, cb};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double m = (double)src_g[k] * mb + (double)src_c[k] * mw;
            int o = (int)(m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m));
            if (weight > 0.0) {
                m = (double)o * (1.0 - weight) + (double)src_c[k] * weight;
                o = (int)(m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m));
            }
            if (weight < 0.0) {
                m = (double)o * (1.0 - (-weight)) + (double)src_g[k] * (-weight);
                o = (int)(m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m));
            }
            out[i * 3 + k] = (uint8_t)o;
        }
    }
}
int launch_restore_color_gradient(const uint8_t* color, const uint8_t* gray, uint8_t* out, int64_t npix, double sat, int tht, double alpha,
                                  double weight, int algo, int return_mask, hipStream_t s) {
    hipLaunchKernelGGL(restore_color_gradient_kernel, dim3(grid_for_px(npix)), dim3(256), 0, s, color, gray, out, npix, sat, tht, alpha, weight,
                       algo, return_mask);
    return (int)hipGetLastError();
}

// ---- image_chroma_tweak (imfilters.py:540-548 -> restcolor.py:288-350): cv2 HSV hue / saturation / value tweak, then the optional
// "hue_adjust" stage (hue range on the TWEAKED hue -> re-tweaked colour, everything else the ORIGINAL pixel, weighted merges) ----
__device__ __forceinline__ int cv_hue_add(int h, double hue_half) {          // nputils.py:330-340 + the uint8 slice assignment
    double t = (double)h + hue_half;
    t = t > 180.0 ? t - 180.0 : t;
    t = t < 0.0 ? t + 180.0 : t;
    return (int)(uint8_t)(long long)t;
}
__device__ __forceinline__ int wmerge1(int a, int b, double w) {
    const double m = (double)a * (1.0 - w) + (double)b * w;
    return (int)(m < 0.0 ? 0.0 : (m > 255.0 ? 255.0 : m));
}
__global__ void chroma_tweak_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, int64_t npix, ChromaTweakArgs a) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int r0 = img[i * 3], g0 = img[i * 3 + 1], b0 = img[i * 3 + 2];
        int h, s, v;
        cv_rgb2hsv(r0, g0, b0, h, s, v);
        if (a.has_hue) h = cv_hue_add(h, a.hue_half);
        s = (int)(uint8_t)(long long)((double)s * a.satc);
        v = (int)(uint8_t)(long long)((double)v * a.brightc);
        int r, g, b;
        if (a.has_adjust == 2) { r = r0; g = g0; b = b0; }      // adjust_chroma (restcolor.py:243-286): no tweak stage, no HSV round trip in front
        else cv_hsv2rgb(h, s, v, r, g, b);
        if (a.has_adjust) {
            int hg, sg, vg;
            cv_rgb2hsv(r, g, b, hg, sg, vg);
            if (a.has_hue2) hg = cv_hue_add(hg, a.hue_half2);
            if (a.has_sat2) sg = (int)(uint8_t)(long long)((double)sg * a.sat2c);
            int gr, gg, gb;
            cv_hsv2rgb(hg, sg, vg, gr, gg, gb);
            bool cond = false;
            for (int k = 0; k < a.n_ranges; ++k) cond |= ((double)h > a.range_lo[k] * 0.5) && ((double)h < a.range_hi[k] * 0.5);
            r = cond ? gr : r0; g = cond ? gg : g0; b = cond ? gb : b0;
            if (a.weight > 0.0) {
                const bool to_gray = !a.has_hue2;
                r = wmerge1(r, to_gray ? gr : r0, a.weight); g = wmerge1(g, to_gray ? gg : g0, a.weight); b = wmerge1(b, to_gray ? gb : b0, a.weight);
            }
            if (a.weight < 0.0) { r = wmerge1(r, r0, -a.weight); g = wmerge1(g, g0, -a.weight); b = wmerge1(b, b0, -a.weight); }
        }
        out[i * 3] = (uint8_t)r; out[i * 3 + 1] = (uint8_t)g; out[i * 3 + 2] = (uint8_t)b;
    }
}
int launch_chroma_tweak(const uint8_t* img, uint8_t* out, int64_t npix, const ChromaTweakArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(chroma_tweak_kernel, dim3(grid_for_px(npix)), dim3(256), 0, s, img, out, npix, a);
    return (int)hipGetLastError();
}

// Eager module load (havc_create, under the library's set-up mutex): the HIP runtime loads a translation unit's code object on the first use
// of one of its kernels; querying one here moves that -- and the big-LDS opt-ins below -- out of the first launch, which may come from
// several host threads at once ("set-up is serialised").
void preload_tweaks() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(chroma_tweak_kernel)); (void)hipGetLastError(); }
