// Internal launcher declarations shared by the HIP translation units of libhavc_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;

struct ConvArgs {
    const half_t* x;       // input NHWC fp16
    const half_t* w;       // packed weights [Npad][Kc][8] fp16
    const float* bias;     // [Npad] or null
    const float* scale;    // [Npad] or null (AFFINE)
    const float* shift;
    const half_t* res;     // residual NHWC fp16 or null
    void* y;               // output (fp16 NHWC / transposed fp16 / u8 RGB)
    int x_cpitch, x_coff;  // elements
    int res_cpitch, res_coff;
    int y_cpitch, y_coff;
    int Hi, Wi, C8;        // input spatial, number of 8-channel chunks
    int Ho, Wo, Co;        // output spatial, channels stored (multiple of 8; 3 for RGB8)
    int kh, kw, stride, pad, dil;
    int pad_w;             // pad along W (== pad except for ConvTranspose parity sub-convs)
    int oss, ooy, oox;     // output pixel = (ho*oss + ooy, wo*oss + oox) of an (Ho*oss) x (Wo*oss) image (oss 1 or 2)
    int Kc, Npad;          // weight matrix: Npad rows, Kc chunks (multiple of 4)
    int M;                 // batch*Ho*Wo
    int flags;
    int pix_pitch;         // transposed store: elements per channel row
    float f0, f1, f2;      // sigmoid range lo/hi, leaky slope
    float mean[3], istd[3];
    int cfg;               // 0 = heuristic; otherwise forced tile configuration (tools/conv_bench.py A/B runs)
    const int2* ktab;      // [Kc] per 16-byte K chunk: .x = byte offset of (tap, cin chunk) from the tap-0 pixel,
                           //      .y = dh | dw << 16 (input-pixel displacement of the tap; pad entries: dh = 0x7fff)
    int C8a;               // chunks of the main K segment (plan.py pack_conv); == C8 when there is no remainder segment
    int tiles_y, tiles_x;  // PS_BLUR / halo conv: 16x16-pixel GEMM-row tiles per frame (stride 15: one row / column of halo)
    const float* fuse_w;   // FUSE_RGB8: fp32 [3][Npad] weights of the fused 1x1 conv, fuse_b: its 3 biases
    const float* fuse_b;
    uint8_t* fuse_rgb;     // FUSE_RGB8: u8 RGB output [M][3]
    int cstore;            // PS_BLUR: channels actually stored per pixel (the packed rows may be padded to a multiple of 64); 0 = all
    float* fuse_out;       // FUSE_PROJ: fp32 output [M][Npad / 256][2]; fuse_w = fp32 [frame][2][256]
    unsigned x_bytes;      // size of the input allocation (buffer descriptor range; OOB lanes read zeros)
    unsigned w_bytes;      // Npad * Kc * 16
    int prio;              // 1: static s_setprio 1 for the younger half of an 8-wave block (HAVC_SETPRIO, default on)
    int splitk;            // > 1: the K range is cut into `splitk` parts, one block per (tile, part) stores its fp32 partial sums to `ws`
    float* ws;             //      [splitk][M][Npad]; splitk_reduce_kernel adds them in a fixed order and runs the usual epilogue
    float pscale;          // HAVC_F_PRECISE: the accumulator is this factor away from the convolution (2^-11 x the weight pre-scale, op.f3)
    int ngroup, rband;     // round 6, set by the pipelined launchers (conv_raster): > 0 = column tiles are walked in groups of `ngroup` inside bands of `rband` row tiles,
                           //      so that the weight panels one XCD has in flight fit its 4 MiB L2 (a 768 -> 3072 GEMM cycles 4.7 MB of weights per row tile otherwise)
};
// grouped tile order for GEMMs whose weight matrix does not fit an XCD's L2.  Same bytes: only the block -> tile map changes.  MEASURED AND LEFT OFF (round 6,
// profiles/r6_raster_group_ab.txt): on ConvNeXt-L's stage-2 pwconv1 (768 -> 3072, 128 frames) it cuts the kernel's HBM-side traffic from 2.13 to 1.76 GB per launch
// (fetch 1.33 -> 0.96 GB) but the launch gets 2 % SLOWER (0.728 -> 0.745 ms; c3 1 339 -> 1 338 frames/s): six weight panels plus the pixel tiles of the 32 blocks an XCD
// has in flight are still more than its 4 MiB, and every pixel tile is now read twice.  HAVC_RASTER_GROUP=1 switches it on (A/B).
inline void conv_raster(ConvArgs& a, int MT, int NT, int BN) {
    static const int on = [] { const char* e = getenv("HAVC_RASTER_GROUP"); return e ? atoi(e) : 0; }();
    a.ngroup = a.rband = 0;
    const double tile_bytes = (double)BN * a.Kc * 16.0;            // one column tile's weight panel
    if (!on || a.splitk > 1 || NT < 4 || MT < 16 || tile_bytes * NT < 3.5e6) return;
    int ng = (int)(2.5e6 / tile_bytes);
    if (ng < 2 || ng >= NT) return;                                  // (a panel of > 1.25 MB per column tile: nothing to group; huge-K layers stream their weights anyway)
    a.ngroup = ng;
    a.rband = (MT + 7) / 8;                                          // one band per XCD share of the row tiles
}
#define HAVC_KTAB_PAD_DH 0x7fff

// returns hipError_t as int
int launch_conv(const ConvArgs& a, hipStream_t s);
int launch_conv_pipe(const ConvArgs& a, int cfg, hipStream_t s);
int launch_conv_pipe_ef(const ConvArgs& a, int cfg, hipStream_t s);   // compile-time epilogue flags; -1 = no such kernel (conv_igemm_pipe_ef.hip)
bool conv_splitk_cfg_ok(int cfg);      // tile configurations that may run with ConvArgs::splitk > 1 (the plain pipelined tiles)
const char* conv_config_name(const ConvArgs& a);

int launch_maxpool3x3s2(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch,
                        int x_coff, int y_cpitch, int y_coff, hipStream_t s);
int launch_blur_resize(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch,
                       int x_coff, int y_cpitch, int y_coff, hipStream_t s);
int launch_affine(const half_t* x, half_t* y, const float* scale, const float* shift, int relu, int64_t npix, int C,
                  int x_cpitch, int x_coff, int y_cpitch, int y_coff, hipStream_t s);
int launch_copy_ch(const half_t* x, half_t* y, int64_t npix, int C, int x_cpitch, int x_coff, int y_cpitch,
                   int y_coff, hipStream_t s);
int launch_prep_rgb8(const uint8_t* rgb, half_t* y0, int y0_cpitch, int y0_coff, half_t* y1, int y1_cpitch,
                     int y1_coff, int64_t npix, hipStream_t s, int y1_fill = 0);   // y1_fill: pad channels behind y1's 8 that may be zeroed too
int launch_subsample2(const half_t* x, half_t* y, int B, int Ho, int Wo, int Hi, int Wi, int C, int x_cpitch, int x_coff,
                      int y_cpitch, int y_coff, hipStream_t s);
int launch_proj2(const half_t* x, int x_cpitch, int x_coff, int C, const float* w, const float* bias, int mode, float mul,
                 float* out, int64_t npix, hipStream_t s);
int launch_bilinear2(const float* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, float mul, hipStream_t s);
int launch_prep_lab_l(const uint8_t* rgb, half_t* y, int y_cpitch, int y_coff, int64_t npix, hipStream_t s, int y_lo = 0);   // y_lo > 0: hi / lo pair (precise)
// Pillow 8bpc resample passes (integer coefficient tables from the host) and the Zhang post-process
int launch_pil_resize_passes(const uint8_t* src, int sw, int sh, uint8_t* tmp, uint8_t* dst, int dw, int dh, int n_frames,
                             const int* hb, const int* hk, int hks, const int* vb, const int* vk, int vks, hipStream_t s);
int launch_zhang_post(const uint8_t* orig, const float* ab, int abH, int abW, uint8_t* out, int n_frames, int w, int h,
                      hipStream_t s);
// attention: qk [B][N][qk_pitch] (f at f_coff, g at g_coff, d channels each), vT [B][dv][npitch],
// x/out NHWC; out = gamma*attn + x
int launch_attention(const half_t* qk, int qk_pitch, int f_coff, int g_coff, int d, const half_t* vT, int dv,
                     int npitch, const half_t* x, int x_cpitch, int x_coff, half_t* out, int o_cpitch, int o_coff,
                     int B, int N, float gamma, hipStream_t s);

// u8 colour filters (device pointers, interleaved RGB)
int launch_blend_u8(const uint8_t* a, const uint8_t* b, float w, uint8_t* out, int64_t nbytes, hipStream_t s);
int launch_yuv_merge(const uint8_t* color, const uint8_t* orig, uint8_t* out, int64_t npix, hipStream_t s);
int launch_chroma_stabilizer(const uint8_t* stable, const uint8_t* inew, double alpha, float weight, uint8_t* out,
                             int64_t npix, hipStream_t s);
int launch_luma_merge(const uint8_t* dark, const uint8_t* white, int mode, double tresh, double grad, uint8_t* out, int64_t npix,
                      hipStream_t s);
int launch_luma_sum(const uint8_t* img, unsigned long long* d_sum, int64_t npix, hipStream_t s);
int launch_chroma_temporal_limiter(const uint8_t* cur, const uint8_t* prv, double alpha, uint8_t* out, int64_t npix, hipStream_t s);
int launch_chroma_stabilizer_adaptive(const uint8_t* stable, const uint8_t* inew, float base_tol, float max_extra, float weight,
                                      uint8_t* out, int w, int h, hipStream_t s);
int launch_color_temporal_stabilizer(const uint8_t* const* frames, const double* weights, int n, uint8_t* out, int64_t npix,
                                     hipStream_t s);
// ddcolor.hip (+ the Lab wrapper kernels in zhang.hip)
int launch_prep_ddcolor(const uint8_t* rgb, half_t* y, int y_cpitch, int y_coff, half_t* y2, int y2_cpitch, int y2_coff, int64_t npix,
                        hipStream_t s, int precise = 0);
int launch_ddcolor_post(const uint8_t* orig, const half_t* ab, int ab_cpitch, int ab_coff, int abH, int abW, uint8_t* out_u8, void* out_planes,
                        int planes_half, int n_frames, int w, int h, hipStream_t s, int ab_lo = 0);          // ab_lo > 0: the ab map is a hi / lo pair tensor
int launch_planar_f_to_rgb8(const void* planes, int is_half, uint8_t* rgb, int64_t npix, hipStream_t s);
int launch_planar_to_rgb8(const uint8_t* planes, uint8_t* rgb, int64_t npix, hipStream_t s);
int launch_rgb8_to_planar(const uint8_t* rgb, uint8_t* planes, int64_t npix, hipStream_t s);
int launch_dwconv7(const half_t* x, const half_t* w, const float* bias, half_t* y, int B, int H, int W, int C, int x_cpitch, int x_coff,
                   int y_cpitch, int y_coff, int w_pitch, hipStream_t s);
bool dwconv7_ln_supported(int C);
int launch_dwconv7_ln(const half_t* x, const half_t* w, const float* bias, const float* gamma, const float* beta, float eps, half_t* y, int B,
                      int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch, hipStream_t s);
int launch_fold_queries(const half_t* e, int e_cpitch, int e_coff, int tok, const float* r, int r_pitch, int nq, float* out, int B, int C,
                        hipStream_t s);
int launch_shuf4_blur_ab(const float* proj, const half_t* img, int img_cpitch, int img_coff, const float* rimg, const float* bias, half_t* y,
                         int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s);
int launch_shuf4_blur_ab_p(const float* proj, const half_t* img, int img_cpitch, int img_coff, const float* rimg, const float* bias, half_t* y,
                         int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s);   // pair image, pair output (precise FUSE_PROJ tail)
int launch_layernorm_c(const half_t* x, half_t* y, const float* gamma, const float* beta, float eps, int64_t npix, int C, int x_cpitch,
                       int x_coff, int y_cpitch, int y_coff, hipStream_t s, int relu = 0);
int launch_mha32(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                 half_t* o, int o_cpitch, int o_coff, int o_tok, int B, int heads, int Lq, int Lk, float scale, hipStream_t s);
int mha32_nsplit(int Lk);
int launch_mha32_split(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok,
                       half_t* o, int o_cpitch, int o_coff, int o_tok, float* part, int B, int heads, int Lq, int Lk, float scale, hipStream_t s);
int launch_pixshuf4_blur(const half_t* x, half_t* y, int B, int Hi, int Wi, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                         hipStream_t s);
// tweaks.hip: image_tweak chain (Pillow hue shift, ImageEnhance Brightness / Contrast / Color, hue-range mask), Y table of
// luma_adjusted_levels, restore_color_gradient.
#define HAVC_MAX_HUE_RANGES 8
struct TweakArgs {
    int hue_offset;               // Pillow hue units (0..255 per turn); 0 = no hue shift
    float brightness, contrast, color;    // ImageEnhance factors; 1 = step skipped
    int mean_l;                   // Contrast: int(mean(L) + 0.5) of the image in front of the contrast step
    int n_ranges;                 // hue-range mask on the ORIGINAL image (degrees, strict inequalities); 0 = none
    double range_lo[HAVC_MAX_HUE_RANGES], range_hi[HAVC_MAX_HUE_RANGES];
};
int launch_image_tweak(const uint8_t* img, uint8_t* out, int64_t npix, const TweakArgs& a, unsigned long long* d_sum, bool sum_only,
                       hipStream_t s);
struct ChromaTweakArgs {
    int has_hue, has_adjust, has_hue2, has_sat2, n_ranges;
    double hue_half, satc, brightc, hue_half2, sat2c, weight;
    double range_lo[HAVC_MAX_HUE_RANGES], range_hi[HAVC_MAX_HUE_RANGES];
};
int launch_chroma_tweak(const uint8_t* img, uint8_t* out, int64_t npix, const ChromaTweakArgs& a, hipStream_t s);
int launch_luma_lut(const uint8_t* img, const uint8_t* d_lut, uint8_t* out, int64_t npix, hipStream_t s);
int launch_restore_color_gradient(const uint8_t* color, const uint8_t* gray, uint8_t* out, int64_t npix, double sat, int tht, double alpha,
                                  double weight, int algo, int return_mask, hipStream_t s);
// separable polyphase resample of interleaved u8 RGB (tap tables from the host; Spline64 = harness stand-in
// for zimg resize.Spline64).  orig != null fuses chroma_post_process (luma of orig, chroma of the resampled).
int launch_resize_passes(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int n_frames, float* tmp,
                         const int* h_start, const float* h_w, int h_taps, const int* v_start, const float* v_w,
                         int v_taps, const uint8_t* orig, hipStream_t s);

// ---- ColorMNet memory kernels (colormnet.hip) ----
int launch_mem_similarity(const float* mk, const float* ms, const float* qk, const float* qe, float* sim, int B, int CK, int N, int HW, hipStream_t s,
                          int64_t mpitch = 0);      // mpitch / vpitch: row pitch of mk / mv in elements (0 = N: the reference's contiguous tensors)
int launch_mem_usage(const int* idx, const float* wgt, unsigned long long* acc, float* usage, int B, int N, int HW, int K, hipStream_t s);
int launch_mem_dense_readout(const float* sim, const float* mv, float* out, int B, int CV, int N, int P, hipStream_t s);
int mem_topk_splits(int N);
int launch_mem_similarity_t(const float* mk, const float* ms, const float* qk, const float* qe, float* simT, int B, int CK, int N, int HW, hipStream_t s,
                            int64_t mpitch = 0);
bool mem_topk_select_supported(int N);
int launch_mem_topk_select_readout(const float* simT, const float* mv, int* idx, float* wgt, float* out, int B, int CV, int N, int HW, int K, hipStream_t s,
                                   int64_t vpitch = 0);
int launch_mem_topk_readout(const float* sim, const float* mv, int* idx, float* wgt, float* cand_val, int* cand_idx, float* out, int B, int CV, int N,
                            int HW, int K, hipStream_t s, int64_t vpitch = 0);
int launch_mem_usage_update(const int* idx, const float* wgt, unsigned long long* acc, float* use, float* life, int from, int N, int HW, int K, hipStream_t s);
int launch_cmn_value_in(const float* img, const float* planes, float* vin, int64_t P, hipStream_t s);
int launch_vec_add(float* y, const float* x, int64_t n, hipStream_t s);
int launch_cmn_frame_in(const uint8_t* rgb, float* lab, float* img, int w, int h, int Wp, int Hp, int pad_l, int pad_t, hipStream_t s);
int launch_cmn_frame_out(const float* l_plane, const float* ab, uint8_t* rgb, int w, int h, int Wp, int Hp, int pad_l, int pad_t, hipStream_t s);
int launch_local_correlation(const float* q, const float* k, float* out, int n, int C, int H, int W, int R, int dil, float qscale, hipStream_t s);
int launch_local_softmax(float* qk, const float* q, const float* rel_w, const float* rel_b, int n, int C, int H, int W, int R, int dil, hipStream_t s);
int launch_local_agg(const float* attn, const float* v, float* agg, int n, int CV, int H, int W, int R, int dil, hipStream_t s);

int launch_range_stats(const void* p, int elem_bytes, int64_t n, unsigned* stats, hipStream_t s);

// ---- eager module load + big-LDS opt-ins, one function per translation unit (called once per device from havc_create) ----
void preload_conv_pipe();
void preload_conv_pipe_ef();
void preload_conv_igemm();
void preload_elementwise();
void preload_zhang();
void preload_attention();
void preload_colorfilters();
void preload_tweaks();
void preload_ddcolor();
void preload_colormnet();
void preload_colormnet_net();
void preload_precise();

// ---- precise mode (precise.hip): the non-conv ops of the DeOldify generators on hi / lo fp16 pairs, fp32 arithmetic ----
int launch_prep_rgb8_p(const uint8_t* rgb, half_t* y0, int y0_cpitch, int y0_coff, half_t* y1, int y1_cpitch, int y1_coff, int64_t npix, hipStream_t s);
int launch_maxpool3x3s2_p(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                          hipStream_t s);
int launch_blur_resize_p(const half_t* x, half_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                         hipStream_t s);
int launch_affine_p(const half_t* x, half_t* y, const float* scale, const float* shift, int relu, int64_t npix, int C, int x_cpitch, int x_coff, int y_cpitch,
                    int y_coff, hipStream_t s);
bool attention_p_supported(int d, int C);
bool attention_pm_supported(int d, int C, int npitch);
int launch_attention_pm(const half_t* qk, int qk_cpitch, int f_coff, int g_coff, int d, int64_t qk_fs, const half_t* vT, int npitch, int64_t v_fs,
                        const half_t* x, int x_cpitch, int x_coff, int64_t x_fs, half_t* out, int o_cpitch, int o_coff, int64_t o_fs, float* stats, int B, int N,
                        int C, float gamma, hipStream_t s);
// precise2.hip: the non-conv ops of DDColor and the Zhang colorizers on hi / lo pairs
int launch_proj2_p(const half_t* x, int x_cpitch, int x_coff, int C, const float* w, const float* bias, int mode, float mul, float* out, int64_t npix,
                   hipStream_t s);
int launch_layernorm_p(const half_t* x, half_t* y, const float* gamma, const float* beta, float eps, int64_t npix, int C, int x_cpitch, int x_coff,
                       int y_cpitch, int y_coff, int relu, hipStream_t s);
bool dwconv7_ln_p_supported(int C);      // precise fused dwconv 7x7 + LayerNorm (ddcolor.hip): 192 / 384 / 768 channels
int launch_dwconv7_ln_p(const half_t* x, const float* w, const float* bias, const float* gamma, const float* beta, float eps, half_t* y, int B,
                        int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff, int w_pitch, hipStream_t s);
int launch_dwconv7_p(const half_t* x, const float* w, const float* bias, half_t* y, int B, int H, int W, int C, int x_cpitch, int x_coff, int y_cpitch,
                     int y_coff, int w_pitch, hipStream_t s);
int launch_mha32_p(const half_t* q, int q_cpitch, int q_coff, int q_tok, const half_t* kv, int kv_cpitch, int k_coff, int v_coff, int kv_tok, half_t* o,
                   int o_cpitch, int o_coff, int o_tok, int B, int heads, int Lq, int Lk, float scale, hipStream_t s);
int launch_fold_queries_p(const half_t* e, int e_cpitch, int e_coff, int tok, const float* r, int r_pitch, int nq, float* out, int B, int C, hipStream_t s);
int launch_shuf4_blur_proj_p(const half_t* x, int x_cpitch, int x_coff, const float* M, const half_t* img, int img_cpitch, int img_coff, const float* rimg,
                             const float* bias, half_t* y, int y_cpitch, int y_coff, int B, int Hi, int Wi, hipStream_t s);
int launch_subsample2_p(const half_t* x, half_t* y, int B, int Ho, int Wo, int Hi, int Wi, int C, int x_cpitch, int x_coff, int y_cpitch, int y_coff,
                        hipStream_t s);
void preload_precise2();
int launch_attention_p(const half_t* qk, int qk_cpitch, int f_coff, int g_coff, int d, int64_t qk_fs, const half_t* h, int h_cpitch, int h_coff, int64_t h_fs,
                       const half_t* x, int x_cpitch, int x_coff, int64_t x_fs, half_t* out, int o_cpitch, int o_coff, int64_t o_fs, float* stats, int B, int N,
                       int C, float gamma, hipStream_t s);

// ---- ColorMNet network kernels (colormnet_net.hip) ----
#ifndef HAVC_EW_SRC_BCAST          // (public values: include/havc_mi355.h)
#define HAVC_EW_SRC_BCAST 1
#define HAVC_EW_RES 2
#define HAVC_EW_RES_BCAST 4
#define HAVC_EW_RELU 8
#define HAVC_EW_DUAL 16
#endif
struct EwArgs {
    const half_t* x; half_t* y; const half_t* res; half_t* y2;
    int B, Hi, Wi, Ho, Wo, C8;
    int x_cp, x_co, y_cp, y_co, r_cp, r_co, y2_cp, y2_co;
    int64_t x_fs, y_fs, r_fs, y2_fs;          // frame strides in elements (0 = the same frame for every batch entry)
    int mode, factor, flags;                  // mode 0 copy, 1 bilinear (align_corners = False), 2 area (factor x factor mean)
    float rh, rw;                             // bilinear source / destination ratios
};
int launch_ew(const EwArgs& a, hipStream_t s);
int launch_dwconv(const half_t* x, const half_t* w, const float* bias, half_t* y, int B, int H, int W, int C, int K, int x_cp, int x_co, int64_t x_fs,
                  int y_cp, int y_co, int64_t y_fs, int w_pitch, hipStream_t s);
int chan_attn_splits(int P, int heads, int c);
int launch_chan_attn(const half_t* q, int q_cp, int q_co, int64_t q_fs, const half_t* k, int k_cp, int k_co, int64_t k_fs, const float* temp,
                     float* part_g, float* part_n, half_t* wout, int64_t w_fs, int w_pitch, int B, int P, int heads, int c, hipStream_t s);
int launch_mha64(const half_t* qkv, int cp, int q_co, int k_co, int v_co, int tok, half_t* o, int o_cp, int o_co, int o_tok, int B, int heads, int L,
                 float scale, hipStream_t s);
int launch_cbam(const half_t* x, int cp, int co, int64_t fs, int B, int H, int W, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                const float* w7, const float* b7, float* scale, float* comp, half_t* y, int y_cp, int y_co, int64_t y_fs, half_t* y2, int y2_cp, int y2_co,
                int64_t y2_fs, hipStream_t s);
int launch_gru(const half_t* v, int cp, int co, int64_t fs, const float* h, float* out, int B, int P, int hd, hipStream_t s);
int launch_cmn_decoder_in(const half_t* g, int g_cp, int g_co, const float* ro, int64_t ro_fs, const float* hid, int64_t hid_fs, half_t* y, half_t* y2, int cp,
                          int co, int64_t fs, int B, int P, int Cg, int CV, int HD, hipStream_t s);
int launch_planar_in(const float* x, int64_t x_fs, half_t* y, int cp, int co, int64_t fs, int B, int P, int C, int span, int pixel_major, int bcast,
                     hipStream_t s);
int launch_planar_out(const half_t* x, int cp, int co, int64_t fs, float* y, int64_t y_fs, int B, int P, int C, int act, hipStream_t s);
// the frame wrapper of ColorMNetRender (zhang.hip: skimage Lab formulas in fp64)
int launch_cmn_rgb_to_lab(const uint8_t* rgb, float* lab, int64_t npix, hipStream_t s);
int launch_cmn_lab_to_rgb(const float* l_plane, const float* ab, uint8_t* rgb, int64_t npix, hipStream_t s);
