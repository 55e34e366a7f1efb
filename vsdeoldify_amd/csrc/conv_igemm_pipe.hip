// Implicit-GEMM convolution, software-pipelined main loop: the kernel the big decoder convs run on.
//
// History in profiles/r1_conv_ablation.txt.  What the measurements forced:
//  * a wave issues in order: address VALU / ds_read / LDS-DMA issue must be INTERLEAVED with the MFMAs in program
//    order or they add to the matrix time instead of hiding under it (ring kernels: MfmaUtil 38 %, zero overlap);
//  * LDS-DMA moves 64-byte row segments at 3.65 TB/s chip-wide but 128-byte segments at 8.5 TB/s, and a row pitch that
//    is not a multiple of 128 B halves it again  =>  K-step 64 with ONE 128-byte line per tile row per stage, channel
//    pitch of every activation buffer a multiple of 64 (plan.py), K ordered so a stage never straddles a tap.
//
// Structure: 256 x 256 block tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64, K-step 64, 2-stage LDS double buffer
// (2 x 64 KiB), ONE barrier per stage.  A stage is 16 steps of 4 MFMAs (v_mfma_f32_16x16x32_f16); each step also
// carries, pinned with sched_barrier: the ds_read of the pixel fragment two steps ahead (3-deep register ring), a weight
// fragment of the second K half, and one 1-KiB LDS-DMA piece (8 rows x 128 B) of the NEXT stage.
// LDS-DMA = buffer_load_dwordx4 ... lds through buffer descriptors: lanes on padding / M tail / K tail / N tail use an
// out-of-range offset and the hardware writes zeros (no zero page, no 64-bit pointer math, no branches).
// The im2col geometry comes from a host-built per-chunk table (ConvArgs::ktab), so stride, dilation, any channel count
// and split K orderings need no per-lane state machine.
// LDS rows are 128 B = 8 chunks; chunk c of row r sits at position c ^ ((r >> 1) & 7): conflict-free for the
// ds_read_b128 lane groups when 16 consecutive rows read chunk (ks*4 + lane>>4).  DMA writes are lane-linear, so the
// involution is applied to the SOURCE chunk each DMA lane fetches.
// EXTRA = 1: 16 extra output columns (the 259-channel tail: Npad = 256 + 16) as 2 more MFMAs per wave per K half.
#include "conv_common.h"
#include <type_traits>

namespace {

constexpr int FN = 4;                         // 16-channel fragments per wave: every wave owns 64 output channels
constexpr unsigned OOB = 0xF0000000u;         // voffset beyond every descriptor range -> DMA writes zeros

// Tile geometry: WM x WN waves, each wave (FM*16 pixels) x 64 channels.
//   <2,4,8>: 256 x 256, 8 waves  (the big decoder / tail convs)
//   <2,2,4>: 128 x 128, 4 waves  (encoder / bottleneck layers: more, smaller tiles; 2 blocks per CU)
//   <1,4,8>: 128 x 256, 4 waves  (few pixels, many channels: the 18x18 / 35x35 layers)
template <int WM, int WN, int FM, int EXTRA>
struct Geo {
    static constexpr int NW = WM * WN;
    static constexpr int BM = WM * FM * 16, BN = WN * 64;
    static constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW;      // 1-KiB DMA pieces per wave per stage
    static constexpr int NS = 2 * FM;                                 // steps per stage
    static constexpr int PIECES = A_IT + B_IT;
    static constexpr int ROWS = BM + BN + 16 * EXTRA;
    static constexpr int STAGE_BYTES = ROWS * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0, "pieces must divide over an even number of waves");
    static_assert(FM >= FN, "second-half weight fragments are prefetched during the first FM steps");
    static_assert(!EXTRA || (WM == 2 && WN == 4 && FM == 8), "extra columns: 256x256 tile only");
    static_assert(NW * FM * 2048 <= LDS_BYTES, "LDS epilogue image");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}

}  // namespace

// ABL (profiling only, wrong results): 1 = no DMA inside the loop, 4 = no MFMA, 5 = no epilogue
// (Tried and dropped: staggering the DMA slots of the two waves sharing a SIMD -- 3-12 % slower, profiles/r1_conv_ablation.txt.)
template <int WM, int WN, int FM, int EXTRA, int ABL = 0>
__global__ void __launch_bounds__(WM* WN * 64) conv_pipe_kernel(const ConvArgs p) {
    using G = Geo<WM, WN, FM, EXTRA>;
    constexpr int NW = G::NW, BM = G::BM, BN = G::BN, A_IT = G::A_IT, B_IT = G::B_IT, NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;

    const int nwg = gridDim.x;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int NT = (p.Npad - 16 * EXTRA + BN - 1) / BN;
    const int m0 = (ABL == 9 ? (pid / NT) % 64 : pid / NT) * BM;      // ABL 9: L2-resident source, DMA ceiling
    const int n0 = (pid % NT) * BN;
    const bool has_extra = EXTRA && (n0 + BN + 16 == p.Npad);
    const int HoWo = p.Ho * p.Wo;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- DMA lane role: row lane>>3 of an 8-row piece, LDS position lane&7, source chunk c = pos ^ f(row) ----
    // wave w owns pieces w, w+NW, w+2NW, ... (same parity => same f) of both the pixel and the weight tile.
    const int r8 = lane >> 3;
    const int c = (lane & 7) ^ ((((wave & 1) << 2) + (r8 >> 1)) & 7);

    unsigned a_base[A_IT];   // byte offset of (pixel row, tap 0, channel 0) in the input buffer (wraps when hi0/wi0 < 0)
    int a_hw0[A_IT];         // hi0 | wi0 << 16 (signed 16-bit each)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (wave + it * NW) * 8 + r8;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        int ho = rem / p.Wo;
        int wo = rem - ho * p.Wo;
        if (p.flags & HAVC_F_PS_BLUR) {                    // GEMM rows = 16x16 pixel tiles, origin (15 ty - 1, 15 tx - 1), clamped
            const int tile = m0 / BM, tt = p.tiles_y * p.tiles_x;
            b = tile / tt;
            const int t = tile - b * tt, ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
            ho = min(max(ty * 15 - 1 + (row >> 4), 0), p.Ho - 1);
            wo = min(max(tx * 15 - 1 + (row & 15), 0), p.Wo - 1);
        }
        const int hi0 = ok ? ho * p.stride - p.pad : -16384, wi0 = wo * p.stride - p.pad_w;
        a_hw0[it] = (hi0 & 0xffff) | (wi0 << 16);
        a_base[it] = (unsigned)((((int64_t)(b * p.Hi + hi0) * p.Wi + wi0) * p.x_cpitch + p.x_coff) * 2);
    }
    unsigned b_voff[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + (wave + it * NW) * 8 + r8;
        b_voff[it] = n < p.Npad ? (unsigned)((n * p.Kc + c) * 16) : OOB;
    }
    // extra 16 weight rows = pieces 32, 33 -> waves 6, 7
    const unsigned x_voff = (unsigned)(((n0 + BN + (wave & 1) * 8 + r8) * p.Kc + c) * 16);
    const bool extra_wave = has_extra && wave >= 6;

    const int KT = p.Kc >> 3;
    const int2* kt_lane = p.ktab + c;                  // this lane's chunk of every stage: kt_lane[kt * 8]

    auto a_voff = [&](int it, int2 e) -> unsigned {
        const int hi = (short)(a_hw0[it] & 0xffff) + (short)(e.y & 0xffff);
        const int wi = (a_hw0[it] >> 16) + (e.y >> 16);
        const bool ok = ((unsigned)hi < (unsigned)p.Hi) & ((unsigned)wi < (unsigned)p.Wi);
        return ok ? a_base[it] + (unsigned)e.x : OOB;
    };

    // ---- fragment read addresses (bytes): per-lane constants + immediates; the second K half is base ^ 64 ----
    const int fsw = (lr >> 1) & 7;
    const int a_l0 = (wm * (FM * 16) + lr) * 128 + ((lg ^ fsw) << 4), a_l1 = a_l0 ^ 64;
    const int b_l0 = BM * 128 + (wn * 64 + lr) * 128 + ((lg ^ fsw) << 4), b_l1 = b_l0 ^ 64;
    const int x_l0 = (BM + BN + lr) * 128 + ((lg ^ fsw) << 4), x_l1 = x_l0 ^ 64;

    float4v acc[FN][FM];
    float4v accx[2];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
    accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: all of stage 0, the first two pieces of stage 1, the first fragments of stage 0 ----
    constexpr int P = G::PIECES;
    constexpr int EARLY = 2;                               // pieces of stage kt+2 issued in the last two steps of stage kt
    constexpr int NSD = NS / 2;                            // the other pieces go out in the first NSD steps of stage kt+1
    auto dma_piece = [&](int q, char* buf, int2 e, int kstage) {
        if (q < A_IT) dma16(rx, buf + (wave + q * NW) * 1024, a_voff(q, e), 0);
        else dma16(rw, buf + BM * 128 + (wave + (q - A_IT) * NW) * 1024, b_voff[q - A_IT], (unsigned)kstage * 128u);
    };
    int2 e_nx = kt_lane[0];
#pragma unroll
    for (int q = 0; q < P; ++q) dma_piece(q, smem, e_nx, 0);
    if (EXTRA && extra_wave) dma16(rw, smem + (BM + BN) * 128 + (wave & 1) * 1024, x_voff, 0);
    e_nx = kt_lane[(KT > 1 ? 1 : 0) * 8];
    if (KT > 1) {
#pragma unroll
        for (int q = 0; q < EARLY; ++q) dma_piece(q, smem + G::STAGE_BYTES, e_nx, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(e_nx.x), "+v"(e_nx.y));         // (see the note at the stage barrier)
    __builtin_amdgcn_s_barrier();

    half8 bf[2][FN];                                   // weight fragments, K half 0 / 1
    half8 xb[2];                                       // extra-column weight fragment
    half8 af[4];                                       // pixel fragment ring (two steps ahead)
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) bf[0][ni] = *reinterpret_cast<const half8*>(smem + b_l0 + ni * 2048);
    if (EXTRA) xb[0] = *reinterpret_cast<const half8*>(smem + x_l0);
    af[0] = *reinterpret_cast<const half8*>(smem + a_l0);
    af[1] = *reinterpret_cast<const half8*>(smem + a_l0 + 2048);

    // One 64-deep stage (NS steps of FN MFMAs).  LVL (compile time): 2 = stages kt+1 and kt+2 exist, 1 = only kt+1,
    // 0 = last stage.  Two LDS buffers; the single barrier of a stage sits at step NS-3, right after the LAST fragment
    // read of this stage's buffer has been issued: behind it (a) the buffer can be refilled (stage kt+2), so the DMA
    // gets the rest of this stage plus most of the next one to land instead of racing the next barrier, and (b) stage
    // kt+1's data is complete, so its first fragments are fetched under the MFMAs of steps NS-2, NS-1 -- the MFMA
    // pipe no longer drains at a stage boundary.
    auto stage = [&](int kt, auto lvl_tag) {
        constexpr int LVL = ABL == 10 ? 0 : ABL == 1 ? (decltype(lvl_tag)::value ? 1 : 0) : decltype(lvl_tag)::value;
        constexpr bool RD = ABL != 10;
        constexpr bool DMA_ON = ABL != 1;
        const char* cur = smem + (kt & 1) * G::STAGE_BYTES;
        char* nxt = smem + ((kt & 1) ^ 1) * G::STAGE_BYTES;
        int2 e_n2 = e_nx;
#pragma unroll
        for (int s = 0; s < NS; ++s) {                 // step s: K half s / FM, pixel fragment s % FM
            const int ks = s / FM, mi = s % FM;
            if (s == 0 && LVL == 2) e_n2 = kt_lane[(kt + 2) * 8];             // K table entry of stage kt+2
            {                                          // (1) pixel fragment two steps ahead (next stage's at the end)
                const int s2 = s + 2;
                if (s2 < NS && RD) af[s2 % 4] = *reinterpret_cast<const half8*>(cur + ((s2 / FM) ? a_l1 : a_l0) + (s2 % FM) * 2048);
            }
            if (s == NS - 3 && LVL >= 1) {             // (B) the barrier of this stage
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                // "use" the table entry here, where nothing is outstanding: otherwise the compiler's own wait for that
                // load lands at the top of the next stage as vmcnt(0) and drains the early pieces issued below.
                asm volatile("" : "+v"(e_n2.x), "+v"(e_n2.y));
                __builtin_amdgcn_s_barrier();
            }
            if (s >= NS - 2 && LVL >= 1) {             // first fragments of stage kt+1
                const int j = s - (NS - 2);
                af[(s + 2) % 4] = *reinterpret_cast<const half8*>(nxt + a_l0 + j * 2048);
#pragma unroll
                for (int ni = 2 * j; ni < 2 * j + 2; ++ni) bf[0][ni] = *reinterpret_cast<const half8*>(nxt + b_l0 + ni * 2048);
                if (EXTRA && j == 1) xb[0] = *reinterpret_cast<const half8*>(nxt + x_l0);
            }
            if (ks == 0 && mi >= FM - FN && RD)        // (2) weight fragments of the second K half
                bf[1][mi - (FM - FN)] = *reinterpret_cast<const half8*>(cur + b_l1 + (mi - (FM - FN)) * 2048);
            if (EXTRA && s == 3) xb[1] = *reinterpret_cast<const half8*>(cur + x_l1);
            if (LVL >= 1 && DMA_ON && s < NSD) {       // (3) the rest of stage kt+1's DMA pieces
#pragma unroll
                for (int q = EARLY + (s * (P - EARLY)) / NSD; q < EARLY + ((s + 1) * (P - EARLY)) / NSD; ++q)
                    dma_piece(q, nxt, e_nx, kt + 1);
            }
            if (EXTRA && LVL >= 1 && DMA_ON && s == NSD) {
                if (extra_wave) dma16(rw, nxt + (BM + BN) * 128 + (wave & 1) * 1024, x_voff, (unsigned)(kt + 1) * 128u);
            }
            if (LVL == 2 && DMA_ON && s >= NS - 2)     //     and, behind the barrier, the first pieces of stage kt+2
                dma_piece(s - (NS - 2), const_cast<char*>(cur), e_n2, kt + 2);
            const half8 a = af[s % 4];                 // (4) the MFMAs of this step
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                if (ABL == 4 || ABL == 9) asm volatile("" ::"v"(bf[ks][ni]), "v"(a));
                else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][ni], a, acc[ni][mi], 0, 0, 0);
            }
            if (EXTRA && has_extra) {                  // extra column fragment x pixel fragment mi: N-wave mi >> 1
                if (wn == (mi >> 1)) accx[mi & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], a, accx[mi & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        e_nx = e_n2;
    };
    int kt = 0;
    for (; kt + 2 < KT; ++kt) stage(kt, std::integral_constant<int, 2>{});
    if (KT >= 2) { stage(kt, std::integral_constant<int, 1>{}); ++kt; }
    stage(kt, std::integral_constant<int, 0>{});

    // ---- epilogue ----
    if (ABL == 5 || ABL == 9 || ABL == 10) {
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) t += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (t == 123.456f) reinterpret_cast<half_t*>(p.y)[0] = (half_t)t;
        return;
    }
    // Main 256 columns: transpose through LDS so every lane stores 16 B and 8 lanes cover one 128-byte line of a pixel
    // (the direct fragment layout gives 8-byte stores in 32-byte runs: 0.29 ms of a 1.6 ms launch, fully exposed at one
    // block per CU).  Phase 1: bias -> ReLU -> affine in registers, fp16, ds_write_b64 into a wave-private
    // [8 fragments][16 pixels][128 B] image (16-byte slots XOR-swizzled by pixel).  Phase 2: ds_read_b128, residual +
    // ReLU, global store.  TRANSPOSED / RGB8 / odd pixel-shuffle widths keep the per-fragment path.
    const bool lds_epi = !(p.flags & (HAVC_F_OUT_TRANSPOSED | HAVC_F_OUT_RGB8)) &&
                         (!(p.flags & HAVC_F_OUT_PIXSHUF) || (p.Co & 63) == 0);
    bool fused_done = false;
    if (!lds_epi) {
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int m = m0 + wm * (FM * 16) + mi * 16 + lr;
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) epilogue_frag(p, acc[ni][mi], m, n0 + wn * 64 + ni * 16 + lg * 4, HoWo);
        }
    } else if (EXTRA && has_extra && (p.flags & HAVC_F_FUSE_RGB8)) {
        // ---- fused layers.11 (1x1 conv -> 3 channels) + SigmoidRange + denormalise + trunc u8 (HAVC_F_FUSE_RGB8) ----
        // r2 = fp16(ReLU(acc + bias) + residual) never leaves the registers: an accumulator fragment (lane = pixel lr,
        // channels lg*4..+3) IS the B operand of v_mfma_f32_16x16x16_f16, so the 1x1 conv is FN MFMAs per pixel
        // fragment with the (3 real of 16) output rows as A.  The 259-channel row of a pixel is spread over the four
        // N-waves and the extra-column fragments; the five partial sums go through LDS and are added in a FIXED
        // order, so a frame colours identically in any batch.
        fused_done = true;
        const bool leaky = p.flags & HAVC_F_LEAKY;
        const int nw0 = n0 + wn * 64;
        half4 wa[FN];
        float4 bvs[FN];
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = nw0 + ni * 16 + lg * 4;
            wa[ni] = half4{0, 0, 0, 0};
            if (lr < 3) {
                const float4 w4 = *reinterpret_cast<const float4*>(p.fuse_w + lr * p.Npad + n);
                wa[ni] = half4{(half_t)w4.x, (half_t)w4.y, (half_t)w4.z, (half_t)w4.w};
            }
            bvs[ni] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4v facc[FM];
#pragma unroll
        for (int mh = 0; mh < FM; mh += 4) {               // four pixel fragments at a time: 16 residual loads in flight
            half4 rr[4][FN];
#pragma unroll
            for (int mj = 0; mj < 4; ++mj) {
                const int m = m0 + wm * (FM * 16) + (mh + mj) * 16 + lr;
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) {
                    rr[mj][ni] = half4{0, 0, 0, 0};
                    if ((p.flags & HAVC_F_RESIDUAL) && m < p.M)
                        rr[mj][ni] = *reinterpret_cast<const half4*>(p.res + out_pixel(p, m, HoWo) * p.res_cpitch + p.res_coff + nw0 + ni * 16 + lg * 4);
                }
            }
#pragma unroll
            for (int mj = 0; mj < 4; ++mj) {
                const int mi = mh + mj;
                facc[mi] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) {
                    const float bb[4] = {bvs[ni].x, bvs[ni].y, bvs[ni].z, bvs[ni].w};
                    half4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[ni][mi][r] + bb[r];
                        if (p.flags & HAVC_F_RELU_PRE) v = v > 0.f ? v : (leaky ? v * p.f2 : 0.f);
                        o[r] = (half_t)(v + (float)rr[mj][ni][r]);    // one rounding, as epilogue_frag
                    }
                    facc[mi] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[ni], o, facc[mi], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // every wave is done reading the last stage
        float* part = reinterpret_cast<float*>(smem);      // [5][BM][3]
        if (lg == 0) {
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                float* q = part + (wn * BM + wm * (FM * 16) + mi * 16 + lr) * 3;
                q[0] = facc[mi][0]; q[1] = facc[mi][1]; q[2] = facc[mi][2];
            }
            const int nx = n0 + BN;                        // extra columns: channels nx .. nx+2 are real (x0's RGB)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = wm * (FM * 16) + (wn * 2 + i) * 16 + lr;
                const int m = m0 + pl;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = accx[i][r] + (p.bias ? p.bias[nx + r] : 0.f);
                    if (p.flags & HAVC_F_RELU_PRE) v = v > 0.f ? v : (leaky ? v * p.f2 : 0.f);
                    if ((p.flags & HAVC_F_RESIDUAL) && m < p.M && nx + r < p.Co)      // as epilogue_frag: one rounding
                        v += (float)p.res[out_pixel(p, m, HoWo) * p.res_cpitch + p.res_coff + nx + r];
                    const float f = (float)(half_t)v;
                    d0 += f * (float)(half_t)p.fuse_w[nx + r];
                    d1 += f * (float)(half_t)p.fuse_w[p.Npad + nx + r];
                    d2 += f * (float)(half_t)p.fuse_w[2 * p.Npad + nx + r];
                }
                float* q = part + (4 * BM + pl) * 3;
                q[0] = d0; q[1] = d1; q[2] = d2;
            }
        }
        __syncthreads();
        for (int pl = threadIdx.x; pl < BM; pl += NW * 64) {
            const int m = m0 + pl;
            if (m >= p.M) continue;
            uint8_t* y = p.fuse_rgb + out_pixel(p, m, HoWo) * 3;
#pragma unroll
            for (int jo = 0; jo < 3; ++jo) {
                float t = p.fuse_b[jo];
#pragma unroll
                for (int w = 0; w < 5; ++w) t += part[(w * BM + pl) * 3 + jo];
                float sg = 1.f / (1.f + __expf(-t));
                sg = sg * (p.f1 - p.f0) + p.f0;
                sg = sg * p.istd[jo] + p.mean[jo];
                sg = fminf(fmaxf(sg, 0.f), 1.f);
                y[jo] = (uint8_t)(int)(sg * 255.f);
            }
        }
    } else {
        __syncthreads();                                   // every wave is done reading the last stage
        char* img = smem + wave * (FM * 2048);
        const bool leaky = p.flags & HAVC_F_LEAKY;
        const int nw0 = n0 + wn * 64;                      // first channel of this wave's 64-channel slice
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = nw0 + ni * 16 + lg * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bv;
            const bool n_ok = n < p.Npad;
            if (p.bias && n_ok) bv = *reinterpret_cast<const float4*>(p.bias + n);
            if ((p.flags & HAVC_F_AFFINE) && n_ok) {
                sc = *reinterpret_cast<const float4*>(p.scale + n);
                sh = *reinterpret_cast<const float4*>(p.shift + n);
            }
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) {
                half4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ni][mi][r] + bb[r];
                    if (p.flags & HAVC_F_RELU_PRE) v = v > 0.f ? v : (leaky ? v * p.f2 : 0.f);
                    if (p.flags & HAVC_F_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                    if (p.flags & HAVC_F_AFFINE) v = v * ss[r] + hh[r];
                    o[r] = (half_t)v;
                }
                const int slot = (ni * 2 + (lg >> 1)) ^ (lr & 7);
                *reinterpret_cast<half4*>(img + mi * 2048 + lr * 128 + slot * 16 + (lg & 1) * 8) = o;
            }
        }
        if ((p.flags & HAVC_F_PS_BLUR) && WM == 2 && WN == 4 && FM == 8) {
            // ---- HAVC_F_PS_BLUR: PixelShuffle(2) + ReplicationPad(1,0,1,0) + AvgPool2d(2,1) straight out of the LDS images ----
            // The block holds the 16x16 low-res tile x (4 sub-pixels x 64 channels); wave (wm, q) owns rows wm*8..+7 of
            // sub-pixel q.  Output = the 30x30 hi-res pixels whose 2x2 source window lies inside the tile.  Same fp16
            // rounding of the shuffled values and the same summation order as elementwise.hip blur_resize_kernel.
            __syncthreads();
            const int tile = m0 / BM, tt = p.tiles_y * p.tiles_x;
            const int b = tile / tt, t = tile - b * tt, ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
            const int h0 = ty * 15 - 1, w0 = tx * 15 - 1, H2 = 2 * p.Ho, W2 = 2 * p.Wo;
            half_t* y = reinterpret_cast<half_t*>(p.y) + p.y_coff + (n0 >> 2);          // 64 output channels per 256-column tile
            for (int it = tid; it < 30 * 30 * 8; it += NW * 64) {
                const int ch = it & 7, pxl = it >> 3, Yl = pxl / 30, Xl = pxl - Yl * 30;
                const int Y = 2 * (h0 + 1) + Yl, X = 2 * (w0 + 1) + Xl;
                if (Y >= H2 || X >= W2) continue;
                const int ay0 = max(Y - 1, 0), ax0 = max(X - 1, 0);
                auto ld = [&](int ay, int ax) -> half8 {
                    const int dy = (ay >> 1) - h0, dx = (ax >> 1) - w0, q = (ay & 1) * 2 + (ax & 1);
                    return *reinterpret_cast<const half8*>(smem + ((dy >> 3) * WN + q) * (FM * 2048) + (dy & 7) * 2048 + dx * 128 + ((ch ^ (dx & 7)) << 4));
                };
                const half8 v00 = ld(ay0, ax0), v01 = ld(ay0, X), v10 = ld(Y, ax0), v11 = ld(Y, X);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v00[e] + (float)v01[e] + (float)v10[e] + (float)v11[e]) * 0.25f);
                *reinterpret_cast<half8*>(y + ((int64_t)(b * H2 + Y) * W2 + X) * p.y_cpitch + ch * 8) = o;
            }
            return;
        }
        // phase 2: lane -> (pixel row lane>>3 (+8), 16-byte channel slot lane&7).  All residual vectors are requested
        // up front (the accumulators are dead by now) so their latency overlaps instead of serialising 16 load->store pairs.
        const int ch = lane & 7;
        const int n = nw0 + ch * 8;
        const bool do_res = (p.flags & HAVC_F_RESIDUAL) && !(p.flags & HAVC_F_OUT_PIXSHUF) && n < p.Co;
        half8 rres[FM][2];
        if (p.flags & HAVC_F_RESIDUAL) {
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    const int m = m0 + wm * (FM * 16) + mi * 16 + (lane >> 3) + hp * 8;
                    rres[mi][hp] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (do_res && m < p.M) rres[mi][hp] = *reinterpret_cast<const half8*>(p.res + out_pixel(p, m, HoWo) * p.res_cpitch + p.res_coff + n);
                }
        }
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                const int px = (lane >> 3) + hp * 8;
                const int m = m0 + wm * (FM * 16) + mi * 16 + px;
                const half8 v = *reinterpret_cast<const half8*>(img + mi * 2048 + px * 128 + ((ch ^ (px & 7)) << 4));
                if (m >= p.M) continue;
                half_t* y = reinterpret_cast<half_t*>(p.y);
                int64_t off;
                if (p.flags & HAVC_F_OUT_PIXSHUF) {
                    const int q = n / p.Co, cc = n - q * p.Co;
                    if (q >= 4) continue;
                    const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    off = ((int64_t)(b * 2 * p.Ho + 2 * ho + (q >> 1)) * (2 * p.Wo) + 2 * wo + (q & 1)) * p.y_cpitch + p.y_coff + cc;
                } else {
                    if (n >= p.Co) continue;
                    off = out_pixel(p, m, HoWo) * p.y_cpitch + p.y_coff + n;
                }
                half8 o = v;
                if (p.flags & (HAVC_F_RESIDUAL | HAVC_F_RELU_POST)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float f = (float)v[e];
                        if (p.flags & HAVC_F_RESIDUAL) f += (float)rres[mi][hp][e];
                        if (p.flags & HAVC_F_RELU_POST) f = f > 0.f ? f : (leaky ? f * p.f2 : 0.f);
                        o[e] = (half_t)f;
                    }
                }
                *reinterpret_cast<half8*>(y + off) = o;
            }
        }
    }
    if (EXTRA && has_extra && !(fused_done)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * (FM * 16) + (wn * 2 + i) * 16 + lr;
            epilogue_frag(p, accx[i], m, n0 + BN + lg * 4, HoWo);
        }
    }
}

template <int WM, int WN, int FM, int EXTRA, int ABL = 0>
static int launch_pipe(const ConvArgs& a, hipStream_t s) {
    using G = Geo<WM, WN, FM, EXTRA>;
    if ((a.Kc & 7) || !a.ktab || a.x_bytes == 0 || a.x_bytes >= OOB || a.w_bytes >= OOB) return (int)hipErrorInvalidValue;
    const int MT = (a.M + G::BM - 1) / G::BM, NT = (a.Npad - 16 * EXTRA + G::BN - 1) / G::BN;
    constexpr int LDS = G::LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipe_kernel<WM, WN, FM, EXTRA, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, FM, EXTRA, ABL>), dim3(MT * NT), dim3(G::NW * 64), LDS, s, a);
    return (int)hipGetLastError();
}

bool conv_pipe_supported(const ConvArgs& a, int extra) {
    return !(a.Kc & 7) && a.ktab && a.x_bytes != 0 && a.x_bytes < OOB && a.w_bytes < OOB && (!extra || (a.Npad - 16) % 256 == 0);
}

int launch_conv_pipe(const ConvArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case 60: return launch_pipe<2, 4, 8, 0>(a, s);        // 256 x 256
        case 61: return launch_pipe<2, 4, 8, 1>(a, s);        // 256 x (256 + 16)
        case 62: return launch_pipe<2, 4, 8, 0, 1>(a, s);     // ablations of cfg 60 (profiling only)
        case 64: return launch_pipe<2, 4, 8, 0, 4>(a, s);
        case 69: return launch_pipe<2, 4, 8, 0, 9>(a, s);
        case 73: return launch_pipe<2, 4, 8, 0, 10>(a, s);
        case 65: return launch_pipe<2, 4, 8, 0, 5>(a, s);
        case 70: return launch_pipe<2, 2, 4, 0>(a, s);        // 128 x 128, 4 waves
        case 71: return launch_pipe<1, 4, 8, 0>(a, s);        // 128 x 256, 4 waves
        case 72: return launch_pipe<1, 2, 4, 0>(a, s);        // 64 x 128, 2 waves
    }
    return (int)hipErrorInvalidValue;
}
