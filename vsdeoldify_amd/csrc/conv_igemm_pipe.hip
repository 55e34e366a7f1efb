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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_pipe_kernel.inc"

// =====================================================================================================================
// conv_halo_kernel: 3x3 stride-1 pad-1 convolutions on 16x16-pixel tiles with HALO REUSE.
// The im2col main loop above fetches every pixel row once per tap: 9 x (256 + 272) rows per 64-channel group, 8.25 LDS-DMA
// pieces per wave per K-64 stage, and it is the ISSUE cost of those pieces that keeps the MFMA pipe at ~50 %
// (profiles/r1_conv_ablation.txt).  Here the 18x18 halo of a 16x16 output tile is staged ONCE per 64-channel group
// (46 pieces) and the nine taps read their pixel fragments from it at shifted row offsets: 4.25 weight pieces + 0.65 halo
// pieces per wave per stage.  Halo rows have a 144-byte pitch (8 data slots + 1 pad slot): 16 consecutive rows then cover all
// sixteen 16-byte slots of a 256-byte bank line for ANY starting row, so every tap's ds_read_b128 is conflict-free with plain
// immediate offsets and no XOR.  LDS: 2 halo buffers (2 x 46 KiB) + 2 weight stages (2 x 34 KiB) = exactly 160 KiB.
// Weights, stage order (group-major, tap inside), barrier placement, fragment rings and the epilogue are those of
// conv_pipe_kernel<2,4,8,EXTRA>; the 1-chunk remainder segment (259 = 256 + 3 channels) is served from a halo tile of the
// remainder channels, each lane group reading the tap its K chunk belongs to.
template <int EXTRA, int ABL = 0>
__global__ void __launch_bounds__(512) conv_halo_kernel(const ConvArgs p) {
    constexpr int WM = 2, WN = 4, FM = 8, NW = 8, BM = 256, BN = 256, B_IT = BN / 8 / NW, NS = 16;
    constexpr int HB = 46 * 1024, WB = (BN + 16 * EXTRA) * 128, H_IT = 6, HP = 144;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wbase = smem + 2 * HB;

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
    const int tile = pid / NT;
    const int m0 = tile * BM;
    const int n0 = (pid % NT) * BN;
    const bool has_extra = EXTRA && (n0 + BN + 16 == p.Npad);
    const int HoWo = p.Ho * p.Wo;
    const int tt = p.tiles_y * p.tiles_x;
    const int tb = tile / tt, trem = tile - tb * tt, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- weight DMA role (as conv_pipe_kernel): row lane>>3 of an 8-row piece, LDS position lane&7, source chunk c ----
    const int r8 = lane >> 3;
    const int c = (lane & 7) ^ ((((wave & 1) << 2) + (r8 >> 1)) & 7);
    unsigned b_voff[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + (wave + it * NW) * 8 + r8;
        b_voff[it] = n < p.Npad ? (unsigned)((n * p.Kc + c) * 16) : OOB;
    }
    const unsigned x_voff = (unsigned)(((n0 + BN + (wave & 1) * 8 + r8) * p.Kc + c) * 16);
    const bool extra_wave = has_extra && wave >= 6;

    // ---- halo DMA role: 16-byte slot s = piece * 64 + lane of the halo buffer -> halo row s / 9, slot s % 9 (8 = pad) ----
    // h_voff: byte offset of (pixel, channel chunk = slot) in the input, OOB outside the image / pad; bit 0 marks slot 0
    // (the only slot the remainder group fetches).
    unsigned h_voff[H_IT];
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
        const int piece = wave + it * NW;
        const int sl = piece * 64 + lane;
        const int row = sl / 9, col = sl - row * 9;
        const int hy = row / 18, hx = row - hy * 18;
        const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
        const bool ok = piece < 46 && row < 324 && col < 8 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int chunk = (col & 4) | ((col & 1) << 1) | ((col >> 1) & 1);               // LDS position col holds source chunk pi(col)
        const unsigned off = (unsigned)((((int64_t)(tb * p.Hi + iy) * p.Wi + ix) * p.x_cpitch + p.x_coff) * 2) + (unsigned)chunk * 16u;
        h_voff[it] = ok ? (off | (col == 0 ? 1u : 0u)) : OOB;
    }
    const int G = p.C8a >> 3;                                                        // full 64-channel groups of the main K segment
    const int KT = p.Kc >> 3;
    const bool has_rem = KT > 9 * G;                                                 // the 1-chunk remainder segment: 2 stages

    // ---- fragment read addresses ----
    const int fsw = (lr >> 1) & 7;
    const int b_l0 = (wn * 64 + lr) * 128 + ((lg ^ fsw) << 4), b_l1 = b_l0 ^ 64;
    const int x_l0 = (BN + lr) * 128 + ((lg ^ fsw) << 4), x_l1 = x_l0 ^ 64;
    // Bank conflicts: a ds_read_b128 is served in lane groups {lr 0-3, 12-15 of lg} + {lr 4-11 of lg+1}.  With a 9-slot row pitch
    // the slot of (row, chunk position) is 9 row + pos (mod 16); MFMA column lr therefore carries pixel lp = sigma(lr) -- even
    // pixels for lr in {0-3, 12-15}, odd ones for lr in {4-11} -- and chunk c sits at position pi(c) = {0,2,1,3,4,6,5,7}[c], so
    // that the two halves of a lane group land on slots of different parity for ANY starting row: conflict-free at every tap.
    const int lp = lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8));
    const int a_row = (wm * 8 * 18 + lp) * HP;                                       // tile row wm*8, pixel lp (tap / mi via immediates)
    int a_cur = a_row + (((lg & 1) << 1) | (lg >> 1)) * 16, a_nxt = a_cur + HB;      // halo buffers 0 / 1; position pi(lg)
    auto tap_off = [](int t) { return ((t / 3) * 18 + (t % 3)) * HP; };            // (1 + dh) * 18 + (1 + dw), in bytes
    // remainder stages: K chunk j of the segment is tap j (chunk 8 = tap 8, chunks 9..15 zero weights): per-lane tap
    const int rem_buf = (G & 1) * HB;
    const int a_rem0 = rem_buf + a_row + tap_off(lg), a_rem1 = rem_buf + a_row + tap_off(4 + lg), a_rem8 = rem_buf + a_row + tap_off(8);

    float4v acc[FN][FM];
    float4v accx[2];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
    accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};

    auto w_piece = [&](int q, char* wbuf, int kstage, bool live = true) {
        dma16(rw, wbuf + (wave + q * NW) * 1024, (ABL == 6 || !live) ? OOB : b_voff[q], (unsigned)kstage * 128u);     // ABL 6: DMA issued, nothing fetched
    };
    auto h_piece = [&](int it, int hoff, int group, bool rem, bool live = true) {
        const unsigned v = h_voff[it];
        const unsigned vr = (v & 1u) ? (v & ~1u) : OOB;                              // remainder group: slot 0 only (OOB has bit 0 clear)
        const unsigned vo = (ABL == 6 || !live) ? OOB : (rem ? vr : (v & ~1u));
        dma16(rx, smem + hoff + (wave + it * NW) * 1024, vo, (unsigned)group * 128u);
    };

    // ---- prologue: halo of group 0, weight stage 0, two early pieces of stage 1 ----
#pragma unroll
    for (int it = 0; it < H_IT; ++it) if (wave + it * NW < 46) h_piece(it, 0, 0, G == 0);
#pragma unroll
    for (int q = 0; q < B_IT; ++q) w_piece(q, Wbase, 0);
    if (EXTRA && extra_wave) dma16(rw, Wbase + BN * 128 + (wave & 1) * 1024, x_voff, 0);
    if (KT > 1) {
#pragma unroll
        for (int q = 0; q < B_IT; ++q) w_piece(q, Wbase + WB, 1);
        if (EXTRA && extra_wave) dma16(rw, Wbase + WB + BN * 128 + (wave & 1) * 1024, x_voff, 128u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    half8 bf[2][FN];
    half8 xb[2];
    half8 af[4];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) bf[0][ni] = *reinterpret_cast<const half8*>(Wbase + b_l0 + ni * 2048);
    if (EXTRA) xb[0] = *reinterpret_cast<const half8*>(Wbase + x_l0);
    {
        const int a0 = G > 0 ? a_cur : a_rem0;                                       // stage 0 is tap 0 (same immediates either way)
        af[0] = *reinterpret_cast<const half8*>(smem + a0);
        af[1] = *reinterpret_cast<const half8*>(smem + a0 + 18 * HP);
    }
    int h_cur_off = 0;                                                               // byte offset of the halo buffer being read

    // One K-64 stage.  T: tap (0..8) of a main group, 9 / 10: the two remainder stages.  a_next: per-lane base of the NEXT
    // stage's first two pixel fragments (its tile rows 0, 1 are at +0 and +18 rows).
    auto stage = [&](int kt, int group, auto t_tag, int a_next) {
        constexpr int T = decltype(t_tag)::value;
        constexpr bool DMA_ON = ABL != 1;
        const char* wcur = Wbase + (kt & 1) * WB;
        char* wnxt = Wbase + ((kt & 1) ^ 1) * WB;
        const bool more2 = kt + 2 < KT;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ks = s / FM, mi = s % FM;
            if (s + 2 < NS) {                              // (1) pixel fragment two steps ahead
                const int s2 = s + 2, k2 = s2 / FM, m2 = s2 % FM;
                int addr;
                if (T < 9) addr = a_cur + tap_off(T) + m2 * 18 * HP + k2 * 64;
                else if (T == 9) addr = (k2 ? a_rem1 : a_rem0) + m2 * 18 * HP;
                else addr = a_rem8 + m2 * 18 * HP;
                af[s2 % 4] = *reinterpret_cast<const half8*>(smem + addr);
            }
            if (s == NS - 3) {                             // (B) the barrier of this stage
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (s >= NS - 2) {                             // first fragments of the next stage
                const int j = s - (NS - 2);
                af[(s + 2) % 4] = *reinterpret_cast<const half8*>(smem + a_next + j * 18 * HP);
#pragma unroll
                for (int ni = 2 * j; ni < 2 * j + 2; ++ni) bf[0][ni] = *reinterpret_cast<const half8*>(wnxt + b_l0 + ni * 2048);
                if (EXTRA && j == 1) xb[0] = *reinterpret_cast<const half8*>(wnxt + x_l0);
            }
            if (ks == 0 && mi >= FM - FN)                  // (2) weight fragments of the second K half
                bf[1][mi - (FM - FN)] = *reinterpret_cast<const half8*>(wcur + b_l1 + (mi - (FM - FN)) * 2048);
            if (EXTRA && s == 3) xb[1] = *reinterpret_cast<const half8*>(wcur + x_l1);
            if (DMA_ON) {                                  // (3) DMA.  Behind the barrier (steps 13..15) the buffer just drained takes
                //     ALL weight pieces of stage kt+2: they have a whole stage to land.  One halo piece of the next group per tap.
                //     No branches: past the last stage / group the pieces are issued with out-of-range offsets (zero fill, no fetch).
                if (s >= NS - 3) {
                    char* wc = const_cast<char*>(wcur);
                    if (s == NS - 3) w_piece(0, wc, kt + 2, more2);
                    if (s == NS - 2) { w_piece(1, wc, kt + 2, more2); w_piece(2, wc, kt + 2, more2); }
                    if (s == NS - 1) {
                        w_piece(3, wc, kt + 2, more2);
                        if (EXTRA) { if (extra_wave) dma16(rw, wc + BN * 128 + (wave & 1) * 1024, more2 ? x_voff : OOB, (unsigned)(kt + 2) * 128u); }
                    }
                }
                if (T < H_IT && s == 6) {
                    if (T < H_IT - 1 || wave + T * NW < 46) h_piece(T, h_cur_off ^ HB, group + 1, group + 1 >= G, group + 1 < G || has_rem);
                }
            }
            const half8 a = af[s % 4];                     // (4) the MFMAs of this step
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                if (ABL == 4) asm volatile("" ::"v"(bf[ks][ni]), "v"(a));
                else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][ni], a, acc[ni][mi], 0, 0, 0);
            }
            if (EXTRA && has_extra) {
                if (wn == (mi >> 1)) accx[mi & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], a, accx[mi & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0;
    for (int g = 0; g < G; ++g) {
        stage(kt++, g, std::integral_constant<int, 0>{}, a_cur + tap_off(1));
        stage(kt++, g, std::integral_constant<int, 1>{}, a_cur + tap_off(2));
        stage(kt++, g, std::integral_constant<int, 2>{}, a_cur + tap_off(3));
        stage(kt++, g, std::integral_constant<int, 3>{}, a_cur + tap_off(4));
        stage(kt++, g, std::integral_constant<int, 4>{}, a_cur + tap_off(5));
        stage(kt++, g, std::integral_constant<int, 5>{}, a_cur + tap_off(6));
        stage(kt++, g, std::integral_constant<int, 6>{}, a_cur + tap_off(7));
        stage(kt++, g, std::integral_constant<int, 7>{}, a_cur + tap_off(8));
        stage(kt++, g, std::integral_constant<int, 8>{}, g + 1 < G ? a_nxt : a_rem0);       // next: tap 0 of the next group / remainder
        const int t = a_cur; a_cur = a_nxt; a_nxt = t;
        h_cur_off ^= HB;
    }
    if (has_rem) {
        stage(kt++, G, std::integral_constant<int, 9>{}, a_rem8);
        stage(kt++, G, std::integral_constant<int, 10>{}, a_rem8);
    }

    auto gm = [&](int local_row) -> int {                  // tile row -> linear output pixel (out_pixel is the identity here)
        const int y = ty * 16 + (local_row >> 4), x = tx * 16 + (local_row & 15);
        return (y < p.Ho && x < p.Wo) ? (tb * p.Ho + y) * p.Wo + x : 0x7fffffff;
    };
#define PF(mi) (mi)
#define HAVC_STAMP(i) do { } while (0)
    const int eflags = p.flags;
    constexpr bool UNIT_STEP = false;                                   // (identity either way; true changes this kernel's register allocation for the worse: 12 spills)
    auto opix = [&](int m) -> int64_t { return (int64_t)m; };          // launch_halo requires out step 1
#include "conv_pipe_epilogue.inc"
#undef PF
#undef HAVC_STAMP
}

template <int EXTRA, int ABL = 0>
static int launch_halo(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    if ((a.Kc & 7) || a.x_bytes == 0 || a.x_bytes >= OOB || a.w_bytes >= OOB || a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 ||
        a.pad_w != 1 || a.dil != 1 || a.oss != 1 || a.Hi != a.Ho || a.Wi != a.Wo)
        return (int)hipErrorInvalidValue;
    const int frames = a.M / (a.Ho * a.Wo);
    a.tiles_y = (a.Ho + 15) / 16;
    a.tiles_x = (a.Wo + 15) / 16;
    const int MT = frames * a.tiles_y * a.tiles_x, NT = (a.Npad - 16 * EXTRA + 255) / 256;
    a.M = MT * 256;
    constexpr int LDS = 2 * 46 * 1024 + 2 * (256 + 16 * EXTRA) * 128;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ensure_lds_optin<conv_halo_kernel<EXTRA, ABL>>(LDS);
    hipLaunchKernelGGL((conv_halo_kernel<EXTRA, ABL>), dim3(MT * NT), dim3(512), LDS, s, a);
    return (int)hipGetLastError();
}

// split-K, second half: sum the parts of every (pixel, 4 channels) in the order 0 .. S-1, then the shared per-fragment epilogue
__global__ void splitk_reduce_kernel(const ConvArgs p) {
    const int n4 = p.Npad >> 2, HoWo = p.Ho * p.Wo;
    const int64_t total = (int64_t)p.M * n4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        float4v acc = *reinterpret_cast<const float4v*>(p.ws + (int64_t)m * p.Npad + n);
        for (int sp = 1; sp < p.splitk; ++sp) {
            const float4v t = *reinterpret_cast<const float4v*>(p.ws + ((int64_t)sp * p.M + m) * p.Npad + n);
            acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
        }
        epilogue_frag(p, acc, m, n, HoWo);
    }
}

template <int WM, int WN, int FM, int EXTRA, int ABL = 0, int EF = -1>
static int launch_pipe(const ConvArgs& a, hipStream_t s) {
    using G = Geo<WM, WN, FM, EXTRA>;
    if ((a.Kc & 7) || !a.ktab || a.x_bytes == 0 || a.x_bytes >= OOB || a.w_bytes >= OOB || (EXTRA && a.Npad != G::BN + 16)) return (int)hipErrorInvalidValue;
    const int MT = (a.M + G::BM - 1) / G::BM, NT = (a.Npad - 16 * EXTRA + G::BN - 1) / G::BN;
    const int SK = a.splitk > 1 ? a.splitk : 1;
    if (SK > 1 && (EXTRA || ABL || !a.ws || (a.Kc >> 3) < 2 * SK || (a.Npad & 3) ||
                   (a.flags & (HAVC_F_PS_BLUR | HAVC_F_FUSE_RGB8 | HAVC_F_FUSE_PROJ | HAVC_F_W_FROM_BUF))))
        return (int)hipErrorInvalidValue;
    constexpr int LDS = G::LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    if (EF >= 0 && ((a.flags & HAVC_EPI_MASK) != EF || a.oss != 1)) return (int)hipErrorInvalidValue;
    ensure_lds_optin<conv_pipe_kernel<WM, WN, FM, EXTRA, ABL, EF>>(LDS);
    ConvArgs ar = a;
    if (!(a.flags & HAVC_F_PS_BLUR)) conv_raster(ar, MT, NT, G::BN);
    hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, FM, EXTRA, ABL, EF>), dim3(MT * NT * SK), dim3(G::NW * 64), LDS, s, ar);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || SK == 1) return (int)e;
    const int64_t work = (int64_t)a.M * (a.Npad >> 2);
    const int64_t gb = (work + 255) / 256;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(gb > 8192 ? 8192 : gb)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// Every PRODUCT instantiation (the heuristic's and the autotuner's tile configurations, the halo conv, the precise epilogues): LDS opt-in + module
// load, eagerly from havc_create under the library's set-up mutex, so that no first launch -- possibly from several host threads -- does either.
void preload_conv_pipe() {
#define PL(WM, WN, FM, EX, ABL) ensure_lds_optin<conv_pipe_kernel<WM, WN, FM, EX, ABL>>(Geo<WM, WN, FM, EX>::LDS_BYTES)
    PL(2, 4, 8, 0, 0); PL(2, 4, 8, 1, 0); PL(2, 2, 4, 0, 0); PL(1, 4, 8, 0, 0); PL(1, 2, 4, 0, 0);
    PL(1, 4, 6, 0, 0); PL(1, 4, 4, 0, 0); PL(2, 2, 6, 0, 0); PL(1, 2, 6, 0, 0); PL(2, 4, 4, 0, 0); PL(2, 4, 6, 0, 0); PL(4, 2, 4, 0, 0);
    PL(4, 1, 4, 0, 0); PL(2, 1, 4, 0, 0);
    PL(2, 4, 8, 0, 30); PL(2, 4, 8, 1, 30); PL(2, 2, 4, 0, 30); PL(1, 4, 8, 0, 30); PL(1, 2, 4, 0, 30); PL(2, 4, 4, 0, 30); PL(4, 2, 4, 0, 30);
    PL(4, 1, 4, 0, 30); PL(2, 1, 4, 0, 30);
#undef PL
    preload_conv_pipe_ef();
    ensure_lds_optin<conv_halo_kernel<0, 0>>(2 * 46 * 1024 + 2 * 256 * 128);
    ensure_lds_optin<conv_halo_kernel<1, 0>>(2 * 46 * 1024 + 2 * 272 * 128);
    (void)hipGetLastError();
}

bool conv_splitk_cfg_ok(int cfg) {
    switch (cfg) { case 60: case 70: case 71: case 72: case 90: case 91: case 92: case 93: case 95: case 96: case 97: case 98: case 99: return true; }
    return false;
}

bool conv_pipe_supported(const ConvArgs& a, int extra) {
    return !(a.Kc & 7) && a.ktab && a.x_bytes != 0 && a.x_bytes < OOB && a.w_bytes < OOB && (!extra || a.Npad == 272);
}

int launch_conv_pipe(const ConvArgs& a, int cfg, hipStream_t s) {
    if (a.flags & HAVC_F_PRECISE) {                            // precise mode: the same main loops with the fp32 hi / lo epilogue (ABL 30)
        switch (cfg) {
            case 60: return launch_pipe<2, 4, 8, 0, 30>(a, s);
            case 61: return launch_pipe<2, 4, 8, 1, 30>(a, s);
            case 70: return launch_pipe<2, 2, 4, 0, 30>(a, s);
            case 71: return launch_pipe<1, 4, 8, 0, 30>(a, s);
            case 72: return launch_pipe<1, 2, 4, 0, 30>(a, s);
            case 96: return launch_pipe<2, 4, 4, 0, 30>(a, s);
            case 98: return launch_pipe<4, 2, 4, 0, 30>(a, s);
            case 99: return launch_pipe<4, 1, 4, 0, 30>(a, s);
            case 92: return launch_pipe<2, 1, 4, 0, 30>(a, s);
        }
        return (int)hipErrorInvalidValue;
    }
    {   // the hot layer kinds on the main tile geometries: kernels with their epilogue flags fixed at compile time (conv_igemm_pipe_ef.hip)
        const int r = launch_conv_pipe_ef(a, cfg, s);
        if (r != -1) return r;
        static const bool trace = getenv("HAVC_EPI_TRACE") != nullptr;        // which (tile, layer kind) pairs still run the run-time-flag kernel
        if (trace) {
            static std::atomic<uint64_t> seen[64];
            const uint64_t key = ((uint64_t)cfg << 32) | (uint32_t)(a.flags & 0xffff) | ((uint64_t)(a.splitk > 1) << 48) | ((uint64_t)(a.oss != 1) << 49) | ((uint64_t)(!a.bias) << 50);
            for (auto& e : seen) {
                uint64_t cur = e.load();
                if (cur == key) break;
                if (cur == 0 && e.compare_exchange_strong(cur, key)) {
                    fprintf(stderr, "havc: run-time-flag conv kernel: cfg %d flags 0x%x splitk %d out_step %d bias %d (M %d, N %d, K %d)\n", cfg, a.flags & 0xffff, a.splitk, a.oss,
                            a.bias != nullptr, a.M, a.Npad, a.Kc * 8);
                    break;
                }
                if (cur == key) break;
            }
        }
    }
    switch (cfg) {
        case 60: return launch_pipe<2, 4, 8, 0>(a, s);        // 256 x 256
        case 61: return launch_pipe<2, 4, 8, 1>(a, s);        // 256 x (256 + 16)
        case 100: return launch_pipe<2, 4, 8, 0, 20>(a, s);   // schedule experiments (same bytes as cfg 60 / 61)
        case 101: return launch_pipe<2, 4, 8, 1, 20>(a, s);
        case 102: return launch_pipe<2, 4, 8, 1, 21>(a, s);
        case 103: return launch_pipe<2, 4, 8, 1, 22>(a, s);
        case 62: return launch_pipe<2, 4, 8, 0, 1>(a, s);     // ablations of cfg 60 (profiling only)
        case 64: return launch_pipe<2, 4, 8, 0, 4>(a, s);
        case 69: return launch_pipe<2, 4, 8, 0, 9>(a, s);
        case 73: return launch_pipe<2, 4, 8, 0, 10>(a, s);
        case 65: return launch_pipe<2, 4, 8, 0, 5>(a, s);
        case 162: return launch_pipe<2, 4, 8, 0, 40, 0>(a, s);   // bias-only epilogue with phase time stamps (tools/conv_timeline.py)
        case 163: return launch_pipe<2, 4, 8, 0, 40, HAVC_F_GELU>(a, s);
        case 164: return launch_pipe<2, 4, 8, 1, 40, HAVC_F_RELU_PRE>(a, s);   // the tail res-block conv (256 x 272 tile)
        case 68: return launch_pipe<2, 4, 8, 0, 12>(a, s);    // generic epilogue without its global stores
        case 78: return launch_pipe<2, 4, 8, 0, 14>(a, s);    // generic epilogue, every store into one 2-MiB window (cache-resident)
        case 66: return launch_pipe<2, 4, 8, 0, 11>(a, s);    // PS_BLUR epilogue without its global stores
        case 67: return launch_pipe<2, 4, 8, 0, 13>(a, s);    // PS_BLUR epilogue with plain instead of non-temporal stores
        case 80: return launch_halo<0>(a, s);                 // 3x3 s1 p1, 16x16-pixel tiles with halo reuse
        case 81: return launch_halo<1>(a, s);
        case 82: return launch_halo<0, 1>(a, s);              // ablations (profiling only)
        case 84: return launch_halo<0, 4>(a, s);
        case 85: return launch_halo<0, 5>(a, s);
        case 86: return launch_halo<0, 6>(a, s);
        case 70: return launch_pipe<2, 2, 4, 0>(a, s);        // 128 x 128, 4 waves
        case 71: return launch_pipe<1, 4, 8, 0>(a, s);        // 128 x 256, 4 waves
        case 72: return launch_pipe<1, 2, 4, 0>(a, s);        // 64 x 128, 2 waves
        // more tile geometries for the autotuner (round 2): layers whose pixel count leaves the standard tiles with a ragged last
        // wave of blocks (16 x 35^2 = 19 600 pixels: 154 tiles of 128 -> 308 blocks on 256 CUs) get a tile height that divides better
        case 90: return launch_pipe<1, 4, 6, 0>(a, s);        //  96 x 256, 4 waves
        case 91: return launch_pipe<1, 4, 4, 0>(a, s);        //  64 x 256, 4 waves
        case 93: return launch_pipe<2, 2, 6, 0>(a, s);        // 192 x 128, 4 waves
        case 95: return launch_pipe<1, 2, 6, 0>(a, s);        //  96 x 128, 2 waves
        case 96: return launch_pipe<2, 4, 4, 0>(a, s);        // 128 x 256, 8 waves
        case 97: return launch_pipe<2, 4, 6, 0>(a, s);        // 192 x 256, 8 waves
        case 98: return launch_pipe<4, 2, 4, 0>(a, s);        // 256 x 128, 8 waves
        case 99: return launch_pipe<4, 1, 4, 0>(a, s);        // 256 x  64, 4 waves: the 64-channel layers (ResNet layer1, the stem)
        case 92: return launch_pipe<2, 1, 4, 0>(a, s);        // 128 x  64, 2 waves
        case 74: return launch_pipe<2, 2, 4, 0, 1>(a, s);     // ablations of cfg 70 (profiling only): no DMA in the loop
        case 75: return launch_pipe<2, 2, 4, 0, 4>(a, s);     //   no MFMA
        case 76: return launch_pipe<2, 2, 4, 0, 5>(a, s);     //   no epilogue
        case 77: return launch_pipe<2, 2, 4, 0, 10>(a, s);    //   MFMA only (no fragment reads, no DMA, no barriers)
    }
    return (int)hipErrorInvalidValue;
}
