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
    static_assert(FM % 2 == 0, "the 4-slot pixel-fragment ring wraps cleanly only when a stage has a multiple of 4 steps");
    static_assert(!EXTRA || (WM == 2 && WN == 4 && FM == 8), "extra columns: 256x256 tile only");
    static_assert(NW * FM * 2048 <= LDS_BYTES, "LDS epilogue image");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}

}  // namespace

// ABL (profiling only, wrong results): 1 = no DMA inside the loop, 4 = no MFMA, 5 = no epilogue; ABL 30 (a product path): the precise epilogue
// (Tried and dropped: staggering the DMA slots of the two waves sharing a SIMD -- 3-12 % slower, profiles/r1_conv_ablation.txt.)
// __launch_bounds__(threads, 2): at most 256 registers per wave for EVERY tile geometry.  The 2- and 4-wave tiles (one wave per SIMD: the compiler may
// use 512 registers, accumulators in AGPRs) came out with 127 v_accvgpr_read / _mov instructions per 32 MFMAs in their main loops -- the allocator
// rotated the accumulators through the AGPR file every stage; capped at 256 the accumulators stay in place (round 4: the 128 x 128 / 128 x 256 /
// 64 x 128 tiles 10 - 25 % faster on long-K shapes, profiles/r4_conv_tiles_register_cap.txt).  HAVC_PIPE_WPE=1 builds the old variant for A/B runs.
#ifndef HAVC_PIPE_WPE
#define HAVC_PIPE_WPE 2
#endif
template <int WM, int WN, int FM, int EXTRA, int ABL = 0>
__global__ void __launch_bounds__(WM* WN * 64, HAVC_PIPE_WPE) conv_pipe_kernel(const ConvArgs p) {
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
    // split-K (HAVC_F_SPLITK): consecutive blocks are the K parts of one tile
    const int SK = (!EXTRA && p.splitk > 1) ? p.splitk : 1;
    const int part = SK > 1 ? pid % SK : 0;
    if (SK > 1) pid /= SK;
    const int m0 = (ABL == 9 ? (pid / NT) % 64 : pid / NT) * BM;      // ABL 9: L2-resident source, DMA ceiling
    const int n0 = (pid % NT) * BN;
    // EXTRA: Npad == 256 + 16 exactly (launch_pipe checks): ONE column tile, which has the extra fragment -- a compile-time fact, so the main loop
    // carries no test for it
    constexpr bool has_extra = EXTRA != 0;
    const int HoWo = p.Ho * p.Wo;
    // EXTRA: the 8 pixel fragments x 1 extra column fragment are shared by the four N-waves of a pixel slab, two pixel fragments each.  Round 4:
    // wave wn walks the pixel fragments ROTATED by 2 wn (accumulator index mi <-> pixel fragment PF(mi) = (mi + 2 wn) & 7), so that its two
    // extra MFMAs sit at the static steps mi = 0, 1 of each K half for EVERY wave: the main loop has no wave-dependent branch any more (it had
    // 16 `if (wn == mi >> 1)` scalar branches per stage around single MFMAs: ~5 % of the kernel).  The rotation is a scalar offset folded into the
    // fragment address the stage toggle already adds; the epilogue maps mi back with PF().
    const int rot = EXTRA ? wn * 2 * 2048 : 0;
    auto aoff = [&](int mi) -> int { return EXTRA ? ((mi * 2048 + rot) & (FM * 2048 - 1)) : mi * 2048; };
#define PF(mi) (EXTRA ? (((mi) + 2 * wn) & (FM - 1)) : (mi))

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
    // this block's stages [kt0, kt1): the whole K, or part `part` of SK with EVEN boundaries (the LDS buffer of stage kt is kt & 1)
    const int kt0 = SK > 1 ? ((KT * part / SK) & ~1) : 0;
    const int kt1 = SK > 1 ? (part + 1 == SK ? KT : ((KT * (part + 1) / SK) & ~1)) : KT;
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
    // ABL 20 / 21 / 22 (round-3 schedule experiments, correct results): 20 = the remaining pieces in the first NS/4 steps instead of NS/2
    // (more time to land before the barrier), 21 = three early pieces (steps NS-3 .. NS-1, the first one right behind the barrier), 22 = both
    constexpr int EARLY = (ABL == 21 || ABL == 22) ? 3 : 2;    // pieces of stage kt+2 issued in the last steps of stage kt
    // the other pieces go out in the first NSD steps of stage kt+1.  (Round 4 re-measured NS / 4 for the rotated extra-column tile: 3 % faster in the
    // isolated conv_bench, nothing in bench.py in a same-box A/B -- and two byte-identical instantiations of the kernel differ by 1 - 2 % in that micro
    // benchmark depending on their position in the code object.  Not adopted: profiles/r4_conv_schedule_ab.txt.)
    constexpr int NSD = (ABL == 20 || ABL == 22) ? NS / 4 : NS / 2;
    auto dma_piece = [&](int q, char* buf, int2 e, int kstage) {
        if (q < A_IT) dma16(rx, buf + (wave + q * NW) * 1024, a_voff(q, e), 0);
        else dma16(rw, buf + BM * 128 + (wave + (q - A_IT) * NW) * 1024, b_voff[q - A_IT], (unsigned)kstage * 128u);
    };
    int2 e_nx = kt_lane[kt0 * 8];
#pragma unroll
    for (int q = 0; q < P; ++q) dma_piece(q, smem, e_nx, kt0);
    if (EXTRA && extra_wave) dma16(rw, smem + (BM + BN) * 128 + (wave & 1) * 1024, x_voff, 0);
    e_nx = kt_lane[(kt1 - kt0 > 1 ? kt0 + 1 : kt0) * 8];
    if (kt1 - kt0 > 1) {
#pragma unroll
        for (int q = 0; q < EARLY; ++q) dma_piece(q, smem + G::STAGE_BYTES, e_nx, kt0 + 1);
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
    af[0] = *reinterpret_cast<const half8*>(smem + a_l0 + aoff(0));
    af[1] = *reinterpret_cast<const half8*>(smem + a_l0 + aoff(1));

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
                if (s2 < NS && RD) af[s2 % 4] = *reinterpret_cast<const half8*>(cur + ((s2 / FM) ? a_l1 : a_l0) + aoff(s2 % FM));
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
                af[(s + 2) % 4] = *reinterpret_cast<const half8*>(nxt + a_l0 + aoff(j));
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
            if (LVL == 2 && DMA_ON && s >= NS - EARLY) //     and, behind the barrier, the first pieces of stage kt+2
                dma_piece(s - (NS - EARLY), const_cast<char*>(cur), e_n2, kt + 2);
            const half8 a = af[s % 4];                 // (4) the MFMAs of this step
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                if (ABL == 4 || ABL == 9) asm volatile("" ::"v"(bf[ks][ni]), "v"(a));
                else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][ni], a, acc[ni][mi], 0, 0, 0);
            }
            if (EXTRA && mi < 2)                       // extra column fragment x this wave's pixel fragments PF(0), PF(1) = 2 wn, 2 wn + 1: static steps
                accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], a, accx[mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        e_nx = e_n2;
    };
    // Static priority for the younger half of an 8-wave block (two waves per SIMD): waves 4-7 lose the VALU / issue arbitration to the older
    // half on every stage (priority, then age: MI355X_MICROARCH.md "Two waves per SIMD", item 4).  ONE s_setprio before the loop, no flips:
    // 5.99 -> 5.83 ms on the tail conv (interleaved same-box A/B, profiles/r3_conv_experiments.txt).  Scheduling only: same bytes.
    if (NW == 8 && p.prio && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    int kt = kt0;
    for (; kt + 2 < kt1; ++kt) stage(kt, std::integral_constant<int, 2>{});
    if (kt1 - kt0 >= 2) { stage(kt, std::integral_constant<int, 1>{}); ++kt; }
    stage(kt, std::integral_constant<int, 0>{});

    auto gm = [&](int local_row) -> int { return m0 + local_row; };
    const int lp = lr;                                     // MFMA column lr carries pixel lr of its fragment
    if (!EXTRA && SK > 1) {                                // split-K: raw fp32 partial sums, lane = pixel lr, channels lg*4 .. +3 of each fragment
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int m = m0 + wm * (FM * 16) + mi * 16 + lr;
            if (m >= p.M) continue;
            float* row = p.ws + ((int64_t)part * p.M + m) * p.Npad;
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + lg * 4;
                if (n < p.Npad) *reinterpret_cast<float4v*>(row + n) = acc[ni][mi];
            }
        }
        return;
    }
#include "conv_pipe_epilogue.inc"
#undef PF
}

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
#include "conv_pipe_epilogue.inc"
#undef PF
}

// Opt a kernel in to > 64 KiB of dynamic LDS.  The attribute is per DEVICE (a process may hold contexts on several GPUs:
// render.get_context caches one per device_index), so the "done" state is a bit per device ordinal, set with an atomic OR
// (VapourSynth worker threads may race here; setting the attribute twice is harmless).
template <auto Kernel>
static void ensure_lds_optin(int lds_bytes) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    done.fetch_or(bit, std::memory_order_release);
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

template <int WM, int WN, int FM, int EXTRA, int ABL = 0>
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
    ensure_lds_optin<conv_pipe_kernel<WM, WN, FM, EXTRA, ABL>>(LDS);
    hipLaunchKernelGGL((conv_pipe_kernel<WM, WN, FM, EXTRA, ABL>), dim3(MT * NT * SK), dim3(G::NW * 64), LDS, s, a);
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
