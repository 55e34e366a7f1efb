// Implicit-GEMM convolution, software-pipelined main loop (the kernel the big decoder convs run on).
//
// Why: PMC + ablation of the ring kernel (profiles/r1_conv_ablation.txt) showed MFMA time (0.72 ms) and
// everything else (im2col address generation + LDS-DMA issue 0.47 ms, fragment reads 0.17 ms, barrier 0.22 ms,
// epilogue) adding up with ZERO overlap: a wave issues in order, and the compiler clustered all address VALU and
// ds_reads ahead of the 32-MFMA block, so the matrix pipe idled 60 % of the time (MfmaUtil 38 %).
//
// Structure: 256 x 256 block tile (L1/TA traffic per flop halves vs 128^2: the vector-memory path is 64 B/clk/CU),
// 8 waves as 2(M) x 4(N), wave tile 128 x 64, K-step 64 = two 32-deep sub-tiles, 2-stage LDS double buffer
// (128 KiB), ONE barrier per 64-deep step.  The body of a stage is 16 "steps" of 4 MFMAs; between the MFMA groups
// each step carries one slice of the next stage's work, pinned in place with sched_barrier:
//   * ds_read of the A fragment two steps ahead (3-deep register ring; B fragments of the next sub-tile
//     are prefetched in the second half of a sub-tile),
//   * one 1-KiB LDS-DMA piece of the NEXT stage (branch-free im2col address: per-lane incremental k-state,
//     out-of-image / K-tail lanes read a zero page),
// so address VALU, LDS latency and DMA issue all sit in the shadow of the matrix pipe.
// EXTRA = 1: 16 extra output columns (the 259-channel tail, Npad = 256 + 16) as 2 more MFMAs per wave per sub-tile.
#include "conv_common.h"
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, NW = 8;   // 8 waves as 2 (M) x 4 (N)
constexpr int FM = 8, FN = 4;            // 16-row / 16-col fragments per wave
constexpr int KSUB = 2;

template <int EXTRA>
struct Geo {
    static constexpr int BNX = BN + 16 * EXTRA;
    static constexpr int SUB = (BM + BNX) * 32;      // halfs per 32-deep sub-tile
    static constexpr int STAGE = SUB * KSUB;
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
};

}  // namespace

// ABL (profiling only, wrong results): 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no MFMA, 5 = no epilogue stores
template <int EXTRA, int ABL = 0>
__global__ void __launch_bounds__(512) conv_pipe_kernel(const ConvArgs p) {
    using G = Geo<EXTRA>;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;

    const int nwg = gridDim.x;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int NT = (p.Npad - 16 * EXTRA + BN - 1) / BN;
    const int m0 = (pid / NT) * BM;
    const int n0 = (pid % NT) * BN;
    const bool has_extra = EXTRA && (n0 + BN + 16 == p.Npad);

    const int HoWo = p.Ho * p.Wo;
    // ---- DMA lane role (see conv_igemm_glds.hip): piece row lane>>2, LDS position lane&3, source chunk j ----
    const int prow = lane >> 2;
    const int j = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const half_t* zero = reinterpret_cast<const half_t*>(havc_zero_page);

    // A rows owned by this lane: pieces `wave` and `wave + 8` of the 16 A pieces
    const half_t* a_ptr[2];
    int64_t a_zoff[2];                               // (zero page - a_ptr) in halfs: select an OFFSET, not a pointer,
    int a_hi0[2], a_wi0[2];                          // so the bounds test compiles to v_cndmask instead of a branch
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (wave + it * NW) * 16 + prow;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_hi0[it] = ok ? hi0 : -(1 << 20);          // an out-of-range row never passes the bounds test
        a_wi0[it] = wi0;
        a_ptr[it] = p.x + ((int64_t)(b * p.Hi * p.Wi) + (int64_t)hi0 * p.Wi + wi0) * p.x_cpitch + p.x_coff;
        a_zoff[it] = ((intptr_t)zero - (intptr_t)a_ptr[it]) / 2;
    }
    // B rows: pieces `wave`, `wave + 8` (+ the extra piece 16 on wave 7)
    const half_t* b_ptr[2];
    int b_stepv[2];                                   // halfs per sub-tile along K (4 chunks); 0 for rows >= Npad (zero page)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int n = n0 + (wave + it * NW) * 16 + prow;
        b_ptr[it] = (n < p.Npad) ? p.w + ((int64_t)n * p.Kc + j) * 8 : zero;
        b_stepv[it] = (n < p.Npad) ? 32 : 0;
    }
    const int b_step = 32;
    const half_t* bx_ptr = p.w + ((int64_t)(n0 + BN + prow) * p.Kc + j) * 8;
    const bool extra_wave = has_extra && wave == NW - 1;

    // ---- branch-free incremental k-state of this lane's chunk (requires C8 >= 4) ----
    int kc8 = j, dh = 0, dw = 0;
    int koff = j * 8;
    const int stepw = p.dil * p.x_cpitch - p.C8 * 8;
    const int steph = p.dil * p.Wi * p.x_cpitch - p.kw * p.dil * p.x_cpitch;
    const int dw_end = p.kw * p.dil, dh_end = p.kh * p.dil;
    auto k_advance = [&]() {
        kc8 += 4;
        koff += 32;
        const bool pw = kc8 >= p.C8;
        kc8 -= pw ? p.C8 : 0;
        koff += pw ? stepw : 0;
        dw += pw ? p.dil : 0;
        const bool qh = dw == dw_end;                  // only possible right after a wrap
        dw = qh ? 0 : dw;
        dh += qh ? p.dil : 0;
        koff += qh ? steph : 0;
    };
    auto a_src = [&](int it) -> const half_t* {
        const int hi = a_hi0[it] + dh, wi = a_wi0[it] + dw;
        const bool ok = (dh < dh_end) & ((unsigned)hi < (unsigned)p.Hi) & ((unsigned)wi < (unsigned)p.Wi);
        return a_ptr[it] + (ok ? (int64_t)koff : a_zoff[it]);
    };

    const int KT = p.Kc >> 3;                         // 64-deep stages

    // ---- fragment read addresses: per-lane constant + immediates ----
    const int a_lds = ((wm * 128 + lr) * 4 + (lg ^ ((4 - ((lr >> 2) & 3)) & 3))) * 8;     // halfs, A frag 0
    const int b_lds = BM * 32 + ((wn * 64 + lr) * 4 + (lg ^ ((4 - ((lr >> 2) & 3)) & 3))) * 8;
    const int x_lds = BM * 32 + ((BN + lr) * 4 + (lg ^ ((4 - ((lr >> 2) & 3)) & 3))) * 8;

    float4v acc[FN][FM];
    float4v accx[2];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
    accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};

    // issue the DMA pieces of one sub-tile (prologue form, not interleaved)
    auto issue_sub = [&](half_t* sub) {
        glds16(a_src(0), sub + wave * 512);
        glds16(a_src(1), sub + (wave + NW) * 512);
        k_advance();
        glds16(b_ptr[0], sub + BM * 32 + wave * 512);
        glds16(b_ptr[1], sub + BM * 32 + (wave + NW) * 512);
        if (EXTRA && extra_wave) glds16(bx_ptr, sub + BM * 32 + 16 * 512);
        b_ptr[0] += b_stepv[0]; b_ptr[1] += b_stepv[1]; bx_ptr += b_step;
    };
    issue_sub(smem);
    issue_sub(smem + G::SUB);

    // one 64-deep stage; MORE (compile time) = the next stage exists and its DMA is issued from inside this one.
    // Straight-line: no branch inside a stage except the per-wave extra-column MFMA.
    auto stage = [&](const half_t* cur, half_t* nxt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value && ABL != 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        half8 bf[2][FN];                               // B fragments of sub-tile 0 / 1
        half8 xb[2];                                   // extra-column B fragment
        half8 af[3];                                   // A fragment ring
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) bf[0][ni] = *reinterpret_cast<const half8*>(cur + b_lds + ni * 16 * 32);
        if (EXTRA) xb[0] = *reinterpret_cast<const half8*>(cur + x_lds);
        af[0] = *reinterpret_cast<const half8*>(cur + a_lds);
        af[1] = *reinterpret_cast<const half8*>(cur + a_lds + 16 * 32);

#pragma unroll
        for (int s = 0; s < 16; ++s) {                 // 16 steps of 4 (+extra) MFMAs
            const int h = s >> 3, mi = s & 7;
            // (1) prefetch the A fragment two steps ahead (within this stage)
            if (s + 2 < 16 && ABL != 2) {
                const int s2 = s + 2;
                af[s2 % 3] = *reinterpret_cast<const half8*>(cur + (s2 >> 3) * G::SUB + a_lds + (s2 & 7) * 16 * 32);
            }
            // (2) prefetch sub-tile 1's B fragments during steps 4..7
            if (s >= 4 && s < 8 && ABL != 2) bf[1][s - 4] = *reinterpret_cast<const half8*>(cur + G::SUB + b_lds + (s - 4) * 16 * 32);
            if (EXTRA && s == 3) xb[1] = *reinterpret_cast<const half8*>(cur + G::SUB + x_lds);
            // (3) one DMA piece of the next stage per step (steps 0..4 -> its sub-tile 0, steps 8..12 -> sub-tile 1)
            if (MORE) {
                const int hs = s >> 3, ps = s & 7;
                half_t* sub = nxt + hs * G::SUB;
                if (ps == 0) glds16(a_src(0), sub + wave * 512);
                if (ps == 1) { glds16(a_src(1), sub + (wave + NW) * 512); k_advance(); }
                if (ps == 2) glds16(b_ptr[0], sub + BM * 32 + wave * 512);
                if (ps == 3) {
                    glds16(b_ptr[1], sub + BM * 32 + (wave + NW) * 512);
                    b_ptr[0] += b_stepv[0]; b_ptr[1] += b_stepv[1];
                }
                if (EXTRA && ps == 4) {
                    if (extra_wave) glds16(bx_ptr, sub + BM * 32 + 16 * 512);
                    bx_ptr += b_step;
                }
            }
            // (4) the MFMAs of this step
            const half8 a = af[ABL == 2 ? s % 2 : s % 3];
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                if (ABL == 4) asm volatile("" :: "v"(bf[ABL == 2 ? 0 : h][ni]), "v"(a));
                else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ABL == 2 ? 0 : h][ni], a, acc[ni][mi], 0, 0, 0);
            }
            if (EXTRA && has_extra) {
                // extra column fragment x row fragment mi: owned by N-wave (mi >> 1)  (2 per wave)
                if (wn == (mi >> 1)) accx[mi & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[h], a, accx[mi & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0;
    for (; kt + 1 < KT; ++kt)
        stage(smem + (kt & 1) * G::STAGE, smem + ((kt & 1) ^ 1) * G::STAGE, std::true_type{});
    stage(smem + (kt & 1) * G::STAGE, smem + ((kt & 1) ^ 1) * G::STAGE, std::false_type{});

    // ---- epilogue ----
    if (ABL == 5) {
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi)
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) t += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (t == 123.456f) reinterpret_cast<half_t*>(p.y)[0] = (half_t)t;
        return;
    }
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * 128 + mi * 16 + lr;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) epilogue_frag(p, acc[ni][mi], m, n0 + wn * 64 + ni * 16 + lg * 4, HoWo);
    }
    if (EXTRA && has_extra) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 128 + (wn * 2 + i) * 16 + lr;
            epilogue_frag(p, accx[i], m, n0 + BN + lg * 4, HoWo);
        }
    }
}

template <int EXTRA, int ABL = 0>
static int launch_pipe(const ConvArgs& a, hipStream_t s) {
    if ((a.Kc & 7) || a.C8 < 4) return (int)hipErrorInvalidValue;
    const int MT = (a.M + BM - 1) / BM, NT = (a.Npad - 16 * EXTRA + BN - 1) / BN;
    constexpr int LDS = Geo<EXTRA>::LDS_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipe_kernel<EXTRA, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_pipe_kernel<EXTRA, ABL>), dim3(MT * NT), dim3(512), LDS, s, a);
    return (int)hipGetLastError();
}

int launch_conv_pipe(const ConvArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case 60: return launch_pipe<0>(a, s);
        case 61: return launch_pipe<1>(a, s);
        case 62: return launch_pipe<0, 1>(a, s);
        case 63: return launch_pipe<0, 2>(a, s);
        case 64: return launch_pipe<0, 4>(a, s);
        case 65: return launch_pipe<0, 5>(a, s);
    }
    return (int)hipErrorInvalidValue;
}
