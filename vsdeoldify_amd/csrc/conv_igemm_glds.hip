// Implicit-GEMM convolution, v2: operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPR
// round trip and no ds_write pass (on gfx950 ds_write_b128 sustains only ~79 B/clk/CU, which made the
// register-staged v1 kernel LDS-bound: 16 KB of tile writes + 32 KB of fragment reads per K-step against
// ~272 MFMA cycles).  Same GEMM view, fragment layout and epilogue as conv_igemm.hip.
//
// LDS-DMA writes lane-linear (wave-uniform base + lane*16 B), so one wave-instruction fills 16 tile rows
// x 4 chunks; the bank-conflict swizzle therefore lives on the SOURCE side: lane (row, pos) fetches chunk
// j = pos ^ g(row) and fragment reads use the same involution.  Padding / M-tail / K-tail lanes fetch from a
// 16-byte zero page instead of being predicated off (the DMA has no per-lane zero fill).
//
// EXTRA: the awkward 259-channel tail (Npad = 272 = 2*128 + 16).  The last N tile carries 16 extra output
// columns whose BM/16 row fragments are spread evenly over the block's waves (each wave reuses X fragments
// it already holds), so MFMA work is exactly proportional to the 272 useful columns.
#include "conv_common.h"

template <int BM, int BN, int WM, int WN, int EXTRA>
__global__ void __launch_bounds__(WM* WN * 64) conv_glds_kernel(const ConvArgs p) {
    constexpr int NW = WM * WN;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int BNX = BN + 16 * EXTRA;
    constexpr int QA = BM / 16, QB = BN / 16;     // 1-KiB DMA pieces per stage (QB excludes the extra piece)
    constexpr int A_IT = QA / NW, B_IT = (QB + NW - 1) / NW;
    constexpr int XF = EXTRA ? FM / WN : 0;       // extra-column row fragments per wave
    static_assert(QA % NW == 0, "A pieces must divide evenly over waves");
    static_assert(!EXTRA || FM % WN == 0, "extra fragments must divide over the N waves");
    constexpr int STAGE = (BM + BNX) * 32;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE];

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
    const int m0 = (pid / NT) * BM;
    const int n0 = (pid % NT) * BN;
    const bool has_extra = EXTRA && (n0 + BN + 16 == p.Npad);

    const int HoWo = p.Ho * p.Wo;
    // DMA lane role: tile row (within a 16-row piece) lane>>2, LDS chunk position lane&3,
    // source chunk j = pos ^ g(row); (row>>2)&3 == (lane>>4)&3 for every piece.
    const int prow = lane >> 2;
    const int j = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const half_t* zero = reinterpret_cast<const half_t*>(havc_zero_page);

    int a_pix[A_IT], a_hi0[A_IT], a_wi0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (wave + it * NW) * 16 + prow;
        const int m = m0 + row;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_hi0[it] = ho * p.stride - p.pad;
        a_wi0[it] = wo * p.stride - p.pad;
        a_pix[it] = b * p.Hi * p.Wi;
    }
    int kc8 = j, kkh = 0, kkw = 0;
    while (kc8 >= p.C8) {
        kc8 -= p.C8;
        if (++kkw == p.kw) { kkw = 0; ++kkh; }
    }
    const int KT = p.Kc >> 2;

    auto issue = [&](int kt, int buf) {
        half_t* As = smem + buf * STAGE;
        half_t* Bs = As + BM * 32;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int hi = a_hi0[it] + kkh * p.dil;
            const int wi = a_wi0[it] + kkw * p.dil;
            const bool ok = a_ok[it] && kkh < p.kh && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            const half_t* src = ok ? p.x + ((int64_t)(a_pix[it] + hi * p.Wi + wi) * p.x_cpitch + p.x_coff + kc8 * 8) : zero;
            glds16(src, As + (wave + it * NW) * 512);
        }
        kc8 += 4;
        while (kc8 >= p.C8) {
            kc8 -= p.C8;
            if (++kkw == p.kw) { kkw = 0; ++kkh; }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            if (QB % NW != 0 && wave + it * NW >= QB) break;
            const int row = (wave + it * NW) * 16 + prow;
            const half_t* src = (n0 + row < p.Npad) ? p.w + ((int64_t)(n0 + row) * p.Kc + kt * 4 + j) * 8 : zero;
            glds16(src, Bs + (wave + it * NW) * 512);
        }
        if (EXTRA && has_extra && wave == NW - 1) {      // the 16 extra weight rows: one more piece, last wave
            const half_t* src = p.w + ((int64_t)(n0 + BN + prow) * p.Kc + kt * 4 + j) * 8;
            glds16(src, Bs + QB * 512);
        }
    };

    float4v acc[FN][FM];
    float4v accx[XF > 0 ? XF : 1];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (XF > 0 ? XF : 1); ++i) accx[i] = float4v{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) issue(kt + 1, buf ^ 1);
        const half_t* As = smem + buf * STAGE;
        const half_t* Bs = As + BM * 32;
        half8 xf[FM];
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int row = wm * (BM / WM) + mi * 16 + lr;
            xf[mi] = *reinterpret_cast<const half8*>(As + (row * 4 + (lg ^ swz2(row))) * 8);
        }
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int row = wn * (BN / WN) + ni * 16 + lr;
            const half8 wf = *reinterpret_cast<const half8*>(Bs + (row * 4 + (lg ^ swz2(row))) * 8);
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (EXTRA && has_extra) {
            const int row = BN + lr;
            const half8 wf = *reinterpret_cast<const half8*>(Bs + (row * 4 + (lg ^ swz2(row))) * 8);
#pragma unroll
            for (int w = 0; w < WN; ++w)          // static xf index: a runtime-indexed vector array goes to scratch
                if (wn == w) {
#pragma unroll
                    for (int i = 0; i < XF; ++i)
                        accx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[w * XF + i], accx[i], 0, 0, 0);
                }
        }
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + lr;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) epilogue_frag(p, acc[ni][mi], m, n0 + wn * (BN / WN) + ni * 16 + lg * 4, HoWo);
    }
    if (EXTRA && has_extra) {
#pragma unroll
        for (int i = 0; i < XF; ++i) {
            const int m = m0 + wm * (BM / WM) + (wn * XF + i) * 16 + lr;
            epilogue_frag(p, accx[i], m, n0 + BN + lg * 4, HoWo);
        }
    }
}

template <int BM, int BN, int WM, int WN, int EXTRA>
static int launch_glds(const ConvArgs& a, hipStream_t s) {
    const int MT = (a.M + BM - 1) / BM, NT = (a.Npad - 16 * EXTRA + BN - 1) / BN;
    hipLaunchKernelGGL((conv_glds_kernel<BM, BN, WM, WN, EXTRA>), dim3(MT * NT), dim3(WM * WN * 64), 0, s, a);
    return (int)hipGetLastError();
}

// ---- v3: 3-buffer LDS ring, counted vmcnt, ONE raw s_barrier per K-step ---------------------------------
// Two tiles of DMA stay in flight across every barrier (the __syncthreads() of the 2-stage kernel drains
// vmcnt(0) each K-step, exposing the full L2/HBM latency at 1-2 waves/SIMD).  Protocol per K-step kt:
//   s_waitcnt vmcnt(P)   this wave's pieces of tile kt have landed (P = its pieces of tile kt+1 still in flight)
//   s_barrier            => every wave's pieces of tile kt landed AND every wave finished reading tile kt-1
//   issue tile kt+2 into the buffer tile kt-1 occupied;  compute tile kt
template <int P>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(P >= 0 && P <= 24, "vmcnt immediate");
    if constexpr (P == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (P == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (P == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (P == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (P == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (P == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (P == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (P == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (P == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (P == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (P == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (P == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (P == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (P == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (P == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (P == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (P == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else if constexpr (P == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (P == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    else if constexpr (P == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (P == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if constexpr (P == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (P == 23) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
    else if constexpr (P == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

// ABL: ablation variants for profiling only (results are WRONG): 1 = no DMA issue in the loop, 2 = no LDS fragment
// reads (fragments kept from the first K-step), 3 = no barrier / vmcnt waits, 4 = no MFMA.
template <int BM, int BN, int WM, int WN, int EXTRA, int KSUB, int STAGES, int ABL = 0>
__global__ void __launch_bounds__(WM* WN * 64) conv_ring_kernel(const ConvArgs p) {
    // KSUB: 32-deep sub-tiles per stage.  KSUB = 2 makes a wave fetch both 64-byte halves of every 128-byte
    // line back to back (K-step 64): with K-step 32 each L2->L1 line fill was used for half its bytes and the
    // kernels plateaued at ~780 TFLOP/s on L2 bandwidth regardless of tile shape (profiles/r1_convbench_*.txt).
    constexpr int NW = WM * WN;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int BNX = BN + 16 * EXTRA;
    constexpr int QA = BM / 16, QB = BN / 16;
    constexpr int A_IT = QA / NW, B_IT = QB / NW;
    constexpr int PIECES = (A_IT + B_IT) * KSUB;    // DMA instructions per wave per stage (+KSUB on the extra wave)
    constexpr int XF = EXTRA ? (BM / 16) / NW : 0;
    static_assert(QA % NW == 0 && QB % NW == 0, "pieces must divide evenly over waves");
    static_assert(!EXTRA || ((BM / 16) % NW == 0 && FM % WN == 0 && XF * WN == FM), "extra fragment split");
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");
    static_assert(PIECES + KSUB <= 24, "wait_vmcnt immediates");
    constexpr int SUB = (BM + BNX) * 32;            // halfs per 32-deep sub-tile
    constexpr int STAGE = SUB * KSUB;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

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
    const int m0 = (pid / NT) * BM;
    const int n0 = (pid % NT) * BN;
    const bool has_extra = EXTRA && (n0 + BN + 16 == p.Npad);
    const bool extra_wave = has_extra && wave == NW - 1;

    const int HoWo = p.Ho * p.Wo;
    const int prow = lane >> 2;
    const int j = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const half_t* zero = reinterpret_cast<const half_t*>(havc_zero_page);

    int a_pix[A_IT], a_hi0[A_IT], a_wi0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (wave + it * NW) * 16 + prow;
        const int m = m0 + row;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_hi0[it] = ho * p.stride - p.pad;
        a_wi0[it] = wo * p.stride - p.pad;
        a_pix[it] = b * p.Hi * p.Wi;
    }
    int kc8 = j, kkh = 0, kkw = 0;
    while (kc8 >= p.C8) {
        kc8 -= p.C8;
        if (++kkw == p.kw) { kkw = 0; ++kkh; }
    }
    const int KT = p.Kc / (4 * KSUB);

    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int h = 0; h < KSUB; ++h) {
            half_t* As = smem + buf * STAGE + h * SUB;
            half_t* Bs = As + BM * 32;
            const int kc0 = (kt * KSUB + h) * 4;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int hi = a_hi0[it] + kkh * p.dil;
                const int wi = a_wi0[it] + kkw * p.dil;
                const bool ok = a_ok[it] && kkh < p.kh && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
                const half_t* src = ok ? p.x + ((int64_t)(a_pix[it] + hi * p.Wi + wi) * p.x_cpitch + p.x_coff + kc8 * 8) : zero;
                glds16(src, As + (wave + it * NW) * 512);
            }
            kc8 += 4;
            while (kc8 >= p.C8) {
                kc8 -= p.C8;
                if (++kkw == p.kw) { kkw = 0; ++kkh; }
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int row = (wave + it * NW) * 16 + prow;
                const half_t* src = (n0 + row < p.Npad) ? p.w + ((int64_t)(n0 + row) * p.Kc + kc0 + j) * 8 : zero;
                glds16(src, Bs + (wave + it * NW) * 512);
            }
            if (EXTRA && extra_wave) {
                const half_t* src = p.w + ((int64_t)(n0 + BN + prow) * p.Kc + kc0 + j) * 8;
                glds16(src, Bs + QB * 512);
            }
        }
    };

    float4v acc[FN][FM];
    float4v accx[XF > 0 ? XF : 1];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (XF > 0 ? XF : 1); ++i) accx[i] = float4v{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) issue(s, s);

    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
        // this wave's pieces of tile kt must have landed; tiles kt+1 .. kt+STAGES-2 may stay in flight
        if (ABL != 3) {
            if (STAGES == 3 && kt + 1 < KT) {
                if (EXTRA && extra_wave) wait_vmcnt<PIECES + KSUB>(); else wait_vmcnt<PIECES>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        if (ABL != 1 && kt + STAGES - 1 < KT) issue(kt + STAGES - 1, buf == 0 ? STAGES - 1 : buf - 1);
#pragma unroll
        for (int h = 0; h < KSUB; ++h) {
            const half_t* As = smem + (ABL == 2 ? 0 : buf * STAGE) + h * SUB;
            const half_t* Bs = As + BM * 32;
            half8 xf[FM];
            if (ABL == 2 && kt > 0) {
#pragma unroll
                for (int mi = 0; mi < FM; ++mi) { xf[mi] = half8{1, 2, 3, 4, 5, 6, 7, 8}; asm volatile("" : "+v"(xf[mi])); }
            } else {
#pragma unroll
                for (int mi = 0; mi < FM; ++mi) {
                    const int row = wm * (BM / WM) + mi * 16 + lr;
                    xf[mi] = *reinterpret_cast<const half8*>(As + (row * 4 + (lg ^ swz2(row))) * 8);
                }
            }
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int row = wn * (BN / WN) + ni * 16 + lr;
                half8 wf;
                if (ABL == 2 && kt > 0) { wf = half8{1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(wf)); }
                else wf = *reinterpret_cast<const half8*>(Bs + (row * 4 + (lg ^ swz2(row))) * 8);
#pragma unroll
                for (int mi = 0; mi < FM; ++mi) {
                    if (ABL == 4) { asm volatile("" :: "v"(wf), "v"(xf[mi])); }
                    else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mi], acc[ni][mi], 0, 0, 0);
                }
            }
            if (EXTRA && has_extra) {
                const int row = BN + lr;
                const half8 wf = *reinterpret_cast<const half8*>(Bs + (row * 4 + (lg ^ swz2(row))) * 8);
#pragma unroll
                for (int w = 0; w < WN; ++w)
                    if (wn == w) {
#pragma unroll
                        for (int i = 0; i < XF; ++i)
                            accx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[w * XF + i], accx[i], 0, 0, 0);
                    }
            }
        }
        buf = buf == STAGES - 1 ? 0 : buf + 1;
    }

#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + lr;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) epilogue_frag(p, acc[ni][mi], m, n0 + wn * (BN / WN) + ni * 16 + lg * 4, HoWo);
    }
    if (EXTRA && has_extra) {
#pragma unroll
        for (int w = 0; w < WN; ++w)
            if (wn == w) {
#pragma unroll
                for (int i = 0; i < XF; ++i) {
                    const int m = m0 + wm * (BM / WM) + (w * XF + i) * 16 + lr;
                    epilogue_frag(p, accx[i], m, n0 + BN + lg * 4, HoWo);
                }
            }
    }
}

template <int BM, int BN, int WM, int WN, int EXTRA, int KSUB, int STAGES, int ABL = 0>
static int launch_ring(const ConvArgs& a, hipStream_t s) {
    if (a.Kc % (4 * KSUB) != 0) return (int)hipErrorInvalidValue;
    const int MT = (a.M + BM - 1) / BM, NT = (a.Npad - 16 * EXTRA + BN - 1) / BN;
    constexpr int LDS = STAGES * KSUB * (BM + BN + 16 * EXTRA) * 32 * 2;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_kernel<BM, BN, WM, WN, EXTRA, KSUB, STAGES, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_ring_kernel<BM, BN, WM, WN, EXTRA, KSUB, STAGES, ABL>), dim3(MT * NT), dim3(WM * WN * 64), LDS, s, a);
    return (int)hipGetLastError();
}

// cfg ids 16.. (v1 register-staged kernels keep 1..15)
int launch_conv_glds(const ConvArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case 16: return launch_glds<128, 128, 2, 2, 0>(a, s);
        case 17: return launch_glds<128, 128, 2, 2, 1>(a, s);
        case 18: return launch_glds<64, 128, 2, 2, 0>(a, s);
        case 19: return launch_glds<64, 64, 2, 2, 0>(a, s);
        case 20: return launch_glds<128, 64, 2, 2, 0>(a, s);
        case 21: return launch_glds<256, 128, 4, 2, 0>(a, s);
        case 22: return launch_glds<256, 128, 4, 2, 1>(a, s);
        case 23: return launch_glds<128, 256, 2, 4, 0>(a, s);
        case 24: return launch_glds<128, 16, 4, 1, 0>(a, s);
        // ring kernels: <BM, BN, WM, WN, EXTRA, KSUB, STAGES>
        case 32: return launch_ring<256, 256, 2, 4, 0, 1, 3>(a, s);
        case 33: return launch_ring<256, 256, 2, 4, 1, 1, 3>(a, s);
        case 34: return launch_ring<256, 128, 4, 2, 0, 1, 3>(a, s);
        case 35: return launch_ring<128, 128, 2, 2, 0, 1, 3>(a, s);
        case 36: return launch_ring<128, 256, 2, 4, 0, 1, 3>(a, s);
        case 40: return launch_ring<256, 256, 2, 4, 0, 2, 2>(a, s);     // K-step 64, full 128-B lines
        case 41: return launch_ring<256, 256, 2, 4, 1, 2, 2>(a, s);
        case 42: return launch_ring<256, 128, 4, 2, 0, 2, 3>(a, s);
        case 43: return launch_ring<256, 128, 4, 2, 0, 2, 2>(a, s);
        case 44: return launch_ring<128, 128, 2, 2, 0, 2, 3>(a, s);
        case 45: return launch_ring<128, 128, 2, 2, 0, 2, 2>(a, s);
        case 46: return launch_ring<128, 256, 2, 4, 0, 2, 3>(a, s);
        case 47: return launch_ring<256, 128, 4, 2, 1, 2, 3>(a, s);
        case 51: return launch_ring<256, 256, 2, 4, 0, 1, 3, 1>(a, s);   // ablations (wrong results, profiling only)
        case 52: return launch_ring<256, 256, 2, 4, 0, 1, 3, 2>(a, s);
        case 53: return launch_ring<256, 256, 2, 4, 0, 1, 3, 3>(a, s);
        case 54: return launch_ring<256, 256, 2, 4, 0, 1, 3, 4>(a, s);
        case 55: return launch_ring<256, 128, 4, 2, 0, 1, 3, 1>(a, s);
        case 56: return launch_ring<256, 128, 4, 2, 0, 1, 3, 2>(a, s);
        case 57: return launch_ring<256, 128, 4, 2, 0, 1, 3, 3>(a, s);
        case 58: return launch_ring<256, 128, 4, 2, 0, 1, 3, 4>(a, s);
    }
    return (int)hipErrorInvalidValue;
}
