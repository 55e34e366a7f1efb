// ColorMNet memory kernels: the two per-frame reads of the exemplar path, fp32 like the reference tensors.
//   memory read      colormnet/model/memory_util.py:7-80 (get_similarity + do_softmax(top_k) + readout; inference/memory_manager.py:58-150)
//   local attention  colormnet/model/attention.py:783-856 (LocalGatedPropagation.forward; :827-835 is the local correlation the
//                    reference takes from the CUDA-only spatial_correlation_sampler wheel)
// Layouts are the reference's: NCHW / [B][C][N] fp32, contiguous.  None of this is GEMM-shaped enough to pay for MFMA at fp32
// (157 TF/s = the vector rate): the similarity is a 64-deep contraction, everything else is selection / gather / window sums --
// LDS-tiled fp32 FMA, coalesced along the query / pixel axis.
#include "kernels.h"
#include <atomic>

#include <cmath>

namespace {

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- similarity[n][q] = (-sum_c mk^2 qe + 2 sum_c mk (qk qe) - sum_c qe qk^2) * ms[n] / sqrt(CK)   (memory_util.py:19-37) ----
// Block = 64 memory entries x 64 queries, 256 threads, thread = 4 x 4 outputs; the three sums are kept apart and combined as the
// reference combines them (-a_sq + two_ab - b_sq).
// TRANSPOSED: the result goes out query-major, simT[b][q][n] (the wave-per-query top-k below reads one contiguous row per query)
template <bool HAS_QE, bool TRANSPOSED = false>
__global__ void __launch_bounds__(256) mem_similarity_kernel(const float* __restrict__ mk, const float* __restrict__ ms, const float* __restrict__ qk,
                                                             const float* __restrict__ qe, float* __restrict__ sim, int CK, int N, int HW, float sqrt_ck,
                                                             int64_t mpitch) {
    // mpitch: row pitch of mk in elements (N for the reference's contiguous [CK][N] tensors; larger when the memory lives in a pre-sized bank)
    __shared__ float Ms[16][64 + 1], Qs[16][64 + 1], Es[16][64 + 1];
    const int b = blockIdx.z, n0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
    const int tid = threadIdx.x, tq = tid & 15, tn = tid >> 4;
    const float* mkb = mk + (int64_t)b * CK * mpitch;
    const float* qkb = qk + (int64_t)b * CK * HW;
    const float* qeb = HAS_QE ? qe + (int64_t)b * CK * HW : nullptr;
    float a_sq[4][4], two_ab[4][4], b_sq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b_sq[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) a_sq[i][j] = two_ab[i][j] = 0.f;
    }
    for (int c0 = 0; c0 < CK; c0 += 16) {
        __syncthreads();
        for (int i = tid; i < 16 * 64; i += 256) {
            const int c = i >> 6, x = i & 63;
            const bool cok = c0 + c < CK;
            Ms[c][x] = (cok && n0 + x < N) ? mkb[(int64_t)(c0 + c) * mpitch + n0 + x] : 0.f;
            Qs[c][x] = (cok && q0 + x < HW) ? qkb[(int64_t)(c0 + c) * HW + q0 + x] : 0.f;
            if (HAS_QE) Es[c][x] = (cok && q0 + x < HW) ? qeb[(int64_t)(c0 + c) * HW + q0 + x] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float m[4], qv[4], ev[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { m[i] = Ms[c][tn * 4 + i]; qv[i] = Qs[c][tq * 4 + i]; ev[i] = HAS_QE ? Es[c][tq * 4 + i] : 1.f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float qe_qk = HAS_QE ? qv[j] * ev[j] : qv[j];
                if (HAS_QE) b_sq[j] += ev[j] * (qv[j] * qv[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (HAS_QE) a_sq[i][j] += (m[i] * m[i]) * ev[j];
                    else if (j == 0) a_sq[i][0] += m[i] * m[i];
                    two_ab[i][j] += m[i] * qe_qk;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + tn * 4 + i;
        if (n >= N) continue;
        const float sc = (ms ? ms[(int64_t)b * N + n] : 1.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + tq * 4 + j;
            if (q >= HW) continue;
            float s = HAS_QE ? (-a_sq[i][j] + 2.f * two_ab[i][j] - b_sq[j]) : (-a_sq[i][0] + 2.f * two_ab[i][j]);
            s = ms ? s * sc / sqrt_ck : s / sqrt_ck;                  // `similarity * ms / math.sqrt(CK)`
            if (TRANSPOSED) sim[((int64_t)b * HW + q) * N + n] = s;
            else sim[((int64_t)b * N + n) * HW + q] = s;
        }
    }
}

// ---- top-k over the memory axis + softmax of the kept values (memory_util.py:41-52): one thread per query, the similarity is
// read with consecutive threads on consecutive queries (coalesced rows); the running top-k set lives in LDS ([k][thread]) with
// its minimum cached in registers.  Output: idx [B][k][HW], w [B][k][HW] (w = exp(v) / sum exp(v), no max subtraction: the
// reference's top-k branch has none). ----
constexpr int TOPK_MAX = 64, TOPK_THREADS = 64, TOPK_MAXSPLIT = 128, TOPK_SLICE = 64;
// Two levels, so that the scan of the N memory elements is spread over the chip (one thread per query alone is HW / 64 = 26 blocks at
// 30 x 54 features): level 1 = block (64 queries, one slice of the memory axis) keeps the slice's top-k per query; level 2 = the same
// selection over the S x k survivors, then the softmax weights.  SRC_IDX: the values come with their memory indices (level 2).
template <bool SRC_IDX, bool FINAL>
__global__ void __launch_bounds__(TOPK_THREADS) mem_topk_kernel(const float* __restrict__ sim, const int* __restrict__ src_idx, int* __restrict__ idx,
                                                                float* __restrict__ wgt, int N, int HW, int K, int slice_len) {
    extern __shared__ float sh[];                                   // val [K][T], then idx [K][T]
    float* val = sh;
    int* ind = reinterpret_cast<int*>(sh + K * TOPK_THREADS);
    const int b = blockIdx.z, sl = blockIdx.y, S = gridDim.y, t = threadIdx.x, q = blockIdx.x * TOPK_THREADS + t;
    if (q >= HW) return;
    const int n0 = sl * slice_len, n1 = min(N, n0 + slice_len), cnt = max(n1 - n0, 0);
    const float* s = sim + (int64_t)b * N * HW + q;
    const int* si = SRC_IDX ? src_idx + (int64_t)b * N * HW + q : nullptr;
    const int first = cnt < K ? cnt : K;
    float vmin = INFINITY;
    int pmin = 0;
    for (int jb = 0; jb < first; jb += 8) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = jb + u < first ? s[(int64_t)(n0 + jb + u) * HW] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb + u;
            if (j >= first) break;
            const float v = v8[u];
            val[j * TOPK_THREADS + t] = v;
            ind[j * TOPK_THREADS + t] = SRC_IDX ? si[(int64_t)(n0 + j) * HW] : n0 + j;
            if (v < vmin) { vmin = v; pmin = j; }
        }
    }
    for (int j = first; j < K; ++j) { val[j * TOPK_THREADS + t] = -INFINITY; ind[j * TOPK_THREADS + t] = 0; }
    for (int nb = n0 + first; nb < n1; nb += 8) {                  // 8 similarity rows in flight (one dependent load per element was 238 us of a 2.2 ms frame)
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = nb + u < n1 ? s[(int64_t)(nb + u) * HW] : -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float v = v8[u];
            const int n = nb + u;
            if (n < n1 && v > vmin) {                               // replace the current minimum, then find the new one
                val[pmin * TOPK_THREADS + t] = v;
                ind[pmin * TOPK_THREADS + t] = SRC_IDX ? si[(int64_t)n * HW] : n;
                vmin = INFINITY;
                for (int j0 = 0; j0 < K; j0 += 8) {                 // 8 LDS reads in flight, then the compares in order (first minimum wins, as before)
                    float w8[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) w8[jj] = j0 + jj < K ? val[(j0 + jj) * TOPK_THREADS + t] : INFINITY;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj)
                        if (w8[jj] < vmin) { vmin = w8[jj]; pmin = j0 + jj; }
                }
            }
        }
    }
    if (!FINAL) {                                                   // survivors of this slice, query-major: [B][HW][S * K], in DESCENDING order
        // (the merge then only ever compares the heads of the slices).  K rounds of arg-max over the K kept values, 8 LDS reads in flight;
        // equal values leave in ascending memory index, i.e. as a single scan would rank them.
        for (int r = 0; r < K; ++r) {
            float best = -INFINITY;
            int bj = -1, bi = 0x7fffffff;
            for (int j0 = 0; j0 < K; j0 += 8) {
                float w8[8];
                int i8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    w8[jj] = j0 + jj < K ? val[(j0 + jj) * TOPK_THREADS + t] : -INFINITY;
                    i8[jj] = j0 + jj < K ? ind[(j0 + jj) * TOPK_THREADS + t] : 0x7fffffff;
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (w8[jj] > best || (w8[jj] == best && w8[jj] != -INFINITY && i8[jj] < bi)) { best = w8[jj]; bj = j0 + jj; bi = i8[jj]; }
            }
            const int64_t o = ((int64_t)b * HW + q) * ((int64_t)S * K) + (int64_t)sl * K + r;
            wgt[o] = best;                                          // -inf marks an empty slot
            idx[o] = bj >= 0 ? bi : 0;
            if (bj >= 0) val[bj * TOPK_THREADS + t] = -INFINITY;
        }
        return;
    }
    float sum = 0.f;
    for (int j = 0; j < first; ++j) {
        const float v = val[j * TOPK_THREADS + t];
        const float e = v == -INFINITY ? 0.f : expf(v);            // exp(v) / sum exp(v), no max subtraction (memory_util.py:44-47)
        val[j * TOPK_THREADS + t] = e;
        sum += e;
    }
    for (int j = 0; j < K; ++j) {
        const int64_t o = ((int64_t)b * K + j) * HW + q;
        idx[o] = ind[j * TOPK_THREADS + t];
        wgt[o] = j < first ? val[j * TOPK_THREADS + t] / sum : 0.f;
    }
}

// level 2: one wave per query merges the S slices, each sorted in descending order: lane l holds the heads of slices l and l + 64, a round is
// one wave-wide arg-max over the heads (ties go to the earlier slice, as in a single scan) and the winning slice advances; then the softmax
// weights.  (Round 2's merge re-scanned all S * K survivors in every round; with 64-element slices that would be 2 000 values x 30 rounds.)
__global__ void __launch_bounds__(64) mem_topk_merge_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int* __restrict__ idx,
                                                            float* __restrict__ wgt, int M, int HW, int K) {
    extern __shared__ float sh[];
    float* cv = sh;
    int* ci = reinterpret_cast<int*>(sh + M);
    __shared__ float sel_v[TOPK_MAX];
    __shared__ int sel_i[TOPK_MAX];
    const int q = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, S = M / K;
    const int64_t base = ((int64_t)b * HW + q) * M;
    for (int i = lane; i < M; i += 64) { cv[i] = cand_val[base + i]; ci[i] = cand_idx[base + i]; }
    __syncthreads();
    int h0 = 0, h1 = 0;                                             // heads of slices lane and lane + 64
    for (int k = 0; k < K; ++k) {
        const float v0 = (lane < S && h0 < K) ? cv[lane * K + h0] : -INFINITY;
        const float v1 = (lane + 64 < S && h1 < K) ? cv[(lane + 64) * K + h1] : -INFINITY;
        float best = v1 > v0 ? v1 : v0;
        int pos = v1 > v0 ? (lane + 64) * K + h1 : lane * K + h0;
        if (best == -INFINITY) pos = M;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int op = __shfl_xor(pos, o);
            if (ov > best || (ov == best && op < pos)) { best = ov; pos = op; }
        }
        if (pos < M) {
            const int sw = pos / K;
            if (sw == lane) ++h0;
            else if (sw == lane + 64) ++h1;
        }
        if (lane == 0) {
            sel_v[k] = best;
            sel_i[k] = pos < M ? ci[pos] : 0;
        }
    }
    __syncthreads();
    const float v = lane < K ? sel_v[lane] : -INFINITY;
    const float e = v == -INFINITY ? 0.f : expf(v);                // exp(v) / sum exp(v), no max subtraction (memory_util.py:44-47)
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < K) {
        const int64_t o = ((int64_t)b * K + lane) * HW + q;
        idx[o] = sel_i[lane];
        wgt[o] = sum > 0.f ? e / sum : 0.f;
    }
}

// ---- top-k, wave per query (round 3): the query's similarity row (query-major simT, N <= 64 NV elements) is loaded ONCE into registers as
// order-preserving integer keys; the k-th largest key is built bit by bit (32 rounds of "how many keys >= candidate": compare + add over the
// registers, one wave reduction per round); then the keys above it and, in ascending memory index, as many of the keys EQUAL to it as are still
// needed are compacted with ballots into the k slots, with their softmax weights exp(v) / sum (no max subtraction: memory_util.py:44-47).
// Exactly k outputs by construction, no candidate lists, no sort; ties resolved as a single ascending scan would.  (The two-level kernels above
// keep a running top-k per thread in LDS: 118 + 27 us per ColorMNet frame at 4 288 memory elements; this one: a tenth of that.)
__device__ __forceinline__ unsigned f2key(float v) {
    const unsigned b = __float_as_uint(v);
    return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
template <int NV>
__global__ void __launch_bounds__(256) mem_topk_select_kernel(const float* __restrict__ simT, int* __restrict__ idx, float* __restrict__ wgt, int N, int HW, int K) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (q >= HW) return;
    const float* row = simT + ((int64_t)b * HW + q) * N;
    unsigned key[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                                  // (clamped, unconditional loads: all of them in flight at once)
        const int n = i * 64 + lane;
        const float v = row[n < N ? n : N - 1];
        key[i] = n < N ? f2key(v) : 0u;                             // 0 is below the key of every float (incl. -inf): padding never wins
    }
    const int Keff = K < N ? K : N;
    unsigned thr = 0u;
    if (N > K) {
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) cnt += key[i] >= cand ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (cnt >= K) thr = cand;
        }
    }
    // how many are strictly above the threshold, and the softmax denominator over the selected set
    int ngt = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) ngt += (key[i] > thr && i * 64 + lane < N) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ngt += __shfl_xor(ngt, o);
    const int need_eq = Keff - ngt;                                 // ties at the threshold that still fit, lowest memory index first
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // softmax denominator over the selected set: the keys above the threshold + need_eq copies of the threshold value itself
    auto unkey = [](unsigned kb) { return __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb); };
    auto expv = [](float v) { return v == -INFINITY ? 0.f : expf(v); };
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (key[i] > thr && i * 64 + lane < N) part += expv(unkey(key[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const float e_thr = need_eq > 0 ? expv(unkey(thr)) : 0.f;       // (no ties wanted, e.g. fewer memory elements than k: thr = 0 is no float's key)
    part += (float)need_eq * e_thr;
    const float inv = part > 0.f ? 1.f / part : 0.f;
    int base_g = 0, base_e = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool live = i * 64 + lane < N;
        const bool g = live && key[i] > thr, e = live && key[i] == thr;
        const unsigned long long mg = __ballot(g), me = __ballot(e);
        int sl = -1;
        if (g) sl = base_g + __popcll(mg & lt);
        else if (e) {
            const int r = base_e + __popcll(me & lt);
            if (r < need_eq) sl = ngt + r;
        }
        base_g += __popcll(mg);
        base_e += __popcll(me);
        if (sl >= 0) {
            const int64_t o = ((int64_t)b * K + sl) * HW + q;
            idx[o] = i * 64 + lane;
            unsigned kk = key[i];
            asm volatile("" : "+v"(kk));                                // (recompute exp here: otherwise NV exponentials of the loop above are kept alive)
            wgt[o] = (g ? expv(unkey(kk)) : e_thr) * inv;
        }
    }
    for (int j = Keff + lane; j < K; j += 64) {                     // fewer memory elements than k: the remaining slots carry no weight
        const int64_t o = ((int64_t)b * K + j) * HW + q;
        idx[o] = 0;
        wgt[o] = 0.f;
    }
}

// The same selection for longer rows (8 192 < N <= 16 384: a clip whose long-term memory has filled up): one block of four waves per query, the
// keys in LDS instead of registers, block-wide counts per round; the compaction is done by wave 0 alone (N / 64 ballots).
__global__ void __launch_bounds__(256) mem_topk_select_lds_kernel(const float* __restrict__ simT, int* __restrict__ idx, float* __restrict__ wgt, int N, int HW, int K) {
    extern __shared__ unsigned keys[];                              // [N]
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x, b = blockIdx.y;
    const float* row = simT + ((int64_t)b * HW + q) * N;
    for (int n = tid; n < N; n += 256) keys[n] = f2key(row[n]);
    __syncthreads();
    const int Keff = K < N ? K : N;
    unsigned thr = 0u;
    if (N > K) {
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int cnt = 0;
            for (int n = tid; n < N; n += 256) cnt += keys[n] >= cand ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0) wcnt[wave] = cnt;
            __syncthreads();
            const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
            if (tot >= K) thr = cand;
        }
    }
    if (wave != 0) return;
    auto unkey = [](unsigned kb) { return __uint_as_float((kb & 0x80000000u) ? (kb ^ 0x80000000u) : ~kb); };
    auto expv = [](float v) { return v == -INFINITY ? 0.f : expf(v); };
    int ngt = 0;
    float part = 0.f;
    for (int n = lane; n < N; n += 64) {
        const unsigned kb = keys[n];
        if (kb > thr) { ++ngt; part += expv(unkey(kb)); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ngt += __shfl_xor(ngt, o); part += __shfl_xor(part, o); }
    const int need_eq = Keff - ngt;
    const float e_thr = need_eq > 0 ? expv(unkey(thr)) : 0.f;
    part += (float)need_eq * e_thr;
    const float inv = part > 0.f ? 1.f / part : 0.f;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int base_g = 0, base_e = 0;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + lane;
        const unsigned kb = n < N ? keys[n] : 0u;
        const bool g = n < N && kb > thr, e = n < N && kb == thr;
        const unsigned long long mg = __ballot(g), me = __ballot(e);
        int sl = -1;
        if (g) sl = base_g + __popcll(mg & lt);
        else if (e) {
            const int r = base_e + __popcll(me & lt);
            if (r < need_eq) sl = ngt + r;
        }
        base_g += __popcll(mg);
        base_e += __popcll(me);
        if (sl >= 0) {
            const int64_t o = ((int64_t)b * K + sl) * HW + q;
            idx[o] = n;
            wgt[o] = (g ? expv(unkey(kb)) : e_thr) * inv;
        }
    }
    for (int j = Keff + lane; j < K; j += 64) {
        const int64_t o = ((int64_t)b * K + j) * HW + q;
        idx[o] = 0;
        wgt[o] = 0.f;
    }
}

// ---- readout: out[cv][q] = sum_j w[j][q] mv[cv][idx[j][q]]   (memory_manager._readout on the sparse affinity) ----
__global__ void __launch_bounds__(256) mem_readout_kernel(const float* __restrict__ mv, const int* __restrict__ idx, const float* __restrict__ wgt,
                                                          float* __restrict__ out, int CV, int64_t N, int HW, int K) {      // N: row pitch of mv
    __shared__ int si[TOPK_MAX][64];
    __shared__ float sw[TOPK_MAX][64];
    const int b = blockIdx.z, q0 = blockIdx.x * 64, cv0 = blockIdx.y * 64;
    const int tid = threadIdx.x, tq = tid & 63, tc = tid >> 6;
    for (int i = tid; i < K * 64; i += 256) {
        const int j = i >> 6, x = i & 63;
        const bool ok = q0 + x < HW;
        si[j][x] = ok ? idx[((int64_t)b * K + j) * HW + q0 + x] : 0;
        sw[j][x] = ok ? wgt[((int64_t)b * K + j) * HW + q0 + x] : 0.f;
    }
    __syncthreads();
    if (q0 + tq >= HW) return;
    for (int c = tc; c < 64 && cv0 + c < CV; c += 4) {
        const float* row = mv + ((int64_t)b * CV + cv0 + c) * N;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) acc += sw[j][tq] * row[si[j][tq]];
        out[((int64_t)b * CV + cv0 + c) * HW + q0 + tq] = acc;
    }
}

// ---- usage of every memory element: the row sums of the sparse affinity (do_softmax(..., return_usage=True), memory_util.py:63-64;
// MemoryManager.match_memory feeds them to KeyValueMemoryStore.update_usage).  Accumulated as 2^-40 fixed point in 64-bit integers:
// integer adds commute, so the sums do not depend on the order the atomics land in (fp32 atomics would make the choice of the
// long-term prototypes, a top-k over these sums, run-to-run dependent). ----
__global__ void mem_usage_accum_kernel(const int* __restrict__ idx, const float* __restrict__ wgt, unsigned long long* __restrict__ acc, int N, int HW,
                                       int K) {
    const int b = blockIdx.y;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * HW) return;
    const float w = wgt[(int64_t)b * K * HW + i];
    if (w > 0.f) atomicAdd(acc + (int64_t)b * N + idx[(int64_t)b * K * HW + i], (unsigned long long)((double)w * 1099511627776.0 + 0.5));
}
__global__ void mem_usage_final_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ usage, int64_t total) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < total) usage[i] = (float)((double)acc[i] * (1.0 / 1099511627776.0));
}
// KeyValueMemoryStore.update_usage in place (kv_memory_store.py:93-101): use_count += usage, life_count += 1, for the elements [from, N)
// ... and the accumulators go back to zero for the next read (round 5: the launcher no longer clears them with a memset in front of every read; the
// caller guarantees zeros before the FIRST use of a buffer, havc_runtime.cpp usage_update_locked)
__global__ void mem_usage_update_kernel(unsigned long long* __restrict__ acc, float* __restrict__ use, float* __restrict__ life, int from, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned long long a = acc[i];
    acc[i] = 0ull;
    if (i >= from) { use[i] = use[i] + (float)((double)a * (1.0 / 1099511627776.0)); life[i] = life[i] + 1.f; }
}
// encode_value's input (network.py:87-101 as colormnet_net.encode_value assembles it): per object i the image (3 planes), its own ab plane and the
// other object's: vin[i] = [img0, img1, img2, m_i, m_(1-i)]
__global__ void cmn_value_in_kernel(const float* __restrict__ img, const float* __restrict__ m, float* __restrict__ vin, int64_t P) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = img[i], b = img[P + i], c = img[2 * P + i], m0 = m[i], m1 = m[P + i];
        vin[i] = a; vin[P + i] = b; vin[2 * P + i] = c; vin[3 * P + i] = m0; vin[4 * P + i] = m1;
        vin[5 * P + i] = a; vin[6 * P + i] = b; vin[7 * P + i] = c; vin[8 * P + i] = m1; vin[9 * P + i] = m0;
    }
}
__global__ void vec_add_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = y[i] + x[i];
}
}  // namespace
// acc [N] must be all zero on entry and is all zero again on exit
int launch_mem_usage_update(const int* idx, const float* wgt, unsigned long long* acc, float* use, float* life, int from, int N, int HW, int K, hipStream_t s) {
    hipLaunchKernelGGL(mem_usage_accum_kernel, dim3(cdiv((int64_t)K * HW, 256), 1), dim3(256), 0, s, idx, wgt, acc, N, HW, K);
    hipLaunchKernelGGL(mem_usage_update_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, acc, use, life, from, N);
    return (int)hipGetLastError();
}
int launch_cmn_value_in(const float* img, const float* planes, float* vin, int64_t P, hipStream_t s) {
    hipLaunchKernelGGL(cmn_value_in_kernel, dim3(cdiv(P, 256) > 2048 ? 2048 : cdiv(P, 256)), dim3(256), 0, s, img, planes, vin, P);
    return (int)hipGetLastError();
}
int launch_vec_add(float* y, const float* x, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256)), dim3(256), 0, s, y, x, n);
    return (int)hipGetLastError();
}
int launch_mem_usage(const int* idx, const float* wgt, unsigned long long* acc, float* usage, int B, int N, int HW, int K, hipStream_t s) {
    hipError_t e = hipMemsetAsync(acc, 0, (size_t)B * N * 8, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mem_usage_accum_kernel, dim3(cdiv((int64_t)K * HW, 256), B), dim3(256), 0, s, idx, wgt, acc, N, HW, K);
    hipLaunchKernelGGL(mem_usage_final_kernel, dim3(cdiv((int64_t)B * N, 256)), dim3(256), 0, s, acc, usage, (int64_t)B * N);
    return (int)hipGetLastError();
}
namespace {

// ---- dense softmax over the memory axis + readout (memory consolidation, memory_manager.py:264-283: do_softmax(similarity) without
// top-k, then v @ affinity).  sim [B][N][P] (P prototypes), mv [B][CV][N] -> out [B][CV][P].  One block per (prototype, batch):
// column max and sum of exp by a block reduction, then every thread owns value channels. ----
__global__ void __launch_bounds__(256) mem_dense_readout_kernel(const float* __restrict__ sim, const float* __restrict__ mv, float* __restrict__ out,
                                                                int CV, int N, int P) {
    __shared__ float red[256];
    __shared__ float wbuf[1024];
    const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* col = sim + (int64_t)b * N * P + p;
    float m = -INFINITY;
    for (int n = tid; n < N; n += 256) m = fmaxf(m, col[(int64_t)n * P]);
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int n = tid; n < N; n += 256) sum += expf(col[(int64_t)n * P] - m);
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    __syncthreads();
    // out[cv] = sum_n mv[cv][n] * exp(sim[n] - m) / sum: weights of 1024 memory elements at a time through LDS
    float acc[8];                                                   // CV <= 2048: thread owns channels tid, tid + 256, ...
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int n0 = 0; n0 < N; n0 += 1024) {
        const int nn = N - n0 < 1024 ? N - n0 : 1024;
        for (int i = tid; i < nn; i += 256) wbuf[i] = expf(col[(int64_t)(n0 + i) * P] - m) * inv;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cv = tid + k * 256;
            if (cv < CV) {
                const float* row = mv + ((int64_t)b * CV + cv) * N + n0;
                float a = acc[k];
                for (int i = 0; i < nn; ++i) a = fmaf(row[i], wbuf[i], a);
                acc[k] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cv = tid + k * 256;
        if (cv < CV) out[((int64_t)b * CV + cv) * P + p] = acc[k];
    }
}
}  // namespace
int launch_mem_dense_readout(const float* sim, const float* mv, float* out, int B, int CV, int N, int P, hipStream_t s) {
    if (CV > 2048) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mem_dense_readout_kernel, dim3(P, B), dim3(256), 0, s, sim, mv, out, CV, N, P);
    return (int)hipGetLastError();
}
namespace {

// ---- local correlation (attention.py:827-835): out[n][(dy+R)*ws+(dx+R)][y*w+x] = sum_c q[n][c][y][x] k[n][c][y+dy*dil][x+dx*dil] ----
// Block = (8 x 8 query pixels, ONE window row dy): thread = pixel p (tid & 63) x a group of 4 window columns (tid >> 6), channels summed in
// ascending order.  The K rows the window row touches (8 rows x (8 + 2 R dil) columns) and the Q tile are staged in LDS LC_CC channels at a
// time.  Round 3: the window rows are spread over the grid -- a 28 x 14 feature map is 8 tiles, and 8 blocks of 4 waves walking all 225
// offsets took 155 us of a 2.2 ms ColorMNet frame; 8 x 15 blocks take a tenth of that.
constexpr int LC_T = 8, LC_CC = 64, LC_MAXWS = 16;     // window up to 16 x 16 columns (max_dis 7: 15): 4 column groups x 4 accumulators per thread
__global__ void __launch_bounds__(256) local_corr_kernel(const float* __restrict__ q, const float* __restrict__ k, float* __restrict__ out, int C, int H,
                                                         int W, int R, int dil, float qscale) {
    extern __shared__ float lds[];                                  // k rows [LC_CC][8][HT + 1], q tile [LC_CC][64]
    const int ws = 2 * R + 1, HT = LC_T + 2 * R * dil, HP = HT + 1;
    float* kt = lds;
    float* qt = lds + LC_CC * LC_T * HP;
    const int n = blockIdx.z / ws, dyi = blockIdx.z % ws, ty0 = blockIdx.y * LC_T, tx0 = blockIdx.x * LC_T;
    const int tid = threadIdx.x, p = tid & 63, py = p >> 3, px = p & 7, grp = tid >> 6;
    const int y = ty0 + py, x = tx0 + px;
    const float* qb = q + (int64_t)n * C * H * W;
    const float* kb = k + (int64_t)n * C * H * W;
    float* ob = out + (int64_t)n * ws * ws * H * W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int row0 = ty0 + (dyi - R) * dil;                         // image row of the K row that pixel row 0 of the tile reads
    for (int c0 = 0; c0 < C; c0 += LC_CC) {
        __syncthreads();
        for (int i0 = tid; i0 < LC_CC * LC_T * HT; i0 += 256 * 8) {                      // 8 loads in flight per thread
            float v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int c = i / (LC_T * HT), r = i - c * (LC_T * HT), hy = r / HT, hx = r - hy * HT;
                const int iy = row0 + hy, ix = tx0 - R * dil + hx;
                dst[u] = i < LC_CC * LC_T * HT ? (c * LC_T + hy) * HP + hx : -1;
                v[u] = (dst[u] >= 0 && c0 + c < C && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? kb[((int64_t)(c0 + c) * H + iy) * W + ix] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (dst[u] >= 0) kt[dst[u]] = v[u];
        }
        for (int i0 = tid; i0 < LC_CC * 64; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int c = i >> 6, pp = i & 63, yy = ty0 + (pp >> 3), xx = tx0 + (pp & 7);
                v[u] = (i < LC_CC * 64 && c0 + c < C && yy < H && xx < W) ? qb[((int64_t)(c0 + c) * H + yy) * W + xx] * qscale : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 256 < LC_CC * 64) qt[i0 + u * 256] = v[u];
        }
        __syncthreads();
        const float* krow = kt + py * HP + px + grp * 4 * dil;
#pragma unroll 4
        for (int c = 0; c < LC_CC; ++c) {
            const float qv = qt[c * 64 + p];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (grp * 4 + j < ws) acc[j] += qv * krow[c * LC_T * HP + j * dil];
        }
    }
    if (y < H && x < W) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (grp * 4 + j < ws) ob[((int64_t)(dyi * ws + grp * 4 + j) * H + y) * W + x] = acc[j];
    }
}

// ---- logits = corr + relative_emb(q) - 1e8 [window position outside the image]; softmax over the window (attention.py:806-846) ----
// in place on the correlation buffer [n][ws*ws][h*w].  Block = 16 pixels x 16 groups of window positions (thread (p, g) owns positions
// g, g + 16, ...: 15 of the 225): the 1x1 relative-embedding conv (225 x C MACs per pixel) and the softmax are spread over 16 threads
// per pixel, the pixel's q vector is shared through LDS, maxima / sums are combined across the groups in a fixed order.
constexpr int LS_PIX = 16, LS_GRP = 16, LS_MAXPOS = 15, LS_MAXC = 256;
__global__ void __launch_bounds__(LS_PIX* LS_GRP) local_softmax_kernel(float* __restrict__ qk, const float* __restrict__ q, const float* __restrict__ rel_w,
                                                                       const float* __restrict__ rel_b, int C, int H, int W, int R, int dil, int n_total) {
    __shared__ float qs[LS_MAXC][LS_PIX];
    __shared__ float red[LS_GRP][LS_PIX];
    const int ws = 2 * R + 1, WW = ws * ws, HWp = H * W;
    const int tid = threadIdx.x, pl = tid & (LS_PIX - 1), g = tid / LS_PIX;
    const int64_t i = blockIdx.x * (int64_t)LS_PIX + pl;
    const bool live = i < (int64_t)n_total * HWp;
    const int n = live ? (int)(i / HWp) : 0, p = live ? (int)(i - (int64_t)n * HWp) : 0, y = p / W, x = p - y * W;
    for (int c = g; c < C; c += LS_GRP) qs[c][pl] = live ? q[((int64_t)n * C + c) * HWp + p] : 0.f;
    __syncthreads();
    float* col = qk + (int64_t)n * WW * HWp + p;
    float v[LS_MAXPOS];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LS_MAXPOS; ++j) {
        const int d = g + j * LS_GRP;
        v[j] = -INFINITY;
        if (d < WW && live) {
            float r = rel_b[d];
            const float* wr = rel_w + (int64_t)d * C;
            for (int c = 0; c < C; ++c) r += wr[c] * qs[c][pl];
            const int yy = y + (d / ws - R) * dil, xx = x + (d % ws - R) * dil;
            float t = col[(int64_t)d * HWp] + r;
            if (!((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)) t -= 1e8f;
            v[j] = t;
            mx = fmaxf(mx, t);
        }
    }
    red[g][pl] = mx;
    __syncthreads();
    mx = red[0][pl];
#pragma unroll
    for (int k = 1; k < LS_GRP; ++k) mx = fmaxf(mx, red[k][pl]);
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LS_MAXPOS; ++j) {
        const float e = v[j] == -INFINITY ? 0.f : expf(v[j] - mx);
        v[j] = e;
        sum += e;
    }
    red[g][pl] = sum;
    __syncthreads();
    sum = red[0][pl];
#pragma unroll
    for (int k = 1; k < LS_GRP; ++k) sum += red[k][pl];
    const float inv = 1.f / sum;
    if (live) {
#pragma unroll
        for (int j = 0; j < LS_MAXPOS; ++j) {
            const int d = g + j * LS_GRP;
            if (d < WW) col[(int64_t)d * HWp] = v[j] * inv;
        }
    }
}

// ---- agg[p][n][cv] = sum_d attn[n][d][p] v[n][cv][p + d]  (local2global + matmul of attention.py:850-853, without the dense map) ----
// Block = 8 x 8 pixels x 32 value channels; the V halo tile and the tile's attention weights are staged in LDS.
constexpr int LA_CC = 32;
__global__ void __launch_bounds__(256) local_agg_kernel(const float* __restrict__ attn, const float* __restrict__ v, float* __restrict__ agg, int CV, int H,
                                                        int W, int R, int dil, int n_total) {
    extern __shared__ float lds[];                                  // v tile [LA_CC][HT][HT + 1], attention [ws*ws][64]
    const int ws = 2 * R + 1, WW = ws * ws, HT = LC_T + 2 * R * dil, HP = HT + 1, HWp = H * W;
    float* vt = lds;
    float* at = lds + LA_CC * HT * HP;
    const int tiles_x = (W + LC_T - 1) / LC_T;
    const int n = blockIdx.z, ty0 = (blockIdx.x / tiles_x) * LC_T, tx0 = (blockIdx.x % tiles_x) * LC_T, c0 = blockIdx.y * LA_CC;
    const int tid = threadIdx.x, p = tid & 63, py = p >> 3, px = p & 7, cg = tid >> 6;
    const float* vb = v + (int64_t)n * CV * HWp;
    // staging with 8 loads in flight per thread (see local_corr_kernel)
    for (int i0 = tid; i0 < LA_CC * HT * HT; i0 += 256 * 8) {
        float vv[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int c = i / (HT * HT), r = i - c * HT * HT, hy = r / HT, hx = r - hy * HT;
            const int iy = ty0 - R * dil + hy, ix = tx0 - R * dil + hx;
            dst[u] = i < LA_CC * HT * HT ? (c * HT + hy) * HP + hx : -1;
            vv[u] = (dst[u] >= 0 && c0 + c < CV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? vb[((int64_t)(c0 + c) * H + iy) * W + ix] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) vt[dst[u]] = vv[u];
    }
    for (int i0 = tid; i0 < WW * 64; i0 += 256 * 8) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int d = i >> 6, pp = i & 63, yy = ty0 + (pp >> 3), xx = tx0 + (pp & 7);
            vv[u] = (i < WW * 64 && yy < H && xx < W) ? attn[((int64_t)n * WW + d) * HWp + yy * W + xx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 256 < WW * 64) at[i0 + u * 256] = vv[u];
    }
    __syncthreads();
    const int y = ty0 + py, x = tx0 + px;
    float acc[LA_CC / 4];
#pragma unroll
    for (int e = 0; e < LA_CC / 4; ++e) acc[e] = 0.f;
    for (int d = 0; d < WW; ++d) {
        const float a = at[d * 64 + p];
        const int off = (py + (d / ws) * dil) * HP + px + (d % ws) * dil;
#pragma unroll
        for (int e = 0; e < LA_CC / 4; ++e) acc[e] += a * vt[(cg * (LA_CC / 4) + e) * HT * HP + off];
    }
    if (y < H && x < W) {
#pragma unroll
        for (int e = 0; e < LA_CC / 4; ++e) {
            const int c = c0 + cg * (LA_CC / 4) + e;
            if (c < CV) agg[((int64_t)(y * W + x) * n_total + n) * CV + c] = acc[e];
        }
    }
}

}  // namespace

int launch_mem_similarity(const float* mk, const float* ms, const float* qk, const float* qe, float* sim, int B, int CK, int N, int HW, hipStream_t s,
                          int64_t mpitch) {
    dim3 grid(cdiv(HW, 64), cdiv(N, 64), B);
    const float sq = sqrtf((float)CK);
    if (mpitch <= 0) mpitch = N;
    if (qe) hipLaunchKernelGGL(mem_similarity_kernel<true>, grid, dim3(256), 0, s, mk, ms, qk, qe, sim, CK, N, HW, sq, mpitch);
    else hipLaunchKernelGGL(mem_similarity_kernel<false>, grid, dim3(256), 0, s, mk, ms, qk, qe, sim, CK, N, HW, sq, mpitch);
    return (int)hipGetLastError();
}

int launch_mem_similarity_t(const float* mk, const float* ms, const float* qk, const float* qe, float* simT, int B, int CK, int N, int HW, hipStream_t s,
                            int64_t mpitch) {
    dim3 grid(cdiv(HW, 64), cdiv(N, 64), B);
    const float sq = sqrtf((float)CK);
    if (mpitch <= 0) mpitch = N;
    if (qe) hipLaunchKernelGGL((mem_similarity_kernel<true, true>), grid, dim3(256), 0, s, mk, ms, qk, qe, simT, CK, N, HW, sq, mpitch);
    else hipLaunchKernelGGL((mem_similarity_kernel<false, true>), grid, dim3(256), 0, s, mk, ms, qk, qe, simT, CK, N, HW, sq, mpitch);
    return (int)hipGetLastError();
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel and DEVICE: done once per device, eagerly from havc_create (preload_colormnet) and -- for a
// caller that reaches a launcher first -- lazily here; never on every launch.
static void colormnet_lds_optin() {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mem_topk_select_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(local_agg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // the two-level merge holds S * K candidates (value + index): 128 slices x K = 64 is exactly 64 KiB of dynamic LDS on top of 512 static bytes
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mem_topk_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipGetLastError();
    done.fetch_or(bit, std::memory_order_release);
}
void preload_colormnet() { colormnet_lds_optin(); }

// the wave-per-query selection holds the row in registers up to 64 x 128 = 8 192 memory elements, in LDS up to 16 384 (beyond: the two-level kernels)
bool mem_topk_select_supported(int N) { return N <= 16384; }

int launch_mem_topk_select_readout(const float* simT, const float* mv, int* idx, float* wgt, float* out, int B, int CV, int N, int HW, int K, hipStream_t s,
                                   int64_t vpitch) {
    if (K < 1 || K > TOPK_MAX || !mem_topk_select_supported(N)) return (int)hipErrorInvalidValue;
    dim3 grid(cdiv(HW, 4), B);
    const int nv = cdiv(N, 64);
    if (nv > 128) {
        colormnet_lds_optin();
        hipLaunchKernelGGL(mem_topk_select_lds_kernel, dim3(HW, B), dim3(256), (size_t)N * 4, s, simT, idx, wgt, N, HW, K);
    } else if (nv <= 32) hipLaunchKernelGGL(mem_topk_select_kernel<32>, grid, dim3(256), 0, s, simT, idx, wgt, N, HW, K);
    else if (nv <= 64) hipLaunchKernelGGL(mem_topk_select_kernel<64>, grid, dim3(256), 0, s, simT, idx, wgt, N, HW, K);
    else if (nv <= 96) hipLaunchKernelGGL(mem_topk_select_kernel<96>, grid, dim3(256), 0, s, simT, idx, wgt, N, HW, K);
    else hipLaunchKernelGGL(mem_topk_select_kernel<128>, grid, dim3(256), 0, s, simT, idx, wgt, N, HW, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mem_readout_kernel, dim3(cdiv(HW, 64), cdiv(CV, 64), B), dim3(256), 0, s, mv, idx, wgt, out, CV, vpitch > 0 ? vpitch : (int64_t)N, HW, K);
    return (int)hipGetLastError();
}

int mem_topk_splits(int N) {                                  // slices of the memory axis at level 1 (1 = single level)
    const int S = (N + TOPK_SLICE - 1) / TOPK_SLICE;          // 64 elements per slice while that gives <= 128 slices (the merge holds two heads per lane)
    return S < 2 ? 1 : (S > TOPK_MAXSPLIT ? TOPK_MAXSPLIT : S);
}
// cand_val / cand_idx: workspace of B * S * K * HW floats / ints each (S = mem_topk_splits(N)); unused when S == 1
int launch_mem_topk_readout(const float* sim, const float* mv, int* idx, float* wgt, float* cand_val, int* cand_idx, float* out, int B, int CV, int N,
                            int HW, int K, hipStream_t s, int64_t vpitch) {
    if (K < 1 || K > TOPK_MAX) return (int)hipErrorInvalidValue;
    const int S = mem_topk_splits(N);
    const size_t lds = (size_t)K * TOPK_THREADS * 8;
    if (S == 1) {
        hipLaunchKernelGGL((mem_topk_kernel<false, true>), dim3(cdiv(HW, TOPK_THREADS), 1, B), dim3(TOPK_THREADS), lds, s, sim, nullptr, idx, wgt, N, HW, K, N);
    } else {
        const int len = (N + S - 1) / S;
        hipLaunchKernelGGL((mem_topk_kernel<false, false>), dim3(cdiv(HW, TOPK_THREADS), S, B), dim3(TOPK_THREADS), lds, s, sim, nullptr, cand_idx, cand_val, N,
                           HW, K, len);
        colormnet_lds_optin();
        hipLaunchKernelGGL(mem_topk_merge_kernel, dim3(HW, B), dim3(64), (size_t)S * K * 8, s, cand_val, cand_idx, idx, wgt, S * K, HW, K);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mem_readout_kernel, dim3(cdiv(HW, 64), cdiv(CV, 64), B), dim3(256), 0, s, mv, idx, wgt, out, CV, vpitch > 0 ? vpitch : (int64_t)N, HW, K);
    return (int)hipGetLastError();
}

int launch_local_correlation(const float* q, const float* k, float* out, int n, int C, int H, int W, int R, int dil, float qscale, hipStream_t s) {
    const int ws = 2 * R + 1, HT = LC_T + 2 * R * dil;
    if (ws > LC_MAXWS || R < 0 || dil < 1) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(LC_CC * LC_T * (HT + 1) + LC_CC * 64) * sizeof(float);
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(local_corr_kernel, dim3(cdiv(W, LC_T), cdiv(H, LC_T), n * ws), dim3(256), lds, s, q, k, out, C, H, W, R, dil, qscale);
    return (int)hipGetLastError();
}

int launch_local_softmax(float* qk, const float* q, const float* rel_w, const float* rel_b, int n, int C, int H, int W, int R, int dil, hipStream_t s) {
    const int ws = 2 * R + 1;
    if (C > LS_MAXC || ws * ws > LS_GRP * LS_MAXPOS) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(local_softmax_kernel, dim3(cdiv((int64_t)n * H * W, LS_PIX)), dim3(LS_PIX * LS_GRP), 0, s, qk, q, rel_w, rel_b, C, H, W, R, dil, n);
    return (int)hipGetLastError();
}

int launch_local_agg(const float* attn, const float* v, float* agg, int n, int CV, int H, int W, int R, int dil, hipStream_t s) {
    const int ws = 2 * R + 1, HT = LC_T + 2 * R * dil;
    const size_t lds = (size_t)(LA_CC * HT * (HT + 1) + ws * ws * 64) * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    colormnet_lds_optin();
    hipLaunchKernelGGL(local_agg_kernel, dim3(cdiv(W, LC_T) * cdiv(H, LC_T), cdiv(CV, LA_CC), n), dim3(256), lds, s, attn, v, agg, CV, H, W, R, dil, n);
    return (int)hipGetLastError();
}
