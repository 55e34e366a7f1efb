// libhavc_mi355.so runtime: contexts, packed weights, plan executor, frame / clip entry points.
// Implements include/havc_mi355.h.  No torch, no Python: plain HIP runtime + the kernels in this directory.
#include "../../include/havc_mi355.h"
#include "kernels.h"
#include "build_stamp.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <string>
#include <vector>

// ---- set-up is serialised process-wide -------------------------------------------------------------------------------------------------
// The reference's glue builds its models from whichever VapourSynth worker thread asks first (vsslib/vsmodels.py:196-233), so contexts,
// weights and nets of SEVERAL models can be created -- and autotuned -- from several host threads at once.  Everything that is set-up rather
// than steady-state work takes this one mutex (after the context's own): context creation (streams, eager code-object load, big-LDS opt-ins,
// the per-queue scratch warm-up below), weight upload, net creation / destruction (hipMalloc / hipFree of the activation arenas), the
// autotuner's trial launches, scratch regrowth and the device / pinned allocators.  Steady-state launches never take it.  History: before
// round 4 concurrent set-up of two models from two threads preceded a "HW Exception: GPU Hang" (1 in ~15 runs).
static std::recursive_mutex g_setup_mu;      // recursive: the autotuner's trial launches may regrow the split-K workspace (ensure_scratch)
// HAVC_SETUP_MUTEX=0 turns the lock into a no-op (tools/setup_stress.py bisects with it; never a production setting)
struct SetupLock {
    bool on;
    SetupLock() {
        static const bool enabled = [] { const char* e = getenv("HAVC_SETUP_MUTEX"); return e ? atoi(e) != 0 : true; }();
        on = enabled;
        if (on) g_setup_mu.lock();
    }
    ~SetupLock() { if (on) g_setup_mu.unlock(); }
    SetupLock(const SetupLock&) = delete;
    SetupLock& operator=(const SetupLock&) = delete;
};

// Kernels that spill use per-queue scratch memory which the HIP runtime sizes on the FIRST launch that needs it (and re-sizes when a later
// kernel needs more).  The largest private segment of any kernel of this library is 624 bytes per lane (conv_igemm_kernel<128,304>): this
// kernel declares more than that and does nothing, so launching it once on each stream of a new context -- under the set-up mutex -- gives
// every hardware queue its final scratch size before any product launch.
__global__ void scratch_warm_kernel(int* out, int n) {
    volatile int a[192];
    for (int i = 0; i < n; ++i) a[i % 192] = i;
    if (n == 12345) out[0] = a[n % 192];
}

// HAVC_STREAM_JITTER=<seed> (race hunting, tools/cmn_race_stress.py): a kernel that does nothing for `ticks` of the 100 MHz wall clock, launched with a
// pseudo-random length in front of the work of the multi-stream entry points (plan slices, the ColorMNet read / short-term attention / side sections,
// device copies).  It moves the streams of a context -- and the look-ahead context -- against each other by up to a few hundred microseconds per call, so
// that an ordering that only holds by timing shows up as different bytes.  Results must not depend on it.  Off (one predictable branch) unless the variable is set.
__global__ void jitter_delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

namespace {

thread_local std::string g_create_error;

struct JitterState {
    bool on = false;
    unsigned max_us = 300;
    std::atomic<uint64_t> rng{0x9E3779B97F4A7C15ull};
    JitterState() {
        const char* e = getenv("HAVC_STREAM_JITTER");
        if (e && *e && atoll(e) != 0) {
            on = true;
            rng = 0x9E3779B97F4A7C15ull * (uint64_t)(atoll(e) + 1);
            if (const char* m = getenv("HAVC_STREAM_JITTER_US")) max_us = (unsigned)std::max(1, atoi(m));
        }
    }
};
JitterState g_jitter;

// one call = at most one delay kernel on `st` (two calls in three launch nothing: the un-delayed interleavings stay in the mix)
inline void stream_jitter(hipStream_t st) {
    if (!g_jitter.on) return;
    uint64_t x = g_jitter.rng.load(std::memory_order_relaxed);
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    g_jitter.rng.store(x, std::memory_order_relaxed);
    if (x % 3) return;
    const unsigned us = 1 + (unsigned)((x >> 20) % g_jitter.max_us);
    hipLaunchKernelGGL(jitter_delay_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)us * 100ull);
    (void)hipGetLastError();
}

struct ResizeTable {
    int taps = 0;
    int* d_start = nullptr;
    float* d_w = nullptr;
};

}  // namespace

struct havc_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // the second generator of a stable/artistic render runs here, concurrently
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_main_done = nullptr, ev_side = nullptr, ev_mark = nullptr;
    bool marked = false;                  // havc_cmn_side_mark recorded ev_mark: the next side section starts behind THAT point of the main stream
    bool stream_exported = false;         // havc_get_stream handed a stream handle out: the streams may not be re-created any more (havc_ctx_set_stream_*)
    bool side = false;                    // between havc_cmn_side_begin / _end: the ColorMNet read (short-term attention, memory read, join) is enqueued on stream2
    size_t acc_clean_sz = 0;              // scratch 17 (usage accumulators of the banked read) holds zeros over this many bytes (0: unknown -> cleared before use)
    struct { float* use = nullptr; float* life = nullptr; int from = 0, N = 0, HW = 0, top_k = 0; } side_usage;   // its usage update, owed until havc_cmn_side_wait
    hipStream_t cur = nullptr;            // stream the plan executor launches on (stream or stream2)
    bool two_streams = true;              // HAVC_TWO_STREAMS=0 serialises the two generators (A/B measurements)
    bool range_check = false;             // HAVC_RANGE_CHECK / havc_range_check_enable: scan every op's destination for inf / NaN / abs-max
    uint64_t nt_store_bytes = 0;          // conv outputs at least this large are written with non-temporal stores (0 = never); HAVC_NT_STORE_MB
    uint64_t desc_limit = 0xE0000000ull;  // bytes one conv launch may address per operand (32-bit buffer descriptors);
                                          // HAVC_DESC_LIMIT_BYTES lowers it so tests reach the frame-chunking path at small sizes
    std::mutex mu;
    std::string err;
    havc_stats stats{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // grow-only scratch (u8 staging + float resample rows): allocated once, reused every call
    void* scratch[18] = {nullptr};         // 0-3 staging / model i-o, 4-5 plane staging, 6 small, 7 resample rows, 8-11 pipelined host clip (and the one-shot
    size_t scratch_sz[18] = {0};           // memory reads), 12 / 13 split-K partial sums of the launches on stream / stream2, 14-17 the BANKED memory read of the
                                           // ColorMNet frame loop (similarity map, top-k indices / weights, usage accumulators): its own slots, because a read that
                                           // runs ahead leaves its top-k lists there until its frame is stepped -- no other entry point may touch them meanwhile
    hipStream_t stream_h2d = nullptr, stream_d2h = nullptr;      // copy streams of havc_colorize_clip_host (created on first use)
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
    std::map<std::pair<int, int>, ResizeTable> resize_tables;
    // per-tag timing
    int timed_tag = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tag_events;
    size_t tag_used = 0;
    double tag_ms = 0;
    int64_t tag_launches = 0;
};

struct havc_weights {
    havc_ctx* ctx;
    uint8_t* d_blob;
    size_t nbytes;
};

struct havc_net {
    havc_ctx* ctx;
    havc_weights* w;
    std::vector<havc_op> ops;
    std::vector<havc_buf> bufdesc;
    std::vector<void*> bufs;
    int in_buf, out_buf, S, max_batch;
    int tail_first = -1;                  // index of the first op tagged 1 (exclusive tail), -1 if none
    int2* d_ktab = nullptr;               // all conv K tables, one allocation
    std::vector<int64_t> ktab_off;        // per op: element offset into d_ktab, -1 for non-conv ops
    const void* in_override = nullptr;
    void* out_override = nullptr;
    unsigned* d_range = nullptr;          // range check: per op {abs-max bits, -, non-finite count (64 bit)}
    bool range_ready = false;             // buffers were zero-filled at creation (never-written padding cannot trip the scan)
    std::vector<float> range_absmax;
    std::vector<int64_t> range_bad;
    std::vector<void*> bound;             // havc_net_bind: caller-owned device memory standing in for a buffer (nullptr = own allocation)
    double flops_per_frame = 0;
};

namespace {

int fail(havc_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(havc_ctx* c, hipError_t e, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? HAVC_E_OOM : HAVC_E_HIP, m);
}

#define HIP_TRY(ctx, expr)                                         \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return hip_fail((ctx), _e, #expr);   \
    } while (0)

// Both streams idle: required before anything either of them may still touch is freed or re-allocated.
hipError_t sync_streams(havc_ctx* c) {
    hipError_t a = hipStreamSynchronize(c->stream);
    hipError_t b = c->stream2 ? hipStreamSynchronize(c->stream2) : hipSuccess;
    return a != hipSuccess ? a : b;
}

int ensure_scratch(havc_ctx* c, int slot, size_t nbytes) {
    if (c->scratch_sz[slot] >= nbytes) return HAVC_OK;
    // a regrow costs a device synchronisation (hipFree / hipMalloc): grow by at least half, so that a request that creeps up call after call
    // (ColorMNet's memory read: the memory gains a frame every fifth frame) stalls the stream a handful of times, not every time
    const size_t exact = (nbytes + 4095) & ~(size_t)4095;
    if (c->scratch[slot]) nbytes = std::max(nbytes, c->scratch_sz[slot] + c->scratch_sz[slot] / 2);
    nbytes = (nbytes + 4095) & ~(size_t)4095;
    SetupLock setup;
    if (c->scratch[slot]) {
        HIP_TRY(c, sync_streams(c));
        (void)hipFree(c->scratch[slot]);
        c->stats.bytes_resident -= (int64_t)c->scratch_sz[slot];
        c->scratch[slot] = nullptr;
        c->scratch_sz[slot] = 0;
    }
    hipError_t e = hipMalloc(&c->scratch[slot], nbytes);
    if (e == hipErrorOutOfMemory && nbytes > exact) {      // the head-room is a convenience: near the memory limit fall back to the exact size
        (void)hipGetLastError();
        nbytes = exact;
        e = hipMalloc(&c->scratch[slot], nbytes);
    }
    if (e != hipSuccess) { c->scratch[slot] = nullptr; return hip_fail(c, e, "workspace allocation"); }
    c->scratch_sz[slot] = nbytes;
    c->stats.bytes_resident += (int64_t)nbytes;
    return HAVC_OK;
}

// ---- Spline64 polyphase tables (Avisynth/zimg Spline64 kernel, support 4, widened when downscaling) ----
double spline64(double x) {
    x = std::fabs(x);
    if (x < 1.0) return ((49.0 / 41.0 * x - 6387.0 / 2911.0) * x - 3.0 / 2911.0) * x + 1.0;
    if (x < 2.0) { x -= 1.0; return ((-24.0 / 41.0 * x + 4032.0 / 2911.0) * x - 2328.0 / 2911.0) * x; }
    if (x < 3.0) { x -= 2.0; return ((6.0 / 41.0 * x - 1008.0 / 2911.0) * x + 582.0 / 2911.0) * x; }
    if (x < 4.0) { x -= 3.0; return ((-1.0 / 41.0 * x + 168.0 / 2911.0) * x - 97.0 / 2911.0) * x; }
    return 0.0;
}

int get_resize_table(havc_ctx* c, int src, int dst, ResizeTable** out) {
    auto key = std::make_pair(src, dst);
    auto it = c->resize_tables.find(key);
    if (it != c->resize_tables.end()) { *out = &it->second; return HAVC_OK; }
    const double scale = (double)dst / (double)src;
    const double fscale = scale < 1.0 ? scale : 1.0;       // kernel stretch for anti-aliasing
    const double support = 4.0 / fscale;
    const int taps = (int)std::ceil(2.0 * support) + 1;
    std::vector<int> start(dst);
    std::vector<float> w((size_t)dst * taps);
    for (int i = 0; i < dst; ++i) {
        const double center = (i + 0.5) / scale - 0.5;
        const int s0 = (int)std::floor(center - support) + 1;
        double sum = 0;
        std::vector<double> tmp(taps);
        for (int t = 0; t < taps; ++t) { tmp[t] = spline64((s0 + t - center) * fscale); sum += tmp[t]; }
        for (int t = 0; t < taps; ++t) w[(size_t)i * taps + t] = (float)(tmp[t] / sum);
        start[i] = s0;
    }
    ResizeTable tb;
    tb.taps = taps;
    SetupLock setup;
    HIP_TRY(c, hipMalloc((void**)&tb.d_start, dst * sizeof(int)));
    HIP_TRY(c, hipMalloc((void**)&tb.d_w, w.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(tb.d_start, start.data(), dst * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(tb.d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    auto res = c->resize_tables.emplace(key, tb);
    *out = &res.first->second;
    return HAVC_OK;
}

int resize_rgb8(havc_ctx* c, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh, int n,
                const uint8_t* d_orig) {
    ResizeTable *th, *tv;
    int rc = get_resize_table(c, sw, dw, &th);
    if (rc) return rc;
    rc = get_resize_table(c, sh, dh, &tv);
    if (rc) return rc;
    rc = ensure_scratch(c, 7, (size_t)n * sh * dw * 3 * sizeof(float));
    if (rc) return rc;
    int e = launch_resize_passes(d_src, sw, sh, d_dst, dw, dh, n, (float*)c->scratch[7], th->d_start, th->d_w, th->taps,
                                 tv->d_start, tv->d_w, tv->taps, d_orig, c->stream);
    c->stats.launches += 2;
    if (e) return hip_fail(c, (hipError_t)e, "resize");
    return HAVC_OK;
}

// ---- Pillow ImagingResample coefficient tables (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc) ----
struct PilTable { int ksize = 0; int* d_bounds = nullptr; int* d_kk = nullptr; };

double pil_filter(int resample, double x) {
    if (x < 0.0) x = -x;
    if (resample == 2) return x < 1.0 ? 1.0 - x : 0.0;                       // BILINEAR
    const double a = -0.5;                                                    // BICUBIC
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int build_pil_table(havc_ctx* c, int in_size, int out_size, int resample, PilTable* tb) {
    const double fsupport = resample == 2 ? 1.0 : 2.0;
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = fsupport * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    std::vector<int> bounds(out_size * 2), kk((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = pil_filter(resample, (x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
        }
        bounds[xx * 2] = xmin; bounds[xx * 2 + 1] = xmax;
    }
    tb->ksize = ksize;
    SetupLock setup;
    HIP_TRY(c, hipMalloc((void**)&tb->d_bounds, bounds.size() * sizeof(int)));
    HIP_TRY(c, hipMalloc((void**)&tb->d_kk, kk.size() * sizeof(int)));
    HIP_TRY(c, hipMemcpy(tb->d_bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(tb->d_kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice));
    return HAVC_OK;
}

void free_pil_table(PilTable& t) { if (t.d_bounds) (void)hipFree(t.d_bounds); if (t.d_kk) (void)hipFree(t.d_kk); t = PilTable{}; }

// Image.resize on device buffers; tmp must hold n*sh*dw*3 bytes
int pil_resize_dev(havc_ctx* c, const uint8_t* d_src, int sw, int sh, uint8_t* d_tmp, uint8_t* d_dst, int dw, int dh, int n, int resample) {
    PilTable th, tv;
    int rc = HAVC_OK;
    if (sw != dw && (rc = build_pil_table(c, sw, dw, resample, &th))) return rc;
    if (sh != dh && (rc = build_pil_table(c, sh, dh, resample, &tv))) { free_pil_table(th); return rc; }
    int e = launch_pil_resize_passes(d_src, sw, sh, d_tmp, d_dst, dw, dh, n, th.d_bounds, th.d_kk, th.ksize, tv.d_bounds, tv.d_kk,
                                     tv.ksize, c->stream);
    c->stats.launches += 2;
    hipError_t se = hipStreamSynchronize(c->stream);       // tables are freed right away (tiny, rebuilt per call)
    free_pil_table(th); free_pil_table(tv);
    if (e) return hip_fail(c, (hipError_t)e, "pil resize");
    if (se != hipSuccess) return hip_fail(c, se, "pil resize sync");
    return HAVC_OK;
}

inline void* bufptr(havc_net* n, int id) {
    if (id == n->in_buf && n->in_override) return const_cast<void*>(n->in_override);
    if (id == n->out_buf && n->out_override) return n->out_override;
    if (!n->bound.empty() && n->bound[id]) return n->bound[id];
    return n->bufs[id];
}

template <typename T>
inline const T* wptr(havc_net* n, int64_t off) {
    return off < 0 ? nullptr : reinterpret_cast<const T*>(n->w->d_blob + off);
}

int run_op(havc_net* n, const havc_op& op, int batch) {
    havc_ctx* c = n->ctx;
    hipStream_t s = c->cur ? c->cur : c->stream;
    int e = 0;
    const bool timed = (op.tag == c->timed_tag && c->timed_tag >= 0 && op.type == HAVC_OP_CONV);
    std::pair<hipEvent_t, hipEvent_t> evp{nullptr, nullptr};
    if (timed) {
        if (c->tag_used == c->tag_events.size()) {
            hipEvent_t a, b;
            HIP_TRY(c, hipEventCreate(&a));
            HIP_TRY(c, hipEventCreate(&b));
            c->tag_events.emplace_back(a, b);
        }
        evp = c->tag_events[c->tag_used++];
        HIP_TRY(c, hipEventRecord(evp.first, s));
    }
    switch (op.type) {
        case HAVC_OP_CONV: {
            ConvArgs a{};
            a.x = (const half_t*)bufptr(n, op.src);
            a.w = wptr<half_t>(n, op.w_off);
            if (op.flags & HAVC_F_W_FROM_BUF) {
                if (op.src2 < 0 || op.src2 >= (int)n->bufs.size() || (op.flags & HAVC_F_RESIDUAL))
                    return fail(c, HAVC_E_INVALID, "conv op: W_FROM_BUF needs a weight buffer in src2 and no residual");
                if ((uint64_t)n->bufdesc[op.src2].elems_per_frame * n->bufdesc[op.src2].elem_bytes < (uint64_t)op.Npad * op.Kc * 16)
                    return fail(c, HAVC_E_INVALID, "conv op: W_FROM_BUF buffer smaller than Npad x Kc x 16 bytes per frame");
                a.w = (const half_t*)bufptr(n, op.src2);
            }
            a.bias = wptr<float>(n, op.bias_off);
            a.scale = wptr<float>(n, op.scale_off);
            a.shift = wptr<float>(n, op.shift_off);
            int rows_per_frame = op.Ho * op.Wo;
            if (op.flags & HAVC_F_PS_BLUR) {
                if (!(op.flags & HAVC_F_OUT_PIXSHUF) || op.kh != 1 || op.kw != 1 || op.stride != 1 || op.pad != 0 || (op.Co & 63) ||
                    op.Npad != 4 * op.Co || (op.Npad & 255) || op.out_step > 1 || (op.flags & (HAVC_F_RESIDUAL | HAVC_F_RELU_POST)))
                    return fail(c, HAVC_E_INVALID, "conv op: PS_BLUR needs a 1x1 stride-1 pixel-shuffle conv with Co % 64 == 0");
                if (op.aux0 < 0 || (op.aux0 & 7) || op.aux0 > op.Co) return fail(c, HAVC_E_INVALID, "conv op: PS_BLUR aux0 = stored channels (multiple of 8, <= Co)");
                a.cstore = op.aux0;
                a.tiles_y = (op.Ho + 14) / 15;
                a.tiles_x = (op.Wo + 14) / 15;
                rows_per_frame = a.tiles_y * a.tiles_x * 256;
            }
            if (op.flags & HAVC_F_FUSE_RGB8) {
                if (op.Npad != 272 || !a.scale || !a.shift || op.aux0 < 0 || op.aux0 >= (int)n->bufs.size())
                    return fail(c, HAVC_E_INVALID, "conv op: FUSE_RGB8 needs Npad 272, fused weights/bias and an RGB8 buffer");
                a.fuse_w = a.scale; a.fuse_b = a.shift; a.scale = a.shift = nullptr;
                a.fuse_rgb = (uint8_t*)bufptr(n, op.aux0);
            }
            uint64_t proj_wf = 0, proj_of = 0;                 // FUSE_PROJ: bytes per frame of the projection matrix / of the output
            if (op.flags & HAVC_F_FUSE_PROJ) {
                if ((op.Npad & 255) || ((op.Ho * op.Wo) & 15) || (op.flags & (HAVC_F_RESIDUAL | HAVC_F_W_FROM_BUF | HAVC_F_OUT_PIXSHUF)) || op.src2 < 0 ||
                    op.src2 >= (int)n->bufs.size() || op.aux0 < 0 || op.aux0 >= (int)n->bufs.size())
                    return fail(c, HAVC_E_INVALID, "conv op: FUSE_PROJ needs Npad % 256 == 0, Ho*Wo % 16 == 0, a matrix buffer (src2) and an output buffer (aux0)");
                proj_wf = (uint64_t)n->bufdesc[op.src2].elems_per_frame * n->bufdesc[op.src2].elem_bytes;
                proj_of = (uint64_t)n->bufdesc[op.aux0].elems_per_frame * n->bufdesc[op.aux0].elem_bytes;
                if (proj_wf != 2 * 256 * 4 || proj_of < (uint64_t)op.Ho * op.Wo * (op.Npad / 256) * 8)
                    return fail(c, HAVC_E_INVALID, "conv op: FUSE_PROJ buffers: fp32 [2][256] matrix per frame, fp32 [Ho*Wo][Npad/256][2] output");
                a.fuse_w = (const float*)bufptr(n, op.src2);
                a.fuse_out = (float*)bufptr(n, op.aux0);
            }
            a.res = (op.flags & HAVC_F_RESIDUAL) ? (const half_t*)bufptr(n, op.src2) : nullptr;
            a.y = bufptr(n, op.dst);
            a.x_cpitch = op.src_cpitch; a.x_coff = op.src_coff;
            a.res_cpitch = op.res_cpitch; a.res_coff = op.res_coff;
            a.y_cpitch = op.dst_cpitch; a.y_coff = op.dst_coff;
            a.Hi = op.Hi; a.Wi = op.Wi; a.C8 = op.Ci / 8;
            a.Ho = op.Ho; a.Wo = op.Wo; a.Co = op.Co;
            a.kh = op.kh; a.kw = op.kw; a.stride = op.stride; a.pad = op.pad; a.dil = op.dil;
            a.pad_w = op.pad + op.pad_w_delta;
            a.oss = op.out_step == 2 ? 2 : 1; a.ooy = op.out_oy; a.oox = op.out_ox;
            a.Kc = op.Kc; a.Npad = op.Npad;
            a.M = batch * op.Ho * op.Wo;   // (PS_BLUR: set below)
            a.flags = op.flags;
            if (c->nt_store_bytes && (uint64_t)batch * n->bufdesc[op.dst].elems_per_frame * n->bufdesc[op.dst].elem_bytes >= c->nt_store_bytes)
                a.flags |= HAVC_F_NT_STORE;
            a.pix_pitch = (op.flags & HAVC_F_PS_BLUR) ? 0 : op.aux0;
            a.C8a = (op.aux1 > 0 && op.aux1 < op.Ci / 8) ? op.aux1 : op.Ci / 8;
            a.f0 = op.f0; a.f1 = op.f1; a.f2 = op.f2;
            a.cfg = op.reserved;
            { static const int prio = [] { const char* e = getenv("HAVC_SETPRIO"); return e ? atoi(e) != 0 : 1; }(); a.prio = prio; }
            a.splitk = HAVC_F_SPLITK_COUNT(op.flags);
            if (a.splitk == 1) a.splitk = 0;
            a.pscale = 1.f;
            if (op.flags & HAVC_F_PRECISE) {
                if ((op.flags & HAVC_F_W_FROM_BUF) || a.splitk ||
                    !(op.f3 > 0.f) || (op.src_cpitch & 15) || (!(op.flags & HAVC_F_OUT_RGB8) && (op.dst_cpitch & 15)))
                    return fail(c, HAVC_E_INVALID, "conv op: PRECISE needs a plain conv (no fused / transposed / split-K form), f3 = accumulator scale > 0, hi|lo pixel rows");
                a.pscale = op.f3;
            }
            if (a.splitk) {
                if ((op.flags & (HAVC_F_PS_BLUR | HAVC_F_FUSE_RGB8 | HAVC_F_FUSE_PROJ | HAVC_F_W_FROM_BUF | HAVC_F_OUT_RGB8)) || (op.Kc >> 3) < 2 * a.splitk)
                    return fail(c, HAVC_E_INVALID, "conv op: SPLITK needs a plain conv with at least 2 K stages per part");
                if (s != c->stream && s != c->stream2) return fail(c, HAVC_E_INVALID, "conv op: SPLITK ops run on the ctx's own streams (one workspace per stream)");
                const int slot = s == c->stream ? 12 : 13;         // the two generators of a stable / artistic render run side by side on the two streams
                const size_t need = (size_t)a.splitk * batch * op.Ho * op.Wo * op.Npad * 4;
                if (int rc2 = ensure_scratch(c, slot, need)) return rc2;
                a.ws = (float*)c->scratch[slot];
            }
            {
                const int oi = (int)(&op - n->ops.data());
                a.ktab = n->ktab_off[oi] >= 0 ? n->d_ktab + n->ktab_off[oi] : nullptr;
                const uint64_t xb = (uint64_t)n->bufdesc[op.src].elems_per_frame * n->bufdesc[op.src].elem_bytes * n->max_batch + 256;
                a.x_bytes = xb < 0xF0000000ull ? (unsigned)xb : 0u;          // 0: descriptor path unavailable
                const uint64_t wb = (uint64_t)op.Npad * op.Kc * 16;
                a.w_bytes = wb < 0xF0000000ull ? (unsigned)wb : 0xFFFFFFFFu;
            }
            const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
            for (int i = 0; i < 3; ++i) { a.mean[i] = mean[i]; a.istd[i] = stdv[i]; }
            if (!a.w || (op.Kc & 7) || a.C8 <= 0 || !a.ktab) return fail(c, HAVC_E_INVALID, "conv op: bad weights / Kc / Ci");
            if ((op.flags & HAVC_F_AFFINE) && (!a.scale || !a.shift)) return fail(c, HAVC_E_INVALID, "conv op: AFFINE without scale/shift");
            {
                // frames per launch: keep every operand within the 32-bit range of a buffer descriptor (and of the
                // kernels' int pixel indices); big batches of the 560x560 tail run as several launches.
                const uint64_t lim = c->desc_limit;
                const uint64_t xf = (uint64_t)n->bufdesc[op.src].elems_per_frame * n->bufdesc[op.src].elem_bytes;
                const uint64_t yf = (uint64_t)n->bufdesc[op.dst].elems_per_frame * n->bufdesc[op.dst].elem_bytes;
                const uint64_t rf = a.res ? (uint64_t)n->bufdesc[op.src2].elems_per_frame * n->bufdesc[op.src2].elem_bytes : 0;
                int chunk = batch;
                const uint64_t big = std::max(xf, std::max(yf, rf));
                if (big * (uint64_t)batch > lim) {
                    chunk = (int)std::max<uint64_t>(1, lim / big);
                    const int nch = (batch + chunk - 1) / chunk;           // equal launches (32 frames of the 560 x 560 tail: 16 + 16, not 18 + 14)
                    chunk = (batch + nch - 1) / nch;
                }
                // A multi-frame launch walks M = frames x rows CONTIGUOUS rows.  A view that covers only part of its buffer's frame (the h0 x w0
                // patch rows of a DINOv2 token buffer whose frame also holds the class row) does not continue into the next frame:
                // one launch per frame, each at its buffer's own frame stride.
                {
                    const bool plain_out = !(op.flags & (HAVC_F_OUT_PIXSHUF | HAVC_F_OUT_TRANSPOSED | HAVC_F_OUT_RGB8 | HAVC_F_FUSE_RGB8 | HAVC_F_FUSE_PROJ | HAVC_F_PS_BLUR)) && a.oss == 1;
                    bool contiguous = (uint64_t)op.Hi * op.Wi * op.src_cpitch == (uint64_t)n->bufdesc[op.src].elems_per_frame;
                    if (plain_out && (uint64_t)op.Ho * op.Wo * op.dst_cpitch != (uint64_t)n->bufdesc[op.dst].elems_per_frame) contiguous = false;
                    if (plain_out && a.res && !(op.flags & HAVC_F_W_FROM_BUF) && (uint64_t)op.Ho * op.Wo * op.res_cpitch != (uint64_t)n->bufdesc[op.src2].elems_per_frame)
                        contiguous = false;
                    if (!contiguous && batch > 1) chunk = 1;
                }
                const char* w0 = (const char*)a.w;
                uint64_t wf = 0;
                if (op.flags & HAVC_F_W_FROM_BUF) {        // per-frame weights taken from an activation buffer: one launch per frame
                    chunk = 1;
                    wf = (uint64_t)n->bufdesc[op.src2].elems_per_frame * n->bufdesc[op.src2].elem_bytes;
                }
                const char* x0 = (const char*)a.x; const char* r0 = (const char*)a.res; char* y0 = (char*)a.y;
                uint8_t* rgb0 = a.fuse_rgb;
                for (int f0 = 0; f0 < batch && e == 0; f0 += chunk) {
                    const int nb = std::min(chunk, batch - f0);
                    if (timed && f0 > 0) {                                 // tag timing is per LAUNCH: close this pair, open the next
                        HIP_TRY(c, hipEventRecord(evp.second, s));
                        if (c->tag_used == c->tag_events.size()) {
                            hipEvent_t ea, eb;
                            HIP_TRY(c, hipEventCreate(&ea));
                            HIP_TRY(c, hipEventCreate(&eb));
                            c->tag_events.emplace_back(ea, eb);
                        }
                        evp = c->tag_events[c->tag_used++];
                        HIP_TRY(c, hipEventRecord(evp.first, s));
                    }
                    a.x = (const half_t*)(x0 + (uint64_t)f0 * xf);
                    a.y = y0 + (uint64_t)f0 * yf;
                    if (r0) a.res = (const half_t*)(r0 + (uint64_t)f0 * rf);
                    if (wf) a.w = (const half_t*)(w0 + (uint64_t)f0 * wf);
                    if (rgb0) a.fuse_rgb = rgb0 + (uint64_t)f0 * op.Ho * op.Wo * 3;
                    if (proj_wf) {
                        a.fuse_w = (const float*)((const char*)bufptr(n, op.src2) + (uint64_t)f0 * proj_wf);
                        a.fuse_out = (float*)((char*)bufptr(n, op.aux0) + (uint64_t)f0 * proj_of);
                    }
                    a.M = nb * rows_per_frame;
                    a.x_bytes = (unsigned)std::min<uint64_t>(xf * (uint64_t)nb + 256, 0xEFFFFFFFull);
                    e = launch_conv(a, s);
                    if (f0 > 0) c->stats.launches += 1;
                }
            }
            break;
        }
        case HAVC_OP_MAXPOOL:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_maxpool3x3s2_p((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ho, op.Wo, op.Ci, op.src_cpitch,
                                          op.src_coff, op.dst_cpitch, op.dst_coff, s);
                break;
            }
            e = launch_maxpool3x3s2((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ho,
                                    op.Wo, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, s);
            break;
        case HAVC_OP_BLUR_RESIZE:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_blur_resize_p((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ho, op.Wo, op.Ci, op.src_cpitch,
                                         op.src_coff, op.dst_cpitch, op.dst_coff, s);
                break;
            }
            e = launch_blur_resize((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ho,
                                   op.Wo, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, s);
            break;
        case HAVC_OP_AFFINE:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_affine_p((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), wptr<float>(n, op.scale_off), wptr<float>(n, op.shift_off),
                                    (op.flags & HAVC_F_RELU_POST) ? 1 : 0, (int64_t)batch * op.Hi * op.Wi, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch,
                                    op.dst_coff, s);
                break;
            }
            e = launch_affine((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), wptr<float>(n, op.scale_off),
                              wptr<float>(n, op.shift_off), (op.flags & HAVC_F_RELU_POST) ? 1 : 0,
                              (int64_t)batch * op.Hi * op.Wi, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch,
                              op.dst_coff, s);
            break;
        case HAVC_OP_COPY_CH:
            e = launch_copy_ch((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), (int64_t)batch * op.Hi * op.Wi,
                               op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, s);
            break;
        case HAVC_OP_ATTENTION:
            if (op.flags & HAVC_F_PRECISE) {
                // aux1 = NHWC value buffer (pixel pitch Kc), kh = fp32 scratch buffer [N][2] per frame (row maximum and sum of the softmax)
                if (op.kh < 0 || op.kh >= (int)n->bufs.size() || n->bufdesc[op.kh].elem_bytes != 4 ||
                    (uint64_t)n->bufdesc[op.kh].elems_per_frame < (uint64_t)op.Hi * op.Wi * 2 || !attention_p_supported(op.aux0, op.Ci))
                    return fail(c, HAVC_E_INVALID, "attention op: PRECISE needs an fp32 [N][2] scratch buffer in kh, d <= 128, C % 128 == 0");
                if (op.flags & HAVC_F_OUT_TRANSPOSED) {
                    // round 5: the value map arrives TRANSPOSED ([2][C][Kc] fp16 per frame: hi plane, lo plane; written by the value conv's transposed precise
                    // epilogue) and the P . H product runs on MFMA with the three-term splitting (csrc/precise.hip pattn_apply_mfma_kernel)
                    if (!attention_pm_supported(op.aux0, op.Ci, op.Kc) || (uint64_t)n->bufdesc[op.aux1].elems_per_frame < (uint64_t)2 * op.Ci * op.Kc ||
                        op.Kc < ((op.Hi * op.Wi + 31) & ~31))
                        return fail(c, HAVC_E_INVALID, "attention op: PRECISE | OUT_TRANSPOSED needs C % 256 == 0 and a [2][C][Kc] value buffer, Kc % 32 == 0, Kc >= N");
                    e = launch_attention_pm((const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff, op.res_coff + op.aux0, op.aux0,
                                            n->bufdesc[op.src2].elems_per_frame, (const half_t*)bufptr(n, op.aux1), op.Kc, n->bufdesc[op.aux1].elems_per_frame,
                                            (const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame,
                                            (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, n->bufdesc[op.dst].elems_per_frame, (float*)bufptr(n, op.kh),
                                            batch, op.Hi * op.Wi, op.Ci, op.f0, s);
                    c->stats.launches += 1;
                    break;
                }
                e = launch_attention_p((const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff, op.res_coff + op.aux0, op.aux0,
                                       n->bufdesc[op.src2].elems_per_frame, (const half_t*)bufptr(n, op.aux1), op.Kc, 0, n->bufdesc[op.aux1].elems_per_frame,
                                       (const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame,
                                       (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, n->bufdesc[op.dst].elems_per_frame, (float*)bufptr(n, op.kh),
                                       batch, op.Hi * op.Wi, op.Ci, op.f0, s);
                c->stats.launches += 1;
                break;
            }
            e = launch_attention((const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff, op.res_coff + op.aux0,
                                 op.aux0, (const half_t*)bufptr(n, op.aux1), op.Ci, op.Kc, (const half_t*)bufptr(n, op.src),
                                 op.src_cpitch, op.src_coff, (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, batch,
                                 op.Hi * op.Wi, op.f0, s);
            break;
        case HAVC_OP_PREP_RGB8:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_prep_rgb8_p((const uint8_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                       op.src2 >= 0 ? (half_t*)bufptr(n, op.src2) : nullptr, op.res_cpitch, op.res_coff, (int64_t)batch * op.Hi * op.Wi, s);
                break;
            }
            e = launch_prep_rgb8((const uint8_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                 op.src2 >= 0 ? (half_t*)bufptr(n, op.src2) : nullptr, op.res_cpitch, op.res_coff,
                                 (int64_t)batch * op.Hi * op.Wi, s, op.aux0 > 0 && op.res_coff + 8 + op.aux0 <= op.res_cpitch ? op.aux0 : 0);
            break;
        case HAVC_OP_SUBSAMPLE2:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_subsample2_p((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Ho, op.Wo, op.Hi, op.Wi, op.Ci,
                                        op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, s);
                break;
            }
            e = launch_subsample2((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Ho, op.Wo, op.Hi, op.Wi, op.Ci,
                                  op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, s);
            break;
        case HAVC_OP_PROJ2:
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_proj2_p((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Ci, wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off),
                                   op.flags & 3, op.f0, (float*)bufptr(n, op.dst), (int64_t)batch * op.Hi * op.Wi, s);
                break;
            }
            e = launch_proj2((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Ci, wptr<float>(n, op.w_off),
                             wptr<float>(n, op.bias_off), op.flags, op.f0, (float*)bufptr(n, op.dst), (int64_t)batch * op.Hi * op.Wi, s);
            break;
        case HAVC_OP_BILINEAR2:
            e = launch_bilinear2((const float*)bufptr(n, op.src), (float*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ho, op.Wo, op.f0, s);
            break;
        case HAVC_OP_PREP_LAB_L:
            e = launch_prep_lab_l((const uint8_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                  (int64_t)batch * op.Hi * op.Wi, s, (op.flags & HAVC_F_PRECISE) ? op.dst_cpitch >> 1 : 0);
            break;
        case HAVC_OP_DWCONV7:
            if (op.w_off < 0 || (op.Ci & 7)) return fail(c, HAVC_E_INVALID, "dwconv7 op: weights / channel count");
            if (op.flags & HAVC_F_PRECISE) {                   // fp32 weights [49][Kc]
                e = launch_dwconv7_p((const half_t*)bufptr(n, op.src), wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off), (half_t*)bufptr(n, op.dst), batch,
                                     op.Hi, op.Wi, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, op.Kc, s);
                break;
            }
            e = launch_dwconv7((const half_t*)bufptr(n, op.src), wptr<half_t>(n, op.w_off), wptr<float>(n, op.bias_off), (half_t*)bufptr(n, op.dst),
                               batch, op.Hi, op.Wi, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff, op.Kc, s);
            break;
        case HAVC_OP_DWCONV7_LN:
            if (op.w_off < 0 || op.scale_off < 0 || op.shift_off < 0 || !dwconv7_ln_supported(op.Ci))
                return fail(c, HAVC_E_INVALID, "dwconv7+layernorm op: weights / gamma / beta / channel count");
            if (op.flags & HAVC_F_PRECISE) {                       // pair tensors, fp32 weights (round 6)
                if (!dwconv7_ln_p_supported(op.Ci)) return fail(c, HAVC_E_INVALID, "dwconv7+layernorm op: the precise form exists for 192 / 384 / 768 channels");
                e = launch_dwconv7_ln_p((const half_t*)bufptr(n, op.src), wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off), wptr<float>(n, op.scale_off),
                                        wptr<float>(n, op.shift_off), op.f0, (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ci, op.src_cpitch,
                                        op.src_coff, op.dst_cpitch, op.dst_coff, op.Kc, s);
                break;
            }
            e = launch_dwconv7_ln((const half_t*)bufptr(n, op.src), wptr<half_t>(n, op.w_off), wptr<float>(n, op.bias_off), wptr<float>(n, op.scale_off),
                                  wptr<float>(n, op.shift_off), op.f0, (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Ci, op.src_cpitch,
                                  op.src_coff, op.dst_cpitch, op.dst_coff, op.Kc, s);
            break;
        case HAVC_OP_FOLD_QUERIES:
            if (op.w_off < 0 || op.Ho < 1 || op.Ho > op.Kc || op.Ho > op.Wi || (op.Ci & 7) || n->bufdesc[op.dst].elem_bytes != 4 ||
                (uint64_t)n->bufdesc[op.dst].elems_per_frame < (uint64_t)2 * op.Ci)
                return fail(c, HAVC_E_INVALID, "fold-queries op: R weights, query count, fp32 [2][Ci] destination");
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_fold_queries_p((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Wi, wptr<float>(n, op.w_off), op.Kc, op.Ho,
                                          (float*)bufptr(n, op.dst), batch, op.Ci, s);
                break;
            }
            e = launch_fold_queries((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Wi, wptr<float>(n, op.w_off), op.Kc, op.Ho,
                                    (float*)bufptr(n, op.dst), batch, op.Ci, s);
            break;
        case HAVC_OP_SHUF4_BLUR_AB:
            if ((op.flags & HAVC_F_PRECISE) && op.Ci == 2) {
                // precise plan with the projection fused into the last_shuf conv (round 6): src = its fp32 [Hi*Wi][16][2] output, image and ab map as pairs
                if (op.w_off < 0 || op.bias_off < 0 || op.src2 < 0 || n->bufdesc[op.src].elem_bytes != 4 ||
                    (uint64_t)n->bufdesc[op.src].elems_per_frame < (uint64_t)op.Hi * op.Wi * 32)
                    return fail(c, HAVC_E_INVALID, "shuffle+blur(ab) op, precise: weights, image view, fp32 [Hi*Wi][16][2] source");
                e = launch_shuf4_blur_ab_p((const float*)bufptr(n, op.src), (const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff,
                                           wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                           batch, op.Hi, op.Wi, s);
                break;
            }
            if (op.flags & HAVC_F_PRECISE) {
                // precise form: src = the last_shuf conv's pair tensor [Hi][Wi][16 x 256], aux0 = the folded projection (fp32 [2][256] per frame, FOLD_QUERIES)
                if (op.w_off < 0 || op.bias_off < 0 || op.src2 < 0 || op.aux0 < 0 || op.aux0 >= (int)n->bufs.size() || n->bufdesc[op.aux0].elem_bytes != 4 ||
                    (uint64_t)n->bufdesc[op.aux0].elems_per_frame < 512 || n->bufdesc[op.src].elem_bytes != 2 || op.Ci != 16 * 256)
                    return fail(c, HAVC_E_INVALID, "shuffle+blur(ab) op, precise: weights, image view, projection buffer (aux0), 4096-channel pair source");
                e = launch_shuf4_blur_proj_p((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, (const float*)bufptr(n, op.aux0),
                                             (const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff, wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off),
                                             (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, batch, op.Hi, op.Wi, s);
                break;
            }
            if (op.w_off < 0 || op.bias_off < 0 || op.src2 < 0 || n->bufdesc[op.src].elem_bytes != 4 ||
                (uint64_t)n->bufdesc[op.src].elems_per_frame < (uint64_t)op.Hi * op.Wi * 32)
                return fail(c, HAVC_E_INVALID, "shuffle+blur(ab) op: weights, image view, fp32 [Hi*Wi][16][2] source");
            e = launch_shuf4_blur_ab((const float*)bufptr(n, op.src), (const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff,
                                     wptr<float>(n, op.w_off), wptr<float>(n, op.bias_off), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                     batch, op.Hi, op.Wi, s);
            break;
        case HAVC_OP_LAYERNORM:
            if (op.scale_off < 0 || op.shift_off < 0) return fail(c, HAVC_E_INVALID, "layernorm op: gamma / beta");
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_layernorm_p((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), wptr<float>(n, op.scale_off), wptr<float>(n, op.shift_off),
                                       op.f0, (int64_t)batch * op.Hi * op.Wi, op.Ci, op.src_cpitch, op.src_coff, op.dst_cpitch, op.dst_coff,
                                       (op.flags & HAVC_F_RELU_POST) ? 1 : 0, s);
                break;
            }
            e = launch_layernorm_c((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), wptr<float>(n, op.scale_off),
                                   wptr<float>(n, op.shift_off), op.f0, (int64_t)batch * op.Hi * op.Wi, op.Ci, op.src_cpitch, op.src_coff,
                                   op.dst_cpitch, op.dst_coff, s, (op.flags & HAVC_F_RELU_POST) ? 1 : 0);
            break;
        case HAVC_OP_MHA:
            if (op.src2 < 0 || op.Ci != op.kh * 32) return fail(c, HAVC_E_INVALID, "mha op: K/V buffer, head dim 32");
            if (op.flags & HAVC_F_PRECISE) {
                e = launch_mha32_p((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Wi, (const half_t*)bufptr(n, op.src2), op.res_cpitch,
                                   op.res_coff, op.aux0, op.Wo, (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, op.Wi, batch, op.kh, op.Hi, op.Ho,
                                   op.f0, s);
                break;
            }
            if (op.aux1 >= 0 && op.aux1 < (int)n->bufs.size() && n->bufdesc[op.aux1].elem_bytes == 4) {
                // keys split over blocks (K / V staged in LDS once per block), partial softmax states in the fp32 buffer aux1
                if ((size_t)n->bufdesc[op.aux1].elems_per_frame < (size_t)op.kh * mha32_nsplit(op.Ho) * op.Hi * 34)
                    return fail(c, HAVC_E_INVALID, "mha op: partial-state buffer too small");
                e = launch_mha32_split((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Wi, (const half_t*)bufptr(n, op.src2),
                                       op.res_cpitch, op.res_coff, op.aux0, op.Wo, (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, op.Wi,
                                       (float*)bufptr(n, op.aux1), batch, op.kh, op.Hi, op.Ho, op.f0, s);
                c->stats.launches += 1;
                break;
            }
            e = launch_mha32((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.Wi, (const half_t*)bufptr(n, op.src2), op.res_cpitch,
                             op.res_coff, op.aux0, op.Wo, (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff, op.Wi, batch, op.kh, op.Hi,
                             op.Ho, op.f0, s);
            break;
        case HAVC_OP_PIXSHUF4_BLUR:
            e = launch_pixshuf4_blur((const half_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), batch, op.Hi, op.Wi, op.Co, op.src_cpitch,
                                     op.src_coff, op.dst_cpitch, op.dst_coff, s);
            break;
        case HAVC_OP_PREP_DDCOLOR:
            e = launch_prep_ddcolor((const uint8_t*)bufptr(n, op.src), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                    op.src2 >= 0 ? (half_t*)bufptr(n, op.src2) : nullptr, op.res_cpitch, op.res_coff,
                                    (int64_t)batch * op.Hi * op.Wi, s, (op.flags & HAVC_F_PRECISE) ? 1 : 0);
            break;
        case HAVC_OP_EW: {
            EwArgs a{};
            const bool dual = op.flags & HAVC_EW_DUAL, res = op.flags & HAVC_EW_RES;
            if ((op.Ci & 7) || op.kh < 0 || op.kh > 2 || (res && op.src2 < 0) || (dual && (op.aux0 < 0 || op.aux0 >= (int)n->bufs.size())) ||
                (op.kh == 0 && (op.Hi != op.Ho || op.Wi != op.Wo)) || (op.kh == 2 && (op.kw < 1 || op.Ho * op.kw != op.Hi || op.Wo * op.kw != op.Wi)))
                return fail(c, HAVC_E_INVALID, "ew op: channels / mode / sizes / buffers");
            a.x = (const half_t*)bufptr(n, op.src); a.y = (half_t*)bufptr(n, op.dst);
            a.res = res ? (const half_t*)bufptr(n, op.src2) : nullptr;
            a.y2 = dual ? (half_t*)bufptr(n, op.aux0) : nullptr;
            a.B = batch; a.Hi = op.Hi; a.Wi = op.Wi; a.Ho = op.Ho; a.Wo = op.Wo; a.C8 = op.Ci / 8;
            a.x_cp = op.src_cpitch; a.x_co = op.src_coff; a.y_cp = op.dst_cpitch; a.y_co = op.dst_coff;
            a.r_cp = op.res_cpitch; a.r_co = op.res_coff; a.y2_cp = op.Kc; a.y2_co = op.aux1;
            a.x_fs = (op.flags & HAVC_EW_SRC_BCAST) ? 0 : n->bufdesc[op.src].elems_per_frame;
            a.y_fs = n->bufdesc[op.dst].elems_per_frame;
            a.r_fs = (!res || (op.flags & HAVC_EW_RES_BCAST)) ? 0 : n->bufdesc[op.src2].elems_per_frame;
            a.y2_fs = dual ? n->bufdesc[op.aux0].elems_per_frame : 0;
            a.mode = op.kh; a.factor = op.kw; a.flags = op.flags; a.rh = op.f0; a.rw = op.f1;
            e = launch_ew(a, s);
            break;
        }
        case HAVC_OP_DWCONV:
            if (op.w_off < 0 || (op.Ci & 7) || (op.kh != 3 && op.kh != 5)) return fail(c, HAVC_E_INVALID, "dwconv op: weights / channels / kernel size");
            e = launch_dwconv((const half_t*)bufptr(n, op.src), wptr<half_t>(n, op.w_off), wptr<float>(n, op.bias_off), (half_t*)bufptr(n, op.dst), batch,
                              op.Hi, op.Wi, op.Ci, op.kh, op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame, op.dst_cpitch, op.dst_coff,
                              n->bufdesc[op.dst].elems_per_frame, op.Kc, s);
            break;
        case HAVC_OP_CHAN_ATTN: {
            const int heads = op.kh, cc = heads > 0 ? op.Ci / heads : 0, P = op.Hi * op.Wi;
            if (heads < 1 || cc * heads != op.Ci || cc > 256 || (cc & 7) || op.src2 < 0 || op.scale_off < 0 || op.aux0 < 0 || op.aux1 < 0 ||
                op.aux0 >= (int)n->bufs.size() || op.aux1 >= (int)n->bufs.size())
                return fail(c, HAVC_E_INVALID, "channel-attention op: heads / channels per head (<= 256) / buffers");
            const int S = chan_attn_splits(P, heads, cc);
            if ((uint64_t)n->bufdesc[op.aux0].elems_per_frame < (uint64_t)heads * S * cc * cc || (uint64_t)n->bufdesc[op.aux1].elems_per_frame < (uint64_t)S * 2 * op.Ci ||
                (uint64_t)n->bufdesc[op.dst].elems_per_frame < (uint64_t)op.Ci * op.Kc * 8)
                return fail(c, HAVC_E_INVALID, "channel-attention op: scratch / matrix buffers too small");
            e = launch_chan_attn((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame,
                                 (const half_t*)bufptr(n, op.src2), op.res_cpitch, op.res_coff, n->bufdesc[op.src2].elems_per_frame, wptr<float>(n, op.scale_off),
                                 (float*)bufptr(n, op.aux0), (float*)bufptr(n, op.aux1), (half_t*)bufptr(n, op.dst), n->bufdesc[op.dst].elems_per_frame,
                                 op.Kc * 8, batch, P, heads, cc, s);
            c->stats.launches += 1;
            break;
        }
        case HAVC_OP_MHA64:
            if (op.Ci != op.kh * 64 || op.Ho < 1 || op.Ho > op.Wi) return fail(c, HAVC_E_INVALID, "mha64 op: head dim 64, live tokens <= tokens per frame");
            e = launch_mha64((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, op.res_coff, op.aux0, op.Wi, (half_t*)bufptr(n, op.dst),
                             op.dst_cpitch, op.dst_coff, op.Wi, batch, op.kh, op.Ho, op.f0, s);
            break;
        case HAVC_OP_CBAM: {
            const int C = op.Ci, Ch = C / 16;
            const bool dual = op.flags & HAVC_EW_DUAL;
            if (op.w_off < 0 || (C & 15) || op.aux0 < 0 || op.aux1 < 0 || (dual && op.src2 < 0) || (uint64_t)n->bufdesc[op.aux0].elems_per_frame != 3 * (uint64_t)C ||
                (uint64_t)n->bufdesc[op.aux1].elems_per_frame < (uint64_t)op.Hi * op.Wi * 2)
                return fail(c, HAVC_E_INVALID, "cbam op: weights / channels / scratch buffers");
            const float* w1 = wptr<float>(n, op.w_off);
            const float *b1 = w1 + (size_t)Ch * C, *w2 = b1 + Ch, *b2 = w2 + (size_t)C * Ch, *w7 = b2 + C, *b7 = w7 + 98;
            e = launch_cbam((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame, batch, op.Hi, op.Wi, C, w1, b1,
                            w2, b2, w7, b7, (float*)bufptr(n, op.aux0), (float*)bufptr(n, op.aux1), (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                            n->bufdesc[op.dst].elems_per_frame, dual ? (half_t*)bufptr(n, op.src2) : nullptr, op.res_cpitch, op.res_coff,
                            dual ? n->bufdesc[op.src2].elems_per_frame : 0, s);
            c->stats.launches += 3;
            break;
        }
        case HAVC_OP_GRU:
            if (op.src2 < 0 || n->bufdesc[op.src2].elem_bytes != 4 || n->bufdesc[op.dst].elem_bytes != 4 || op.Co < 1)
                return fail(c, HAVC_E_INVALID, "gru op: fp32 planar hidden buffers");
            e = launch_gru((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame, (const float*)bufptr(n, op.src2),
                           (float*)bufptr(n, op.dst), batch, op.Hi * op.Wi, op.Co, s);
            break;
        case HAVC_OP_PLANAR_IN:
            if (n->bufdesc[op.src].elem_bytes != 4 || (op.Co & 7) || op.Ci > op.Co) return fail(c, HAVC_E_INVALID, "planar-in op: fp32 source, Ci <= Co, Co % 8 == 0");
            e = launch_planar_in((const float*)bufptr(n, op.src), n->bufdesc[op.src].elems_per_frame, (half_t*)bufptr(n, op.dst), op.dst_cpitch, op.dst_coff,
                                 n->bufdesc[op.dst].elems_per_frame, batch, op.Hi * op.Wi, op.Ci, op.Co, op.flags & 1, (op.flags & 2) ? 1 : 0, s);
            break;
        case HAVC_OP_CMN_DECODER_IN:
            if (op.src2 < 0 || op.src2 >= (int)n->bufs.size() || op.aux0 < 0 || op.aux0 >= (int)n->bufs.size() || op.aux1 < 0 || op.aux1 >= (int)n->bufs.size() ||
                n->bufdesc[op.src2].elem_bytes != 4 || n->bufdesc[op.aux0].elem_bytes != 4 || n->bufdesc[op.src].elem_bytes != 2 ||
                n->bufdesc[op.aux1].elem_bytes != 2 || n->bufdesc[op.aux1].elems_per_frame != n->bufdesc[op.dst].elems_per_frame || op.kh < 8 || op.kw < 8 ||
                op.Co != op.Ci + op.kh + op.kw || op.dst_coff + op.Co > op.dst_cpitch)
                return fail(c, HAVC_E_INVALID, "decoder-in op: fp16 features, fp32 planar readout (kh channels) and hidden (kw), twin destination buffers, Co = Ci + kh + kw");
            e = launch_cmn_decoder_in((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, (const float*)bufptr(n, op.src2), n->bufdesc[op.src2].elems_per_frame,
                                      (const float*)bufptr(n, op.aux0), n->bufdesc[op.aux0].elems_per_frame, (half_t*)bufptr(n, op.dst), (half_t*)bufptr(n, op.aux1),
                                      op.dst_cpitch, op.dst_coff, n->bufdesc[op.dst].elems_per_frame, batch, op.Hi * op.Wi, op.Ci, op.kh, op.kw, s);
            break;
        case HAVC_OP_PLANAR_OUT:
            if (n->bufdesc[op.dst].elem_bytes != 4 || op.kh < 0 || op.kh > 3) return fail(c, HAVC_E_INVALID, "planar-out op: fp32 destination, activation 0..3");
            e = launch_planar_out((const half_t*)bufptr(n, op.src), op.src_cpitch, op.src_coff, n->bufdesc[op.src].elems_per_frame, (float*)bufptr(n, op.dst),
                                  n->bufdesc[op.dst].elems_per_frame, batch, op.Hi * op.Wi, op.Ci, op.kh, s);
            break;
        default:
            return fail(c, HAVC_E_INVALID, "unknown op type");
    }
    c->stats.launches += 1;
    if (e) return hip_fail(c, (hipError_t)e, "kernel launch");
    if (timed) HIP_TRY(c, hipEventRecord(evp.second, s));
    return HAVC_OK;
}

int run_ops_locked(havc_net* n, int first, int count, int batch) {
    if (batch < 1 || batch > n->max_batch) return fail(n->ctx, HAVC_E_INVALID, "batch out of range");
    if (first < 0 || count < 0 || first + count > (int)n->ops.size()) return fail(n->ctx, HAVC_E_INVALID, "op range");
    havc_ctx* c = n->ctx;
    const bool check = c->range_check;
    if (check && !n->range_ready) return fail(c, HAVC_E_INVALID, "range check: enable it before the net is created (its buffers must start zero-filled)");
    hipStream_t s = c->cur ? c->cur : c->stream;
    stream_jitter(s);
    for (int i = first; i < first + count; ++i) {
        int rc = run_op(n, n->ops[i], batch);
        if (rc) return rc;
        if (check) {
            const havc_op& op = n->ops[i];
            const havc_buf& bd = n->bufdesc[op.dst];
            int e = launch_range_stats(bufptr(n, op.dst), bd.elem_bytes, (int64_t)batch * bd.elems_per_frame, n->d_range + 4 * (size_t)i, s);
            if (e) return hip_fail(c, (hipError_t)e, "range-check kernel");
        }
    }
    if (check) {
        std::vector<unsigned> host((size_t)n->ops.size() * 4);
        HIP_TRY(c, hipStreamSynchronize(s));
        HIP_TRY(c, hipMemcpy(host.data(), n->d_range, host.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemset(n->d_range, 0, host.size() * 4));
        n->range_absmax.assign(n->ops.size(), 0.f);
        n->range_bad.assign(n->ops.size(), 0);
        int first_bad = -1;
        for (size_t i = 0; i < n->ops.size(); ++i) {
            float f;
            memcpy(&f, &host[4 * i], 4);
            unsigned long long b;
            memcpy(&b, &host[4 * i + 2], 8);
            n->range_absmax[i] = f;
            n->range_bad[i] = (int64_t)b;
            if (b && first_bad < 0) first_bad = (int)i;
        }
        if (first_bad >= 0) {
            char msg[256];
            snprintf(msg, sizeof msg, "range check: op %d (type %d, tag %d) left %lld non-finite values in buffer %d -- an activation exceeded the fp16 range "
                     "(65504); largest finite magnitude there %.4g", first_bad, n->ops[first_bad].type, n->ops[first_bad].tag, (long long)n->range_bad[first_bad],
                     n->ops[first_bad].dst, (double)n->range_absmax[first_bad]);
            return fail(c, HAVC_E_RANGE, msg);
        }
    }
    return HAVC_OK;
}

int net_run_rgb8_locked(havc_net* n, const uint8_t* d_in, uint8_t* d_out, int batch, hipStream_t on = nullptr) {
    n->in_override = d_in;
    n->out_override = d_out;
    n->ctx->cur = on;
    int rc = run_ops_locked(n, 0, (int)n->ops.size(), batch);
    n->ctx->cur = nullptr;
    n->in_override = nullptr;
    n->out_override = nullptr;
    if (rc == HAVC_OK) {
        n->ctx->stats.total_flops += n->flops_per_frame * batch;
    }
    return rc;
}

struct Timer {
    havc_ctx* c;
    explicit Timer(havc_ctx* ctx) : c(ctx) { (void)hipEventRecord(c->ev0, c->stream); }
    int finish() {
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        HIP_TRY(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.last_ms = ms;
        c->stats.total_ms += ms;
        return HAVC_OK;
    }
};

// colour + blend tail at S x S for a batch already on the device.
// d_in: source frames; d_v / d_s: raw colour of the video / second model (d_s may be null); result -> d_out.
int deoldify_tail(havc_ctx* c, const uint8_t* d_in, uint8_t* d_v, uint8_t* d_s, float video_weight, int post_process,
                  uint8_t* d_out, int64_t npix) {
    int e;
    if (!d_s) {
        if (post_process) { e = launch_yuv_merge(d_v, d_in, d_out, npix, c->stream); c->stats.launches++; }
        else e = (int)hipMemcpyAsync(d_out, d_v, npix * 3, hipMemcpyDeviceToDevice, c->stream);
        if (e) return hip_fail(c, (hipError_t)e, "tail");
        return HAVC_OK;
    }
    if (post_process) {
        e = launch_yuv_merge(d_v, d_in, d_v, npix, c->stream);
        if (e) return hip_fail(c, (hipError_t)e, "yuv_merge");
        e = launch_yuv_merge(d_s, d_in, d_s, npix, c->stream);
        if (e) return hip_fail(c, (hipError_t)e, "yuv_merge");
        c->stats.launches += 2;
    }
    // Image.blend(img_second, img_video, video_weight)  (deoldify/visualize.py:129,135)
    e = launch_blend_u8(d_s, d_v, video_weight, d_out, npix * 3, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "blend");
    return HAVC_OK;
}

// Two generators per frame (video + stable/artistic).  Their encoder / bottleneck / decoder-block phases fill well under
// 256 CUs at small batch, so those phases of the two networks run CONCURRENTLY on two streams; the GPU-filling tails
// (ops from the first one tagged 1 = the 560x560 res-block convs onward) then run one after the other, each alone:
//   stream : A.small ------------\  wait(B.small) -> A.tail -> record(A.done) ............ wait(B.done)
//   stream2: wait(fork) B.small --/--------------------------- wait(A.done) -> B.tail -> record(B.done)
int run_generators(havc_ctx* c, havc_net* video, havc_net* second, const uint8_t* d_in, uint8_t* d_v, uint8_t* d_s, int b) {
    int rc;
    if (!second) return net_run_rgb8_locked(video, d_in, d_v, b);
    const int ta = video->tail_first, tb = second->tail_first;
    if (!c->two_streams || ta < 0 || tb < 0) {
        if ((rc = net_run_rgb8_locked(video, d_in, d_v, b))) return rc;
        return net_run_rgb8_locked(second, d_in, d_s, b);
    }
    auto part = [&](havc_net* n, const uint8_t* in, uint8_t* out, int first, int count, hipStream_t on) {
        n->in_override = in; n->out_override = out; c->cur = on;
        int r = run_ops_locked(n, first, count, b);
        n->in_override = nullptr; n->out_override = nullptr; c->cur = nullptr;
        return r;
    };
    // After the fork stream2 holds kernels that use the caller's scratch and the second net's buffers: on ANY failure both
    // streams are drained before the error is returned, so a later realloc / free cannot race with queued work.
    auto body = [&]() -> int {
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        if ((rc = part(video, d_in, d_v, 0, ta, nullptr))) return rc;
        if ((rc = part(second, d_in, d_s, 0, tb, c->stream2))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));                        // B.small done
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if ((rc = part(video, d_in, d_v, ta, (int)video->ops.size() - ta, nullptr))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_main_done, c->stream));                    // A.tail done
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_main_done, 0));
        if ((rc = part(second, d_in, d_s, tb, (int)second->ops.size() - tb, c->stream2))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));                        // B.tail done
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        return HAVC_OK;
    };
    if ((rc = body())) {
        const std::string keep = c->err;
        (void)sync_streams(c);
        (void)hipGetLastError();
        c->err = keep;
        return rc;
    }
    c->stats.total_flops += (video->flops_per_frame + second->flops_per_frame) * b;
    return HAVC_OK;
}

}  // namespace

extern "C" {

const char* havc_version(void) { return "havc_mi355 0.1.0 (gfx950)"; }

static void havc_destroy_unlocked(havc_ctx* c);

int havc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static void preload_device_locked(int dev) {
    static uint64_t done = 0;                              // guarded by g_setup_mu
    if (done & (1ull << (dev & 63))) return;
    preload_conv_pipe(); preload_conv_igemm(); preload_elementwise(); preload_zhang(); preload_attention(); preload_colorfilters();
    preload_tweaks(); preload_ddcolor(); preload_colormnet(); preload_colormnet_net(); preload_precise(); preload_precise2();
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(scratch_warm_kernel));
    (void)hipGetLastError();
    done |= 1ull << (dev & 63);
}

int havc_create(havc_ctx** out, int device_id) {
    if (!out) return fail(nullptr, HAVC_E_INVALID, "out is NULL");
    *out = nullptr;
    SetupLock setup;
    int n = havc_device_count();
    if (n <= 0) return fail(nullptr, HAVC_E_NODEVICE, "no HIP device visible (libhavc_mi355 needs an MI355X / gfx950 GPU)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, HAVC_E_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, HAVC_E_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, HAVC_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    havc_ctx* c = new havc_ctx();
    c->dev = device_id;
    if (const char* e = getenv("HAVC_TWO_STREAMS")) c->two_streams = atoi(e) != 0;
    if (const char* e = getenv("HAVC_NT_STORE_MB")) c->nt_store_bytes = strtoull(e, nullptr, 0) << 20;
    if (const char* e = getenv("HAVC_RANGE_CHECK")) c->range_check = atoi(e) != 0;
    if (const char* e = getenv("HAVC_DESC_LIMIT_BYTES")) {
        const unsigned long long v = strtoull(e, nullptr, 0);
        if (v >= 4096 && v <= 0xE0000000ull) c->desc_limit = v;
    }
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_main_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail(nullptr, HAVC_E_HIP, "failed to create stream/events");
    }
    static const bool eager = [] { const char* e = getenv("HAVC_EAGER_SETUP"); return e ? atoi(e) != 0 : true; }();
    if (eager) {
        preload_device_locked(device_id);
        for (hipStream_t st : {c->stream, c->stream2}) hipLaunchKernelGGL(scratch_warm_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr, 0);
        if (hipGetLastError() != hipSuccess || sync_streams(c) != hipSuccess) {
            (void)hipGetLastError();
            havc_destroy_unlocked(c);
            return fail(nullptr, HAVC_E_HIP, "scratch warm-up launch failed");
        }
    }
    *out = c;
    return HAVC_OK;
}

void havc_destroy(havc_ctx* c) {
    if (!c) return;
    SetupLock setup;
    havc_destroy_unlocked(c);
}

static void havc_destroy_unlocked(havc_ctx* c) {
    (void)hipSetDevice(c->dev);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream_h2d) {
        (void)hipStreamSynchronize(c->stream_h2d);
        (void)hipStreamSynchronize(c->stream_d2h);
        for (int k = 0; k < 2; ++k) { (void)hipEventDestroy(c->ev_up[k]); (void)hipEventDestroy(c->ev_comp[k]); (void)hipEventDestroy(c->ev_down[k]); }
        (void)hipStreamDestroy(c->stream_h2d);
        (void)hipStreamDestroy(c->stream_d2h);
    }
    for (int i = 0; i < 18; ++i)
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (auto& kv : c->resize_tables) { (void)hipFree(kv.second.d_start); (void)hipFree(kv.second.d_w); }
    for (auto& p : c->tag_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamSynchronize(c->stream2);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    (void)hipEventDestroy(c->ev_main_done);
    (void)hipEventDestroy(c->ev_side);
    (void)hipEventDestroy(c->ev_mark);
    (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* havc_last_error(const havc_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int havc_synchronize(havc_ctx* c) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, sync_streams(c));
    return HAVC_OK;
}

int havc_get_stats(havc_ctx* c, havc_stats* out) {
    if (!c || !out) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    *out = c->stats;
    return HAVC_OK;
}

int havc_reset_stats(havc_ctx* c) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    int64_t res = c->stats.bytes_resident;
    c->stats = havc_stats{};
    c->stats.bytes_resident = res;
    c->tag_used = 0;
    c->tag_ms = 0;
    c->tag_launches = 0;
    return HAVC_OK;
}

int havc_weights_load(havc_ctx* c, const void* blob, size_t nbytes, havc_weights** out) {
    if (!c || !blob || !out || nbytes == 0) return fail(c, HAVC_E_INVALID, "weights_load: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    SetupLock setup;
    HIP_TRY(c, hipSetDevice(c->dev));
    uint8_t* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, nbytes));
    hipError_t e = hipMemcpy(d, blob, nbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return hip_fail(c, e, "weights H2D"); }
    c->stats.bytes_resident += (int64_t)nbytes;
    *out = new havc_weights{c, d, nbytes};
    return HAVC_OK;
}

void havc_weights_free(havc_weights* w) {
    if (!w) return;
    std::lock_guard<std::mutex> lk(w->ctx->mu);
    SetupLock setup;
    (void)sync_streams(w->ctx);
    (void)hipFree(w->d_blob);
    w->ctx->stats.bytes_resident -= (int64_t)w->nbytes;
    delete w;
}

int havc_net_create(havc_ctx* c, havc_weights* w, const havc_op* ops, int n_ops, const havc_buf* bufs, int n_bufs,
                    int in_buf, int out_buf, int S, int max_batch, havc_net** out) {
    if (!c || !w || !ops || !bufs || !out || n_ops <= 0 || n_bufs <= 0 || max_batch < 1)
        return fail(c, HAVC_E_INVALID, "net_create: bad args");
    if (in_buf < 0 || in_buf >= n_bufs || out_buf < 0 || out_buf >= n_bufs) return fail(c, HAVC_E_INVALID, "net_create: in/out buffer id");
    std::lock_guard<std::mutex> lk(c->mu);
    SetupLock setup;
    HIP_TRY(c, hipSetDevice(c->dev));
    for (int i = 0; i < n_ops; ++i) {
        const havc_op& o = ops[i];
        auto bad = [&](int id) { return id < -1 || id >= n_bufs; };
        if (bad(o.src) || bad(o.src2) || bad(o.dst) || o.src < 0 || o.dst < 0) return fail(c, HAVC_E_INVALID, "net_create: op buffer id out of range");
        if (o.type == HAVC_OP_ATTENTION && (o.aux1 < 0 || o.aux1 >= n_bufs)) return fail(c, HAVC_E_INVALID, "net_create: attention V^T buffer id");
        if (o.type == HAVC_OP_CONV && (o.flags & (HAVC_F_RESIDUAL | HAVC_F_W_FROM_BUF)) && o.src2 < 0)
            return fail(c, HAVC_E_INVALID, "net_create: conv op with RESIDUAL / W_FROM_BUF needs a buffer in src2");
        for (int64_t off : {o.w_off, o.bias_off, o.scale_off, o.shift_off})
            if (off >= 0 && ((size_t)off >= w->nbytes || (off & 15))) return fail(c, HAVC_E_INVALID, "net_create: weight offset out of range / unaligned");
    }
    havc_net* n = new havc_net();
    n->ctx = c; n->w = w;
    n->ops.assign(ops, ops + n_ops);
    n->bufdesc.assign(bufs, bufs + n_bufs);
    n->bufs.assign(n_bufs, nullptr);
    n->in_buf = in_buf; n->out_buf = out_buf; n->S = S; n->max_batch = max_batch;
    for (int i = 0; i < n_ops; ++i) n->flops_per_frame += (double)ops[i].flops;
    for (int i = 0; i < n_ops; ++i)
        if (ops[i].tag == 1) { n->tail_first = i; break; }
    size_t ktab_bytes = 0;
    // ---- per-conv K tables: chunk kidx -> (byte offset from the tap-0 pixel, tap displacement) -----------------
    // K order (must match plan.py pack_conv): main segment = chunks [0, C8a) of every tap, then chunks [C8a, C8), each
    // segment padded to a multiple of 8 chunks.  A main segment with C8a % 8 == 0 is CHANNEL-GROUP MAJOR (group of 8
    // chunks, then tap): the 9 taps of one 128-byte line group are consecutive stages -> re-reads hit L1/L2.
    {
        std::vector<int2> host;
        n->ktab_off.assign(n_ops, -1);
        for (int i = 0; i < n_ops; ++i) {
            const havc_op& o = ops[i];
            if (o.type != HAVC_OP_CONV) continue;
            n->ktab_off[i] = (int64_t)host.size();
            const int C8 = o.Ci / 8, C8a = (o.aux1 > 0 && o.aux1 < C8) ? o.aux1 : C8;
            const size_t start = host.size();
            auto emit = [&](int kh, int kw, int c) {
                int2 e;
                e.x = ((kh * o.dil * o.Wi + kw * o.dil) * o.src_cpitch + c * 8) * 2;
                e.y = (int)(((unsigned)(kh * o.dil) & 0xffffu) | ((unsigned)(kw * o.dil) << 16));
                host.push_back(e);
            };
            auto emit_seg = [&](int c_lo, int c_hi) {
                if (c_hi <= c_lo) return;
                if (c_lo == 0 && c_hi % 8 == 0) {
                    for (int g = 0; g < c_hi / 8; ++g)
                        for (int kh = 0; kh < o.kh; ++kh)
                            for (int kw = 0; kw < o.kw; ++kw)
                                for (int c = 0; c < 8; ++c) emit(kh, kw, g * 8 + c);
                } else {
                    for (int kh = 0; kh < o.kh; ++kh)
                        for (int kw = 0; kw < o.kw; ++kw)
                            for (int c = c_lo; c < c_hi; ++c) emit(kh, kw, c);
                }
                while ((host.size() - start) % 8) host.push_back(int2{0, HAVC_KTAB_PAD_DH});
            };
            emit_seg(0, C8a);
            emit_seg(C8a, C8);
            if (o.flags & HAVC_F_PRECISE) {
                // precise conv: the K walk above three times -- x_hi (x 2^11 w_hi), x_hi (x 2^11 w_lo), x_lo (x w_hi; the lo plane of a pixel
                // sits src_cpitch / 2 elements = src_cpitch BYTES behind its hi plane) -- matching plan.py pack_conv(precise=True)
                const size_t len = host.size() - start;
                for (int rep = 1; rep < 3; ++rep)
                    for (size_t k = 0; k < len; ++k) {
                        int2 e = host[start + k];
                        if (rep == 2 && (short)(e.y & 0xffff) != HAVC_KTAB_PAD_DH) e.x += o.src_cpitch;
                        host.push_back(e);
                    }
            }
            if ((int)(host.size() - start) != o.Kc) {
                delete n;
                return fail(c, HAVC_E_INVALID, "net_create: conv op Kc does not match its K layout (Ci, kh, kw, aux1)");
            }
        }
        if (!host.empty()) {
            hipError_t e = hipMalloc((void**)&n->d_ktab, host.size() * sizeof(int2));
            if (e == hipSuccess) e = hipMemcpy(n->d_ktab, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (n->d_ktab) (void)hipFree(n->d_ktab);
                delete n;
                return hip_fail(c, e, "K table upload");
            }
            ktab_bytes = host.size() * sizeof(int2);
            c->stats.bytes_resident += (int64_t)ktab_bytes;
        }
    }
    for (int i = 0; i < n_bufs; ++i) {
        // +256 B tail so a predicated-off 16-byte vector address is never formed past the allocation
        size_t nb = (size_t)bufs[i].elems_per_frame * bufs[i].elem_bytes * max_batch + 256;
        hipError_t e = hipMalloc(&n->bufs[i], nb);
        if (e == hipSuccess && (bufs[i].zero_init || c->range_check)) e = hipMemset(n->bufs[i], 0, nb);
        if (e != hipSuccess) {
            for (int k = 0; k <= i; ++k) {
                if (!n->bufs[k]) continue;
                (void)hipFree(n->bufs[k]);
                if (k < i) c->stats.bytes_resident -= (int64_t)((size_t)bufs[k].elems_per_frame * bufs[k].elem_bytes * max_batch + 256);
            }
            if (n->d_ktab) {
                (void)hipFree(n->d_ktab);
                c->stats.bytes_resident -= (int64_t)ktab_bytes;
            }
            delete n;
            return hip_fail(c, e, "activation buffer allocation");
        }
        c->stats.bytes_resident += (int64_t)nb;
    }
    if (c->range_check) {
        if (hipMalloc((void**)&n->d_range, (size_t)n_ops * 16) != hipSuccess || hipMemset(n->d_range, 0, (size_t)n_ops * 16) != hipSuccess)
            return fail(c, HAVC_E_HIP, "range-check buffer allocation");
        n->range_ready = true;
    }
    *out = n;
    return HAVC_OK;
}

void havc_net_free(havc_net* n) {
    if (!n) return;
    std::lock_guard<std::mutex> lk(n->ctx->mu);
    SetupLock setup;
    (void)sync_streams(n->ctx);
    if (n->d_ktab) {
        (void)hipFree(n->d_ktab);
        size_t chunks = 0;
        for (const havc_op& o : n->ops) if (o.type == HAVC_OP_CONV) chunks += (size_t)o.Kc;
        n->ctx->stats.bytes_resident -= (int64_t)(chunks * sizeof(int2));
    }
    if (n->d_range) (void)hipFree(n->d_range);
    for (size_t i = 0; i < n->bufs.size(); ++i) {
        if (n->bufs[i]) (void)hipFree(n->bufs[i]);
        n->ctx->stats.bytes_resident -= (int64_t)((size_t)n->bufdesc[i].elems_per_frame * n->bufdesc[i].elem_bytes * n->max_batch + 256);
    }
    delete n;
}

int havc_net_run_rgb8(havc_net* n, const uint8_t* d_in, uint8_t* d_out, int batch) {
    if (!n || !d_in || !d_out) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    Timer t(c);
    int rc = net_run_rgb8_locked(n, d_in, d_out, batch);
    if (rc) return rc;
    c->stats.frames += batch;
    return t.finish();
}

int havc_net_upload(havc_net* n, int buf, const void* host, size_t nbytes) {
    if (!n || buf < 0 || buf >= (int)n->bufs.size() || !host) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (nbytes > (size_t)n->bufdesc[buf].elems_per_frame * n->bufdesc[buf].elem_bytes * n->max_batch) return fail(c, HAVC_E_INVALID, "upload too large");
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(n->bufs[buf], host, nbytes, hipMemcpyHostToDevice));
    return HAVC_OK;
}

int havc_net_download(havc_net* n, int buf, void* host, size_t nbytes) {
    if (!n || buf < 0 || buf >= (int)n->bufs.size() || !host) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (nbytes > (size_t)n->bufdesc[buf].elems_per_frame * n->bufdesc[buf].elem_bytes * n->max_batch) return fail(c, HAVC_E_INVALID, "download too large");
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(host, n->bufs[buf], nbytes, hipMemcpyDeviceToHost));
    return HAVC_OK;
}

int havc_range_check_enable(havc_ctx* c, int enable) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    c->range_check = enable != 0;
    return HAVC_OK;
}

int havc_net_range_stats(havc_net* n, float* abs_max, int64_t* non_finite, int n_ops) {
    if (!n || !abs_max || !non_finite || n_ops != (int)n->ops.size()) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(n->ctx->mu);
    if (n->range_absmax.size() != n->ops.size()) return fail(n->ctx, HAVC_E_INVALID, "range stats: no checked run of this net yet");
    for (int i = 0; i < n_ops; ++i) { abs_max[i] = n->range_absmax[i]; non_finite[i] = n->range_bad[i]; }
    return HAVC_OK;
}

int havc_net_bind(havc_net* n, int buf, void* device_ptr) {
    if (!n || buf < 0 || buf >= (int)n->bufs.size()) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(n->ctx->mu);
    if (n->bound.empty()) n->bound.assign(n->bufs.size(), nullptr);
    n->bound[buf] = device_ptr;
    return HAVC_OK;
}

int havc_net_enqueue_ops(havc_net* n, int first_op, int n_ops, int batch) {
    if (!n) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    return run_ops_locked(n, first_op, n_ops, batch);
}

void* havc_get_stream(havc_ctx* c) {
    if (!c) return nullptr;
    std::lock_guard<std::mutex> lk(c->mu);
    c->stream_exported = true;            // (a wrapped torch ExternalStream would dangle if the stream were re-created: ADVICE r5)
    return (void*)c->stream;
}

int havc_net_run_ops(havc_net* n, int first_op, int n_ops, int batch) {
    if (!n) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    Timer t(c);
    int rc = run_ops_locked(n, first_op, n_ops, batch);
    if (rc) return rc;
    return t.finish();
}

// Per-op choice of the conv tile configuration by measurement.  Every candidate computes the same bytes (same packed K order,
// same MFMA sequence per output), so tuning never changes a result; ops with the same shape signature are tuned once.
// tile configurations an op may run with (all produce the same bytes); empty = the op has exactly one legal configuration
static std::vector<int> tune_candidates(const havc_op& op) {
    if (op.type != HAVC_OP_CONV || (op.flags & (HAVC_F_PS_BLUR | HAVC_F_FUSE_RGB8 | HAVC_F_OUT_RGB8 | HAVC_F_FUSE_PROJ))) return {};
    if (HAVC_F_SPLITK_COUNT(op.flags) > 1) {                               // split-K: the plain pipelined tiles (same bytes for a given count)
        std::vector<int> sk = {0};
        if (op.Npad % 256 == 0 || op.Npad % 256 >= 192) for (int k : {60, 71, 90, 91, 96, 97}) sk.push_back(k);
        for (int k : {70, 72, 93, 95, 98}) sk.push_back(k);
        if (op.Npad <= 192) { sk.push_back(99); sk.push_back(92); }
        return sk;
    }
    std::vector<int> cand = {0};
    if (op.Npad <= 16) return cand;                                        // thin N: the 128x16 kernel only
    if (op.flags & HAVC_F_PRECISE) {
        // precise convs: the PIPELINED tile geometries instantiated with the precise epilogue, and nothing else: nothing is rounded to fp16
        // before the hi / lo split, so the bytes follow the fp32 accumulation order -- the pipelined tiles share one K walk, the register-staged
        // kernels (cfg 1, 2, 3, 7) and whatever the heuristic (cfg 0) picks per batch size do not.  The heuristic stays only where no pipelined tile fits.
        std::vector<int> pc;
        if (op.Npad % 256 == 0 || op.Npad % 256 >= 192) { for (int k : {60, 71, 96}) pc.push_back(k); }
        if (op.Npad == 272) pc.push_back(61);
        if (op.Npad % 128 == 0 || op.Npad % 128 >= 96 || op.Npad > 256) { for (int k : {70, 72, 98}) pc.push_back(k); }
        if (op.Npad % 64 == 0 && op.Npad <= 192) { pc.push_back(99); pc.push_back(92); }
        if (pc.empty()) { for (int k : {72, 92}) pc.push_back(k); }
        return pc;
    }
    // column tiles of 256 / 128 channels: also when the last tile is >= 75 % full (ConvNeXt pwconv2 at stage 0, 768 -> 192: the
    // 256 x 256 tile beats every narrower one by 30 %)
    if (op.Npad % 256 == 0 || op.Npad % 256 >= 192) { for (int k : {60, 71, 90, 91, 96, 97}) cand.push_back(k); }
    if (op.Npad == 272) cand.push_back(61);                                // (launch_pipe<.., EXTRA = 1> takes exactly 256 + 16 columns)
    // 128-wide tiles for every wide layer: the DynamicUnetDeep (artistic) channel counts 304 / 320 / 672 / 1344 fit no tile exactly and
    // a partly empty last tile on the pipelined kernel still beats the register-staged kernels there
    if (op.Npad % 128 == 0 || op.Npad % 128 >= 96 || op.Npad > 256) { for (int k : {70, 72, 93, 95, 98}) cand.push_back(k); }
    if (op.Npad % 64 == 0 && op.Npad <= 192) { cand.push_back(99); cand.push_back(92); }      // 64-wide pipelined tiles (ResNet layer1, stem)
    for (int k : {1, 2, 3, 7}) cand.push_back(k);                          // register-staged 128x128 / 128x64 / 64x64 / 64x128
    return cand;
}

int havc_net_set_cfg(havc_net* n, int op_index, int cfg) {
    if (!n || op_index < 0 || op_index >= (int)n->ops.size()) return HAVC_E_INVALID;
    const std::vector<int> cand = tune_candidates(n->ops[op_index]);
    if (std::find(cand.begin(), cand.end(), cfg) == cand.end())
        return cfg == 0 ? HAVC_OK : fail(n->ctx, HAVC_E_INVALID, "net_set_cfg: not a legal tile configuration for this op");
    std::lock_guard<std::mutex> lk(n->ctx->mu);
    n->ops[op_index].reserved = cfg;
    return HAVC_OK;
}

int havc_net_autotune(havc_net* n, int batch, int* n_changed) {
    if (!n) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    // The process-wide set-up mutex is taken PER TRIAL (round 5): a trial launch is the first launch of most kernel instantiations, which is
    // what the mutex serialises -- but held over the whole tuning run (seconds) it stalled every other thread's havc_dev_alloc / scratch regrowth.
    HIP_TRY(c, hipSetDevice(c->dev));
    if (batch < 1 || batch > n->max_batch) return fail(c, HAVC_E_INVALID, "autotune: batch out of range");
    HIP_TRY(c, sync_streams(c));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    std::map<std::vector<int64_t>, int> seen;
    int changed = 0, rc = HAVC_OK;
    const havc_stats keep = c->stats;
    for (size_t i = 0; i < n->ops.size() && rc == HAVC_OK; ++i) {
        havc_op& op = n->ops[i];
        if (tune_candidates(op).empty()) continue;                          // not a conv, or one legal config
        const std::vector<int64_t> sig = {op.flags, op.Hi, op.Wi, op.Ci, op.Ho, op.Wo, op.Co, op.kh, op.kw, op.stride, op.pad, op.dil, op.Kc,
                                          op.Npad, op.src_cpitch, op.dst_cpitch, op.res_cpitch, op.aux1, op.out_step};
        auto it = seen.find(sig);
        if (it != seen.end()) { if (op.reserved != it->second) { op.reserved = it->second; ++changed; } continue; }
        const std::vector<int> cand = tune_candidates(op);
        const int before = op.reserved;
        int best = before;
        float best_ms = 1e30f;
        for (int cfg : cand) {
            SetupLock setup;
            op.reserved = cfg;
            float tot = 0.f;
            bool ok = true;
            for (int rep = 0; rep < 4 && ok; ++rep) {                      // rep 0 = warm-up
                if (hipEventRecord(e0, c->stream) != hipSuccess) { ok = false; break; }
                if (run_op(n, op, batch) != HAVC_OK) { ok = false; (void)hipGetLastError(); break; }
                if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { ok = false; break; }
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { ok = false; break; }
                if (rep) tot += ms;
            }
            if (ok && (best_ms > 1e29f || tot < best_ms * 0.98f)) { best_ms = tot; best = cfg; }   // 2 % hysteresis towards earlier candidates
        }
        op.reserved = best;
        seen[sig] = best;
        if (best != before) ++changed;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->stats = keep;
    c->err.clear();
    if (n_changed) *n_changed = changed;
    return rc;
}

int havc_device_name(havc_ctx* c, char* buf, int nbuf) {
    if (!c || !buf || nbuf < 2) return HAVC_E_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->dev) != hipSuccess) return fail(c, HAVC_E_HIP, "hipGetDeviceProperties failed");
    snprintf(buf, (size_t)nbuf, "%s/%s/%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return HAVC_OK;
}

int havc_net_get_cfg(havc_net* n, int op_index) {
    if (!n || op_index < 0 || op_index >= (int)n->ops.size()) return HAVC_E_INVALID;
    return n->ops[op_index].reserved;
}

int havc_net_profile(havc_net* n, int batch, float* ms_per_op, int n_ops) {
    if (!n || !ms_per_op || n_ops != (int)n->ops.size()) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    if (batch < 1 || batch > n->max_batch) return fail(c, HAVC_E_INVALID, "batch out of range");
    std::vector<hipEvent_t> ev(n_ops + 1);
    for (auto& e : ev) HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipEventRecord(ev[0], c->stream));
    int rc = HAVC_OK;
    for (int i = 0; i < n_ops && rc == HAVC_OK; ++i) {
        rc = run_op(n, n->ops[i], batch);
        if (rc == HAVC_OK && hipEventRecord(ev[i + 1], c->stream) != hipSuccess) rc = HAVC_E_HIP;
    }
    if (rc == HAVC_OK) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = HAVC_E_HIP;
        for (int i = 0; i < n_ops && rc == HAVC_OK; ++i)
            if (hipEventElapsedTime(&ms_per_op[i], ev[i], ev[i + 1]) != hipSuccess) rc = HAVC_E_HIP;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc == HAVC_E_HIP) return fail(c, rc, "profile failed");
    return rc;
}

// ---- pointer-agnostic operands ---------------------------------------------------------------------------------------
// Every frame / filter entry point accepts HOST or DEVICE pointers for its image operands (unified addressing tells them apart).
// Host operands are staged through the ctx scratch buffers and the call blocks until the result is back in host memory; device
// operands (havc_dev_alloc, or any hipMalloc of this device) are used in place, nothing is copied and the call only ENQUEUES
// work on the ctx stream (havc_synchronize / a later host-output call / havc_dev_download order against it).  This is what
// lets a whole HAVC merge graph run without leaving HBM (vsdeoldify_amd/device.py).
static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

static int stage_in(havc_ctx* c, int slot, const void* p, size_t nbytes, const uint8_t** d) {
    if (is_device_ptr(p)) { *d = (const uint8_t*)p; return HAVC_OK; }
    int rc = ensure_scratch(c, slot, nbytes);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->scratch[slot], p, nbytes, hipMemcpyHostToDevice, c->stream));
    *d = (const uint8_t*)c->scratch[slot];
    return HAVC_OK;
}

static int stage_out_ptr(havc_ctx* c, int slot, void* p, size_t nbytes, uint8_t** d, bool* host) {
    *host = !is_device_ptr(p);
    if (!*host) { *d = (uint8_t*)p; return HAVC_OK; }
    int rc = ensure_scratch(c, slot, nbytes);
    if (rc) return rc;
    *d = (uint8_t*)c->scratch[slot];
    return HAVC_OK;
}

static int stage_out(havc_ctx* c, void* p, const uint8_t* d, size_t nbytes, bool host) {
    if (!host) return HAVC_OK;
    HIP_TRY(c, hipMemcpyAsync(p, d, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HAVC_OK;
}

// two-input (b may be NULL) / one-output per-pixel filter: stage, launch, hand back
extern "C++" {
template <typename Launch, typename Pre>
static int run_filter(havc_ctx* c, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t nbytes, const char* what, Launch launch, Pre pre) {
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const uint8_t *da = nullptr, *db = nullptr;
    uint8_t* dout = nullptr;
    bool host = false;
    int rc;
    if ((rc = pre())) return rc;
    if ((rc = stage_in(c, 0, a, nbytes, &da))) return rc;
    if (b && (rc = stage_in(c, 1, b, nbytes, &db))) return rc;
    if ((rc = stage_out_ptr(c, 2, out, nbytes, &dout, &host))) return rc;
    const int e = launch(da, db, dout);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, what);
    return stage_out(c, out, dout, nbytes, host);
}
template <typename Launch>
static int run_filter(havc_ctx* c, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t nbytes, const char* what, Launch launch) {
    return run_filter(c, a, b, out, nbytes, what, launch, []() { return HAVC_OK; });
}
}  // extern "C++"

int havc_deoldify_frames(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, int post_process,
                         const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames) {
    if (!c || !video || !rgb_in || !rgb_out || n_frames < 0) return fail(c, HAVC_E_INVALID, "deoldify_frames: bad args");
    if (video->ctx != c || (second && (second->ctx != c || second->S != video->S))) return fail(c, HAVC_E_INVALID, "deoldify_frames: nets from another ctx / size");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int S = video->S;
    const int64_t npix1 = (int64_t)S * S;
    int maxb = video->max_batch;
    if (second) maxb = std::min(maxb, second->max_batch);
    const size_t fb = (size_t)npix1 * 3;
    const bool in_dev = is_device_ptr(rgb_in), out_dev = is_device_ptr(rgb_out);
    int rc;
    for (int slot = 0; slot < 4; ++slot) {
        if ((slot == 0 && in_dev) || (slot == 3 && out_dev)) continue;
        if ((rc = ensure_scratch(c, slot, fb * maxb))) return rc;
    }
    uint8_t *d_v = (uint8_t*)c->scratch[1], *d_s = (uint8_t*)c->scratch[2];
    Timer t(c);
    for (int f0 = 0; f0 < n_frames; f0 += maxb) {
        const int b = std::min(maxb, n_frames - f0);
        const uint8_t* d_in = in_dev ? rgb_in + (size_t)f0 * fb : (const uint8_t*)c->scratch[0];
        uint8_t* d_out = out_dev ? rgb_out + (size_t)f0 * fb : (uint8_t*)c->scratch[3];
        if (!in_dev) HIP_TRY(c, hipMemcpyAsync(c->scratch[0], rgb_in + (size_t)f0 * fb, fb * b, hipMemcpyHostToDevice, c->stream));
        if ((rc = run_generators(c, video, second, d_in, d_v, d_s, b))) return rc;
        if ((rc = deoldify_tail(c, d_in, d_v, second ? d_s : nullptr, video_weight, post_process, d_out, npix1 * b))) return rc;
        if (!out_dev) {
            HIP_TRY(c, hipMemcpyAsync(rgb_out + (size_t)f0 * fb, d_out, fb * b, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    c->stats.frames += n_frames;
    return out_dev ? HAVC_OK : t.finish();
}

// ---- frame coalescer: the reference calls get_transformed_image ONCE PER FRAME from several VapourSynth worker threads (vsmodels.py:201-230,
// one call per std.ModifyFrame selector).  A batch of one leaves the encoder at 5 blocks on 256 CUs; here concurrent callers are
// merged: the first caller to arrive leads, waits up to wait_us (or until `callers` requests are queued), runs ONE havc_deoldify_frames
// over everything queued and hands every caller its frame.  Bytes are those of a call of its own (all tile configurations and batch sizes
// produce the same result).  ----
struct havc_batcher {
    havc_ctx* ctx = nullptr;
    int kind = 0;                                              // 0 DeOldify (video [+ second]), 1 DDColor, 2 Zhang: `video` is the net
    int width = 0, height = 0;                                 // frame size (kinds 1, 2; kind 0: S x S)
    havc_net *video = nullptr, *second = nullptr;
    float video_weight = 0.f;
    int post_process = 1, S = 0, max_batch = 1, wait_us = 200, callers = 0;
    size_t fb = 0;
    struct Req { const uint8_t* in; uint8_t* out; int rc; bool done; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Req*> q;
    bool leader = false;
    int inflight = 0;                                          // callers inside havc_batcher_submit (free waits for them to LEAVE, not only for an empty queue)
    uint8_t *h_in = nullptr, *h_out = nullptr;                 // pinned [max_batch][S * S * 3]
    int64_t calls = 0, batches = 0;
};

int havc_batcher_create(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, int post_process, int wait_us, int callers,
                        havc_batcher** out) {
    if (!c || !video || !out || video->ctx != c || (second && (second->ctx != c || second->S != video->S)))
        return fail(c, HAVC_E_INVALID, "batcher_create: bad args / nets of another ctx or size");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    auto* b = new havc_batcher();
    b->ctx = c; b->video = video; b->second = second; b->video_weight = video_weight; b->post_process = post_process;
    b->S = video->S;
    b->max_batch = second ? std::min(video->max_batch, second->max_batch) : video->max_batch;
    b->wait_us = wait_us < 0 ? 0 : wait_us;
    b->callers = callers;
    b->fb = (size_t)b->S * b->S * 3;
    if (hipHostMalloc((void**)&b->h_in, b->fb * b->max_batch, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&b->h_out, b->fb * b->max_batch, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (b->h_in) (void)hipHostFree(b->h_in);
        delete b;
        return fail(c, HAVC_E_OOM, "batcher_create: pinned staging");
    }
    *out = b;
    return HAVC_OK;
}

int havc_batcher_create_frames(havc_ctx* c, int kind, havc_net* net, int width, int height, int wait_us, int callers, havc_batcher** out) {
    if (!c || !net || !out || net->ctx != c || (kind != 1 && kind != 2) || width <= 0 || height <= 0)
        return fail(c, HAVC_E_INVALID, "batcher_create_frames: kind 1 (DDColor) or 2 (Zhang), a net of this ctx, a frame size");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    auto* b = new havc_batcher();
    b->ctx = c; b->kind = kind; b->video = net; b->width = width; b->height = height; b->S = net->S;
    b->max_batch = net->max_batch;
    b->wait_us = wait_us < 0 ? 0 : wait_us;
    b->callers = callers;
    b->fb = (size_t)width * height * 3;
    if (hipHostMalloc((void**)&b->h_in, b->fb * b->max_batch, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&b->h_out, b->fb * b->max_batch, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (b->h_in) (void)hipHostFree(b->h_in);
        delete b;
        return fail(c, HAVC_E_OOM, "batcher_create_frames: pinned staging");
    }
    *out = b;
    return HAVC_OK;
}

void havc_batcher_free(havc_batcher* b) {
    if (!b) return;
    {
        std::unique_lock<std::mutex> lk(b->m);
        // a follower woken by the leader still has to re-acquire b->m before it returns: wait until every submitter has left
        b->cv.wait(lk, [&] { return !b->leader && b->q.empty() && b->inflight == 0; });
    }
    (void)hipHostFree(b->h_in);
    (void)hipHostFree(b->h_out);
    delete b;
}

int havc_batcher_stats(havc_batcher* b, int64_t* calls, int64_t* batches) {
    if (!b) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(b->m);
    if (calls) *calls = b->calls;
    if (batches) *batches = b->batches;
    return HAVC_OK;
}

int havc_batcher_submit(havc_batcher* b, const uint8_t* rgb_in, uint8_t* rgb_out) {
    if (!b || !rgb_in || !rgb_out) return HAVC_E_INVALID;
    havc_batcher::Req r{rgb_in, rgb_out, HAVC_OK, false};
    std::unique_lock<std::mutex> lk(b->m);
    ++b->inflight;
    b->q.push_back(&r);
    ++b->calls;
    b->cv.notify_all();                                        // a leader collecting its batch re-checks the queue length
    while (!r.done) {
        if (b->leader) { b->cv.wait(lk); continue; }
        b->leader = true;
        const int want = b->callers > 0 ? std::min(b->callers, b->max_batch) : b->max_batch;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(b->wait_us);
        while ((int)b->q.size() < want && b->cv.wait_until(lk, deadline) != std::cv_status::timeout) {}
        std::vector<havc_batcher::Req*> batch;
        while (!b->q.empty() && (int)batch.size() < b->max_batch) { batch.push_back(b->q.front()); b->q.pop_front(); }
        lk.unlock();
        const int n = (int)batch.size();
        for (int i = 0; i < n; ++i) memcpy(b->h_in + (size_t)i * b->fb, batch[i]->in, b->fb);
        const int rc = b->kind == 0   ? havc_deoldify_frames(b->ctx, b->video, b->second, b->video_weight, b->post_process, b->h_in, b->h_out, n)
                       : b->kind == 1 ? havc_ddcolor_frames(b->ctx, b->video, b->h_in, b->h_out, n, b->width, b->height)
                                      : havc_zhang_frames(b->ctx, b->video, b->h_in, b->h_out, n, b->width, b->height);
        if (rc == HAVC_OK)
            for (int i = 0; i < n; ++i) memcpy(batch[i]->out, b->h_out + (size_t)i * b->fb, b->fb);
        lk.lock();
        for (auto* q : batch) { q->rc = rc; q->done = true; }
        ++b->batches;
        b->leader = false;
        b->cv.notify_all();
    }
    if (--b->inflight == 0) b->cv.notify_all();                // (still under b->m) havc_batcher_free may be waiting for the last caller
    return r.rc;
}

int havc_pil_resize(havc_ctx* c, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int resample) {
    if (!c || !src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || (resample != 2 && resample != 3))
        return fail(c, HAVC_E_INVALID, "pil_resize: bad args (resample must be 2 = BILINEAR or 3 = BICUBIC)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    int rc;
    const size_t sb = (size_t)sw * sh * 3, tb = (size_t)sh * dw * 3, db = (size_t)dw * dh * 3;
    const uint8_t* d_src;
    uint8_t* d_dst;
    bool host;
    if ((rc = stage_in(c, 0, src, sb, &d_src)) || (rc = ensure_scratch(c, 1, tb)) || (rc = stage_out_ptr(c, 2, dst, db, &d_dst, &host))) return rc;
    if ((rc = pil_resize_dev(c, d_src, sw, sh, (uint8_t*)c->scratch[1], d_dst, dw, dh, 1, resample))) return rc;
    return stage_out(c, dst, d_dst, db, host);
}

int havc_zhang_frames(havc_ctx* c, havc_net* net, const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames, int width, int height) {
    if (!c || !net || !rgb_in || !rgb_out || n_frames < 0 || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "zhang_frames: bad args");
    if (net->ctx != c || net->bufdesc[net->out_buf].elem_bytes != 4) return fail(c, HAVC_E_INVALID, "zhang_frames: not a Zhang net of this ctx");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int S = net->S, maxb = net->max_batch;
    const size_t fb = (size_t)width * height * 3, sq = (size_t)S * S * 3;
    const bool in_dev = is_device_ptr(rgb_in), out_dev = is_device_ptr(rgb_out);
    int rc;
    if ((!in_dev && (rc = ensure_scratch(c, 0, fb * maxb))) || (rc = ensure_scratch(c, 1, (size_t)height * S * 3 * maxb)) ||
        (rc = ensure_scratch(c, 2, sq * maxb)) || (!out_dev && (rc = ensure_scratch(c, 3, fb * maxb)))) return rc;
    uint8_t *d_tmp = (uint8_t*)c->scratch[1], *d_sq = (uint8_t*)c->scratch[2];
    Timer t(c);
    for (int f0 = 0; f0 < n_frames; f0 += maxb) {
        const int b = std::min(maxb, n_frames - f0);
        const uint8_t* d_in = in_dev ? rgb_in + (size_t)f0 * fb : (const uint8_t*)c->scratch[0];
        uint8_t* d_out = out_dev ? rgb_out + (size_t)f0 * fb : (uint8_t*)c->scratch[3];
        if (!in_dev) HIP_TRY(c, hipMemcpyAsync(c->scratch[0], rgb_in + (size_t)f0 * fb, fb * b, hipMemcpyHostToDevice, c->stream));
        if ((rc = pil_resize_dev(c, d_in, width, height, d_tmp, d_sq, S, S, b, 3))) return rc;                  // PIL BICUBIC -> 256x256
        net->in_override = d_sq;
        rc = run_ops_locked(net, 0, (int)net->ops.size(), b);
        net->in_override = nullptr;
        if (rc) return rc;
        c->stats.total_flops += net->flops_per_frame * b;
        int e = launch_zhang_post(d_in, (const float*)net->bufs[net->out_buf], S, S, d_out, b, width, height, c->stream);
        c->stats.launches++;
        if (e) return hip_fail(c, (hipError_t)e, "zhang post");
        if (!out_dev) {
            HIP_TRY(c, hipMemcpyAsync(rgb_out + (size_t)f0 * fb, d_out, fb * b, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    c->stats.frames += n_frames;
    return out_dev ? HAVC_OK : t.finish();
}

// shared body of the DDColor entry points.  out_planes != NULL: float / half planar output (the RGBS / RGBH shape of
// vsddcolor.ddcolor) instead of interleaved u8.
static int ddcolor_batch_locked(havc_ctx* c, havc_net* net, const uint8_t* d_in, uint8_t* d_out_u8, void* d_out_planes, int planes_half,
                                int b, int width, int height) {
    const int S = net->S;
    const int ab_pitch = (int)(net->bufdesc[net->out_buf].elems_per_frame / ((size_t)S * S));
    const bool squash = width != S || height != S;
    const size_t sq = (size_t)S * S * 3;
    int rc;
    const uint8_t* d_net_in = d_in;
    if (squash) {                                           // frame != input_size: Pillow BILINEAR to S x S (build's choice)
        if ((rc = ensure_scratch(c, 1, (size_t)height * S * 3 * b)) || (rc = ensure_scratch(c, 2, sq * b))) return rc;
        d_net_in = (uint8_t*)c->scratch[2];
        if ((rc = pil_resize_dev(c, d_in, width, height, (uint8_t*)c->scratch[1], (uint8_t*)c->scratch[2], S, S, b, 2))) return rc;
    }
    net->in_override = d_net_in;
    rc = run_ops_locked(net, 0, (int)net->ops.size(), b);
    net->in_override = nullptr;
    if (rc) return rc;
    c->stats.total_flops += net->flops_per_frame * b;
    const bool precise = !net->ops.empty() && (net->ops[0].flags & HAVC_F_PRECISE);          // precise nets: the ab map is a hi / lo pair tensor
    int e = launch_ddcolor_post(d_in, (const half_t*)net->bufs[net->out_buf], ab_pitch, 0, S, S, d_out_u8, d_out_planes, planes_half, b, width, height, c->stream,
                                precise ? ab_pitch >> 1 : 0);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "ddcolor post");
    return HAVC_OK;
}

int havc_ddcolor_frames(havc_ctx* c, havc_net* net, const uint8_t* rgb_in, uint8_t* rgb_out, int n_frames, int width, int height) {
    if (!c || !net || !rgb_in || !rgb_out || n_frames < 0 || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "ddcolor_frames: bad args");
    if (net->ctx != c || net->bufdesc[net->out_buf].elem_bytes != 2) return fail(c, HAVC_E_INVALID, "ddcolor_frames: not a DDColor net of this ctx");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int maxb = net->max_batch;
    const size_t fb = (size_t)width * height * 3;
    const bool in_dev = is_device_ptr(rgb_in), out_dev = is_device_ptr(rgb_out);
    int rc;
    if ((!in_dev && (rc = ensure_scratch(c, 0, fb * maxb))) || (!out_dev && (rc = ensure_scratch(c, 3, fb * maxb)))) return rc;
    Timer t(c);
    for (int f0 = 0; f0 < n_frames; f0 += maxb) {
        const int b = std::min(maxb, n_frames - f0);
        const uint8_t* d_in = in_dev ? rgb_in + (size_t)f0 * fb : (const uint8_t*)c->scratch[0];
        uint8_t* d_out = out_dev ? rgb_out + (size_t)f0 * fb : (uint8_t*)c->scratch[3];
        if (!in_dev) HIP_TRY(c, hipMemcpyAsync(c->scratch[0], rgb_in + (size_t)f0 * fb, fb * b, hipMemcpyHostToDevice, c->stream));
        if ((rc = ddcolor_batch_locked(c, net, d_in, d_out, nullptr, 0, b, width, height))) return rc;
        if (!out_dev) {
            HIP_TRY(c, hipMemcpyAsync(rgb_out + (size_t)f0 * fb, d_out, fb * b, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    c->stats.frames += n_frames;
    return out_dev ? HAVC_OK : t.finish();
}

int havc_ddcolor_frame_planar_f(havc_ctx* c, havc_net* net, const void* const in_planes[3], int in_stride_bytes, void* const out_planes[3],
                                int out_stride_bytes, int is_half, int width, int height) {
    if (!c || !net || !in_planes || !out_planes || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "ddcolor_frame_planar_f: bad args");
    for (int p = 0; p < 3; ++p) if (!in_planes[p] || !out_planes[p]) return fail(c, HAVC_E_INVALID, "ddcolor_frame_planar_f: NULL plane");
    if (net->ctx != c || net->bufdesc[net->out_buf].elem_bytes != 2) return fail(c, HAVC_E_INVALID, "ddcolor_frame_planar_f: not a DDColor net of this ctx");
    const size_t esz = is_half ? 2 : 4, row = (size_t)width * esz;
    if ((size_t)in_stride_bytes < row || (size_t)out_stride_bytes < row) return fail(c, HAVC_E_INVALID, "ddcolor_frame_planar_f: stride smaller than a row");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t plane = row * height, fb = (size_t)width * height * 3;
    int rc;
    if ((rc = ensure_scratch(c, 0, fb)) || (rc = ensure_scratch(c, 4, 3 * plane)) || (rc = ensure_scratch(c, 5, 3 * plane))) return rc;
    uint8_t* d_planes_in = (uint8_t*)c->scratch[4];
    uint8_t* d_planes_out = (uint8_t*)c->scratch[5];
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync(d_planes_in + p * plane, row, in_planes[p], (size_t)in_stride_bytes, row, (size_t)height, hipMemcpyDefault, c->stream));
    // RGBH / RGBS full range [0, 1] -> the u8 frame the float clip was cast from (vsmodels.py:354,358): round(x * 255)
    int e = launch_planar_f_to_rgb8(d_planes_in, is_half, (uint8_t*)c->scratch[0], (int64_t)width * height, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "planar float -> rgb8");
    if ((rc = ddcolor_batch_locked(c, net, (const uint8_t*)c->scratch[0], nullptr, d_planes_out, is_half, 1, width, height))) return rc;
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync(out_planes[p], (size_t)out_stride_bytes, d_planes_out + p * plane, row, row, (size_t)height, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stats.frames += 1;
    return HAVC_OK;
}

int havc_blend(havc_ctx* c, const uint8_t* a, const uint8_t* b, float w, uint8_t* out, int width, int height) {
    if (!c || !a || !b || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "blend: bad args");
    const size_t nb = (size_t)width * height * 3;
    return run_filter(c, a, b, out, nb, "blend", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_blend_u8(da, db, w, dout, (int64_t)nb, c->stream); });
}

int havc_chroma_post_process(havc_ctx* c, const uint8_t* color, const uint8_t* orig, uint8_t* out, int width, int height) {
    if (!c || !color || !orig || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "chroma_post_process: bad args");
    return run_filter(c, color, orig, out, (size_t)width * height * 3, "chroma_post_process", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_yuv_merge(da, db, dout, (int64_t)width * height, c->stream); });
}

int havc_chroma_stabilizer(havc_ctx* c, const uint8_t* img_stable, const uint8_t* img_new, double alpha, double weight,
                           uint8_t* out, int width, int height) {
    if (!c || !img_stable || !img_new || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "chroma_stabilizer: bad args");
    return run_filter(c, img_stable, img_new, out, (size_t)width * height * 3, "chroma_stabilizer", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_chroma_stabilizer(da, db, alpha, (float)weight, dout, (int64_t)width * height, c->stream); });
}

int havc_chroma_stabilizer_adaptive(havc_ctx* c, const uint8_t* img_stable, const uint8_t* img_new, double base_tol, double max_extra,
                                    double weight, uint8_t* out, int width, int height) {
    if (!c || !img_stable || !img_new || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "chroma_stabilizer_adaptive: bad args");
    if (out == img_stable) return fail(c, HAVC_E_INVALID, "chroma_stabilizer_adaptive: out must not alias img_stable (Laplacian neighbourhood)");
    return run_filter(c, img_stable, img_new, out, (size_t)width * height * 3, "chroma_stabilizer_adaptive", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_chroma_stabilizer_adaptive(da, db, (float)base_tol, (float)max_extra, (float)weight, dout, width, height, c->stream); });
}

int havc_chroma_temporal_limiter(havc_ctx* c, const uint8_t* cur, const uint8_t* prv, double alpha, uint8_t* out, int width, int height) {
    if (!c || !cur || !prv || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "chroma_temporal_limiter: bad args");
    return run_filter(c, cur, prv, out, (size_t)width * height * 3, "chroma_temporal_limiter", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_chroma_temporal_limiter(da, db, alpha, dout, (int64_t)width * height, c->stream); });
}

int havc_image_luma_merge(havc_ctx* c, const uint8_t* img_dark, const uint8_t* img_white, int mode, double tresh, double grad, uint8_t* out,
                          int width, int height) {
    if (!c || !img_dark || !img_white || !out || width <= 0 || height <= 0 || mode < 0 || mode > 3) return fail(c, HAVC_E_INVALID, "image_luma_merge: bad args");
    return run_filter(c, img_dark, img_white, out, (size_t)width * height * 3, "image_luma_merge", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_luma_merge(da, db, mode, tresh, grad, dout, (int64_t)width * height, c->stream); });
}

int havc_image_luma(havc_ctx* c, const uint8_t* img, int width, int height, double* mean_y) {
    if (!c || !img || !mean_y || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "image_luma: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t nb = (size_t)width * height * 3;
    int rc;
    const uint8_t* din;
    if ((rc = stage_in(c, 0, img, nb, &din)) || (rc = ensure_scratch(c, 6, 256))) return rc;
    int e = launch_luma_sum(din, (unsigned long long*)c->scratch[6], (int64_t)width * height, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "luma sum");
    unsigned long long sum = 0;
    HIP_TRY(c, hipMemcpyAsync(&sum, c->scratch[6], sizeof(sum), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mean_y = (double)sum / ((double)width * (double)height);
    return HAVC_OK;
}

int havc_image_tweak(havc_ctx* c, const uint8_t* img, uint8_t* out, int width, int height, int hue_offset, float brightness, float contrast,
                     float color, const double* hue_ranges, int n_ranges) {
    if (!c || !img || !out || width <= 0 || height <= 0 || n_ranges < 0 || n_ranges > HAVC_MAX_HUE_RANGES || (n_ranges && !hue_ranges))
        return fail(c, HAVC_E_INVALID, "image_tweak: bad args (at most 8 hue ranges)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t nb = (size_t)width * height * 3;
    const int64_t npix = (int64_t)width * height;
    int rc;
    const uint8_t* din;
    uint8_t* dout;
    bool host;
    if ((rc = stage_in(c, 0, img, nb, &din)) || (rc = stage_out_ptr(c, 2, out, nb, &dout, &host)) || (rc = ensure_scratch(c, 6, 256))) return rc;
    TweakArgs a{};
    a.hue_offset = hue_offset; a.brightness = brightness; a.contrast = contrast; a.color = color; a.mean_l = 0; a.n_ranges = n_ranges;
    for (int k = 0; k < n_ranges; ++k) { a.range_lo[k] = hue_ranges[2 * k]; a.range_hi[k] = hue_ranges[2 * k + 1]; }
    if (contrast != 1.f) {
        // ImageEnhance.Contrast: degenerate = solid int(mean(L) + 0.5) of the image as it enters the step
        int e = launch_image_tweak(din, dout, npix, a, (unsigned long long*)c->scratch[6], true, c->stream);
        c->stats.launches++;
        if (e) return hip_fail(c, (hipError_t)e, "image_tweak (L sum)");
        unsigned long long sum = 0;
        HIP_TRY(c, hipMemcpyAsync(&sum, c->scratch[6], sizeof(sum), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        a.mean_l = (int)((double)sum / (double)npix + 0.5);
    }
    int e = launch_image_tweak(din, dout, npix, a, (unsigned long long*)c->scratch[6], false, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "image_tweak");
    return stage_out(c, out, dout, nb, host);
}

int havc_image_chroma_tweak(havc_ctx* c, const uint8_t* img, uint8_t* out, int width, int height, double sat, double bright, int hue,
                            int has_adjust, const double* hue_ranges, int n_ranges, double adj_sat, int adj_hue, double adj_weight) {
    if (!c || !img || !out || width <= 0 || height <= 0 || n_ranges < 0 || n_ranges > HAVC_MAX_HUE_RANGES || (has_adjust && (!hue_ranges || n_ranges < 1)))
        return fail(c, HAVC_E_INVALID, "image_chroma_tweak: bad args (1..8 hue ranges with an adjust stage)");
    auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    ChromaTweakArgs a{};
    a.has_hue = hue != 0; a.hue_half = 0.5 * (double)std::min(std::max(hue, -360), 360);
    a.satc = clampd(sat, 0.0, 10.0); a.brightc = clampd(1.0 + bright, 0.0, 10.0);
    a.has_adjust = has_adjust == 2 ? 2 : (has_adjust != 0); a.n_ranges = has_adjust ? n_ranges : 0;
    if (has_adjust == 2) { a.has_hue = 0; a.satc = 1.0; a.brightc = 1.0; }
    for (int k = 0; k < a.n_ranges; ++k) { a.range_lo[k] = hue_ranges[2 * k]; a.range_hi[k] = hue_ranges[2 * k + 1]; }
    a.has_hue2 = adj_hue != 0; a.hue_half2 = 0.5 * (double)std::min(std::max(adj_hue, -360), 360);
    a.has_sat2 = adj_sat != 1.0; a.sat2c = clampd(adj_sat, 0.0, 10.0);
    a.weight = adj_weight;
    return run_filter(c, img, nullptr, out, (size_t)width * height * 3, "image_chroma_tweak", [&](const uint8_t* da, const uint8_t*, uint8_t* dout) {
        return launch_chroma_tweak(da, dout, (int64_t)width * height, a, c->stream); });
}

int havc_luma_lut(havc_ctx* c, const uint8_t* img, const uint8_t* lut256, uint8_t* out, int width, int height) {
    if (!c || !img || !lut256 || !out || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "luma_lut: bad args");
    return run_filter(c, img, nullptr, out, (size_t)width * height * 3, "luma_lut",
                      [&](const uint8_t* da, const uint8_t*, uint8_t* dout) {
                          return launch_luma_lut(da, (const uint8_t*)c->scratch[6], dout, (int64_t)width * height, c->stream); },
                      [&]() -> int {
                          int rc = ensure_scratch(c, 6, 256);
                          if (rc) return rc;
                          HIP_TRY(c, hipMemcpyAsync(c->scratch[6], lut256, 256, hipMemcpyDefault, c->stream));
                          return HAVC_OK; });
}

int havc_restore_color_gradient(havc_ctx* c, const uint8_t* img_color, const uint8_t* img_gray, uint8_t* out, int width, int height, double sat,
                                int tht, double weight, double alpha, int algo, int return_mask) {
    if (!c || !img_color || !img_gray || !out || width <= 0 || height <= 0 || algo < 0 || algo > 2)
        return fail(c, HAVC_E_INVALID, "restore_color_gradient: bad args (algo 0..2)");
    return run_filter(c, img_color, img_gray, out, (size_t)width * height * 3, "restore_color_gradient", [&](const uint8_t* da, const uint8_t* db, uint8_t* dout) {
        return launch_restore_color_gradient(da, db, dout, (int64_t)width * height, sat, tht, alpha, weight, algo, return_mask, c->stream); });
}

int havc_color_temporal_stabilizer(havc_ctx* c, const uint8_t* const* frames, const double* weights, int n, uint8_t* out, int width, int height) {
    if (!c || !frames || !weights || !out || n < 1 || n > 9 || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "color_temporal_stabilizer: bad args (1..9 frames)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t nb = (size_t)width * height * 3;
    int rc;
    if ((rc = ensure_scratch(c, 0, nb * n))) return rc;
    uint8_t* dout;
    bool host;
    if ((rc = stage_out_ptr(c, 2, out, nb, &dout, &host))) return rc;
    const uint8_t* d_frames[9];
    for (int k = 0; k < n; ++k) {
        if (!frames[k]) return fail(c, HAVC_E_INVALID, "color_temporal_stabilizer: NULL frame");
        if (is_device_ptr(frames[k])) { d_frames[k] = frames[k]; continue; }
        uint8_t* d = (uint8_t*)c->scratch[0] + nb * k;
        HIP_TRY(c, hipMemcpyAsync(d, frames[k], nb, hipMemcpyHostToDevice, c->stream));
        d_frames[k] = d;
    }
    int e = launch_color_temporal_stabilizer(d_frames, weights, n, dout, (int64_t)width * height, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "color_temporal_stabilizer");
    return stage_out(c, out, dout, nb, host);
}

// one batch of the HAVC_colorizer(method=0) clip flow on device frames (the body of havc_colorize_clip[_host])
static int colorize_batch_locked(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, const uint8_t* src, uint8_t* dst, int b,
                                 int width, int height) {
    const int S = video->S;
    const int64_t npix1 = (int64_t)S * S;
    const size_t fb = (size_t)npix1 * 3;
    uint8_t *d_sq = (uint8_t*)c->scratch[0], *d_v = (uint8_t*)c->scratch[1], *d_s = (uint8_t*)c->scratch[2], *d_col = (uint8_t*)c->scratch[3];
    int rc;
    if (width == S && height == S) {
        HIP_TRY(c, hipMemcpyAsync(d_sq, src, fb * b, hipMemcpyDeviceToDevice, c->stream));
    } else if ((rc = resize_rgb8(c, src, width, height, d_sq, S, S, b, nullptr))) return rc;
    if ((rc = run_generators(c, video, second, d_sq, d_v, d_s, b))) return rc;
    if ((rc = deoldify_tail(c, d_sq, d_v, second ? d_s : nullptr, video_weight, 1, d_col, npix1 * b))) return rc;
    // Spline64 back to full size fused with vs_recover_clip_luma (chroma_post_process vs the source frame)
    return resize_rgb8(c, d_col, S, S, dst, width, height, b, src);
}

int havc_colorize_clip(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, const uint8_t* d_src,
                       uint8_t* d_dst, int n_frames, int width, int height) {
    if (!c || !video || !d_src || !d_dst || n_frames < 0 || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "colorize_clip: bad args");
    if (video->ctx != c || (second && (second->ctx != c || second->S != video->S))) return fail(c, HAVC_E_INVALID, "colorize_clip: nets from another ctx / size");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int S = video->S;
    int maxb = video->max_batch;
    if (second) maxb = std::min(maxb, second->max_batch);
    const size_t fb = (size_t)S * S * 3, fbig = (size_t)width * height * 3;
    int rc;
    for (int slot = 0; slot < 4; ++slot)
        if ((rc = ensure_scratch(c, slot, fb * maxb))) return rc;
    Timer t(c);
    for (int f0 = 0; f0 < n_frames; f0 += maxb) {
        const int b = std::min(maxb, n_frames - f0);
        if ((rc = colorize_batch_locked(c, video, second, video_weight, d_src + (size_t)f0 * fbig, d_dst + (size_t)f0 * fbig, b, width, height))) return rc;
    }
    c->stats.frames += n_frames;
    return t.finish();
}

// Host frames in, host frames out, PIPELINED: batch i+1 is uploaded and batch i-1 downloaded on two copy streams while batch i
// is on the compute stream (double-buffered device staging, events between the three streams).  With pinned host memory
// (havc_host_alloc) the copies are true DMA transfers: 6.2 MB per 1080p frame each way against 63 GB/s of PCIe Gen5.
int havc_colorize_clip_host(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, const uint8_t* h_src, uint8_t* h_dst,
                            int n_frames, int width, int height) {
    if (!c || !video || !h_src || !h_dst || n_frames < 0 || width <= 0 || height <= 0) return fail(c, HAVC_E_INVALID, "colorize_clip_host: bad args");
    if (video->ctx != c || (second && (second->ctx != c || second->S != video->S))) return fail(c, HAVC_E_INVALID, "colorize_clip_host: nets from another ctx / size");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int S = video->S;
    int maxb = video->max_batch;
    if (second) maxb = std::min(maxb, second->max_batch);
    const size_t fb = (size_t)S * S * 3, fbig = (size_t)width * height * 3;
    int rc;
    for (int slot = 0; slot < 4; ++slot)
        if ((rc = ensure_scratch(c, slot, fb * maxb))) return rc;
    for (int slot = 8; slot < 12; ++slot)                              // 8, 9: source batches; 10, 11: result batches
        if ((rc = ensure_scratch(c, slot, fbig * maxb))) return rc;
    if (!c->stream_h2d) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_h2d, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_d2h, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_comp[k], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_down[k], hipEventDisableTiming));
        }
    }
    auto body = [&]() -> int {
        int i = 0;
        for (int f0 = 0; f0 < n_frames; f0 += maxb, ++i) {
            const int b = std::min(maxb, n_frames - f0), s = i & 1;
            uint8_t *src = (uint8_t*)c->scratch[8 + s], *dst = (uint8_t*)c->scratch[10 + s];
            if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream_h2d, c->ev_comp[s], 0));      // batch i-2 has consumed this source slot
            HIP_TRY(c, hipMemcpyAsync(src, h_src + (size_t)f0 * fbig, fbig * b, hipMemcpyHostToDevice, c->stream_h2d));
            HIP_TRY(c, hipEventRecord(c->ev_up[s], c->stream_h2d));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up[s], 0));
            if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_down[s], 0));           // batch i-2's result has left this slot
            int r = colorize_batch_locked(c, video, second, video_weight, src, dst, b, width, height);
            if (r) return r;
            HIP_TRY(c, hipEventRecord(c->ev_comp[s], c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream_d2h, c->ev_comp[s], 0));
            HIP_TRY(c, hipMemcpyAsync(h_dst + (size_t)f0 * fbig, dst, fbig * b, hipMemcpyDeviceToHost, c->stream_d2h));
            HIP_TRY(c, hipEventRecord(c->ev_down[s], c->stream_d2h));
        }
        return HAVC_OK;
    };
    rc = body();
    const std::string keep = c->err;
    hipError_t e1 = hipStreamSynchronize(c->stream_h2d), e2 = sync_streams(c), e3 = hipStreamSynchronize(c->stream_d2h);
    if (rc) { (void)hipGetLastError(); c->err = keep; return rc; }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return hip_fail(c, e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "colorize_clip_host drain");
    c->stats.frames += n_frames;
    return HAVC_OK;
}

int havc_host_alloc(havc_ctx* c, size_t nbytes, void** out) {
    if (!c || !out) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    SetupLock setup;
    HIP_TRY(c, hipHostMalloc(out, nbytes, hipHostMallocDefault));
    return HAVC_OK;
}

int havc_host_free(havc_ctx* c, void* p) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, sync_streams(c));
    SetupLock setup;
    HIP_TRY(c, hipHostFree(p));
    return HAVC_OK;
}

// ---- planar <-> interleaved (vsslib/vsutils.py:60-110: frame_to_image / image_to_frame / frame_to_np_array / np_array_to_frame) ----
int havc_planar_to_rgb8(havc_ctx* c, const uint8_t* const planes[3], int stride, uint8_t* rgb, int width, int height) {
    if (!c || !planes || !rgb || width <= 0 || height <= 0 || stride < width) return fail(c, HAVC_E_INVALID, "planar_to_rgb8: bad args");
    for (int p = 0; p < 3; ++p) if (!planes[p]) return fail(c, HAVC_E_INVALID, "planar_to_rgb8: NULL plane");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t plane = (size_t)width * height, nb = plane * 3;
    int rc;
    uint8_t* dout;
    bool host;
    if ((rc = ensure_scratch(c, 4, nb)) || (rc = stage_out_ptr(c, 2, rgb, nb, &dout, &host))) return rc;
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync((uint8_t*)c->scratch[4] + p * plane, (size_t)width, planes[p], (size_t)stride, (size_t)width, (size_t)height, hipMemcpyDefault, c->stream));
    int e = launch_planar_to_rgb8((const uint8_t*)c->scratch[4], dout, (int64_t)plane, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "planar_to_rgb8");
    return stage_out(c, rgb, dout, nb, host);
}

int havc_rgb8_to_planar(havc_ctx* c, const uint8_t* rgb, uint8_t* const planes[3], int stride, int width, int height) {
    if (!c || !planes || !rgb || width <= 0 || height <= 0 || stride < width) return fail(c, HAVC_E_INVALID, "rgb8_to_planar: bad args");
    for (int p = 0; p < 3; ++p) if (!planes[p]) return fail(c, HAVC_E_INVALID, "rgb8_to_planar: NULL plane");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t plane = (size_t)width * height, nb = plane * 3;
    int rc;
    const uint8_t* din;
    if ((rc = stage_in(c, 0, rgb, nb, &din)) || (rc = ensure_scratch(c, 4, nb))) return rc;
    int e = launch_rgb8_to_planar(din, (uint8_t*)c->scratch[4], (int64_t)plane, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "rgb8_to_planar");
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync(planes[p], (size_t)stride, (uint8_t*)c->scratch[4] + p * plane, (size_t)width, (size_t)width, (size_t)height, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HAVC_OK;
}

// One VapourSynth frame through ModelImageRender: the selector body of vs_sc_deoldify (vsslib/vsmodels.py:214-230) --
// frame_to_image, get_transformed_image, image_to_frame -- with the plane <-> interleaved shuffles on the GPU.
int havc_deoldify_frame_planar(havc_ctx* c, havc_net* video, havc_net* second, float video_weight, int post_process,
                               const uint8_t* const in_planes[3], int in_stride, uint8_t* const out_planes[3], int out_stride) {
    if (!c || !video || !in_planes || !out_planes) return fail(c, HAVC_E_INVALID, "deoldify_frame_planar: bad args");
    if (video->ctx != c || (second && (second->ctx != c || second->S != video->S))) return fail(c, HAVC_E_INVALID, "deoldify_frame_planar: nets from another ctx / size");
    const int S = video->S;
    if (in_stride < S || out_stride < S) return fail(c, HAVC_E_INVALID, "deoldify_frame_planar: stride smaller than the render size");
    for (int p = 0; p < 3; ++p) if (!in_planes[p] || !out_planes[p]) return fail(c, HAVC_E_INVALID, "deoldify_frame_planar: NULL plane");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t plane = (size_t)S * S, fb = plane * 3;
    int rc;
    for (int slot = 0; slot < 5; ++slot)
        if ((rc = ensure_scratch(c, slot, fb))) return rc;
    uint8_t *d_in = (uint8_t*)c->scratch[0], *d_v = (uint8_t*)c->scratch[1], *d_s = (uint8_t*)c->scratch[2], *d_out = (uint8_t*)c->scratch[3],
            *d_pl = (uint8_t*)c->scratch[4];
    Timer t(c);
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync(d_pl + p * plane, (size_t)S, in_planes[p], (size_t)in_stride, (size_t)S, (size_t)S, hipMemcpyDefault, c->stream));
    int e = launch_planar_to_rgb8(d_pl, d_in, (int64_t)plane, c->stream);
    if (e) return hip_fail(c, (hipError_t)e, "planar_to_rgb8");
    if ((rc = run_generators(c, video, second, d_in, d_v, d_s, 1))) return rc;
    if ((rc = deoldify_tail(c, d_in, d_v, second ? d_s : nullptr, video_weight, post_process, d_out, (int64_t)plane))) return rc;
    e = launch_rgb8_to_planar(d_out, d_pl, (int64_t)plane, c->stream);
    c->stats.launches += 2;
    if (e) return hip_fail(c, (hipError_t)e, "rgb8_to_planar");
    for (int p = 0; p < 3; ++p)
        HIP_TRY(c, hipMemcpy2DAsync(out_planes[p], (size_t)out_stride, d_pl + p * plane, (size_t)S, (size_t)S, (size_t)S, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stats.frames += 1;
    return t.finish();
}

int havc_spline64_resize(havc_ctx* c, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, const uint8_t* luma_from) {
    return havc_spline64_resize_n(c, src, sw, sh, dst, dw, dh, luma_from, 1);
}

int havc_spline64_resize_n(havc_ctx* c, const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, const uint8_t* luma_from, int n_frames) {
    if (!c || !src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || n_frames < 1) return fail(c, HAVC_E_INVALID, "spline64_resize: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t sb = (size_t)sw * sh * 3 * n_frames, db = (size_t)dw * dh * 3 * n_frames;
    int rc;
    const uint8_t *d_src, *d_luma = nullptr;
    uint8_t* d_dst;
    bool host;
    if ((rc = stage_in(c, 0, src, sb, &d_src)) || (rc = stage_out_ptr(c, 2, dst, db, &d_dst, &host)) ||
        (luma_from && (rc = stage_in(c, 3, luma_from, db, &d_luma)))) return rc;
    if (sw == dw && sh == dh && !luma_from) {
        HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, sb, hipMemcpyDeviceToDevice, c->stream));
    } else if ((rc = resize_rgb8(c, d_src, sw, sh, d_dst, dw, dh, n_frames, d_luma)))
        return rc;
    return stage_out(c, dst, d_dst, db, host);
}

int havc_dev_alloc(havc_ctx* c, size_t nbytes, void** out) {
    if (!c || !out) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    SetupLock setup;
    HIP_TRY(c, hipMalloc(out, nbytes));
    return HAVC_OK;
}
int havc_dev_free(havc_ctx* c, void* p) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, sync_streams(c));
    SetupLock setup;
    HIP_TRY(c, hipFree(p));
    return HAVC_OK;
}
int havc_dev_upload(havc_ctx* c, void* d_dst, const void* h_src, size_t nbytes) {
    if (!c || !d_dst || !h_src) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HAVC_OK;
}
int havc_dev_download(havc_ctx* c, void* h_dst, const void* d_src, size_t nbytes) {
    if (!c || !h_dst || !d_src) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HAVC_OK;
}

// ---- ColorMNet memory kernels.  fp32 operands in the reference's layouts, host or device pointers. ----
static int stage_in_f(havc_ctx* c, int slot, const void* p, size_t nbytes, const float** d) {
    const uint8_t* q = nullptr;
    int rc = stage_in(c, slot, p, nbytes, &q);
    *d = reinterpret_cast<const float*>(q);
    return rc;
}

int havc_memory_read_topk(havc_ctx* c, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out, int B, int CK,
                          int CV, int N, int HW, int top_k) {
    return havc_memory_read_topk_usage(c, mk, ms, qk, qe, mv, out, nullptr, B, CK, CV, N, HW, top_k);
}

int havc_memory_read_topk_usage(havc_ctx* c, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out,
                                float* usage, int B, int CK, int CV, int N, int HW, int top_k) {
    if (!c || !mk || !qk || !mv || !out || B < 1 || CK < 1 || CV < 1 || N < 1 || HW < 1 || top_k < 1 || top_k > 64)
        return fail(c, HAVC_E_INVALID, "memory_read_topk: bad args (top_k 1..64)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t fmk = (size_t)B * CK * N * 4, fq = (size_t)B * CK * HW * 4, fms = (size_t)B * N * 4, fmv = (size_t)B * CV * N * 4, fo = (size_t)B * CV * HW * 4;
    const float *d_mk, *d_ms = nullptr, *d_qk, *d_qe = nullptr, *d_mv;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in_f(c, 0, mk, fmk, &d_mk)) || (rc = stage_in_f(c, 1, qk, fq, &d_qk)) || (rc = stage_in_f(c, 3, mv, fmv, &d_mv)) ||
        (ms && (rc = stage_in_f(c, 4, ms, fms, &d_ms))) || (qe && (rc = stage_in_f(c, 5, qe, fq, &d_qe))) || (rc = stage_out_ptr(c, 2, out, fo, &d_out, &host)))
        return rc;
    // scratch 8: similarity [B][N][HW]; 9: top-k indices; 10: top-k weights
    // scratch 9 / 10 also hold the level-1 survivors of the two-level top-k behind the final lists: [B][k][HW] + [B][S][k][HW]
    const size_t lst = (size_t)B * top_k * HW, cand = lst * (mem_topk_splits(N) > 1 ? mem_topk_splits(N) : 0);
    if ((rc = ensure_scratch(c, 8, (size_t)B * N * HW * 4)) || (rc = ensure_scratch(c, 9, (lst + cand) * 4)) ||
        (rc = ensure_scratch(c, 10, (lst + cand) * 4))) return rc;
    // wave-per-query selection on a query-major similarity (HAVC_TOPK_WAVE=0: the two-level kernels; also beyond 16 384 memory elements)
    static const bool wave_topk = [] { const char* e = getenv("HAVC_TOPK_WAVE"); return !e || atoi(e) != 0; }();
    int e;
    if (wave_topk && mem_topk_select_supported(N)) {
        e = launch_mem_similarity_t(d_mk, d_ms, d_qk, d_qe, (float*)c->scratch[8], B, CK, N, HW, c->stream);
        if (!e) e = launch_mem_topk_select_readout((const float*)c->scratch[8], d_mv, (int*)c->scratch[9], (float*)c->scratch[10], (float*)d_out, B, CV, N, HW,
                                                   top_k, c->stream);
    } else {
        e = launch_mem_similarity(d_mk, d_ms, d_qk, d_qe, (float*)c->scratch[8], B, CK, N, HW, c->stream);
        if (!e) e = launch_mem_topk_readout((const float*)c->scratch[8], d_mv, (int*)c->scratch[9], (float*)c->scratch[10], (float*)c->scratch[10] + lst,
                                            (int*)c->scratch[9] + lst, (float*)d_out, B, CV, N, HW, top_k, c->stream);
    }
    c->stats.launches += 3;
    if (e) return hip_fail(c, (hipError_t)e, "memory_read_topk");
    if (usage) {                                               // row sums of the sparse affinity (do_softmax(..., return_usage=True))
        uint8_t* d_us;
        bool uhost;
        if ((rc = ensure_scratch(c, 11, (size_t)B * N * 8)) || (rc = stage_out_ptr(c, 6, usage, (size_t)B * N * 4, &d_us, &uhost))) return rc;
        e = launch_mem_usage((const int*)c->scratch[9], (const float*)c->scratch[10], (unsigned long long*)c->scratch[11], (float*)d_us, B, N, HW, top_k,
                             c->stream);
        c->stats.launches += 2;
        if (e) return hip_fail(c, (hipError_t)e, "memory_read_topk (usage)");
        if ((rc = stage_out(c, usage, d_us, (size_t)B * N * 4, uhost))) return rc;
    }
    return stage_out(c, out, d_out, fo, host);
}

int havc_memory_dense_readout(havc_ctx* c, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out, int B, int CK,
                              int CV, int N, int P) {
    if (!c || !mk || !qk || !mv || !out || B < 1 || CK < 1 || CV < 1 || CV > 2048 || N < 1 || P < 1)
        return fail(c, HAVC_E_INVALID, "memory_dense_readout: bad args (CV <= 2048)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t fmk = (size_t)B * CK * N * 4, fq = (size_t)B * CK * P * 4, fms = (size_t)B * N * 4, fmv = (size_t)B * CV * N * 4, fo = (size_t)B * CV * P * 4;
    const float *d_mk, *d_ms = nullptr, *d_qk, *d_qe = nullptr, *d_mv;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in_f(c, 0, mk, fmk, &d_mk)) || (rc = stage_in_f(c, 1, qk, fq, &d_qk)) || (rc = stage_in_f(c, 3, mv, fmv, &d_mv)) ||
        (ms && (rc = stage_in_f(c, 4, ms, fms, &d_ms))) || (qe && (rc = stage_in_f(c, 5, qe, fq, &d_qe))) || (rc = stage_out_ptr(c, 2, out, fo, &d_out, &host)) ||
        (rc = ensure_scratch(c, 8, (size_t)B * N * P * 4)))
        return rc;
    int e = launch_mem_similarity(d_mk, d_ms, d_qk, d_qe, (float*)c->scratch[8], B, CK, N, P, c->stream);
    if (!e) e = launch_mem_dense_readout((const float*)c->scratch[8], d_mv, (float*)d_out, B, CV, N, P, c->stream);
    c->stats.launches += 2;
    if (e) return hip_fail(c, (hipError_t)e, "memory_dense_readout");
    return stage_out(c, out, d_out, fo, host);
}

int havc_memory_similarity(havc_ctx* c, const float* mk, const float* ms, const float* qk, const float* qe, float* sim, int B, int CK, int N, int HW) {
    if (!c || !mk || !qk || !sim || B < 1 || CK < 1 || N < 1 || HW < 1) return fail(c, HAVC_E_INVALID, "memory_similarity: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t fmk = (size_t)B * CK * N * 4, fq = (size_t)B * CK * HW * 4, fms = (size_t)B * N * 4, fo = (size_t)B * N * HW * 4;
    const float *d_mk, *d_ms = nullptr, *d_qk, *d_qe = nullptr;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in_f(c, 0, mk, fmk, &d_mk)) || (rc = stage_in_f(c, 1, qk, fq, &d_qk)) || (ms && (rc = stage_in_f(c, 4, ms, fms, &d_ms))) ||
        (qe && (rc = stage_in_f(c, 5, qe, fq, &d_qe))) || (rc = stage_out_ptr(c, 2, sim, fo, &d_out, &host))) return rc;
    int e = launch_mem_similarity(d_mk, d_ms, d_qk, d_qe, (float*)d_out, B, CK, N, HW, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "memory_similarity");
    return stage_out(c, sim, d_out, fo, host);
}

int havc_local_correlation(havc_ctx* c, const float* q, const float* k, float* out, int n, int C, int H, int W, int max_dis, int dilation, float q_scale) {
    if (!c || !q || !k || !out || n < 1 || C < 1 || H < 1 || W < 1 || max_dis < 0 || max_dis > 7 || dilation < 1)
        return fail(c, HAVC_E_INVALID, "local_correlation: bad args (max_dis 0..7, dilation >= 1)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int ws = 2 * max_dis + 1;
    const size_t fi = (size_t)n * C * H * W * 4, fo = (size_t)n * ws * ws * H * W * 4;
    const float *d_q, *d_k;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in_f(c, 0, q, fi, &d_q)) || (rc = stage_in_f(c, 1, k, fi, &d_k)) || (rc = stage_out_ptr(c, 2, out, fo, &d_out, &host))) return rc;
    int e = launch_local_correlation(d_q, d_k, (float*)d_out, n, C, H, W, max_dis, dilation, q_scale, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "local_correlation");
    return stage_out(c, out, d_out, fo, host);
}

int havc_local_attention(havc_ctx* c, const float* q, const float* k, const float* v, const float* rel_w, const float* rel_b, float* agg, float* attn,
                         int n, int C, int CV, int H, int W, int max_dis, int dilation) {
    if (!c || !q || !k || !v || !rel_w || !rel_b || !agg || n < 1 || C < 1 || CV < 1 || H < 1 || W < 1 || max_dis < 0 || max_dis > 7 || dilation < 1)
        return fail(c, HAVC_E_INVALID, "local_attention: bad args (max_dis 0..7, dilation >= 1)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const int ws = 2 * max_dis + 1, WW = ws * ws;
    const size_t fi = (size_t)n * C * H * W * 4, fv = (size_t)n * CV * H * W * 4, fa = (size_t)n * WW * H * W * 4, fo = (size_t)H * W * n * CV * 4;
    const float *d_q, *d_k, *d_v, *d_rw, *d_rb;
    uint8_t *d_agg, *d_attn;
    bool host_agg, host_attn = false;
    int rc;
    if ((rc = stage_in_f(c, 0, q, fi, &d_q)) || (rc = stage_in_f(c, 1, k, fi, &d_k)) || (rc = stage_in_f(c, 3, v, fv, &d_v)) ||
        (rc = stage_in_f(c, 4, rel_w, (size_t)WW * C * 4, &d_rw)) || (rc = stage_in_f(c, 5, rel_b, (size_t)WW * 4, &d_rb)) ||
        (rc = stage_out_ptr(c, 2, agg, fo, &d_agg, &host_agg))) return rc;
    if (attn) { if ((rc = stage_out_ptr(c, 8, attn, fa, &d_attn, &host_attn))) return rc; }
    else { if ((rc = ensure_scratch(c, 8, fa))) return rc; d_attn = (uint8_t*)c->scratch[8]; }
    // q / T with T = sqrt(d_att) = sqrt(C) (attention.py:742, 809); the relative embedding is taken from the UNSCALED q (:806)
    int e = launch_local_correlation(d_q, d_k, (float*)d_attn, n, C, H, W, max_dis, dilation, 1.0f / sqrtf((float)C), c->stream);
    if (!e) e = launch_local_softmax((float*)d_attn, d_q, d_rw, d_rb, n, C, H, W, max_dis, dilation, c->stream);
    if (!e) e = launch_local_agg((const float*)d_attn, d_v, (float*)d_agg, n, CV, H, W, max_dis, dilation, c->stream);
    c->stats.launches += 3;
    if (e) return hip_fail(c, (hipError_t)e, "local_attention");
    if (attn && host_attn) {
        HIP_TRY(c, hipMemcpyAsync(attn, d_attn, fa, hipMemcpyDeviceToHost, c->stream));
        if (!host_agg) HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return stage_out(c, agg, d_agg, fo, host_agg);
}

// ---- ColorMNetRender's frame transforms (colormnet_render.py:285-301, 276-279) ----
int havc_colormnet_rgb_to_lab(havc_ctx* c, const uint8_t* rgb, float* lab, int width, int height) {
    if (!c || !rgb || !lab || width < 1 || height < 1) return fail(c, HAVC_E_INVALID, "colormnet_rgb_to_lab: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t npix = (size_t)width * height;
    const uint8_t* d_in;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in(c, 0, rgb, npix * 3, &d_in)) || (rc = stage_out_ptr(c, 2, lab, npix * 12, &d_out, &host))) return rc;
    int e = launch_cmn_rgb_to_lab(d_in, (float*)d_out, (int64_t)npix, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "colormnet_rgb_to_lab");
    return stage_out(c, lab, d_out, npix * 12, host);
}

int havc_colormnet_lab_to_rgb(havc_ctx* c, const float* l_plane, const float* ab, uint8_t* rgb, int width, int height) {
    if (!c || !l_plane || !ab || !rgb || width < 1 || height < 1) return fail(c, HAVC_E_INVALID, "colormnet_lab_to_rgb: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t npix = (size_t)width * height;
    const float *d_l, *d_ab;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_in_f(c, 0, l_plane, npix * 4, &d_l)) || (rc = stage_in_f(c, 1, ab, npix * 8, &d_ab)) || (rc = stage_out_ptr(c, 2, rgb, npix * 3, &d_out, &host)))
        return rc;
    int e = launch_cmn_lab_to_rgb(d_l, d_ab, d_out, (int64_t)npix, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "colormnet_lab_to_rgb");
    return stage_out(c, rgb, d_out, npix * 3, host);
}

// ---- ColorMNet: the per-frame step without tensor bookkeeping between the kernels (include/havc_mi355.h, "fast step") -----------------------
int havc_cmn_frame_in(havc_ctx* c, const uint8_t* rgb, float* lab, float* img, int width, int height, int padded_w, int padded_h, int pad_left, int pad_top) {
    if (!c || !rgb || !lab || width < 1 || height < 1 || padded_w < width + pad_left || padded_h < height + pad_top || pad_left < 0 || pad_top < 0 ||
        is_device_ptr(lab) == false || (img && !is_device_ptr(img)))
        return fail(c, HAVC_E_INVALID, "cmn_frame_in: bad args (lab / img are device buffers)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const uint8_t* d_in;
    int rc;
    if ((rc = stage_in(c, 0, rgb, (size_t)width * height * 3, &d_in))) return rc;
    int e = launch_cmn_frame_in(d_in, lab, img, width, height, img ? padded_w : width, img ? padded_h : height, img ? pad_left : 0, img ? pad_top : 0, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "cmn_frame_in");
    return HAVC_OK;
}

int havc_cmn_frame_out(havc_ctx* c, const float* l_plane, const float* ab_padded, uint8_t* rgb, int width, int height, int padded_w, int padded_h,
                       int pad_left, int pad_top) {
    if (!c || !l_plane || !ab_padded || !rgb || width < 1 || height < 1 || padded_w < width + pad_left || padded_h < height + pad_top || pad_left < 0 || pad_top < 0 ||
        !is_device_ptr(l_plane) || !is_device_ptr(ab_padded))
        return fail(c, HAVC_E_INVALID, "cmn_frame_out: bad args (the Lab planes are device buffers)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const size_t nb = (size_t)width * height * 3;
    uint8_t* d_out;
    bool host;
    int rc;
    if ((rc = stage_out_ptr(c, 2, rgb, nb, &d_out, &host))) return rc;
    int e = launch_cmn_frame_out(l_plane, ab_padded, d_out, width, height, padded_w, padded_h, pad_left, pad_top, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "cmn_frame_out");
    return stage_out(c, rgb, d_out, nb, host);
}

// use_count += usage, life_count += 1 from the top-k lists in scratch 15 / 16.  The accumulators (scratch 17) are kept at zero BETWEEN reads by the update kernel
// itself; they are cleared here only when the buffer is new (first use, re-grown).  Slots 14 - 17 belong to the banked read alone.
static int usage_update_locked(havc_ctx* c, float* use, float* life, int from, int N, int HW, int top_k, hipStream_t st) {
    if (c->acc_clean_sz != c->scratch_sz[17] || !c->acc_clean_sz) {
        hipError_t m = hipMemsetAsync(c->scratch[17], 0, c->scratch_sz[17], st);
        if (m != hipSuccess) return (int)m;
        c->acc_clean_sz = c->scratch_sz[17];
        c->stats.launches += 1;
    }
    c->stats.launches += 2;
    return launch_mem_usage_update((const int*)c->scratch[15], (const float*)c->scratch[16], (unsigned long long*)c->scratch[17], use, life, from, N, HW, top_k, st);
}

int havc_memory_read_banked(havc_ctx* c, const float* mk, const float* ms, const float* qk, const float* qe, const float* mv, float* out, float* use_count,
                            float* life_count, int usage_from, int CK, int CV, int N, int64_t pitch, int HW, int top_k) {
    if (!c || !mk || !qk || !mv || !out || CK < 1 || CV < 1 || N < 1 || HW < 1 || pitch < N || top_k < 1 || top_k > 64 || usage_from < 0 ||
        (use_count && !life_count))
        return fail(c, HAVC_E_INVALID, "memory_read_banked: bad args (top_k 1..64, pitch >= N)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    int rc;
    const size_t lst = (size_t)top_k * HW, cand = lst * (mem_topk_splits(N) > 1 ? mem_topk_splits(N) : 0);
    if ((rc = ensure_scratch(c, 14, (size_t)N * HW * 4)) || (rc = ensure_scratch(c, 15, (lst + cand) * 4)) || (rc = ensure_scratch(c, 16, (lst + cand) * 4))) return rc;
    const hipStream_t st = c->side ? c->stream2 : c->stream;       // a read-ahead (havc_cmn_side_begin) runs next to the previous frame's decoder
    stream_jitter(st);
    static const bool wave_topk = [] { const char* e = getenv("HAVC_TOPK_WAVE"); return !e || atoi(e) != 0; }();
    int e;
    if (wave_topk && mem_topk_select_supported(N)) {
        e = launch_mem_similarity_t(mk, ms, qk, qe, (float*)c->scratch[14], 1, CK, N, HW, st, pitch);
        if (!e) e = launch_mem_topk_select_readout((const float*)c->scratch[14], mv, (int*)c->scratch[15], (float*)c->scratch[16], out, 1, CV, N, HW, top_k, st, pitch);
    } else {
        e = launch_mem_similarity(mk, ms, qk, qe, (float*)c->scratch[14], 1, CK, N, HW, st, pitch);
        if (!e) e = launch_mem_topk_readout((const float*)c->scratch[14], mv, (int*)c->scratch[15], (float*)c->scratch[16], (float*)c->scratch[16] + lst,
                                            (int*)c->scratch[15] + lst, out, 1, CV, N, HW, top_k, st, pitch);
    }
    c->stats.launches += 3;
    if (!e && use_count) {
        if ((rc = ensure_scratch(c, 17, (size_t)N * 8))) return rc;
        if (c->side) {
            // a read that runs ahead must not touch the counters before its frame is really stepped (a caller may leave the announced order): the
            // top-k lists stay in scratch 15 / 16 until the next read, havc_cmn_side_wait(apply = 1) launches the update from them on the main stream
            c->side_usage.use = use_count; c->side_usage.life = life_count; c->side_usage.from = usage_from;
            c->side_usage.N = N; c->side_usage.HW = HW; c->side_usage.top_k = top_k;
        } else {
            e = usage_update_locked(c, use_count, life_count, usage_from, N, HW, top_k, st);
        }
    }
    if (e) return hip_fail(c, (hipError_t)e, "memory_read_banked");
    return HAVC_OK;
}

int havc_ctx_set_stream_cus(havc_ctx* c, int n_cus) {
    if (!c || n_cus < 1) return fail(c, HAVC_E_INVALID, "set_stream_cus: n_cus >= 1");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side) return fail(c, HAVC_E_INVALID, "set_stream_cus: inside a side section");
    if (c->stream_exported) return fail(c, HAVC_E_INVALID, "set_stream_cus: a stream handle of this context has been handed out (havc_get_stream): call it before");
    HIP_TRY(c, hipSetDevice(c->dev));
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->dev));
    const int total = prop.multiProcessorCount;
    if (n_cus > total) n_cus = total;
    // bit i of the mask = CU i in the driver's enumeration, which deals consecutive bits round-robin over the XCDs: the first n bits are n / 8 CUs of every XCD
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    HIP_TRY(c, sync_streams(c));
    hipStream_t a = nullptr, b = nullptr;
    if (hipExtStreamCreateWithCUMask(&a, (uint32_t)mask.size(), mask.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&b, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipStreamDestroy(a);
        return fail(c, HAVC_E_HIP, "set_stream_cus: hipExtStreamCreateWithCUMask failed");
    }
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream2);
    c->stream = a;
    c->stream2 = b;
    return HAVC_OK;
}

int havc_memory_read_reserve(havc_ctx* c, int N_max, int HW, int top_k) {
    if (!c || N_max < 1 || HW < 1 || top_k < 1 || top_k > 64) return fail(c, HAVC_E_INVALID, "memory_read_reserve: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    // the largest request havc_memory_read_banked can make for N <= N_max: the candidate lists of the two-level selection peak at the largest slice count
    size_t splits = 1;
    for (int n = 64; n <= N_max + 63; n += 64) splits = std::max(splits, (size_t)mem_topk_splits(std::min(n, N_max)));
    const size_t lst = (size_t)top_k * HW, cand = lst * (splits > 1 ? splits : 0);
    int rc;
    if ((rc = ensure_scratch(c, 14, (size_t)N_max * HW * 4)) || (rc = ensure_scratch(c, 15, (lst + cand) * 4)) || (rc = ensure_scratch(c, 16, (lst + cand) * 4)) ||
        (rc = ensure_scratch(c, 17, (size_t)N_max * 8)))
        return rc;
    return HAVC_OK;
}

int havc_cmn_short_term(havc_ctx* c, havc_net* net, int first_op, int n_ops, int agg_buf, int short_buf, const float* q, const float* k, const float* v,
                        const float* rel_w, const float* rel_b, float* agg, float* attn, float* short_out, int C, int CV, int H, int W, int max_dis) {
    if (!c || !net || net->ctx != c || !q || !k || !v || !rel_w || !rel_b || !agg || !attn || !short_out || C < 1 || CV < 1 || H < 1 || W < 1 || max_dis < 0 ||
        max_dis > 7 || agg_buf < 0 || agg_buf >= (int)net->bufs.size() || short_buf < 0 || short_buf >= (int)net->bufs.size())
        return fail(c, HAVC_E_INVALID, "cmn_short_term: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    // fork: stream2 starts behind everything the main stream holds so far (the producers of q / k / v), runs the local attention and the plan's
    // `short` slice there, and records the join event havc_cmn_join_add waits for -- the main stream is free for the memory read meanwhile
    if (!c->side) {                                                // (a read-ahead is on stream2 from its first launch on)
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    stream_jitter(c->stream2);
    int e = launch_local_correlation(q, k, attn, 1, C, H, W, max_dis, 1, 1.0f / sqrtf((float)C), c->stream2);
    if (!e) e = launch_local_softmax(attn, q, rel_w, rel_b, 1, C, H, W, max_dis, 1, c->stream2);
    if (!e) e = launch_local_agg(attn, v, agg, 1, CV, H, W, max_dis, 1, c->stream2);
    c->stats.launches += 3;
    if (e) { (void)hipStreamSynchronize(c->stream2); return hip_fail(c, (hipError_t)e, "cmn_short_term"); }
    if (net->bound.empty()) net->bound.assign(net->bufs.size(), nullptr);
    net->bound[agg_buf] = agg;
    net->bound[short_buf] = short_out;
    c->cur = c->stream2;
    int rc = run_ops_locked(net, first_op, n_ops, 1);
    c->cur = nullptr;
    if (rc) { (void)hipStreamSynchronize(c->stream2); return rc; }
    HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));
    return HAVC_OK;
}

int havc_cmn_join_add(havc_ctx* c, float* readout, const float* short_out, int64_t n) {
    if (!c || !readout || !short_out || n < 1) return fail(c, HAVC_E_INVALID, "cmn_join_add: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    const hipStream_t st = c->side ? c->stream2 : c->stream;
    if (!c->side) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    stream_jitter(st);
    int e = launch_vec_add(readout, short_out, n, st);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "cmn_join_add");
    return HAVC_OK;
}

int havc_cmn_side_mark(havc_ctx* c) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side) return fail(c, HAVC_E_INVALID, "cmn_side_mark: inside a side section");
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipEventRecord(c->ev_mark, c->stream));             // everything the main stream holds NOW: the previous read, the banks, the look-ahead keys it waited for
    c->marked = true;
    return HAVC_OK;
}

int havc_cmn_side_begin(havc_ctx* c) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side) return fail(c, HAVC_E_INVALID, "cmn_side_begin: already inside a side section");
    if (c->side_usage.use) return fail(c, HAVC_E_INVALID, "cmn_side_begin: the previous section has not been waited for");
    HIP_TRY(c, hipSetDevice(c->dev));
    // behind the main stream's work up to the last havc_cmn_side_mark (what was enqueued after it -- this frame's decoder -- runs NEXT to the section), or,
    // without a mark, behind everything it holds
    if (!c->marked) HIP_TRY(c, hipEventRecord(c->ev_mark, c->stream));
    c->marked = false;
    HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_mark, 0));
    stream_jitter(c->stream2);
    c->side = true;
    return HAVC_OK;
}

int havc_ctx_set_stream_priority(havc_ctx* c, int level) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side) return fail(c, HAVC_E_INVALID, "set_stream_priority: inside a side section");
    if (c->stream_exported) return fail(c, HAVC_E_INVALID, "set_stream_priority: a stream handle of this context has been handed out (havc_get_stream): call it before");
    HIP_TRY(c, hipSetDevice(c->dev));
    int least = 0, greatest = 0;
    HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = level < 0 ? least : level > 0 ? greatest : 0;
    HIP_TRY(c, sync_streams(c));
    hipStream_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio) != hipSuccess || hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipStreamDestroy(a);
        return fail(c, HAVC_E_HIP, "set_stream_priority: hipStreamCreateWithPriority failed");
    }
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream2);
    c->stream = a;
    c->stream2 = b;
    return HAVC_OK;
}

int havc_cmn_side_end(havc_ctx* c) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->side) return fail(c, HAVC_E_INVALID, "cmn_side_end: no side section open");
    c->side = false;
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipEventRecord(c->ev_side, c->stream2));
    return HAVC_OK;
}

int havc_cmn_side_wait(havc_ctx* c, int apply_usage) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side) return fail(c, HAVC_E_INVALID, "cmn_side_wait: inside a side section");
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side, 0));
    stream_jitter(c->stream);
    auto u = c->side_usage;
    c->side_usage = {};
    if (apply_usage && u.use) {
        int e = usage_update_locked(c, u.use, u.life, u.from, u.N, u.HW, u.top_k, c->stream);
        if (e) return hip_fail(c, (hipError_t)e, "cmn_side_wait: usage update");
    }
    return HAVC_OK;
}

int havc_cmn_value_in(havc_ctx* c, const float* image, const float* planes, float* value_in, int64_t pixels) {
    if (!c || !image || !planes || !value_in || pixels < 1) return fail(c, HAVC_E_INVALID, "cmn_value_in: bad args");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    int e = launch_cmn_value_in(image, planes, value_in, pixels, c->stream);
    c->stats.launches++;
    if (e) return hip_fail(c, (hipError_t)e, "cmn_value_in");
    return HAVC_OK;
}

int havc_dev_copy_2d(havc_ctx* c, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows) {
    if (!c || !d_dst || !d_src || width_bytes == 0 || rows == 0 || dst_pitch < width_bytes || src_pitch < width_bytes) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    stream_jitter(c->stream);
    HIP_TRY(c, hipMemcpy2DAsync(d_dst, dst_pitch, d_src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, c->stream));
    return HAVC_OK;
}

int havc_net_bind_many(havc_net* n, int count, const int32_t* bufs, void* const* device_ptrs) {
    if (!n || count < 0 || (count && (!bufs || !device_ptrs))) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(n->ctx->mu);
    for (int i = 0; i < count; ++i)
        if (bufs[i] < 0 || bufs[i] >= (int)n->bufs.size()) return fail(n->ctx, HAVC_E_INVALID, "net_bind_many: buffer id out of range");
    if (n->bound.empty()) n->bound.assign(n->bufs.size(), nullptr);
    for (int i = 0; i < count; ++i) n->bound[bufs[i]] = device_ptrs[i];
    return HAVC_OK;
}

int havc_net_enqueue_slices(havc_net* n, int count, const int32_t* first_op, const int32_t* n_ops, const int32_t* batch) {
    if (!n || count < 0 || (count && (!first_op || !n_ops || !batch))) return HAVC_E_INVALID;
    havc_ctx* c = n->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    for (int i = 0; i < count; ++i)
        if (int rc = run_ops_locked(n, first_op[i], n_ops[i], batch[i])) return rc;
    return HAVC_OK;
}

const char* havc_build_stamp(void) { return HAVC_BUILD_STAMP; }

int havc_debug_stream_jitter(int seed, int max_us) {
    g_jitter.on = false;
    if (seed != 0) {
        g_jitter.rng.store(0x9E3779B97F4A7C15ull * (uint64_t)((int64_t)seed + 1) | 1ull);
        g_jitter.max_us = (unsigned)std::max(1, max_us);
        g_jitter.on = true;
    }
    return HAVC_OK;
}

int havc_dev_copy(havc_ctx* c, void* d_dst, const void* d_src, size_t nbytes) {
    if (!c || !d_dst || !d_src) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice, c->stream));     // ordered on the ctx stream, does not block
    return HAVC_OK;
}

int havc_tag_timing_enable(havc_ctx* c, int tag, int enable) {
    if (!c) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    c->timed_tag = enable ? tag : -1;
    c->tag_used = 0;
    return HAVC_OK;
}

int havc_tag_timing_read(havc_ctx* c, double* avg_ms, int64_t* launches) {
    if (!c || !avg_ms || !launches) return HAVC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double total = 0;
    for (size_t i = 0; i < c->tag_used; ++i) {
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->tag_events[i].first, c->tag_events[i].second));
        total += ms;
    }
    *launches = (int64_t)c->tag_used;
    *avg_ms = c->tag_used ? total / (double)c->tag_used : 0.0;
    return HAVC_OK;
}

}  // extern "C"
