// MFMA-only loops at full occupancy on random operands: sustained TFLOP/s of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 at
// the package power limit (same FLOPs, same accumulator registers).  hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int ZERO>
__global__ void __launch_bounds__(512) k(const half8* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    half8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = ZERO ? half8{0, 0, 0, 0, 0, 0, 0, 0} : src[(t * 12 + i) & 0xfffff];
    for (int i = 0; i < 8; ++i) b[i] = ZERO ? half8{0, 0, 0, 0, 0, 0, 0, 0} : src[(t * 12 + 4 + i) & 0xfffff];
    float s = 0.f;
    if (MODE == 0) {                       // 32 accumulator fragments of 16x16: 4 weight frags x 8 pixel frags, K = 32 per MFMA
        f4 acc[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {                               // 8 accumulator fragments of 32x32: 2 x 4, K = 16 per MFMA: two MFMAs per K-32 -> same FLOPs
        f16v acc[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * kk], b[j + 4 * kk], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) out[t] = s;
}

template <int MODE, int ZERO>
static void run(const half8* d, float* o, const char* name) {
    const int iters = 20000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, ZERO>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, ZERO>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)blocks * 8 * iters * 32 * (16.0 * 16 * 32 * 2);
    printf("%-34s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    const size_t n = 1 << 20;
    std::vector<_Float16> h(n * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    half8* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 24);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    run<0, 0>(d, o, "16x16x32 f16, random operands");
    run<1, 0>(d, o, "32x32x16 f16, random operands");
    run<0, 1>(d, o, "16x16x32 f16, zero operands");
    run<1, 1>(d, o, "32x32x16 f16, zero operands");
    run<0, 0>(d, o, "16x16x32 f16, random operands");
    run<1, 0>(d, o, "32x32x16 f16, random operands");
    return 0;
}
