// LDS-DMA (global_load_lds_dwordx4) gather-pattern micro-benchmark: how fast can a CU pull L2-resident data
// into LDS when each wave-instruction covers (a) 16 rows x 64 B, (b) 8 rows x 128 B, (c) 4 rows x 256 B, (d) 1 KiB contiguous.
// Rows are `pitch` bytes apart (like NHWC pixels).  Build: hipcc --offload-arch=gfx950 -O3 tools/dma_bench.hip -o tools/bin/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef WIN
#define WIN 128
#endif
typedef _Float16 half_t;

template <int SEG>   // bytes contiguous per row per instruction
__global__ void __launch_bounds__(512) dma_kernel(const char* src, int pitch, int rows_total, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per instruction
    const int r = lane / LPR, c = lane % LPR;
    // each block walks its own window of rows; 8 waves x 4 instr per "stage" = 32 KiB, like the conv kernel
    size_t base_row = ((size_t)blockIdx.x * 977) % (rows_total - 4096);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t row = base_row + (size_t)((it * 4 + q) * 8 + wave) * RPI % WIN + r;
            const char* g = src + row * pitch + ((it * 7 + q) % (pitch / SEG)) * SEG + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds + ((q * 8 + wave) * 1024) % 65536), 16, 0, 0);
        }
        if ((it & 3) == 3) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[lane];
}

// same traffic through VGPRs: global_load_dwordx4 (+ optional ds_write_b128 into the same LDS image)
template <int SEG, int WRITE_LDS>
__global__ void __launch_bounds__(512) reg_kernel(const char* src, int pitch, int rows_total, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;
    const int r = lane / LPR, c = lane % LPR;
    size_t base_row = ((size_t)blockIdx.x * 977) % (rows_total - 4096);
    uint4 accv = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t row = base_row + (size_t)((it * 4 + q) * 8 + wave) * RPI % WIN + r;
            v[q] = *reinterpret_cast<const uint4*>(src + row * pitch + ((it * 7 + q) % (pitch / SEG)) * SEG + c * 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (WRITE_LDS) *reinterpret_cast<uint4*>(lds + ((q * 8 + wave) * 1024) % 65536 + lane * 16) = v[q];
            else { accv.x ^= v[q].x; accv.y ^= v[q].y; accv.z ^= v[q].z; accv.w ^= v[q].w; }
        }
    }
    __syncthreads();
    if (accv.x == 0x12345 || threadIdx.x == 0) sink[blockIdx.x] = lds[lane] + accv.y;
}

template <int SEG, int WRITE_LDS>
void run_reg(const char* d, int pitch, int rows, int* sink, const char* name) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(reg_kernel<SEG, WRITE_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((reg_kernel<SEG, WRITE_LDS>), dim3(blocks), dim3(512), 65536, 0, d, pitch, rows, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)blocks * iters * 4 * 8 * 1024;
    printf("%-40s pitch %4d: %7.3f ms  %7.2f TB/s  %6.1f B/clk/CU@2.1GHz\n", name, pitch, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.1e9);
}

template <int SEG>
void run(const char* d, int pitch, int rows, int* sink, const char* name) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(dma_kernel<SEG>, dim3(blocks), dim3(512), 65536, 0, d, pitch, rows, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)blocks * iters * 4 * 8 * 1024;
    printf("%-28s pitch %4d: %7.3f ms  %7.2f TB/s  %6.1f B/clk/CU@2.1GHz\n", name, pitch, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.1e9);
}

int main() {
    const int rows = 1 << 20; const int pitch_max = 528;
    char* d; int* sink;
    hipMalloc(&d, (size_t)rows * pitch_max); hipMemset(d, 1, (size_t)rows * pitch_max); hipMalloc(&sink, 4096);
    for (int pitch : {512, 528}) {
        run<64>(d, pitch, rows, sink, "16 rows x 64 B per instr");
        run<128>(d, pitch, rows, sink, "8 rows x 128 B per instr");
        run<256>(d, pitch, rows, sink, "4 rows x 256 B per instr");
    }
    run<1024>(d, 1024, rows / 2, sink, "1 KiB contiguous per instr");
    run_reg<64, 0>(d, 512, rows, sink, "VGPR load only, 16 rows x 64 B");
    run_reg<128, 0>(d, 512, rows, sink, "VGPR load only, 8 rows x 128 B");
    run_reg<128, 1>(d, 512, rows, sink, "VGPR load + ds_write_b128, 8 x 128 B");
    run_reg<64, 1>(d, 512, rows, sink, "VGPR load + ds_write_b128, 16 x 64 B");
    run_reg<1024, 0>(d, 1024, rows / 2, sink, "VGPR load only, 1 KiB contiguous");
    return 0;
}
