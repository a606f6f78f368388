// What does each resource cost in WATTS on this part?  Four steady-state loops, each run for a few seconds while the caller samples
// rocm-smi (tests/perf/ubench/power.sh): the 16-bit hourglass step sits at the socket's 1 400 W cap, so the design question
// is energy per operation, not cycles.
//   hipcc -O2 --offload-arch=gfx950 power.hip -o power && ./power <mode> <seconds>
//   modes: mfma_rand (v_mfma_f32_32x32x16_f16 on random operands, 2 waves / SIMD), mfma_zero (the same on zeros), hbm (16-byte streaming
//   reads of a 4 GB buffer), l2 (every workgroup re-reads the same 2 MB: L2 -> CU traffic), lds (ds_read_b128 from a 64 KB tile, 2 waves / SIMD),
//   mix (mfma_rand + lds in one loop)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ __launch_bounds__(256, 2) void mfma_kernel(const u32x4* seed, float* sink, int iters, int zero) {
    u32x4 a = zero ? u32x4{0, 0, 0, 0} : seed[threadIdx.x], b = zero ? u32x4{0, 0, 0, 0} : seed[256 + threadIdx.x];
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[k], 0, 0, 0);
        a.x ^= (unsigned)i & (zero ? 0u : 0x00010001u);   // keep the operands moving a little (no overflow: tiny mantissa flips)
    }
    float s = 0;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256, 2) void lds_kernel(float* sink, int iters, int with_mfma, const u32x4* seed) {
    __shared__ u32x4 tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = seed[i & 511];
    __syncthreads();
    u32x4 v = {0, 0, 0, 0};
    f32x16 acc[2] = {};
    unsigned idx = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 t = tile[(idx + 64 * k) & 4095];
            v ^= t;
            if (with_mfma) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, t), __builtin_bit_cast(f16x8, seed ? t : v), acc[k & 1], 0, 0, 0);
        }
        idx += 256;
    }
    float s = acc[0][0] + acc[1][3];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u || s == 1.5f) sink[0] = 1.0f;
}

__global__ __launch_bounds__(256) void hbm_kernel(const u32x4* __restrict__ buf, size_t n, float* sink) {
    u32x4 v = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v ^= buf[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = 1.0f;
}

// every workgroup walks the same small buffer: L2 hits (the weight-stream pattern of the ring kernels), 8 x 16 bytes per lane and round
__global__ __launch_bounds__(256) void l2_kernel(const u32x4* __restrict__ buf, size_t n, int rounds, float* sink) {
    u32x4 v = {0, 0, 0, 0};
    size_t i = ((size_t)blockIdx.x * 977 + threadIdx.x) % n;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v ^= buf[i];
            i += 256;
            if (i >= n) i -= n;
        }
    }
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = 1.0f;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "mfma_rand";
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    u32x4* seed;
    float* sink;
    hipMalloc(&seed, 512 * sizeof(u32x4));
    hipMalloc(&sink, 64);
    u32x4 h[512];
    srand(1);
    for (auto& q : h) {   // half values in [0.5, 2): finite products, realistic toggling
        for (int c = 0; c < 4; ++c) {
            const unsigned lo = 0x3800u + (rand() & 0x07ff), hi = 0x3800u + (rand() & 0x07ff) + ((rand() & 1) << 15);
            q[c] = lo | (hi << 16);
        }
    }
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    u32x4* big = nullptr;
    const size_t nbig = (size_t)4 << 30 >> 4;
    const bool l2mode = !strcmp(mode, "l2");
    const size_t nsmall = (size_t)2 << 20 >> 4;   // l2: every workgroup streams the same 2 MB (resident in every XCD's 4 MB L2) again and again
    if (!strcmp(mode, "hbm") || l2mode) {
        hipMalloc(&big, nbig * 16);
        hipMemset(big, 1, nbig * 16);
    }
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double work = 0;
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (!strcmp(mode, "mfma_rand") || !strcmp(mode, "mfma_zero")) {
            hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, 0, seed, sink, 20000, !strcmp(mode, "mfma_zero"));
            work += 512.0 * 4 * 20000 * 4 * 32768;   // FLOP
        } else if (!strcmp(mode, "lds") || !strcmp(mode, "mix")) {
            hipLaunchKernelGGL(lds_kernel, dim3(512), dim3(256), 0, 0, sink, 20000, !strcmp(mode, "mix"), seed);
            work += 512.0 * 4 * 20000 * 4 * 1024;    // LDS bytes
        } else if (l2mode) {
            hipLaunchKernelGGL(l2_kernel, dim3(2048), dim3(256), 0, 0, big, nsmall, 64, sink);
            work += 2048.0 * 64 * 256 * 16 * 8;      // bytes through L2 -> CU
        } else {
            hipLaunchKernelGGL(hbm_kernel, dim3(2048), dim3(256), 0, 0, big, nbig, sink);
            work += (double)nbig * 16;               // bytes
        }
        ++launches;
        hipDeviceSynchronize();
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f s, %ld launches, %.3e units/s (FLOP/s for mfma*, bytes/s otherwise)\n", mode, dt, launches, work / dt);
    return 0;
}
