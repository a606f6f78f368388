// LDS read bandwidth per CU: ds_read_b128 / ds_read_b64, 4 or 8 waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int W, int PATTERN>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // PATTERN 0: lane * W (linear);  1: MFMA fragment of a swizzled 64-byte-row stage: row = lane & 31, chunk = lane >> 5
    unsigned off;
    if (PATTERN == 0) off = lane * W;
    else { const int r = lane & 31, c = lane >> 5; off = (r >> 2) * 256 + ((((r & 3) << 2 | c) ^ ((r >> 3) & 3)) << 4); }
    const unsigned char* base = smem + off + (wave & 3) * 4096;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[16];
        const unsigned char* b2 = base + (it & 7) * 512;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (W == 16) v[u] = *reinterpret_cast<const u32x4*>(b2 + u * 2048);
            else { u32x2 t = *reinterpret_cast<const u32x2*>(b2 + u * 2048); v[u] = u32x4{t[0], t[1], 0, 0}; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u];
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W, int PATTERN>
void run(int waves, unsigned* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<W, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<W, PATTERN><<<256, waves * 64, 100 * 1024>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c[256];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
    // s_memtime counts at 100 MHz constant clock on gfx9?  report both raw ticks and bytes/tick
    const double bytes = (double)iters * 16 * waves * 64 * W;
    printf("W=%2d pattern=%d waves=%d: %.0f ticks, %.1f bytes/tick/CU\n", W, PATTERN, waves, avg, bytes / avg);
}

int main() {
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    // calibrate the s_memtime tick: time a kernel with events
    for (int waves : {4, 8, 16}) {
        run<16, 0>(waves, out, cyc); run<16, 1>(waves, out, cyc); run<8, 0>(waves, out, cyc);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<16, 0><<<256, 256, 100 * 1024>>>(out, cyc, 20000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %.3f ms for %llu ticks -> %.1f MHz tick; 4 waves b128: %.1f GB/s/CU\n", ms, c0, c0 / ms / 1e3, 20000.0 * 16 * 256 * 16 / ms / 1e6);
    return 0;
}
