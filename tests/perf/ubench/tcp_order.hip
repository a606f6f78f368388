// Is the CU's vector-memory return path in order ACROSS waves?  Wave 0 measures the latency of an L2-resident load while
// waves 1..3 (other SIMDs, same CU) stream HBM misses (mode 1), stream L2 hits (mode 2) or stay idle (mode 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void k(const unsigned char* big, size_t big_bytes, const unsigned char* small, int mode, int iters,
                                            unsigned long long* lat, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    if (wave == 0) {
        unsigned long long total = 0;
        for (int it = 0; it < iters; ++it) {
            const unsigned char* p = small + ((size_t)((it * 64 + lane) * 16) & 0xffff);
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            u32x4 v = *reinterpret_cast<const volatile u32x4*>(p);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            acc ^= v;
            total += t1 - t0;
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) lat[blockIdx.x] = total / iters;
    } else if (mode != 0) {
        // each of waves 1..3 keeps 16 x 1 KB loads in flight
        const size_t span = mode == 1 ? big_bytes : (size_t)1 << 20;
        size_t off = ((size_t)blockIdx.x * 3 + (wave - 1)) * ((size_t)iters * 64 * 1024) % span;
        for (int it = 0; it < iters * 4; ++it) {
            u32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const u32x4*>(big + (off + (size_t)u * 1024 + lane * 16) % span);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc ^= v[u];
            off += 16 * 1024;
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    const size_t big = (size_t)2 << 30;
    unsigned char *b, *s; unsigned long long* lat; unsigned* sink;
    hipMalloc(&b, big); hipMemset(b, 1, big); hipMalloc(&s, 1 << 16); hipMemset(s, 2, 1 << 16);
    hipMalloc(&lat, 256 * 8); hipMalloc(&sink, 256 * 256 * 4);
    for (int wgs : {256, 8}) {
        for (int mode = 0; mode < 3; ++mode) {
            k<<<wgs, 256>>>(b, big, s, mode, 2000, lat, sink);
            hipDeviceSynchronize();
            unsigned long long l[256];
            hipMemcpy(l, lat, 8 * wgs, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < wgs; ++i) avg += l[i]; avg /= wgs;
            printf("wgs=%3d %s: L2-resident load latency seen by wave 0 = %.0f ticks\n", wgs,
                   mode == 0 ? "other waves idle        " : mode == 1 ? "other waves stream HBM  " : "other waves stream L2   ", avg);
        }
    }
    return 0;
}
