// Per-CU load throughput of the "MFMA B-operand straight from global" pattern (lane = pixel, 16 bytes of its 512-byte channel
// vector per load) against contiguous streaming, at different bytes-per-pixel-per-visit granularities.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// G = bytes of one pixel consumed per visit (64: the kernel today; 128: one cache line; 256; 512); DEPTH = visits in flight
template <int G, int DEPTH, bool CONTIG>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* buf, size_t region_bytes, int regions_per_wg, unsigned* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    constexpr int KT = 3;                 // row tiles of 32 pixels per wave
    constexpr int NJ = G / 32;            // load instructions per row tile and visit
    constexpr int VISITS = 512 / G;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < regions_per_wg; ++r) {
        const unsigned char* reg = buf + ((size_t)blockIdx.x * regions_per_wg + r) * region_bytes;   // 384 pixels x 512 bytes
        u32x4 v[DEPTH][KT][NJ];
        auto issue = [&](int visit, int slot) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned char* p;
                    if (CONTIG) p = reg + (size_t)((wave * KT + kt) * VISITS + visit) * (32 * G) + j * 1024 + lane * 16;
                    else p = reg + (size_t)((wave + 4 * kt) * 32 + l31) * 512 + visit * G + j * 32 + half * 16;
                    v[slot][kt][j] = *reinterpret_cast<const u32x4*>(p);
                }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, d);
#pragma unroll
        for (int visit = 0; visit < VISITS; ++visit) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc ^= v[visit % DEPTH][kt][j];
            if (visit + DEPTH < VISITS) issue(visit + DEPTH, visit % DEPTH);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int G, int DEPTH, bool CONTIG>
void run(const unsigned char* buf, size_t total, int wgs, unsigned* out, unsigned long long* cyc, const char* what) {
    const size_t region = 384 * 512;
    const int per = (int)(total / region / wgs);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<G, DEPTH, CONTIG><<<wgs, 256>>>(buf, region, per, out, cyc);   // warm
    hipEventRecord(a);
    k<G, DEPTH, CONTIG><<<wgs, 256>>>(buf, region, per, out, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static unsigned long long c[256];
    hipMemcpy(c, cyc, 8 * wgs, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < wgs; ++i) avg += c[i]; avg /= wgs;
    printf("%-8s G=%3d depth=%d wgs=%3d total=%5zu MB: %7.0f ticks/region, %5.1f B/tick/CU, %6.2f TB/s\n", what, G, DEPTH, wgs, total >> 20,
           avg / per, (double)region * per / avg, (double)region * per * wgs / ms / 1e9);
}

int main() {
    const size_t big = (size_t)3 << 30;
    unsigned char* buf; unsigned* out; unsigned long long* cyc;
    hipMalloc(&buf, big); hipMemset(buf, 1, big); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    for (size_t total : {big, (size_t)48 << 20}) {
        for (int wgs : {256, 32}) {
            run<64, 3, true>(buf, total, wgs, out, cyc, "contig");
            run<64, 3, false>(buf, total, wgs, out, cyc, "frag");
            run<64, 6, false>(buf, total, wgs, out, cyc, "frag");
            run<128, 2, false>(buf, total, wgs, out, cyc, "frag");
            run<128, 3, false>(buf, total, wgs, out, cyc, "frag");
            run<256, 2, false>(buf, total, wgs, out, cyc, "frag");
            run<512, 1, false>(buf, total, wgs, out, cyc, "frag");
        }
    }
    return 0;
}
