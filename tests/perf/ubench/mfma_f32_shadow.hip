// What does an instruction cost when ONE wave per SIMD issues it between two independent v_mfma_f32_32x32x2_f32 (64 cycles each)?
// The Winograd tail (csrc/hg_bt_wino_f32.h) runs one wave per SIMD (256 accumulator registers), so nothing but the wave's own MFMAs can hide
// its VALU / LDS / VMEM instructions.  16 accumulators round-robin (no dependent pair closer than 16 MFMAs), K filler instructions behind every
// MFMA, fenced so that the order survives; time per MFMA relative to the MFMA-only loop.
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_shadow.hip -o /tmp/mfma_f32_shadow && /tmp/mfma_f32_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc[16];
    for (int q = 0; q < 16; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
    f32x4 v4 = {a, b, a, b};
    double d0 = a, d1 = b;   // (register pairs for the packed add)
    const float* gp = in + threadIdx.x * 4;
    float* lp = lds + threadIdx.x * 4;
    const unsigned voff = threadIdx.x * 16;
    float* op = out + (blockIdx.x * 256 + threadIdx.x) * 4;
    const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x >> 6) * 1024);
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1" : "+v"(x0) : "v"(x1), "v"(x2), "v"(x3));
            if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1" : "+v"(x0) : "v"(x1), "v"(x2), "v"(x3));
            if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1\n"
                                        "v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %2, %2, %1" : "+v"(x0) : "v"(x1), "v"(x2), "v"(x3));
            if (MODE == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)lp) : "memory");
            if (MODE == 6) asm volatile("ds_write_b128 %1, %0" : : "v"(v4), "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)lp) : "memory");
            if (MODE == 7) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v4) : "v"(gp) : "memory");
            if (MODE == 8) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
            if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(x0) : "v"(x1));
            if (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d0) : "v"(d1));
            if (MODE == 11) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v4) : "v"(voff), "s"(in) : "memory");                 // scalar base + 32-bit lane offset
            if (MODE == 12) asm volatile("global_load_dword %0, %1, off" : "=v"(x0) : "v"(gp) : "memory");
            if (MODE == 13) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(op), "v"(v4) : "memory");
            if (MODE == 14) asm volatile("global_store_dword %0, %1, off" : : "v"(op), "v"(x1) : "memory");
            if (MODE == 15) asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(v4), "s"(out) : "memory");
            if (MODE == 16) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(ldsbase) : "memory");            // LDS-DMA, vector address
            if (MODE == 17) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(in), "s"(ldsbase) : "memory");  // LDS-DMA, scalar base
            if (MODE == 18) x0 = acc[(q + 8) & 15][q & 15] + x1;                                                                               // one accumulator element read by the VALU
            if (MODE == 19) asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %2, %2, %3\n s_nop 0\n v_add_f32 %0, %0, %3\n s_nop 0\n v_add_f32 %2, %2, %1" : "+v"(x0) : "v"(x1), "v"(x2), "v"(x3));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 5 || MODE == 6 || MODE == 7 || (MODE >= 11 && MODE <= 17)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = x0 + x2 + v4[0] + v4[3] + (float)d0;
    for (int q = 0; q < 16; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20);
    hipMalloc(&out, 256 * 256 * 16);
    hipMemset(in, 0, 1 << 20);
    const int iters = 20000;   // x 16 MFMAs x 64 cycles = 20.5 M cycles ~ 8.6 ms at 2.4 GHz
    const char* names[] = {"MFMA only", "+1 v_add_f32", "+4 v_add_f32", "+8 v_add_f32", "+16 v_add_f32", "+1 ds_read_b128", "+1 ds_write_b128", "+1 global_load_dwordx4", "+4 s_nop", "+1 v_mov_b32", "+1 v_pk_add_f32",
                           "+1 global_load_dwordx4 saddr", "+1 global_load_dword", "+1 global_store_dwordx4", "+1 global_store_dword", "+1 global_store_dwordx4 saddr",
                           "+1 LDS-DMA x4 vaddr", "+1 LDS-DMA x4 saddr", "+1 VALU reading an acc", "+4 v_add_f32, s_nop between"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float base = 0;
    for (int mode = 0; mode <= 19; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: k<0><<<256, 256>>>(in, out, iters); break;
                case 1: k<1><<<256, 256>>>(in, out, iters); break;
                case 2: k<2><<<256, 256>>>(in, out, iters); break;
                case 3: k<3><<<256, 256>>>(in, out, iters); break;
                case 4: k<4><<<256, 256>>>(in, out, iters); break;
                case 5: k<5><<<256, 256>>>(in, out, iters); break;
                case 6: k<6><<<256, 256>>>(in, out, iters); break;
                case 7: k<7><<<256, 256>>>(in, out, iters); break;
                case 8: k<8><<<256, 256>>>(in, out, iters); break;
                case 9: k<9><<<256, 256>>>(in, out, iters); break;
                case 10: k<10><<<256, 256>>>(in, out, iters); break;
                case 11: k<11><<<256, 256>>>(in, out, iters); break;
                case 12: k<12><<<256, 256>>>(in, out, iters); break;
                case 13: k<13><<<256, 256>>>(in, out, iters); break;
                case 14: k<14><<<256, 256>>>(in, out, iters); break;
                case 15: k<15><<<256, 256>>>(in, out, iters); break;
                case 16: k<16><<<256, 256>>>(in, out, iters); break;
                case 17: k<17><<<256, 256>>>(in, out, iters); break;
                case 18: k<18><<<256, 256>>>(in, out, iters); break;
                case 19: k<19><<<256, 256>>>(in, out, iters); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (mode == 0) base = best;
        printf("%-26s %8.3f ms  %.3f x MFMA-only  = %+6.1f cycles per MFMA (of 64)\n", names[mode], best, best / base, 64.0 * (best / base - 1.0));
    }
    return 0;
}
