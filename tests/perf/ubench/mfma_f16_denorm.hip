// Does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs (|x| < 2^-14), or flush them to zero?  Decides the form of the low part of a
// two-way f16 split of an fp32 operand (x = hi + lo): an unscaled lo is subnormal whenever |x| < 2^-3.
//   hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o mfma_f16_denorm && ./mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const _Float16* a, const _Float16* b, float* out) {
    // A[i][k]: lane (i = l & 31, k = 8 (l >> 5) + e); B[k][j]: lane (j = l & 31, k = 8 (l >> 5) + e)
    const int l = threadIdx.x;
    f16x8 av, bv;
    for (int e = 0; e < 8; ++e) {
        av[e] = a[(l & 31) * 16 + 8 * (l >> 5) + e];
        bv[e] = b[(8 * (l >> 5) + e) * 32 + (l & 31)];
    }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

int main() {
    std::vector<_Float16> a(32 * 16), b(16 * 32);
    // row i of A: every element 2^-(10 + i / 2) (normal down to 2^-14, subnormal below: 2^-15 .. 2^-24 are representable subnormals)
    for (int i = 0; i < 32; ++i)
        for (int kk = 0; kk < 16; ++kk) a[i * 16 + kk] = (_Float16)std::ldexp(1.0f, -(10 + i / 2));
    for (int kk = 0; kk < 16; ++kk)
        for (int j = 0; j < 32; ++j) b[kk * 32 + j] = (_Float16)(j < 16 ? 1.0f : std::ldexp(1.0f, -(10 + j / 4)));   // right half: small B as well
    _Float16 *da, *db;
    float* dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dout, 32 * 32 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    std::vector<float> out(32 * 32);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) {
        const float av = (float)a[i * 16];
        for (int j : {0, 20, 31}) {
            const float expect = 16.0f * av * (float)b[j];
            const float got = out[i * 32 + j];
            const bool ok = got == expect;
            bad += !ok;
            if (j == 0 || !ok) printf("A = 2^%d (%s) x B = %g: expect %.6e got %.6e %s\n", -(10 + i / 2), (10 + i / 2) > 14 ? "subnormal" : "normal", (float)b[j], expect, got, ok ? "" : "<-- DIFFERENT");
        }
    }
    printf(bad ? "RESULT: f16 subnormal inputs are NOT kept exactly (%d mismatches)\n" : "RESULT: f16 subnormal inputs are kept (exact products, %d mismatches)\n", bad);
    return 0;
}
