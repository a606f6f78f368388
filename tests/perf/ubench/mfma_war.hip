// Does a VALU write to an MFMA's SrcA/SrcB register right behind the MFMA corrupt the product on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define LOADAB "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n v_mov_b32 v44, %5\n v_mov_b32 v45, %6\n v_mov_b32 v46, %7\n v_mov_b32 v47, %8\n s_nop 7\n"
#define OPS : "+v"(acc) : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(junk) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65"
#define MF "v_mfma_f32_32x32x16_bf16 %0, v[40:43], v[44:47], %0\n"
#define TAIL "s_nop 15\n s_nop 15\n s_nop 15"
template <int MODE>
__global__ void k(const u32x4* a, const u32x4* b, f32x16* out, unsigned junk) {
    u32x4 av = a[threadIdx.x], bv = b[threadIdx.x];
    f32x16 acc = {};
    if (MODE == 0) asm volatile(LOADAB MF TAIL OPS);
    if (MODE == 1) asm volatile(LOADAB MF "v_mov_b32 v44, %9\n v_mov_b32 v47, %9\n" TAIL OPS);                 // B dwords 0, 3 right behind
    if (MODE == 2) asm volatile(LOADAB MF "v_pk_add_f32 v[44:45], v[44:45], v[46:47]\n v_pk_add_f32 v[46:47], v[40:41], v[42:43]\n v_pk_add_f32 v[40:41], v[44:45], v[44:45]\n v_pk_add_f32 v[42:43], v[44:45], v[44:45]\n" TAIL OPS);
    if (MODE == 3) asm volatile(LOADAB "v_mfma_f32_32x32x16_bf16 v[50:65], v[40:43], v[44:47], 0\n" MF "v_mov_b32 v44, %9\n v_mov_b32 v45, %9\n v_mov_b32 v46, %9\n v_mov_b32 v47, %9\n v_mov_b32 v40, %9\n v_mov_b32 v41, %9\n v_mov_b32 v42, %9\n v_mov_b32 v43, %9\n" TAIL OPS);
    if (MODE == 4) asm volatile(LOADAB "v_mfma_f32_32x32x16_bf16 v[50:65], v[40:43], v[44:47], 0\n v_mfma_f32_32x32x16_bf16 v[50:65], v[40:43], v[44:47], v[50:65]\n" MF "v_add_f32 v44, v44, v44\n v_add_f32 v45, v44, v44\n" TAIL OPS);
    // RAW: VALU writes a source register of the MFMA that issues right behind it (no wait states in between)
    if (MODE == 10) asm volatile(LOADAB "v_pk_max_i16 v44, %9, 0\n v_pk_max_i16 v47, %9, 0\n v_pk_max_i16 v40, %9, 0\n s_nop 7\n" MF TAIL OPS);
    if (MODE == 11) asm volatile(LOADAB "v_pk_max_i16 v44, %9, 0\n v_pk_max_i16 v47, %9, 0\n v_pk_max_i16 v40, %9, 0\n" MF TAIL OPS);
    if (MODE == 12) asm volatile(LOADAB "v_pk_max_i16 v40, %9, 0\n v_pk_max_i16 v44, %9, 0\n v_pk_max_i16 v47, %9, 0\n s_nop 0\n" MF TAIL OPS);
    if (MODE == 13) asm volatile(LOADAB "v_mov_b32 v44, %9\n v_mov_b32 v47, %9\n v_mov_b32 v40, %9\n" MF TAIL OPS);
    out[threadIdx.x] = acc;
}

int main() {
    std::vector<unsigned> a(64 * 4), b(64 * 4);
    srand(1);
    auto bf = []() { float f = (rand() % 2001 - 1000) / 500.0f; unsigned u; memcpy(&u, &f, 4); return u >> 16; };
    for (auto& x : a) x = bf() | bf() << 16;
    for (auto& x : b) x = bf() | bf() << 16;
    u32x4 *da, *db; f32x16* dout;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 64 * 64);
    hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
    std::vector<float> ref(1024), got(1024);
    auto run = [&](int mode, unsigned junk) {
        switch (mode) {
            case 0: k<0><<<1, 64>>>(da, db, dout, junk); break;
            case 1: k<1><<<1, 64>>>(da, db, dout, junk); break;
            case 2: k<2><<<1, 64>>>(da, db, dout, junk); break;
            case 3: k<3><<<1, 64>>>(da, db, dout, junk); break;
            case 4: k<4><<<1, 64>>>(da, db, dout, junk); break;
            case 10: k<10><<<1, 64>>>(da, db, dout, junk); break;
            case 11: k<11><<<1, 64>>>(da, db, dout, junk); break;
            case 12: k<12><<<1, 64>>>(da, db, dout, junk); break;
            case 13: k<13><<<1, 64>>>(da, db, dout, junk); break;
        }
    };
    auto cmp = [&](int mode, int refmode, unsigned junk) {
        run(refmode, junk); hipMemcpy(ref.data(), dout, 4096, hipMemcpyDeviceToHost);
        run(mode, junk); hipMemcpy(got.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
        printf("mode %d vs %d: %d of 1024 outputs differ\n", mode, refmode, bad);
    };
    for (int m = 1; m <= 4; ++m) cmp(m, 0, 0x7fc07fc0u);
    for (int m = 11; m <= 13; ++m) cmp(m, 10, 0x3f803f80u);
    cmp(10, 0, 0x3f803f80u);   // (sanity: the overwritten operands do change the product)
    return 0;
}
