// Micro-benchmark: issue rate of the integer VALU ops the DP sweep is made of, 32-bit vs packed
// 16-bit, on gfx950.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(int* out, int n, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    const int g = seed - 6, e = seed - 2;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {  // v_add_u32 + v_max_i32, 8 independent chains
                a0 = max(a0 + g, a1 + e); a1 = max(a1 + g, a2 + e); a2 = max(a2 + g, a3 + e); a3 = max(a3 + g, a4 + e);
                a4 = max(a4 + g, a5 + e); a5 = max(a5 + g, a6 + e); a6 = max(a6 + g, a7 + e); a7 = max(a7 + g, a0 + e);
            } else if (MODE == 1) {  // v_pk_add_i16 + v_pk_max_i16
#define PK(x, y) { s16x2 p = __builtin_bit_cast(s16x2, x) + __builtin_bit_cast(s16x2, g); s16x2 q = __builtin_bit_cast(s16x2, y) + __builtin_bit_cast(s16x2, e); \
                   s16x2 r = __builtin_elementwise_max(p, q); x = __builtin_bit_cast(int, r); }
                PK(a0, a1) PK(a1, a2) PK(a2, a3) PK(a3, a4) PK(a4, a5) PK(a5, a6) PK(a6, a7) PK(a7, a0)
            } else {  // v_cmp_gt + v_cndmask (+add)
                a0 = (a1 + e > a0 + g) ? a1 + e : a0 + g; a1 = (a2 + e > a1 + g) ? a2 + e : a1 + g; a2 = (a3 + e > a2 + g) ? a3 + e : a2 + g;
                a3 = (a4 + e > a3 + g) ? a4 + e : a3 + g; a4 = (a5 + e > a4 + g) ? a5 + e : a4 + g; a5 = (a6 + e > a5 + g) ? a6 + e : a5 + g;
                a6 = (a7 + e > a6 + g) ? a7 + e : a6 + g; a7 = (a0 + e > a7 + g) ? a0 + e : a7 + g;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    int* d; hipMalloc(&d, 256 * 4 * waves_per_simd * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    k<MODE><<<blocks, 256>>>(d, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * 16 * 8 * 3;  // per wave: 3 VALU per chain step
    printf("%-28s waves/SIMD=%d  %.2f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (instr * waves_per_simd));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) { run<0>("v_add_u32/v_max_i32", w); run<1>("v_pk_add_i16/v_pk_max_i16", w); run<2>("v_add/v_cmp/v_cndmask", w); }
    return 0;
}
