// Micro-benchmark: issue cost of the individual VALU opcodes the packed sweep is made of (gfx950).
// 8 independent registers per wave, 4 waves per SIMD, so the figure is throughput, not latency;
// DEP=1 makes every instruction depend on the previous one (latency-bound for one wave, but 4 waves
// per SIMD still interleave).   hipcc --offload-arch=gfx950 -O3 op_rate.hip -o op_rate && ./op_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(NAME, ASM)                                                                          \
    __global__ void NAME(int* out, int n, int seed) {                                            \
        int a[8], b = seed * 3 + (int)threadIdx.x, c = seed + 0x00070007;                        \
        for (int q = 0; q < 8; ++q) a[q] = (int)threadIdx.x * (q + 3) + seed;                    \
        for (int i = 0; i < n; ++i) {                                                            \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                      \
                _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(a[q]) : "v"(b), "v"(c)); \
            }                                                                                    \
        }                                                                                        \
        int r = 0;                                                                               \
        for (int q = 0; q < 8; ++q) r ^= a[q];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
    }
BODY(k_add, "v_add_u32 %0, %0, %1")
BODY(k_pk_add, "v_pk_add_i16 %0, %0, %1")
BODY(k_pk_sub, "v_pk_sub_i16 %0, %0, %1")
BODY(k_pk_max, "v_pk_max_i16 %0, %0, %1")
BODY(k_pk_minu, "v_pk_min_u16 %0, %0, %1")
BODY(k_pk_mad, "v_pk_mad_i16 %0, %0, %1, %2")
BODY(k_perm, "v_perm_b32 %0, %0, %1, %2")
BODY(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
BODY(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
BODY(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
BODY(k_lshr, "v_lshrrev_b32 %0, 3, %0")
BODY(k_xor, "v_xor_b32 %0, %0, %1")
BODY(k_dpp, "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_dpp_bc, "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")
BODY(k_mov, "v_mov_b32 %0, %1")
typedef void (*kern_t)(int*, int, int);
static void run(const char* name, kern_t k, int wps) {
    int* d; hipMalloc(&d, 256 * 4 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000, blocks = 256 * wps;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * 64;
    printf("%-14s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, wps, ms * 1e-3 * 2.4e9 / (instr * wps));
    hipFree(d);
}
int main() {
#define R(N) run(#N, N, 4);
    R(k_add) R(k_pk_add) R(k_pk_sub) R(k_pk_max) R(k_pk_minu) R(k_pk_mad) R(k_perm) R(k_and_or) R(k_lshl_or) R(k_bfi) R(k_lshr) R(k_xor)
    R(k_dpp) R(k_dpp_bc) R(k_mov)
    run("k_pk_add", k_pk_add, 1); run("k_and_or", k_and_or, 1); run("k_dpp", k_dpp, 1);
    return 0;
}
