// Micro-benchmark: issue cost of the individual VALU opcodes the packed sweep is made of (gfx950), in SHADER
// CYCLES MEASURED ON THE DEVICE -- no assumed clock.  Every wave reads the core-clock counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) before and after its instruction stream; the figure printed is
//     cycles per wave-instruction per SIMD = elapsed core cycles / (instructions per wave * waves per SIMD)
// (the SLOWEST wave: the arbiter favours old waves, which finish early), next to the clock the loop really ran at (core cycles / 100 MHz ticks).
// 8 independent registers per wave, so the figure is issue throughput, not dependent-chain latency.
// 256-thread workgroups = one wave per SIMD per workgroup; `wps` workgroups per CU = waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 op_rate.hip -o op_rate && ./op_rate [out.json]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define BODY(NAME, ASM)                                                                          \
    __global__ void NAME(int* out, unsigned long long* clk, int n, int seed) {                   \
        int a[8], b = seed * 3 + (int)threadIdx.x, c = seed + 0x00070007;                        \
        for (int q = 0; q < 8; ++q) a[q] = (int)threadIdx.x * (q + 3) + seed;                    \
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();         \
        for (int i = 0; i < n; ++i) {                                                            \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                      \
                _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(a[q]) : "v"(b), "v"(c)); \
            }                                                                                    \
        }                                                                                        \
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();         \
        int r = 0;                                                                               \
        for (int q = 0; q < 8; ++q) r ^= a[q];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; } \
    }
BODY(k_add_u32, "v_add_u32 %0, %0, %1")
BODY(k_max_i32, "v_max_i32 %0, %0, %1")
BODY(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
BODY(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
BODY(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
BODY(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
BODY(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
BODY(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
BODY(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
BODY(k_pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
BODY(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
BODY(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
BODY(k_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
BODY(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
BODY(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
BODY(k_xor_b32, "v_xor_b32 %0, %0, %1")
BODY(k_max_dpp_shr, "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_max_dpp_bc, "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")
BODY(k_mov_b32, "v_mov_b32 %0, %1")
BODY(k_sub_u32, "v_sub_u32 %0, %0, %1")
BODY(k_and_b32, "v_and_b32 %0, %0, %1")
BODY(k_or_b32, "v_or_b32 %0, %0, %1")
BODY(k_lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
BODY(k_ashrrev_i32, "v_ashrrev_i32 %0, 1, %0")
BODY(k_max_u32, "v_max_u32 %0, %0, %1")
BODY(k_min_i32, "v_min_i32 %0, %0, %1")
BODY(k_max_f32, "v_max_f32 %0, %0, %1")
BODY(k_min_f32, "v_min_f32 %0, %0, %1")
BODY(k_add_f32, "v_add_f32 %0, %0, %1")
BODY(k_mul_f32, "v_mul_f32 %0, %0, %1")
BODY(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
BODY(k_cmp_gt_i32, "v_cmp_gt_i32 vcc, %0, %1")
// (round 4: v_cndmask_b32 read 22.8 cycles in round 3 with a VCC nobody had written -- these tell whether that was the opcode)
BODY(k_cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[8:9]")
BODY(k_cmp_then_cndmask, "v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
BODY(k_cmp_e64_then_cndmask_e64, "v_cmp_gt_i32_e64 s[8:9], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[8:9]")
BODY(k_max_i16, "v_max_i16 %0, %0, %1")
BODY(k_max_u16, "v_max_u16 %0, %0, %1")
BODY(k_add_u16, "v_add_u16 %0, %0, %1")
BODY(k_max_f16, "v_max_f16 %0, %0, %1")
BODY(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
BODY(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
BODY(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
BODY(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
BODY(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
BODY(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
BODY(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
BODY(k_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0")
BODY(k_pk_ashrrev_i16, "v_pk_ashrrev_i16 %0, 1, %0")
BODY(k_mov_dpp_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_add_dpp_shr, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_maxf_dpp_shr, "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_mov_dpp_wshr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
BODY(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
BODY(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 8")
BODY(k_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 8")
BODY(k_alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
BODY(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
BODY(k_add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 1")
BODY(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
BODY(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
BODY(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
BODY(k_pk_max_i16_opsel, "v_pk_max_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")
BODY(k_mix_sub_pkmax, "v_sub_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2")
BODY(k_mix_sub_pkmaxf16, "v_sub_u32 %0, %0, %1\n\tv_pk_max_f16 %0, %0, %2")
BODY(k_add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
BODY(k_max_i16_sdwa, "v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
typedef void (*kern_t)(int*, unsigned long long*, int, int);
static FILE* g_json = nullptr;
static bool g_first = true;
static const char* g_only = nullptr;
static void run(const char* name, kern_t k, int wps) {
    if (g_only && !strstr(name, g_only)) return;
    const int n = 20000, blocks = 256 * wps;
    int* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    unsigned long long* c; hipMalloc(&c, (size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, c, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 2);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, mhz;
    for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2 * b]); if (h[2 * b + 1]) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double instr = (double)n * 64;   // per wave
    // (the arbiter is oldest-wave-first: early waves finish early, so the MEDIAN wave under-reports; the slowest wave's span
    //  covers the whole stream of its SIMD and agrees with the launch's wall time)
    const double cpi = cyc.back() / (instr * wps), clock = mhz.empty() ? 0 : mhz[mhz.size() / 2];
    // cross-check from the host side: the launch's wall time at the measured clock
    const double cpi_wall = ms * 1e-3 * clock * 1e6 / (instr * wps);
    printf("%-16s waves/SIMD=%d  %.3f cycles per wave-instruction per SIMD  (clock %.0f MHz; %.3f from the launch's wall time, %.2f ms)\n",
           name, wps, cpi, clock, cpi_wall, ms);
    if (g_json) {
        fprintf(g_json, "%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instruction\": %.4f, \"clock_mhz\": %.1f, \"cycles_from_wall\": %.4f, \"ms\": %.3f}",
                g_first ? "" : ",", name, wps, cpi, clock, cpi_wall, ms);
        g_first = false;
    }
    hipFree(d); hipFree(c);
}
int main(int argc, char** argv) {
    if (argc > 2) g_only = argv[2];   // (only kernels whose name holds this substring)
    if (argc > 1) { g_json = fopen(argv[1], "w"); if (g_json) fprintf(g_json, "["); }
#define R(N) run(#N, N, 4);
    R(k_add_u32) R(k_max_i32) R(k_max3_i32) R(k_add3_u32) R(k_fma_f32)
    R(k_pk_add_i16) R(k_pk_sub_i16) R(k_pk_max_i16) R(k_pk_min_u16) R(k_pk_mad_i16) R(k_perm_b32) R(k_and_or_b32) R(k_lshl_or_b32)
    R(k_bfi_b32) R(k_lshrrev_b32) R(k_xor_b32) R(k_max_dpp_shr) R(k_max_dpp_bc) R(k_mov_b32)
    R(k_sub_u32) R(k_and_b32) R(k_or_b32) R(k_lshlrev_b32) R(k_ashrrev_i32) R(k_max_u32) R(k_min_i32) R(k_max_f32) R(k_min_f32) R(k_add_f32) R(k_mul_f32)
    R(k_cndmask_b32) R(k_cndmask_e64_sgpr) R(k_cmp_then_cndmask) R(k_cmp_e64_then_cndmask_e64) R(k_cmp_gt_i32) R(k_max_i16) R(k_max_u16) R(k_add_u16) R(k_max_f16) R(k_pk_max_u16) R(k_pk_add_u16) R(k_pk_max_f16) R(k_pk_min_f16)
    R(k_pk_add_f16) R(k_pk_mul_f16) R(k_pk_fma_f16) R(k_pk_lshlrev_b16) R(k_pk_ashrrev_i16) R(k_mov_dpp_shr) R(k_add_dpp_shr) R(k_maxf_dpp_shr) R(k_mov_dpp_wshr)
    R(k_mad_u32_u24) R(k_mul_u32_u24) R(k_bfe_u32) R(k_alignbit_b32) R(k_alignbyte_b32) R(k_lshl_add_u32) R(k_add_lshl_u32) R(k_med3_i32) R(k_min3_u32) R(k_max3_f32)
    R(k_pk_max_i16_opsel) R(k_mix_sub_pkmax) R(k_mix_sub_pkmaxf16) R(k_add_u32_sdwa) R(k_max_i16_sdwa)
    for (int w : {1, 2, 8}) { run("k_pk_add_i16", k_pk_add_i16, w); run("k_pk_max_i16", k_pk_max_i16, w); run("k_perm_b32", k_perm_b32, w); run("k_max_dpp_shr", k_max_dpp_shr, w); run("k_add_u32", k_add_u32, w); }
    if (g_json) { fprintf(g_json, "\n]\n"); fclose(g_json); }
    return 0;
}
