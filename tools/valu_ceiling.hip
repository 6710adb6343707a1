// Micro-benchmarks behind bench.py's `valu_frac` and `copy_GBs` ("measure the ceiling you divide by"):
//   1. how many wave64 vector instructions one SIMD of an MI355X issues per cycle on the instruction mix of the codec's kernels
//      (f32 add / mul, v_cndmask, DPP row moves, v_readlane, i32 add / shift), as independent streams (eight accumulators per
//      wave) and as one dependent chain, at 1, 2, 4 and 8 waves per SIMD;
//   2. the device-to-device copy bandwidth a plain 16-byte-per-lane kernel reaches (SURVEY 8d's "measured device-copy bandwidth").
// Stand-alone: hipcc --offload-arch=gfx950 -O3 -o valu_ceiling tools/valu_ceiling.hip && ./valu_ceiling > profiles/r04_valu_ceiling.json
// Every kernel runs `iters` rounds of an unrolled block of 1 024 instructions (16 x 64) between two s_memtime stamps (shader-clock
// cycles) on lane 0 of each wave; a launch is one workgroup of 4 W waves per CU (W per SIMD).  The loop's own cost -- counter, compare,
// taken branch: measured with an empty body, ~50 cycles per round for a lone wave -- is 1 % of a round of 1 024 instructions and is
// subtracted (round 3 ran 64 instructions per round and did not: its figures were up to 19 % high).  Reported: wave-instructions per
// SIMD per cycle = W * instructions of one wave / (cycles of the slowest wave: a SIMD serves its waves oldest first), its reciprocal,
// and the shader clock the kernel ran at (delta s_memtime / delta s_memrealtime x 100 MHz, median over the waves).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// eight independent streams x 128 = 1 024 instructions per block, ONE asm statement (between two asm statements the compiler pads with
// an s_nop); the asm is volatile and touches only its own operands
#define REP8(x) x
#define R128 ".rept 128\n"
#define ENDR ".endr\n"
#define BLOCK_INDEP(OP)                                                                                                   \
    REP8(asm volatile(R128 OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) ENDR                                                  \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                   \
                      : "v"(k) : "vcc");)
#define BLOCK_CHAIN(OP)                                                                                                   \
    REP8(asm volatile(R128 OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) ENDR                                                  \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                   \
                      : "v"(k) : "vcc");)
#define OP_FADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_FMUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CNDMASK_S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define OP_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_IADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_SHIFT(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
// the mix: per eight instructions 2 f32 add, 2 f32 mul, 2 selects, 1 DPP move, 1 integer add (the analysis kernels' VALU histogram)
// (the selects in the SGPR-pair form the kernels' selects mostly have; a VOP2 select on vcc that does not directly follow the compare that
// wrote vcc is a case of its own, rows "v_cndmask_b32 (vcc)": 22.8 cycles)
#define OP_MIX(i) "v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                  "v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_add_u32 %7, %7, %8\n"
#define BLOCK_MIX REP8(asm volatile(R128 OP_MIX(0) ENDR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s20", "s21");)
// v_readlane writes a scalar register: eight different ones per group
#define BLOCK_READLANE                                                                                                      \
    REP8(asm volatile(R128 "v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %1, 2\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 4\n" \
                      "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 6\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 8\n" ENDR \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
                      : "v"(k) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, BODY)                                                                                  \
    __global__ void NAME(unsigned long long *cycles, float *sink, int iters) {                              \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const float k = 1.0001f;                                                                            \
        __syncthreads();                                                                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  \
        _Pragma("unroll 1") for (int i = 0; i < iters; i++) { BODY }                                 \
        asm volatile("s_nop 0" ::: "memory");                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
        if ((threadIdx.x & 63) == 0) {                                                                      \
            const size_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                           \
            cycles[2 * w] = t1 - t0;                                                                        \
            cycles[2 * w + 1] = r1 - r0;                                                                    \
        }                                                                                                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;                              \
    }
KERNEL(k_empty, asm volatile("" ::: "memory");)
KERNEL(k_fadd, BLOCK_INDEP(OP_FADD))
KERNEL(k_fmul, BLOCK_INDEP(OP_FMUL))
KERNEL(k_cndmask, BLOCK_INDEP(OP_CNDMASK))
#define BLOCK_CNDMASK_S                                                                                                   \
    REP8(asm volatile(R128 OP_CNDMASK_S(0) OP_CNDMASK_S(1) OP_CNDMASK_S(2) OP_CNDMASK_S(3) OP_CNDMASK_S(4) OP_CNDMASK_S(5) OP_CNDMASK_S(6) OP_CNDMASK_S(7) ENDR \
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                   \
                      : "v"(k) : "s20", "s21");)
KERNEL(k_cndmask_s, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); BLOCK_CNDMASK_S)
KERNEL(k_dpp, BLOCK_INDEP(OP_DPP))
KERNEL(k_iadd, BLOCK_INDEP(OP_IADD))
KERNEL(k_shift, BLOCK_INDEP(OP_SHIFT))
KERNEL(k_mul24, BLOCK_INDEP(OP_MUL24))
KERNEL(k_mullo, BLOCK_INDEP(OP_MULLO))
KERNEL(k_readlane, BLOCK_READLANE)
KERNEL(k_mix, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); BLOCK_MIX)
KERNEL(k_fadd_chain, BLOCK_CHAIN(OP_FADD))
KERNEL(k_cndmask_chain, BLOCK_CHAIN(OP_CNDMASK))

__global__ void k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

typedef void (*kern_t)(unsigned long long *, float *, int);
struct Case { const char *name; kern_t k; const char *what; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 128;             // x 1 024 instructions per wave
    const double instr = 1024.0 * iters;
    unsigned long long *d_cycles;
    float *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * (size_t)cus * 32 * 2));
    CHECK(hipMalloc(&d_sink, 64));
    const Case cases[] = {
        {"v_add_f32", k_fadd, "8 independent streams"}, {"v_mul_f32", k_fmul, "8 independent streams"},
        {"v_cndmask_b32 (vcc)", k_cndmask, "8 independent streams"}, {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_s, "8 independent streams"}, {"v_mov_b32_dpp row_shr:1", k_dpp, "8 independent streams"},
        {"v_add_u32", k_iadd, "8 independent streams"}, {"v_lshlrev_b32", k_shift, "8 independent streams"},
        {"v_mul_u32_u24", k_mul24, "8 independent streams"}, {"v_mul_lo_u32", k_mullo, "8 independent streams"},
        {"v_readlane_b32", k_readlane, "8 independent streams"},
        {"mix (2 add, 2 mul, 2 cndmask_e64, 1 dpp, 1 iadd per 8)", k_mix, "8 independent streams"},
        {"v_add_f32 dependent chain", k_fadd_chain, "1 chain"}, {"v_cndmask_b32 dependent chain", k_cndmask_chain, "1 chain"},
    };
    printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_khz_reported\": %d,\n", prop.name, prop.gcnArchName, cus,
           prop.clockRate);
    printf("  \"method\": \"one workgroup of 4W waves per CU, %d x 1024 instructions per wave between two s_memtime stamps, the empty loop's cycles subtracted; wave_instr_per_simd_cycle = W * instructions / cycles of the slowest wave; clock = 100 MHz x delta s_memtime / delta s_memrealtime, median over the waves\",\n", iters);
    // the loop's own cost per round, by waves per SIMD (empty body)
    double overhead[9] = {0};
    for (int W : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * W, wg_threads = threads > 1024 ? 1024 : threads, grid = cus * (threads / wg_threads), waves = grid * (wg_threads / 64);
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_empty, dim3(grid), dim3(wg_threads), 0, nullptr, d_cycles, d_sink, iters * 16);  // (16 x the rounds: a measurable run)
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h((size_t)waves * 2);
        CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int w = 0; w < waves; w++) mx = std::max(mx, h[2 * (size_t)w]);
        overhead[W] = (double)mx / (iters * 16.0);  // per round of the loop
    }
    printf("  \"loop_overhead_cycles_per_round\": {\"1\": %.1f, \"2\": %.1f, \"4\": %.1f, \"8\": %.1f},\n", overhead[1], overhead[2], overhead[4], overhead[8]);
    printf("  \"valu\": [\n");
    bool first = true;
    for (const Case &c : cases) {
        for (int W : {1, 2, 4, 8}) {
            const int threads = 64 * 4 * W;
            if (threads > 1024 && W == 8) {
                // 8 waves per SIMD = 32 waves per CU: two workgroups of 1024 threads per CU
            }
            const int wg_threads = threads > 1024 ? 1024 : threads;
            const int grid = cus * (threads / wg_threads);
            const int waves = grid * (wg_threads / 64);
            for (int rep = 0; rep < 3; rep++) {  // the last repetition counts (clocks ramped up)
                hipLaunchKernelGGL(c.k, dim3(grid), dim3(wg_threads), 0, nullptr, d_cycles, d_sink, iters);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> hh((size_t)waves * 2), h((size_t)waves);
            CHECK(hipMemcpy(hh.data(), d_cycles, sizeof(unsigned long long) * hh.size(), hipMemcpyDeviceToHost));
            std::vector<double> mhz((size_t)waves);
            for (int w = 0; w < waves; w++) {
                h[(size_t)w] = hh[2 * (size_t)w];
                mhz[(size_t)w] = hh[2 * (size_t)w + 1] ? 100.0 * (double)hh[2 * (size_t)w] / (double)hh[2 * (size_t)w + 1] : 0.0;
            }
            std::sort(h.begin(), h.end());
            std::sort(mhz.begin(), mhz.end());
            const double raw_mx = (double)h.back();
            const double med = (double)h[h.size() / 2] - overhead[W] * iters, mx = raw_mx - overhead[W] * iters;
            // the waves of a SIMD are served oldest first: the last one to finish marks the time the SIMD needed for all W streams
            printf("%s    {\"instruction\": \"%s\", \"streams\": \"%s\", \"waves_per_simd\": %d, \"wave_instr_per_simd_cycle\": %.4f, "
                   "\"cycles_per_wave_instr\": %.3f, \"median_wave_cycles\": %.0f, \"max_wave_cycles\": %.0f, \"max_wave_cycles_with_loop\": %.0f, \"clock_MHz\": %.0f}",
                   first ? "" : ",\n", c.name, c.what, W, W * instr / mx, mx / (W * instr), med, mx, raw_mx, mhz[mhz.size() / 2]);
            first = false;
        }
    }
    printf("\n  ],\n");
    // device copy bandwidth
    {
        const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(float4);
        float4 *a, *b;
        CHECK(hipMalloc(&a, bytes));
        CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 1, bytes));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_copy16, dim3(cus * 16), dim3(256), 0, nullptr, (const float4 *)a, b, n);
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("  \"copy\": {\"kernel\": \"16 bytes per lane, grid-stride, 1 GiB read + 1 GiB written\", \"ms\": %.4f, \"GBs_read_plus_write\": %.1f, \"GBs_one_way\": %.1f}\n",
               best, 2.0 * bytes / (best * 1e-3) / 1e9, (double)bytes / (best * 1e-3) / 1e9);
    }
    printf("}\n");
    return 0;
}
