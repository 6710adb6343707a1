// What single instructions and short idioms cost a wave of the lane-per-frame kernels (one wave per SIMD) and of the wave-per-stream
// kernels (four): shader-clock cycles per instruction, measured as 500 rounds of an unrolled block of 256 idioms (ONE asm statement: the
// compiler pads between two of them with an s_nop; the loop's own ~32 cycles per round are under 3 % of the cheapest block and are
// subtracted) between two s_memtime stamps.
// The lane-per-frame kernels are issue-bound (filler instructions cost their full time), so their time is the sum
// of these numbers -- this table says which idiom to write.
// Stand-alone: hipcc --offload-arch=gfx950 -O3 -o instr_cost tools/instr_cost.hip && ./instr_cost > profiles/r04_instr_cost.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#define REP8(x) x
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(lds), "v"(pq) : "vcc", "s20", "s21", "s22", "s23", "s24", "memory"
// eight copies of an idiom on eight independent registers, 32 times: 256 idioms per block
#define BLOCK(S0, S1, S2, S3, S4, S5, S6, S7) REP8(asm volatile(".rept 32\n" S0 S1 S2 S3 S4 S5 S6 S7 ".endr\n" OPERANDS);)
#define EACH(OP) BLOCK(OP(0), OP(1), OP(2), OP(3), OP(4), OP(5), OP(6), OP(7))

#define KERNEL(NAME, PRE, BODY)                                                                             \
    __global__ void NAME(unsigned long long *cycles, unsigned *sink, int iters) {                           \
        __shared__ unsigned lds_buf[1024];                                                                  \
        lds_buf[threadIdx.x & 1023] = threadIdx.x;                                                          \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const unsigned k = 37u, lds = (threadIdx.x & 63) * 4;  /* one bank per lane */                                             \
        const unsigned long long pq = 0x3f8000003f800000ull + threadIdx.x;  /* a register PAIR (%10): packed-f32 operand of the mixes */    \
        __syncthreads();                                                                                    \
        PRE;                                                                                                \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        _Pragma("unroll 1") for (int i = 0; i < iters; i++) { BODY }                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 0" ::: "memory");                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds_buf[5] == 12345678u) sink[0] = a0;                  \
    }
#define NOPRE ((void)0)
#define SETMASKS asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333\n s_mov_b64 s[22:23], 0x0f0f" ::: "vcc", "s20", "s21", "s22", "s23")

// ---- selects and compares
#define I_CND_VCC(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_CND_E64_VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define I_CND_E64_S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_CMP_VCC(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n"
#define I_CMP_S(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n"
#define I_SEL_VCC(i) "v_cmp_lt_u32_e32 vcc, %8, %" #i "\n v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_SEL_S(i) "v_cmp_lt_u32_e64 s[20:21], %8, %" #i "\n s_nop 1\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_SEL_S2(i) "v_cmp_lt_u32_e64 s[20:21], %8, %" #i "\n v_cmp_lt_u32_e64 s[22:23], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_SEL_ARITH(i) "v_sub_u32 %" #i ", %" #i ", %8\n v_ashrrev_i32 %" #i ", 31, %" #i "\n v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define I_SEL_MINMAX(i) "v_min_u32 %" #i ", %" #i ", %8\n"
// a VOP2 select on vcc in its surroundings: right behind the compare that wrote vcc; the second select on the same compare; a select on
// a vcc written long ago, between other instructions
#define I_SEL_VCC2(i) "v_cmp_lt_u32_e32 vcc, %8, %" #i "\n v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define I_SEL_VCC_GAP(i) "v_cmp_lt_u32_e32 vcc, %8, %" #i "\n v_add_u32 %" #i ", %" #i ", %8\n v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_CND_VCC_ADD(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_add_u32 %" #i ", %" #i ", %8\n"
#define I_SEL_S_NONOP(i) "v_cmp_lt_u32_e64 s[20:21], %8, %" #i "\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define I_SEL_S_2(i) "v_cmp_lt_u32_e64 s[20:21], %8, %" #i "\n v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
KERNEL(k_sel_vcc2, NOPRE, EACH(I_SEL_VCC2))
KERNEL(k_sel_vcc_gap, NOPRE, EACH(I_SEL_VCC_GAP))
KERNEL(k_cnd_vcc_add, SETMASKS, EACH(I_CND_VCC_ADD))
KERNEL(k_sel_s_nonop, NOPRE, EACH(I_SEL_S_NONOP))
KERNEL(k_sel_s_2, NOPRE, EACH(I_SEL_S_2))
KERNEL(k_cnd_vcc, SETMASKS, EACH(I_CND_VCC))
KERNEL(k_cnd_e64_vcc, SETMASKS, EACH(I_CND_E64_VCC))
KERNEL(k_cnd_e64_s, SETMASKS, EACH(I_CND_E64_S))
KERNEL(k_cmp_vcc, NOPRE, EACH(I_CMP_VCC))
KERNEL(k_cmp_s, NOPRE, EACH(I_CMP_S))
KERNEL(k_sel_vcc, NOPRE, EACH(I_SEL_VCC))
KERNEL(k_sel_s, NOPRE, EACH(I_SEL_S))
KERNEL(k_sel_s2, NOPRE, EACH(I_SEL_S2))
KERNEL(k_sel_arith, NOPRE, EACH(I_SEL_ARITH))
KERNEL(k_min, NOPRE, EACH(I_SEL_MINMAX))
// ---- plain integer operations
#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define I_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define I_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 3, 1\n"
#define I_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MUL24_SDWA(i) "v_mul_u32_u24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define I_ADDC(i) "v_addc_co_u32_e64 %" #i ", s[22:23], %" #i ", %8, s[20:21]\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
KERNEL(k_add, NOPRE, EACH(I_ADD))
KERNEL(k_add3, NOPRE, EACH(I_ADD3))
KERNEL(k_lshladd, NOPRE, EACH(I_LSHLADD))
KERNEL(k_bfi, NOPRE, EACH(I_BFI))
KERNEL(k_bfe, NOPRE, EACH(I_BFE))
KERNEL(k_ashr, NOPRE, EACH(I_ASHR))
KERNEL(k_andor, NOPRE, EACH(I_ANDOR))
KERNEL(k_mad24, NOPRE, EACH(I_MAD24))
KERNEL(k_mul24_sdwa, NOPRE, EACH(I_MUL24_SDWA))
KERNEL(k_addc, SETMASKS, EACH(I_ADDC))
KERNEL(k_mov, NOPRE, EACH(I_MOV))
// ---- f32 arithmetic: VOP2 encodings, the VOP3 forms of the same operations, packed pairs, DPP operands (round 6: what does a flop cost?)
#define I_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define I_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_FMACF(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define I_FMAF(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_ADDF_E64(i) "v_add_f32_e64 %" #i ", %" #i ", %8\n"
#define I_ADDF_NEG(i) "v_add_f32_e64 %" #i ", -%" #i ", %8\n"
#define I_MULF_E64(i) "v_mul_f32_e64 %" #i ", %" #i ", %8\n"
#define I_ADDF_DPP(i) "v_add_f32_dpp %" #i ", %8, %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CVT_F32_I32(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define I_RCPF(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define I_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
KERNEL(k_addf, NOPRE, EACH(I_ADDF))
KERNEL(k_subf, NOPRE, EACH(I_SUBF))
KERNEL(k_mulf, NOPRE, EACH(I_MULF))
KERNEL(k_maxf, NOPRE, EACH(I_MAXF))
KERNEL(k_fmacf, NOPRE, EACH(I_FMACF))
KERNEL(k_fmaf, NOPRE, EACH(I_FMAF))
KERNEL(k_addf_e64, NOPRE, EACH(I_ADDF_E64))
KERNEL(k_addf_neg, NOPRE, EACH(I_ADDF_NEG))
KERNEL(k_mulf_e64, NOPRE, EACH(I_MULF_E64))
KERNEL(k_addf_dpp, NOPRE, EACH(I_ADDF_DPP))
KERNEL(k_mov_dpp, NOPRE, EACH(I_MOV_DPP))
KERNEL(k_cvt, NOPRE, EACH(I_CVT_F32_I32))
KERNEL(k_rcp, NOPRE, EACH(I_RCPF))
KERNEL(k_mullo, NOPRE, EACH(I_MULLO))
KERNEL(k_lshl, NOPRE, EACH(I_LSHL))
KERNEL(k_and, NOPRE, EACH(I_AND))
KERNEL(k_xor, NOPRE, EACH(I_XOR))
KERNEL(k_subu, NOPRE, EACH(I_SUBU))
KERNEL(k_maxi, NOPRE, EACH(I_MAXI))
// packed f32: two IEEE operations per lane and instruction on 64-bit register pairs (un-fused add / mul: bit-exact under -ffp-contract=off)
typedef float lc3_f2 __attribute__((ext_vector_type(2)));
#define OPERANDS_PK : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q), "v"(r) : "memory"
#define BLOCK_PK(S0, S1, S2, S3, S4, S5, S6, S7) asm volatile(".rept 32\n" S0 S1 S2 S3 S4 S5 S6 S7 ".endr\n" OPERANDS_PK);
#define EACH_PK(OP) BLOCK_PK(OP(0), OP(1), OP(2), OP(3), OP(4), OP(5), OP(6), OP(7))
#define KERNEL_PK(NAME, BODY)                                                                               \
    __global__ void NAME(unsigned long long *cycles, unsigned *sink, int iters) {                           \
        const float t = (float)threadIdx.x;                                                                 \
        lc3_f2 p0 = {t, t + 1}, p1 = {t + 2, t + 3}, p2 = {t + 4, t + 5}, p3 = {t + 6, t + 7}, p4 = {t + 8, t + 9}, p5 = {t + 10, t + 11}, \
               p6 = {t + 12, t + 13}, p7 = {t + 14, t + 15};                                                \
        const lc3_f2 q = {1.0000001f, 0.9999999f}, r = {1e-9f, -1e-9f};                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        _Pragma("unroll 1") for (int i = 0; i < iters; i++) { BODY }                                        \
        asm volatile("s_nop 0" ::: "memory");                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
        const lc3_f2 z = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;                                             \
        if (z.x + z.y == 12345.678f) sink[0] = 1;                                                           \
    }
#define I_PK_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_PK_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define I_PK_FMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_PK_ADD_NEG(i) "v_pk_add_f32 %" #i ", %" #i ", %8 neg_lo:[0,1] neg_hi:[1,0]\n"
#define I_PK_MUL_SWAP(i) "v_pk_mul_f32 %" #i ", %" #i ", %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
KERNEL_PK(k_pk_add, EACH_PK(I_PK_ADD))
KERNEL_PK(k_pk_mul, EACH_PK(I_PK_MUL))
KERNEL_PK(k_pk_fma, EACH_PK(I_PK_FMA))
KERNEL_PK(k_pk_add_neg, EACH_PK(I_PK_ADD_NEG))
KERNEL_PK(k_pk_mul_swap, EACH_PK(I_PK_MUL_SWAP))
// ---- run-length sweep: N plain VOP2 f32 additions (on up to eight independent registers), then ONE instruction of another class, repeated.
// Which class takes the SIMD out of its two-cycle rate, and how long a pure run has to be before the rate is back
#define R1 I_ADDF(0)
#define R2 I_ADDF(0) I_MULF(1)
#define R3 I_ADDF(0) I_MULF(1) I_ADDF(2)
#define R4 I_ADDF(0) I_MULF(1) I_ADDF(2) I_MULF(3)
#define R6 R4 I_ADDF(4) I_MULF(5)
#define R8 R4 I_ADDF(4) I_MULF(5) I_ADDF(6) I_MULF(7)
#define R12 R8 R4
#define R16 R8 R8
#define R24 R8 R8 R8
#define R32 R16 R16
#define X_CND "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
#define X_DPP "v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define X_VOP3 "v_add3_u32 %7, %7, %8, %9\n"
#define X_CMP "v_cmp_lt_u32_e64 s[20:21], %7, %8\n"
#define X_LDS "ds_read_b32 %7, %9\n"
#define X_NONE ""
// (second batch: what else shares a wave's instruction stream with the arithmetic)
#define X_DPP_QUAD "v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define X_DPP_ADD "v_add_f32_dpp %7, %8, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define X_DPP_BC "v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define X_DPP_OWN "v_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define X_DPP_WSHR "v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define X_DPP_NOP "v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
#define X_READLANE "v_readlane_b32 s24, %7, 3\n"
#define X_READFIRST "v_readfirstlane_b32 s24, %7\n"
#define X_BPERM "ds_bpermute_b32 %7, %9, %8\n"
#define X_SWIZZLE "ds_swizzle_b32 %7, %8 offset:swizzle(SWAP,1)\n"
#define X_PERMLANE "v_permlane32_swap_b32 %7, %6\n"
#define X_SDWA "v_mov_b32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define X_CVT "v_cvt_f32_i32 %7, %7\n"
#define X_RCP "v_rcp_f32 %7, %7\n"
#define X_MAXF "v_max_f32 %7, %7, %8\n"
#define X_LSHL "v_lshlrev_b32 %7, 3, %7\n"
#define X_PK "v_pk_add_f32 %10, %10, %10\n"
#define X_PK2 "v_pk_mul_f32 %10, %10, %10\n v_pk_add_f32 %10, %10, %10\n"
#define X_SALU "s_add_u32 s24, s24, 1\n"
#define X_SALU4 "s_add_u32 s24, s24, 1\n s_and_b32 s24, s24, 0xff\n s_lshl_b32 s24, s24, 1\n s_cmp_lg_u32 s24, 0\n"
#define X_WAIT "s_waitcnt lgkmcnt(0)\n"
#define X_CND_VCC "v_cndmask_b32_e32 %7, %7, %8, vcc\n"
#define X_MOV "v_mov_b32 %7, %8\n"
#define R64 R32 R32
#define R128 R64 R64
#define RUNBLOCK(R, X) asm volatile(".rept 16\n" R X ".endr\n" OPERANDS);
#define RUN4_KERNELS(TAG, X)                             \
    KERNEL(k_r4_##TAG##_2, SETMASKS, RUNBLOCK(R2, X))    \
    KERNEL(k_r4_##TAG##_8, SETMASKS, RUNBLOCK(R8, X))    \
    KERNEL(k_r4_##TAG##_32, SETMASKS, RUNBLOCK(R32, X))  \
    KERNEL(k_r4_##TAG##_128, SETMASKS, RUNBLOCK(R128, X))
#define RUN_KERNELS(TAG, X)                        \
    KERNEL(k_run_##TAG##_1, SETMASKS, RUNBLOCK(R1, X))   \
    KERNEL(k_run_##TAG##_2, SETMASKS, RUNBLOCK(R2, X))   \
    KERNEL(k_run_##TAG##_3, SETMASKS, RUNBLOCK(R3, X))   \
    KERNEL(k_run_##TAG##_4, SETMASKS, RUNBLOCK(R4, X))   \
    KERNEL(k_run_##TAG##_6, SETMASKS, RUNBLOCK(R6, X))   \
    KERNEL(k_run_##TAG##_8, SETMASKS, RUNBLOCK(R8, X))   \
    KERNEL(k_run_##TAG##_12, SETMASKS, RUNBLOCK(R12, X)) \
    KERNEL(k_run_##TAG##_16, SETMASKS, RUNBLOCK(R16, X)) \
    KERNEL(k_run_##TAG##_24, SETMASKS, RUNBLOCK(R24, X)) \
    KERNEL(k_run_##TAG##_32, SETMASKS, RUNBLOCK(R32, X))
RUN_KERNELS(none, X_NONE)
RUN_KERNELS(cnd, X_CND)
RUN_KERNELS(dpp, X_DPP)
RUN_KERNELS(vop3, X_VOP3)
RUN_KERNELS(cmp, X_CMP)
RUN_KERNELS(lds, X_LDS)
RUN4_KERNELS(dppq, X_DPP_QUAD)
RUN4_KERNELS(dppadd, X_DPP_ADD)
RUN4_KERNELS(dppbc, X_DPP_BC)
RUN4_KERNELS(dppown, X_DPP_OWN)
RUN4_KERNELS(dpp, X_DPP)
RUN4_KERNELS(dppwshr, X_DPP_WSHR)
RUN4_KERNELS(dppnop, X_DPP_NOP)
RUN4_KERNELS(readlane, X_READLANE)
RUN4_KERNELS(readfirst, X_READFIRST)
RUN4_KERNELS(bperm, X_BPERM)
RUN4_KERNELS(swizzle, X_SWIZZLE)
RUN4_KERNELS(permlane, X_PERMLANE)
RUN4_KERNELS(sdwa, X_SDWA)
RUN4_KERNELS(cvt, X_CVT)
RUN4_KERNELS(rcp, X_RCP)
RUN4_KERNELS(maxf, X_MAXF)
RUN4_KERNELS(lshl, X_LSHL)
RUN4_KERNELS(pk, X_PK)
RUN4_KERNELS(pk2, X_PK2)
RUN4_KERNELS(salu, X_SALU)
RUN4_KERNELS(salu4, X_SALU4)
RUN4_KERNELS(wait, X_WAIT)
RUN4_KERNELS(cndvcc, X_CND_VCC)
RUN4_KERNELS(mov, X_MOV)
// ---- scalar side: mask logic, a not-taken branch, an exec-masked region, wait states
#define I_SNOP(i) "s_nop 1\n"
KERNEL(k_snop, NOPRE, EACH(I_SNOP))
// eight v_add each followed by a branch that is never taken (exec is never zero)
#define I_ADD_BR(i) "v_add_u32 %" #i ", %" #i ", %8\n s_cbranch_execz 1f\n1:\n"
KERNEL(k_add_branch, NOPRE, EACH(I_ADD_BR))
// ---- LDS: eight reads in flight then one wait (throughput), a read waited for at once (round trip), byte stores
#define I_DSR32(i) "ds_read_b32 %" #i ", %9\n"
#define I_DSR128W(i) "ds_read_b32 %" #i ", %9\n s_waitcnt lgkmcnt(0)\n"
#define I_DSW8(i) "ds_write_b8 %9, %" #i "\n"
KERNEL(k_dsr32, NOPRE, BLOCK(I_DSR32(0), I_DSR32(1), I_DSR32(2), I_DSR32(3), I_DSR32(4), I_DSR32(5), I_DSR32(6), I_DSR32(7) "s_waitcnt lgkmcnt(0)\n"))
KERNEL(k_dsr32_wait, NOPRE, EACH(I_DSR128W))
KERNEL(k_dsw8, NOPRE, EACH(I_DSW8))
// a dependent LDS read: the address is the value just read (round-trip latency of a pointer chase)
#define I_DSCHASE(i) "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0\n"
KERNEL(k_dschase, a0 = lds, EACH(I_DSCHASE))

typedef void (*kern_t)(unsigned long long *, unsigned *, int);
struct Case { const char *name; kern_t k; int instrs_per_idiom; const char *what; };

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 500;
    unsigned long long *d_cycles;
    unsigned *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * (size_t)cus * 32));
    CHECK(hipMalloc(&d_sink, 64));
    const Case cases[] = {
        {"v_cndmask_b32_e32 (vcc)", k_cnd_vcc, 1, "select on vcc, VOP2 encoding"},
        {"v_cndmask_b32_e64 (vcc)", k_cnd_e64_vcc, 1, "select on vcc, VOP3 encoding"},
        {"v_cndmask_b32_e64 (s[20:21])", k_cnd_e64_s, 1, "select on an SGPR pair"},
        {"v_cmp_lt_u32_e32 -> vcc", k_cmp_vcc, 1, "compare writing vcc"},
        {"v_cmp_lt_u32_e64 -> s[20:21]", k_cmp_s, 1, "compare writing an SGPR pair"},
        {"v_cmp_e32 + v_cndmask_e32 through vcc", k_sel_vcc, 2, "a select: two instructions"},
        {"v_cmp_e64 + s_nop 1 + v_cndmask_e64 through s[20:21]", k_sel_s, 3, "a select as the compiler emits it: three instructions"},
        {"v_cmp_e64 + v_cndmask_e64 through s[20:21], no s_nop", k_sel_s_nonop, 2, "the same without the compiler's wait states (the hardware interlocks)"},
        {"v_cmp_e64 + 2 x v_cndmask_e64 through s[20:21], no s_nop", k_sel_s_2, 3, "one compare, two selects"},
        {"v_cmp_e32 + 2 x v_cndmask_e32 through vcc", k_sel_vcc2, 3, "one compare, two selects on vcc"},
        {"v_cmp_e32 + v_add_u32 + v_cndmask_e32 through vcc", k_sel_vcc_gap, 3, "an instruction between the compare and its select"},
        {"v_cndmask_b32_e32 (vcc) + v_add_u32", k_cnd_vcc_add, 2, "selects on a vcc written long ago, between other instructions"},
        {"v_cmp_e64 + v_cmp_e64 + v_cndmask_e64", k_sel_s2, 3, "a select with another compare in the wait-state slot"},
        {"v_sub + v_ashrrev 31 + v_bfi", k_sel_arith, 3, "a select without a mask register: three instructions"},
        {"v_min_u32", k_min, 1, ""},
        {"v_add_u32", k_add, 1, ""}, {"v_add3_u32", k_add3, 1, ""}, {"v_lshl_add_u32", k_lshladd, 1, ""}, {"v_bfi_b32", k_bfi, 1, ""},
        {"v_bfe_i32", k_bfe, 1, ""}, {"v_ashrrev_i32", k_ashr, 1, ""}, {"v_and_or_b32", k_andor, 1, ""}, {"v_mad_u32_u24", k_mad24, 1, ""},
        {"v_mul_u32_u24_sdwa", k_mul24_sdwa, 1, ""}, {"v_addc_co_u32_e64", k_addc, 1, ""}, {"v_mov_b32", k_mov, 1, ""},
        {"v_add_f32", k_addf, 1, "VOP2"}, {"v_sub_f32", k_subf, 1, "VOP2"}, {"v_mul_f32", k_mulf, 1, "VOP2"}, {"v_max_f32", k_maxf, 1, "VOP2"},
        {"v_fmac_f32", k_fmacf, 1, "VOP2 (fused: not usable where the reference rounds twice)"}, {"v_fma_f32", k_fmaf, 1, "VOP3"},
        {"v_add_f32_e64", k_addf_e64, 1, "the VOP3 encoding of a plain add"}, {"v_add_f32_e64 with a neg modifier", k_addf_neg, 1, "VOP3 (source modifier)"},
        {"v_mul_f32_e64", k_mulf_e64, 1, "the VOP3 encoding of a plain multiply"},
        {"v_add_f32_dpp quad_perm", k_addf_dpp, 1, "a DPP operand on an arithmetic instruction"}, {"v_mov_b32_dpp row_shr", k_mov_dpp, 1, ""},
        {"v_cvt_f32_i32", k_cvt, 1, "VOP1"}, {"v_rcp_f32", k_rcp, 1, "VOP1, transcendental unit"}, {"v_mul_lo_u32", k_mullo, 1, "VOP3"},
        {"v_lshlrev_b32", k_lshl, 1, "VOP2"}, {"v_and_b32", k_and, 1, "VOP2"}, {"v_xor_b32", k_xor, 1, "VOP2"}, {"v_sub_u32", k_subu, 1, "VOP2"},
        {"v_max_i32", k_maxi, 1, "VOP2"},
        {"v_pk_add_f32", k_pk_add, 1, "two f32 additions per lane"}, {"v_pk_mul_f32", k_pk_mul, 1, "two f32 multiplications per lane"},
        {"v_pk_fma_f32", k_pk_fma, 1, "two fused multiply-adds per lane"},
        {"v_pk_add_f32 neg_lo:[0,1] neg_hi:[1,0]", k_pk_add_neg, 1, "add in one half, subtract in the other (a butterfly's re / im)"},
        {"v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", k_pk_mul_swap, 1, "second operand's halves swapped (the cross terms of a complex product)"},
        {"s_nop 1", k_snop, 1, ""},
        {"v_add_u32 + s_cbranch_execz (not taken)", k_add_branch, 2, ""},
        {"ds_read_b32 x8 then one wait", k_dsr32, 8, "eight reads in flight; per read"},
        {"ds_read_b32 + s_waitcnt", k_dsr32_wait, 1, "round trip of one read"},
        {"ds_write_b8", k_dsw8, 1, ""},
        {"ds_read_b32 pointer chase", k_dschase, 1, "read, wait, mask: dependent round trip"},
    };
    printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d,\n", prop.name, prop.gcnArchName, cus);
    printf("  \"method\": \"one workgroup of 4W waves per CU; %d rounds of 256 idioms (one asm statement) on eight independent registers between two s_memtime stamps, 32 cycles per round of loop overhead subtracted; cycles per idiom = cycles of the slowest wave / (W x idioms)\",\n", iters);
    printf("  \"idioms\": [\n");
    bool first = true;
    for (const Case &c : cases) {
        double per[3] = {0, 0, 0};
        int wi = 0;
        for (int W : {1, 2, 4}) {
            const int threads = 64 * 4 * W;
            const int grid = cus, waves = grid * (threads / 64);
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(c.k, dim3(grid), dim3(threads), 0, nullptr, d_cycles, d_sink, iters);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h((size_t)waves);
            CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * (size_t)waves, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double idioms = 256.0 * iters;
            per[wi++] = ((double)h.back() - 32.0 * iters) / (W * idioms);
        }
        const double n = 1.0;
        printf("%s    {\"idiom\": \"%s\", \"instructions\": %d, \"cycles_lone_wave\": %.2f, \"cycles_per_wave_at_2_per_simd\": %.2f, \"cycles_per_wave_at_4_per_simd\": %.2f, \"note\": \"%s\"}",
               first ? "" : ",\n", c.name, c.instrs_per_idiom, per[0] / n, per[1] / n, per[2] / n, c.what);
        first = false;
    }
    printf("\n  ],\n");
    // run-length sweep
    struct Run { const char *tag; const char *what; kern_t k[10]; };
    const int lens[10] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
#define RUNS(TAG) {k_run_##TAG##_1, k_run_##TAG##_2, k_run_##TAG##_3, k_run_##TAG##_4, k_run_##TAG##_6, k_run_##TAG##_8, k_run_##TAG##_12, k_run_##TAG##_16, k_run_##TAG##_24, k_run_##TAG##_32}
    const Run runs[] = {{"none", "nothing (the pure run)", RUNS(none)}, {"v_cndmask_b32_e64", "a select on an SGPR pair", RUNS(cnd)},
                        {"v_mov_b32_dpp", "a DPP move", RUNS(dpp)}, {"v_add3_u32", "a three-operand VOP3", RUNS(vop3)},
                        {"v_cmp_lt_u32_e64", "a compare into an SGPR pair", RUNS(cmp)}, {"ds_read_b32", "an LDS read (not waited for)", RUNS(lds)}};
    printf("  \"run_length_sweep\": {\"what\": \"N alternating v_add_f32 / v_mul_f32 (VOP2, independent registers) followed by ONE instruction of another class, the block repeated 16 x 500 times; cycles per BLOCK and per instruction by waves per SIMD\", \"rows\": [\n");
    first = true;
    for (const Run &r : runs)
        for (int li = 0; li < 10; li++) {
            const int n = lens[li], per_block = n + (r.tag[0] == 'n' ? 0 : 1);
            double cyc[3];
            int wi = 0;
            for (int W : {1, 2, 4}) {
                const int threads = 64 * 4 * W, waves = cus * (threads / 64);
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(r.k[li], dim3(cus), dim3(threads), 0, nullptr, d_cycles, d_sink, iters);
                    CHECK(hipGetLastError());
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<unsigned long long> h((size_t)waves);
                CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * (size_t)waves, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                cyc[wi++] = ((double)h.back() - 32.0 * iters) / (W * 16.0 * iters);
            }
            printf("%s    {\"other\": \"%s\", \"run\": %d, \"cycles_per_block\": {\"1\": %.2f, \"2\": %.2f, \"4\": %.2f}, \"cycles_per_instruction\": {\"1\": %.2f, \"2\": %.2f, \"4\": %.2f}}",
                   first ? "" : ",\n", r.tag, n, cyc[0], cyc[1], cyc[2], cyc[0] / per_block, cyc[1] / per_block, cyc[2] / per_block);
            first = false;
        }
    printf("\n  ]},\n");
    // second batch: runs of 2 / 8 / 32 / 128 with one (or a few) instructions of many more classes
    struct Run4 { const char *tag; int extra; kern_t k[4]; };
    const int lens4[4] = {2, 8, 32, 128};
#define RUNS4(TAG) {k_r4_##TAG##_2, k_r4_##TAG##_8, k_r4_##TAG##_32, k_r4_##TAG##_128}
    const Run4 runs4[] = {
        {"v_mov_b32_dpp row_shr:1 (source never written)", 1, RUNS4(dpp)}, {"v_mov_b32_dpp quad_perm", 1, RUNS4(dppq)}, {"v_add_f32_dpp quad_perm", 1, RUNS4(dppadd)},
        {"v_mov_b32_dpp row_shr:1 bound_ctrl", 1, RUNS4(dppbc)}, {"v_mov_b32_dpp row_shr:1 of a register the run just wrote", 1, RUNS4(dppown)},
        {"v_mov_b32_dpp wave_shr:1 (the kernels' neighbour exchange)", 1, RUNS4(dppwshr)}, {"v_mov_b32_dpp row_shr:1 + s_nop 1", 2, RUNS4(dppnop)},
        {"v_readlane_b32", 1, RUNS4(readlane)}, {"v_readfirstlane_b32", 1, RUNS4(readfirst)}, {"ds_bpermute_b32 (not waited for)", 1, RUNS4(bperm)},
        {"ds_swizzle_b32 (not waited for)", 1, RUNS4(swizzle)}, {"v_permlane32_swap", 1, RUNS4(permlane)}, {"v_mov_b32_sdwa", 1, RUNS4(sdwa)},
        {"v_cvt_f32_i32", 1, RUNS4(cvt)}, {"v_rcp_f32", 1, RUNS4(rcp)}, {"v_max_f32", 1, RUNS4(maxf)}, {"v_lshlrev_b32", 1, RUNS4(lshl)},
        {"v_pk_add_f32", 1, RUNS4(pk)}, {"v_pk_mul_f32 + v_pk_add_f32", 2, RUNS4(pk2)}, {"s_add_u32", 1, RUNS4(salu)}, {"four SALU instructions", 4, RUNS4(salu4)},
        {"s_waitcnt lgkmcnt(0) (nothing outstanding)", 1, RUNS4(wait)}, {"v_cndmask_b32_e32 on vcc", 1, RUNS4(cndvcc)}, {"v_mov_b32", 1, RUNS4(mov)}};
    printf("  \"run_length_sweep_2\": {\"what\": \"as above with runs of 2 / 8 / 32 / 128 and more classes of the other instruction(s); cycles per block by waves per SIMD, and beside them what the block would cost at 2 cycles per VOP2 of the run alone (2 N)\", \"rows\": [\n");
    first = true;
    for (const Run4 &r : runs4)
        for (int li = 0; li < 4; li++) {
            const int n = lens4[li];
            double cyc[3];
            int wi = 0;
            for (int W : {1, 2, 4}) {
                const int threads = 64 * 4 * W, waves = cus * (threads / 64);
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(r.k[li], dim3(cus), dim3(threads), 0, nullptr, d_cycles, d_sink, iters);
                    CHECK(hipGetLastError());
                    CHECK(hipDeviceSynchronize());
                }
                std::vector<unsigned long long> h((size_t)waves);
                CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * (size_t)waves, hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                cyc[wi++] = ((double)h.back() - 32.0 * iters) / (W * 16.0 * iters);
            }
            printf("%s    {\"other\": \"%s\", \"other_instructions\": %d, \"run\": %d, \"run_alone_at_2_cycles\": %d, \"cycles_per_block\": {\"1\": %.2f, \"2\": %.2f, \"4\": %.2f}}",
                   first ? "" : ",\n", r.tag, r.extra, n, 2 * n, cyc[0], cyc[1], cyc[2]);
            first = false;
        }
    printf("\n  ]}\n}\n");
    return 0;
}
