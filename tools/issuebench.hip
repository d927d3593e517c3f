// issuebench.hip — what does one instruction of each kind COST on gfx950 when several wavefronts share a SIMD?
// The render kernels are bound by instruction issue in the dense regime; this measures the
// issue rates the compute roofline of bench.py is priced with: cycles per wave-instruction per SIMD for the VALU
// forms the kernels use (full rate, quarter rate, packed, cross-lane), per CU for SALU / SMEM, and whether VALU and
// SALU streams of different wavefronts overlap.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issuebench tools/issuebench.hip && /tmp/issuebench
// Every kernel: 256-thread workgroups (one wavefront per SIMD), W workgroups per CU (W wavefronts per SIMD), each
// wavefront runs ITERS x 64 copies of the instruction (8 independent chains). Reported: shader cycles per instruction
// per SIMD = wavefront cycles / (ITERS * 64 * W), from s_memtime, and the wall-clock equivalent.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HIP_OK(x)                                                        \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
             "v56", "v57", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", \
             "s54", "s55", "s56", "s57", "vcc", "memory"

// BODY: 8 instructions on independent registers; the kernel runs it 8 x ITERS times.
#define BENCH_KERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, const int* mem) {   \
        asm volatile(                                                                                   \
            "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n"       \
            "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n"       \
            "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n"       \
            "v_mov_b32 v52, 3\n v_mov_b32 v53, 5\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"               \
            "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n"                                                     \
            "s_mov_b32 s40, 1\n s_mov_b32 s41, 2\n s_mov_b32 s42, 3\n s_mov_b32 s43, 4\n"               \
            "s_mov_b32 s44, 5\n s_mov_b32 s45, 6\n s_mov_b32 s46, 7\n s_mov_b32 s47, 8\n"               \
            "s_mov_b32 s48, 0\n s_mov_b32 s49, 0\n s_mov_b32 s50, 0\n s_mov_b32 s51, 0\n"               \
            "s_mov_b32 s52, 0\n s_mov_b32 s53, 0\n s_mov_b32 s54, 0\n s_mov_b32 s55, 0\n" ::: CLOB);    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                 \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(".rept 8\n" BODY ".endr\n" ::"s"(mem) : CLOB);                                 \
        }                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                 \
        if ((threadIdx.x & 63) == 0) {                                                                  \
            const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);                               \
            out[w * 2] = t1 - t0;                                                                       \
            out[w * 2 + 1] = r1 - r0;                                                                   \
        }                                                                                               \
    }

// R8_n: the instruction on 8 independent register sets (v40..v47 / s40..s47), the varying register number n times
#define RI1(a, b, n) a #n b "\n"
#define RI2(a, b, c, n) a #n b #n c "\n"
#define R8_1(a, b) RI1(a, b, 40) RI1(a, b, 41) RI1(a, b, 42) RI1(a, b, 43) RI1(a, b, 44) RI1(a, b, 45) RI1(a, b, 46) RI1(a, b, 47)
#define R8_2(a, b, c) RI2(a, b, c, 40) RI2(a, b, c, 41) RI2(a, b, c, 42) RI2(a, b, c, 43) RI2(a, b, c, 44) RI2(a, b, c, 45) RI2(a, b, c, 46) RI2(a, b, c, 47)

BENCH_KERNEL(k_000, R8_2("v_add_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_001, R8_2("v_sub_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_002, R8_2("v_mul_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_003, R8_2("v_add_u32 v", ", v", ", v52"))
BENCH_KERNEL(k_004, R8_2("v_sub_u32 v", ", v", ", v52"))
BENCH_KERNEL(k_005, R8_2("v_and_b32 v", ", v", ", v52"))
BENCH_KERNEL(k_006, R8_2("v_or_b32 v", ", v", ", v52"))
BENCH_KERNEL(k_007, R8_2("v_xor_b32 v", ", v", ", v52"))
BENCH_KERNEL(k_008, R8_2("v_max_i32 v", ", v", ", v52"))
BENCH_KERNEL(k_009, R8_2("v_min_i32 v", ", v", ", v52"))
BENCH_KERNEL(k_010, R8_2("v_max_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_011, R8_2("v_min_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_012, R8_2("v_mul_i32_i24 v", ", v", ", v52"))
BENCH_KERNEL(k_013, R8_2("v_mul_u32_u24 v", ", v", ", v52"))
BENCH_KERNEL(k_014, R8_2("v_mul_lo_u32 v", ", v", ", v52"))
BENCH_KERNEL(k_015, R8_2("v_mul_hi_u32 v", ", v", ", v52"))
BENCH_KERNEL(k_016, R8_2("v_fmac_f32 v", ", v", ", v50"))
BENCH_KERNEL(k_017, R8_2("v_lshlrev_b32 v", ", 2, v", ""))
BENCH_KERNEL(k_018, R8_2("v_ashrrev_i32 v", ", 2, v", ""))
BENCH_KERNEL(k_019, R8_2("v_add_u32 v", ", 1, v", ""))
BENCH_KERNEL(k_020, R8_2("v_add_u32 v", ", 0x12345, v", ""))
BENCH_KERNEL(k_021, R8_2("v_add_f32 v", ", 1.0, v", ""))
BENCH_KERNEL(k_022, R8_2("v_add_u32 v", ", s40, v", ""))
BENCH_KERNEL(k_023, R8_2("v_mul_f32 v", ", s40, v", ""))
BENCH_KERNEL(k_024, R8_1("v_mov_b32 v", ", v50"))
BENCH_KERNEL(k_025, R8_1("v_mov_b32 v", ", 0"))
BENCH_KERNEL(k_026, R8_1("v_mov_b32 v", ", s40"))
BENCH_KERNEL(k_027, R8_2("v_fma_f32 v", ", v", ", v50, v51"))
BENCH_KERNEL(k_028, R8_2("v_fma_f32 v", ", v", ", s40, v51"))
BENCH_KERNEL(k_029, R8_2("v_add_f32_e64 v", ", |v", "|, v50"))
BENCH_KERNEL(k_030, R8_2("v_mad_u32_u24 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_031, R8_2("v_mad_i32_i24 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_032, R8_2("v_add3_u32 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_033, R8_2("v_lshl_add_u32 v", ", v", ", 2, v53"))
BENCH_KERNEL(k_034, R8_2("v_lshl_or_b32 v", ", v", ", 8, v53"))
BENCH_KERNEL(k_035, R8_2("v_and_or_b32 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_036, R8_2("v_or3_b32 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_037, R8_2("v_bfe_u32 v", ", v", ", 4, 8"))
BENCH_KERNEL(k_038, R8_2("v_bfe_i32 v", ", v", ", 0, 16"))
BENCH_KERNEL(k_039, R8_2("v_med3_f32 v", ", v", ", v50, v51"))
BENCH_KERNEL(k_040, R8_2("v_min3_f32 v", ", v", ", v50, v51"))
BENCH_KERNEL(k_041, R8_2("v_max3_i32 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_042, R8_2("v_perm_b32 v", ", v", ", v52, v53"))
BENCH_KERNEL(k_043, R8_2("v_alignbit_b32 v", ", v", ", v52, 8"))
BENCH_KERNEL(k_044, R8_1("v_cvt_f32_i32 v", ", v52"))
BENCH_KERNEL(k_045, R8_1("v_cvt_i32_f32 v", ", v50"))
BENCH_KERNEL(k_046, R8_1("v_cvt_f32_ubyte0 v", ", v52"))
BENCH_KERNEL(k_047, R8_1("v_cvt_f32_ubyte1 v", ", v52"))
BENCH_KERNEL(k_048, R8_1("v_cvt_f32_u32 v", ", v52"))
BENCH_KERNEL(k_049, R8_2("v_rcp_f32 v", ", v", ""))
BENCH_KERNEL(k_050, R8_2("v_rcp_iflag_f32 v", ", v", ""))
BENCH_KERNEL(k_051, R8_1("v_cmp_lt_i32 vcc, v", ", v52"))
BENCH_KERNEL(k_052, R8_1("v_cmp_lt_u32 vcc, v", ", v52"))
BENCH_KERNEL(k_053, R8_1("v_cmp_lt_f32 vcc, v", ", v50"))
BENCH_KERNEL(k_054, R8_1("v_cmp_lt_i32 vcc, s40, v", ""))
BENCH_KERNEL(k_055, R8_1("v_cmp_class_f32 vcc, v", ", v52"))
BENCH_KERNEL(k_056, "v_cmp_lt_i32 s[40:41], v40, v52\n v_cmp_lt_i32 s[42:43], v41, v52\n v_cmp_lt_i32 s[44:45], v42, v52\n v_cmp_lt_i32 s[46:47], v43, v52\n" "v_cmp_lt_i32 s[48:49], v44, v52\n v_cmp_lt_i32 s[50:51], v45, v52\n v_cmp_lt_i32 s[52:53], v46, v52\n v_cmp_lt_i32 s[54:55], v47, v52\n")
BENCH_KERNEL(k_057, R8_2("v_cndmask_b32 v", ", v", ", v50, vcc"))
BENCH_KERNEL(k_058, R8_2("v_cndmask_b32_e64 v", ", v", ", v50, s[48:49]"))
BENCH_KERNEL(k_059, R8_2("v_cndmask_b32 v", ", v51, v50, vcc ; v", ""))
BENCH_KERNEL(k_060, "v_cmp_lt_i32 vcc, v40, v52\n v_cndmask_b32 v41, v41, v50, vcc\n v_cndmask_b32 v42, v42, v50, vcc\n v_cndmask_b32 v43, v43, v50, vcc\n" "v_cmp_lt_i32 vcc, v44, v52\n v_cndmask_b32 v45, v45, v50, vcc\n v_cndmask_b32 v46, v46, v50, vcc\n v_cndmask_b32 v47, v47, v50, vcc\n")
BENCH_KERNEL(k_061, "v_cmp_lt_i32 vcc, v40, v52\n s_and_saveexec_b64 s[40:41], vcc\n v_mov_b32 v41, v50\n v_mov_b32 v42, v50\n v_mov_b32 v43, v50\n s_or_b64 exec, exec, s[40:41]\n s_nop 0\n s_nop 0\n")
BENCH_KERNEL(k_062, "v_pk_mul_f32 v[40:41], v[40:41], v[50:51]\n v_pk_mul_f32 v[42:43], v[42:43], v[50:51]\n v_pk_mul_f32 v[44:45], v[44:45], v[50:51]\n v_pk_mul_f32 v[46:47], v[46:47], v[50:51]\n" "v_pk_mul_f32 v[48:49], v[48:49], v[50:51]\n v_pk_mul_f32 v[54:55], v[54:55], v[50:51]\n v_pk_mul_f32 v[56:57], v[56:57], v[50:51]\n v_pk_mul_f32 v[40:41], v[40:41], v[50:51]\n")
BENCH_KERNEL(k_063, "v_pk_add_f32 v[40:41], v[40:41], v[50:51]\n v_pk_add_f32 v[42:43], v[42:43], v[50:51]\n v_pk_add_f32 v[44:45], v[44:45], v[50:51]\n v_pk_add_f32 v[46:47], v[46:47], v[50:51]\n" "v_pk_add_f32 v[48:49], v[48:49], v[50:51]\n v_pk_add_f32 v[54:55], v[54:55], v[50:51]\n v_pk_add_f32 v[56:57], v[56:57], v[50:51]\n v_pk_add_f32 v[40:41], v[40:41], v[50:51]\n")
BENCH_KERNEL(k_064, "v_pk_fma_f32 v[40:41], v[40:41], v[50:51], v[50:51]\n v_pk_fma_f32 v[42:43], v[42:43], v[50:51], v[50:51]\n v_pk_fma_f32 v[44:45], v[44:45], v[50:51], v[50:51]\n v_pk_fma_f32 v[46:47], v[46:47], v[50:51], v[50:51]\n" "v_pk_fma_f32 v[48:49], v[48:49], v[50:51], v[50:51]\n v_pk_fma_f32 v[54:55], v[54:55], v[50:51], v[50:51]\n v_pk_fma_f32 v[56:57], v[56:57], v[50:51], v[50:51]\n v_pk_fma_f32 v[40:41], v[40:41], v[50:51], v[50:51]\n")
BENCH_KERNEL(k_065, "v_pk_add_f32 v[40:41], s[40:41], v[50:51]\n v_pk_add_f32 v[42:43], s[42:43], v[50:51]\n v_pk_add_f32 v[44:45], s[44:45], v[50:51]\n v_pk_add_f32 v[46:47], s[46:47], v[50:51]\n" "v_pk_add_f32 v[48:49], s[40:41], v[50:51]\n v_pk_add_f32 v[54:55], s[42:43], v[50:51]\n v_pk_add_f32 v[56:57], s[44:45], v[50:51]\n v_pk_add_f32 v[40:41], s[46:47], v[50:51]\n")
BENCH_KERNEL(k_066, "v_pk_mul_f32 v[40:41], v[40:41], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[42:43], v[42:43], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[44:45], v[44:45], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[46:47], v[46:47], v[50:51] op_sel_hi:[1,0]\n" "v_pk_mul_f32 v[48:49], v[48:49], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[54:55], v[54:55], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[56:57], v[56:57], v[50:51] op_sel_hi:[1,0]\n v_pk_mul_f32 v[40:41], v[40:41], v[50:51] op_sel_hi:[1,0]\n")
BENCH_KERNEL(k_067, "v_lshl_add_u64 v[40:41], v[40:41], 2, v[50:51]\n v_lshl_add_u64 v[42:43], v[42:43], 2, v[50:51]\n v_lshl_add_u64 v[44:45], v[44:45], 2, v[50:51]\n v_lshl_add_u64 v[46:47], v[46:47], 2, v[50:51]\n" "v_lshl_add_u64 v[48:49], v[48:49], 2, v[50:51]\n v_lshl_add_u64 v[54:55], v[54:55], 2, v[50:51]\n v_lshl_add_u64 v[56:57], v[56:57], 2, v[50:51]\n v_lshl_add_u64 v[40:41], v[40:41], 2, v[50:51]\n")
BENCH_KERNEL(k_068, "v_mad_u64_u32 v[40:41], s[48:49], v52, v53, v[40:41]\n v_mad_u64_u32 v[42:43], s[48:49], v52, v53, v[42:43]\n v_mad_u64_u32 v[44:45], s[48:49], v52, v53, v[44:45]\n v_mad_u64_u32 v[46:47], s[48:49], v52, v53, v[46:47]\n" "v_mad_u64_u32 v[48:49], s[48:49], v52, v53, v[48:49]\n v_mad_u64_u32 v[54:55], s[48:49], v52, v53, v[54:55]\n v_mad_u64_u32 v[56:57], s[48:49], v52, v53, v[56:57]\n v_mad_u64_u32 v[40:41], s[48:49], v52, v53, v[40:41]\n")
BENCH_KERNEL(k_069, "v_readlane_b32 s40, v40, 3\n v_readlane_b32 s41, v41, 5\n v_readlane_b32 s42, v42, 7\n v_readlane_b32 s43, v43, 9\n" "v_readlane_b32 s44, v44, 11\n v_readlane_b32 s45, v45, 13\n v_readlane_b32 s46, v46, 15\n v_readlane_b32 s47, v47, 17\n")
BENCH_KERNEL(k_070, "v_readlane_b32 s40, v40, s48\n v_readlane_b32 s41, v41, s48\n v_readlane_b32 s42, v42, s48\n v_readlane_b32 s43, v43, s48\n" "v_readlane_b32 s44, v44, s48\n v_readlane_b32 s45, v45, s48\n v_readlane_b32 s46, v46, s48\n v_readlane_b32 s47, v47, s48\n")
BENCH_KERNEL(k_071, "v_readfirstlane_b32 s40, v40\n v_readfirstlane_b32 s41, v41\n v_readfirstlane_b32 s42, v42\n v_readfirstlane_b32 s43, v43\n" "v_readfirstlane_b32 s44, v44\n v_readfirstlane_b32 s45, v45\n v_readfirstlane_b32 s46, v46\n v_readfirstlane_b32 s47, v47\n")
BENCH_KERNEL(k_072, "v_writelane_b32 v40, s40, 3\n v_writelane_b32 v41, s41, 5\n v_writelane_b32 v42, s42, 7\n v_writelane_b32 v43, s43, 9\n" "v_writelane_b32 v44, s44, 11\n v_writelane_b32 v45, s45, 13\n v_writelane_b32 v46, s46, 15\n v_writelane_b32 v47, s47, 17\n")
BENCH_KERNEL(k_073, R8_2("v_mov_b32_dpp v", ", v50 row_shr:1 row_mask:0xf bank_mask:0xf ; v", ""))
BENCH_KERNEL(k_074, R8_2("v_add_u32_dpp v", ", v52, v", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
BENCH_KERNEL(k_075, R8_2("v_add_u32_sdwa v", ", v", ", v52 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"))
BENCH_KERNEL(k_076, "ds_bpermute_b32 v40, v54, v50\n ds_bpermute_b32 v41, v54, v50\n ds_bpermute_b32 v42, v54, v50\n ds_bpermute_b32 v43, v54, v50\n" "ds_bpermute_b32 v44, v54, v50\n ds_bpermute_b32 v45, v54, v50\n ds_bpermute_b32 v46, v54, v50\n ds_bpermute_b32 v47, v54, v50\n s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_077, "ds_swizzle_b32 v40, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v41, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v42, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v43, v50 offset:swizzle(SWAP,1)\n" "ds_swizzle_b32 v44, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v45, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v46, v50 offset:swizzle(SWAP,1)\n ds_swizzle_b32 v47, v50 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_078, "global_load_dword v40, v56, %0\n global_load_dword v41, v56, %0\n global_load_dword v42, v56, %0\n global_load_dword v43, v56, %0\n" "global_load_dword v44, v56, %0\n global_load_dword v45, v56, %0\n global_load_dword v46, v56, %0\n global_load_dword v47, v56, %0\n s_waitcnt vmcnt(0)\n")
BENCH_KERNEL(k_079, "global_load_dwordx4 v[40:43], v56, %0\n global_load_dwordx4 v[44:47], v56, %0\n global_load_dwordx4 v[40:43], v56, %0\n global_load_dwordx4 v[44:47], v56, %0\n s_waitcnt vmcnt(0)\n" "global_load_dwordx4 v[40:43], v56, %0\n global_load_dwordx4 v[44:47], v56, %0\n global_load_dwordx4 v[40:43], v56, %0\n global_load_dwordx4 v[44:47], v56, %0\n s_waitcnt vmcnt(0)\n")
BENCH_KERNEL(k_080, R8_2("s_add_u32 s", ", s", ", 1"))
BENCH_KERNEL(k_081, R8_2("s_mul_i32 s", ", s", ", 3"))
BENCH_KERNEL(k_082, R8_2("s_lshl_b32 s", ", s", ", 1"))
BENCH_KERNEL(k_083, R8_2("s_sext_i32_i16 s", ", s", ""))
BENCH_KERNEL(k_084, R8_2("s_bfe_u32 s", ", s", ", 0x80004"))
BENCH_KERNEL(k_085, "s_and_b64 s[40:41], s[40:41], s[48:49]\n s_or_b64 s[42:43], s[42:43], s[48:49]\n s_andn2_b64 s[44:45], s[44:45], s[48:49]\n s_and_b64 s[46:47], s[46:47], s[48:49]\n" "s_or_b64 s[50:51], s[50:51], s[48:49]\n s_and_b64 s[52:53], s[52:53], s[48:49]\n s_or_b64 s[54:55], s[54:55], s[48:49]\n s_and_b64 s[40:41], s[40:41], s[48:49]\n")
BENCH_KERNEL(k_086, "s_ff1_i32_b64 s40, s[48:49]\n s_ff1_i32_b64 s41, s[48:49]\n s_ff1_i32_b64 s42, s[48:49]\n s_ff1_i32_b64 s43, s[48:49]\n" "s_ff1_i32_b64 s44, s[48:49]\n s_ff1_i32_b64 s45, s[48:49]\n s_ff1_i32_b64 s46, s[48:49]\n s_ff1_i32_b64 s47, s[48:49]\n")
BENCH_KERNEL(k_087, "s_bcnt1_i32_b64 s40, s[48:49]\n s_bcnt1_i32_b64 s41, s[48:49]\n s_bcnt1_i32_b64 s42, s[48:49]\n s_bcnt1_i32_b64 s43, s[48:49]\n" "s_bcnt1_i32_b64 s44, s[48:49]\n s_bcnt1_i32_b64 s45, s[48:49]\n s_bcnt1_i32_b64 s46, s[48:49]\n s_bcnt1_i32_b64 s47, s[48:49]\n")
BENCH_KERNEL(k_088, "s_cmp_lt_i32 s40, s41\n s_cselect_b32 s42, s43, s44\n s_cmp_lt_i32 s45, s46\n s_cselect_b32 s47, s48, s49\n" "s_cmp_lt_i32 s40, s41\n s_cselect_b32 s50, s43, s44\n s_cmp_lt_i32 s45, s46\n s_cselect_b32 s51, s48, s49\n")
BENCH_KERNEL(k_089, "s_cmp_eq_u32 s48, 1\n s_cbranch_scc1 9f\n s_cmp_eq_u32 s48, 1\n s_cbranch_scc1 9f\n" "s_cmp_eq_u32 s48, 1\n s_cbranch_scc1 9f\n s_cmp_eq_u32 s48, 1\n s_cbranch_scc1 9f\n 9:\n")
BENCH_KERNEL(k_090, "s_branch 1f\n s_nop 0\n 1: s_branch 2f\n s_nop 0\n 2: s_branch 3f\n s_nop 0\n 3: s_branch 4f\n s_nop 0\n 4: \n" "s_branch 5f\n s_nop 0\n 5: s_branch 6f\n s_nop 0\n 6: s_branch 7f\n s_nop 0\n 7: s_branch 8f\n s_nop 0\n 8:\n")
BENCH_KERNEL(k_091, "s_and_saveexec_b64 s[40:41], s[48:49]\n s_or_b64 exec, exec, s[40:41]\n s_and_saveexec_b64 s[42:43], s[48:49]\n s_or_b64 exec, exec, s[42:43]\n" "s_and_saveexec_b64 s[44:45], s[48:49]\n s_or_b64 exec, exec, s[44:45]\n s_and_saveexec_b64 s[46:47], s[48:49]\n s_or_b64 exec, exec, s[46:47]\n")
BENCH_KERNEL(k_092, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
BENCH_KERNEL(k_093, "s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_094, "s_load_dwordx8 s[40:47], %0, 0x0\n s_load_dwordx8 s[48:55], %0, 0x20\n s_load_dwordx8 s[40:47], %0, 0x40\n s_load_dwordx8 s[48:55], %0, 0x60\n" "s_load_dwordx8 s[40:47], %0, 0x0\n s_load_dwordx8 s[48:55], %0, 0x20\n s_load_dwordx8 s[40:47], %0, 0x40\n s_load_dwordx8 s[48:55], %0, 0x60\n s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_095, "s_load_dword s40, %0, 0x0\n s_load_dword s41, %0, 0x20\n s_load_dword s42, %0, 0x40\n s_load_dword s43, %0, 0x60\n" "s_load_dword s44, %0, 0x0\n s_load_dword s45, %0, 0x20\n s_load_dword s46, %0, 0x40\n s_load_dword s47, %0, 0x60\n s_waitcnt lgkmcnt(0)\n")
BENCH_KERNEL(k_096, "v_add_f32 v40, v40, v50\n s_add_u32 s40, s40, 1\n v_add_f32 v41, v41, v50\n s_add_u32 s41, s41, 1\n" "v_add_f32 v42, v42, v50\n s_add_u32 s42, s42, 1\n v_add_f32 v43, v43, v50\n s_add_u32 s43, s43, 1\n")
BENCH_KERNEL(k_097, "v_min_f32 v40, v40, v50\n s_add_u32 s40, s40, 1\n v_min_f32 v41, v41, v50\n s_add_u32 s41, s41, 1\n" "v_min_f32 v42, v42, v50\n s_add_u32 s42, s42, 1\n v_min_f32 v43, v43, v50\n s_add_u32 s43, s43, 1\n")
BENCH_KERNEL(k_098, "v_add_f32 v40, v40, v50\n v_add_f32 v41, v41, v50\n v_add_f32 v42, v42, v50\n s_add_u32 s40, s40, 1\n" "v_add_f32 v43, v43, v50\n v_add_f32 v44, v44, v50\n v_add_f32 v45, v45, v50\n s_add_u32 s41, s41, 1\n")
BENCH_KERNEL(k_099, "v_min_f32 v40, v40, v50\n v_add_f32 v41, v41, v50\n v_min_f32 v42, v42, v50\n v_add_f32 v43, v43, v50\n" "v_min_f32 v44, v44, v50\n v_add_f32 v45, v45, v50\n v_min_f32 v46, v46, v50\n v_add_f32 v47, v47, v50\n")

typedef void (*kern_t)(unsigned long long*, int, const int*);
struct Case {
    const char* name;
    kern_t fn;
    const char* unit;  // what a "per-SIMD" figure means for it
};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"instructions_per_wave\": %d, \"results\": [\n", prop.gcnArchName, cus, iters, iters * 64);
    unsigned long long* out;
    int* mem;
    HIP_OK(hipMalloc(&out, sizeof(unsigned long long) * 2 * 4 * cus * 8));
    HIP_OK(hipMalloc(&mem, 4096));
    HIP_OK(hipMemset(mem, 0, 4096));
    const Case cases[] = {
        {"v_add_f32", k_000, "valu"},
        {"v_sub_f32", k_001, "valu"},
        {"v_mul_f32", k_002, "valu"},
        {"v_add_u32", k_003, "valu"},
        {"v_sub_u32", k_004, "valu"},
        {"v_and_b32", k_005, "valu"},
        {"v_or_b32", k_006, "valu"},
        {"v_xor_b32", k_007, "valu"},
        {"v_max_i32", k_008, "valu"},
        {"v_min_i32", k_009, "valu"},
        {"v_max_f32", k_010, "valu"},
        {"v_min_f32", k_011, "valu"},
        {"v_mul_i32_i24", k_012, "valu"},
        {"v_mul_u32_u24", k_013, "valu"},
        {"v_mul_lo_u32", k_014, "valu"},
        {"v_mul_hi_u32", k_015, "valu"},
        {"v_fmac_f32", k_016, "valu"},
        {"v_lshlrev_b32 (const shift)", k_017, "valu"},
        {"v_ashrrev_i32 (const shift)", k_018, "valu"},
        {"v_add_u32 inline const", k_019, "valu"},
        {"v_add_u32 literal", k_020, "valu"},
        {"v_add_f32 inline const", k_021, "valu"},
        {"v_add_u32 sgpr", k_022, "valu"},
        {"v_mul_f32 sgpr", k_023, "valu"},
        {"v_mov_b32 vgpr", k_024, "valu"},
        {"v_mov_b32 inline const", k_025, "valu"},
        {"v_mov_b32 sgpr", k_026, "valu"},
        {"v_fma_f32", k_027, "valu"},
        {"v_fma_f32 one sgpr", k_028, "valu"},
        {"v_add_f32 |abs| (e64)", k_029, "valu"},
        {"v_mad_u32_u24", k_030, "valu"},
        {"v_mad_i32_i24", k_031, "valu"},
        {"v_add3_u32", k_032, "valu"},
        {"v_lshl_add_u32", k_033, "valu"},
        {"v_lshl_or_b32", k_034, "valu"},
        {"v_and_or_b32", k_035, "valu"},
        {"v_or3_b32", k_036, "valu"},
        {"v_bfe_u32", k_037, "valu"},
        {"v_bfe_i32", k_038, "valu"},
        {"v_med3_f32", k_039, "valu"},
        {"v_min3_f32", k_040, "valu"},
        {"v_max3_i32", k_041, "valu"},
        {"v_perm_b32", k_042, "valu"},
        {"v_alignbit_b32", k_043, "valu"},
        {"v_cvt_f32_i32", k_044, "valu"},
        {"v_cvt_i32_f32", k_045, "valu"},
        {"v_cvt_f32_ubyte0", k_046, "valu"},
        {"v_cvt_f32_ubyte1", k_047, "valu"},
        {"v_cvt_f32_u32", k_048, "valu"},
        {"v_rcp_f32", k_049, "valu"},
        {"v_rcp_iflag_f32", k_050, "valu"},
        {"v_cmp_lt_i32 -> vcc", k_051, "valu"},
        {"v_cmp_lt_u32 -> vcc", k_052, "valu"},
        {"v_cmp_lt_f32 -> vcc", k_053, "valu"},
        {"v_cmp_lt_i32 sgpr,v -> vcc", k_054, "valu"},
        {"v_cmp_class_f32 -> vcc", k_055, "valu"},
        {"v_cmp_lt_i32 -> sgpr pair (e64)", k_056, "valu"},
        {"v_cndmask_b32 vcc (e32)", k_057, "valu"},
        {"v_cndmask_b32 sgpr pair (e64)", k_058, "valu"},
        {"v_cndmask_b32 vcc, other dst", k_059, "valu"},
        {"1 v_cmp + 3 v_cndmask", k_060, "valu"},
        {"1 v_cmp + saveexec + 3 v_mov + restore", k_061, "mix"},
        {"v_pk_mul_f32", k_062, "valu"},
        {"v_pk_add_f32", k_063, "valu"},
        {"v_pk_fma_f32", k_064, "valu"},
        {"v_pk_add_f32 sgpr pair", k_065, "valu"},
        {"v_pk_mul_f32 op_sel broadcast", k_066, "valu"},
        {"v_lshl_add_u64", k_067, "valu"},
        {"v_mad_u64_u32", k_068, "valu"},
        {"v_readlane_b32", k_069, "valu"},
        {"v_readlane_b32 sgpr lane", k_070, "valu"},
        {"v_readfirstlane_b32", k_071, "valu"},
        {"v_writelane_b32", k_072, "valu"},
        {"v_mov_b32 dpp row_shr:1", k_073, "valu"},
        {"v_add_u32 dpp quad_perm", k_074, "valu"},
        {"v_add_u32 sdwa", k_075, "valu"},
        {"ds_bpermute_b32", k_076, "lds"},
        {"ds_swizzle_b32", k_077, "lds"},
        {"global_load_dword (cache hit, 8 in flight)", k_078, "vmem"},
        {"global_load_dwordx4 (cache hit, 4 in flight)", k_079, "vmem"},
        {"s_add_u32", k_080, "salu"},
        {"s_mul_i32", k_081, "salu"},
        {"s_lshl_b32", k_082, "salu"},
        {"s_sext_i32_i16", k_083, "salu"},
        {"s_bfe_u32", k_084, "salu"},
        {"s_and/or/andn2_b64", k_085, "salu"},
        {"s_ff1_i32_b64", k_086, "salu"},
        {"s_bcnt1_i32_b64", k_087, "salu"},
        {"s_cmp + s_cselect_b32", k_088, "salu"},
        {"s_cmp + s_cbranch not taken", k_089, "salu"},
        {"s_branch taken (each skips one instruction)", k_090, "salu"},
        {"s_and_saveexec + s_or exec pairs", k_091, "salu"},
        {"s_nop 0", k_092, "salu"},
        {"s_waitcnt (nothing outstanding)", k_093, "salu"},
        {"s_load_dwordx8 (scalar cache hit, 8 in flight)", k_094, "smem"},
        {"s_load_dword (scalar cache hit, 8 in flight)", k_095, "smem"},
        {"1 v_add_f32 : 1 s_add_u32", k_096, "mix"},
        {"1 v_min_f32 : 1 s_add_u32", k_097, "mix"},
        {"3 v_add_f32 : 1 s_add_u32", k_098, "mix"},
        {"1 v_min_f32 : 1 v_add_f32", k_099, "mix"},
    };
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    const int n_cases = (int)(sizeof(cases) / sizeof(cases[0]));
    for (int c = 0; c < n_cases; c++) {
        std::printf(" {\"op\": \"%s\", \"kind\": \"%s\"", cases[c].name, cases[c].unit);
        for (int w : {1, 4, 8}) {
            const int blocks = cus * w;
            // warm-up, then the measured launch
            hipLaunchKernelGGL(cases[c].fn, dim3(blocks), dim3(256), 0, 0, out, iters / 10 + 1, mem);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(cases[c].fn, dim3(blocks), dim3(256), 0, 0, out, iters, mem);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipDeviceSynchronize());
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
            HIP_OK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> cyc, ghz;
            for (size_t i = 0; i < h.size(); i += 2) {
                cyc.push_back((double)h[i]);
                if (h[i + 1] > 0) ghz.push_back((double)h[i] / ((double)h[i + 1] * 10.0) );  // cycles per ns (100 MHz ticks)
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(ghz.begin(), ghz.end());
            const double med = cyc[cyc.size() / 2];
            const double n_inst = (double)iters * 64.0;
            const double g = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
            // per SIMD: w wavefronts each issued n_inst instructions; the launch took `ms` at `g` GHz (the wall-clock
            // figure is the one to use: with many wavefronts per SIMD they do not all run from start to end together)
            std::printf(", \"w%d\": {\"cyc_per_inst_per_simd\": %.3f, \"wave_cyc_per_inst\": %.2f, \"ghz\": %.3f, \"wall_us\": %.1f}", w,
                        (double)ms * 1e6 * g / (n_inst * w), med / n_inst, g, ms * 1e3);
        }
        std::printf("}%s\n", c + 1 < n_cases ? "," : "");
    }
    std::printf("]}\n");
    return 0;
}
