// VALU issue-rate microbenchmark for gfx950 (developer tool): how many cycles does a wave64 instruction of each kind
// occupy its SIMD?  8 independent dependency chains per wave, no memory traffic.
//
// Layouts: every block has 256*k threads, i.e. k waves on each of the CU's 4 SIMDs (the hardware deals a workgroup's
// waves round-robin over the SIMDs), and `bpc` blocks per CU are launched, so waves per SIMD = k*bpc whatever the
// dispatcher does with single waves.  Two clocks per row:
//   * "event": hipEvent time of the whole launch x the NOMINAL clock / instructions per SIMD;
//   * "s_memtime": shader-clock ticks between the first and the last instruction of one wave of the launch, divided by
//     that wave's instructions and multiplied by the waves sharing its SIMD -- independent of the clock the chip holds.
// Control rows (v_fma_f32 / v_add_f32 / v_mul_f32) are the instructions /opt/skills/guides/MI355X_MICROARCH.md quotes
// at 2 cycles per wave64 instruction once two waves share a SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHAIN8(OP)                                                                                   \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                       \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(b), "v"(c));

#define OP_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_MAX_U32(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define OP_AND_B32(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_LSHLREV(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_PK_MAX_U16(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define OP_PK_MIN_I16(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_ADD_U16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define OP_PK_SUB_U16C(i) "v_pk_sub_u16 %" #i ", %" #i ", %8 clamp\n"
#define OP_PK_MAD_I16(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %9 op_sel_hi:[0,1,1]\n"
#define OP_PK_LSHL(i) "v_pk_lshlrev_b16 %" #i ", 8, %" #i " op_sel_hi:[0,1]\n"
#define OP_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define OP_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define OP_MAX3(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN_U16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define OP_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_DPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0\n"
#define OP_PK_ADD_F16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define OP_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"

#define ROWS(X)                                                                                                       \
  X(0, OP_FMA_F32, "v_fma_f32 (control)") X(1, OP_ADD_F32, "v_add_f32 (control)") X(2, OP_MUL_F32, "v_mul_f32 (control)") \
  X(3, OP_ADD_U32, "v_add_u32") X(4, OP_MAX_U32, "v_max_u32") X(5, OP_AND_B32, "v_and_b32") X(6, OP_LSHLREV, "v_lshlrev_b32") \
  X(7, OP_MOV, "v_mov_b32") X(8, OP_PK_MAX_U16, "v_pk_max_u16") X(9, OP_PK_MIN_I16, "v_pk_min_i16")                  \
  X(10, OP_PK_ADD_U16, "v_pk_add_u16") X(11, OP_PK_SUB_U16C, "v_pk_sub_u16 clamp") X(12, OP_PK_MAD_I16, "v_pk_mad_i16") \
  X(13, OP_PK_LSHL, "v_pk_lshlrev_b16") X(14, OP_ALIGNBYTE, "v_alignbyte_b32") X(15, OP_LSHL_OR, "v_lshl_or_b32")    \
  X(16, OP_BCNT, "v_bcnt_u32_b32") X(17, OP_MAX3, "v_max3_u32") X(18, OP_ADD3, "v_add3_u32") X(19, OP_MIN_U16, "v_min_u16") \
  X(20, OP_DOT4, "v_dot4_u32_u8") X(21, OP_CNDMASK, "v_cndmask_b32") X(22, OP_MUL24, "v_mul_u32_u24")                \
  X(23, OP_MAD_U24, "v_mad_u32_u24") X(24, OP_MULLO, "v_mul_lo_u32") X(25, OP_PERM, "v_perm_b32")                    \
  X(26, OP_DPP, "v_add_u32 dpp row_shr") X(27, OP_SDWA, "v_add_u32 sdwa") X(28, OP_PK_ADD_F16, "v_pk_add_f16")       \
  X(29, OP_CMP, "v_cmp_lt_u32 vcc")
#define OPX_V_SUB_U32(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define OPX_V_OR_B32(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define OPX_V_XOR_B32(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OPX_V_MIN_U32(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define OPX_V_MIN_I32(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define OPX_V_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define OPX_V_LSHRREV_B32(i) "v_lshrrev_b32 %" #i ", %" #i ", %8\n"
#define OPX_V_ASHRREV_I32(i) "v_ashrrev_i32 %" #i ", %" #i ", %8\n"
#define OPX_V_ADD_U16(i) "v_add_u16 %" #i ", %" #i ", %8\n"
#define OPX_V_SUB_U16(i) "v_sub_u16 %" #i ", %" #i ", %8\n"
#define OPX_V_MAX_U16(i) "v_max_u16 %" #i ", %" #i ", %8\n"
#define OPX_V_MIN_I16(i) "v_min_i16 %" #i ", %" #i ", %8\n"
#define OPX_V_MAX_I16(i) "v_max_i16 %" #i ", %" #i ", %8\n"
#define OPX_V_MUL_LO_U16(i) "v_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define OPX_V_LSHLREV_B16(i) "v_lshlrev_b16 %" #i ", %" #i ", %8\n"
#define OPX_V_LSHRREV_B16(i) "v_lshrrev_b16 %" #i ", %" #i ", %8\n"
#define OPX_V_ADD_F16(i) "v_add_f16 %" #i ", %" #i ", %8\n"
#define OPX_V_MUL_F16(i) "v_mul_f16 %" #i ", %" #i ", %8\n"
#define OPX_V_MAX_F16(i) "v_max_f16 %" #i ", %" #i ", %8\n"
#define OPX_V_MAX_F32(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define OPX_V_MIN_F32(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OPX_V_SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define OPX_V_FMAC_F32(i) "v_fmac_f32 %" #i ", %" #i ", %8\n"
#define OPX_V_MUL_I32_I24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define OPX_V_XNOR_B32(i) "v_xnor_b32 %" #i ", %" #i ", %8\n"
#define OPX_V_PK_MUL_LO_U16(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define OPX_V_PK_MAX_I16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define OPX_V_PK_SUB_I16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define OPX_V_PK_ASHRREV_I16(i) "v_pk_ashrrev_i16 %" #i ", %" #i ", %8\n"
#define OPX_V_ADD_U32_E64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define OPX_V_AND_OR_B32(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_OR3_B32(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_XAD_U32(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_BFE_U32(i) "v_bfe_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_BFI_B32(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MIN3_U32(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MED3_U32(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_SAD_U8(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_SAD_U32(i) "v_sad_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_ADD_LSHL_U32(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_LSHL_ADD_U32(i) "v_lshl_add_u32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MSAD_U8(i) "v_msad_u8 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MAD_U16(i) "v_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_FMA_F16(i) "v_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MAX3_F32(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MED3_F32(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_MAD_I32_I24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_PK_FMA_F16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define OPX_V_NOT_B32(i) "v_not_b32 %" #i ", %" #i "\n"
#define OPX_V_BFREV_B32(i) "v_bfrev_b32 %" #i ", %" #i "\n"
#define OPX_V_FFBH_U32(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define OPX_V_CVT_F32_U32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define OPX_V_CVT_F32_UBYTE0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define OPX_V_CVT_U32_F32(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define OPX_V_RNDNE_F32(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define OPX_V_MOV_B32_DPPX(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OPX_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[8:9]\n"
#define OPX_ADD_SGPR(i) "v_add_u32 %" #i ", s8, %" #i "\n"
#define OPX_ADD_LIT(i) "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define OPX_AND_LIT(i) "v_and_b32 %" #i ", 0xff00ff, %" #i "\n"
#define OPX_CMP_U16(i) "v_cmp_lt_u16 vcc, %" #i ", %8\n"
#define OPX_CMP64(i) "v_cmp_lt_u32_e64 s[10:11], %" #i ", %8\n"
#define OPX_MOV_SDWA(i) "v_mov_b32_sdwa %" #i ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n"
#define OPX_OR_SDWA(i) "v_or_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define ROWS2(X) \
  X(30, OPX_V_SUB_U32, "v_sub_u32") \
  X(31, OPX_V_OR_B32, "v_or_b32") \
  X(32, OPX_V_XOR_B32, "v_xor_b32") \
  X(33, OPX_V_MIN_U32, "v_min_u32") \
  X(34, OPX_V_MIN_I32, "v_min_i32") \
  X(35, OPX_V_MAX_I32, "v_max_i32") \
  X(36, OPX_V_LSHRREV_B32, "v_lshrrev_b32") \
  X(37, OPX_V_ASHRREV_I32, "v_ashrrev_i32") \
  X(38, OPX_V_ADD_U16, "v_add_u16") \
  X(39, OPX_V_SUB_U16, "v_sub_u16") \
  X(40, OPX_V_MAX_U16, "v_max_u16") \
  X(41, OPX_V_MIN_I16, "v_min_i16") \
  X(42, OPX_V_MAX_I16, "v_max_i16") \
  X(43, OPX_V_MUL_LO_U16, "v_mul_lo_u16") \
  X(44, OPX_V_LSHLREV_B16, "v_lshlrev_b16") \
  X(45, OPX_V_LSHRREV_B16, "v_lshrrev_b16") \
  X(46, OPX_V_ADD_F16, "v_add_f16") \
  X(47, OPX_V_MUL_F16, "v_mul_f16") \
  X(48, OPX_V_MAX_F16, "v_max_f16") \
  X(49, OPX_V_MAX_F32, "v_max_f32") \
  X(50, OPX_V_MIN_F32, "v_min_f32") \
  X(51, OPX_V_SUB_F32, "v_sub_f32") \
  X(52, OPX_V_FMAC_F32, "v_fmac_f32") \
  X(53, OPX_V_MUL_I32_I24, "v_mul_i32_i24") \
  X(54, OPX_V_XNOR_B32, "v_xnor_b32") \
  X(55, OPX_V_PK_MUL_LO_U16, "v_pk_mul_lo_u16") \
  X(56, OPX_V_PK_MAX_I16, "v_pk_max_i16") \
  X(57, OPX_V_PK_SUB_I16, "v_pk_sub_i16") \
  X(58, OPX_V_PK_ASHRREV_I16, "v_pk_ashrrev_i16") \
  X(60, OPX_V_ADD_U32_E64, "v_add_u32_e64 (VOP3 enc)") \
  X(61, OPX_V_AND_OR_B32, "v_and_or_b32") \
  X(62, OPX_V_OR3_B32, "v_or3_b32") \
  X(63, OPX_V_XAD_U32, "v_xad_u32") \
  X(64, OPX_V_BFE_U32, "v_bfe_u32") \
  X(65, OPX_V_BFI_B32, "v_bfi_b32") \
  X(66, OPX_V_MIN3_U32, "v_min3_u32") \
  X(67, OPX_V_MED3_U32, "v_med3_u32") \
  X(68, OPX_V_SAD_U8, "v_sad_u8") \
  X(69, OPX_V_SAD_U32, "v_sad_u32") \
  X(70, OPX_V_ADD_LSHL_U32, "v_add_lshl_u32") \
  X(71, OPX_V_LSHL_ADD_U32, "v_lshl_add_u32") \
  X(72, OPX_V_MSAD_U8, "v_msad_u8") \
  X(73, OPX_V_MAD_U16, "v_mad_u16") \
  X(74, OPX_V_FMA_F16, "v_fma_f16") \
  X(75, OPX_V_MAX3_F32, "v_max3_f32") \
  X(76, OPX_V_MED3_F32, "v_med3_f32") \
  X(77, OPX_V_MAD_I32_I24, "v_mad_i32_i24") \
  X(78, OPX_V_PK_FMA_F16, "v_pk_fma_f16") \
  X(79, OPX_V_NOT_B32, "v_not_b32") \
  X(80, OPX_V_BFREV_B32, "v_bfrev_b32") \
  X(81, OPX_V_FFBH_U32, "v_ffbh_u32") \
  X(82, OPX_V_CVT_F32_U32, "v_cvt_f32_u32") \
  X(83, OPX_V_CVT_F32_UBYTE0, "v_cvt_f32_ubyte0") \
  X(84, OPX_V_CVT_U32_F32, "v_cvt_u32_f32") \
  X(85, OPX_V_RNDNE_F32, "v_rndne_f32") \
  X(86, OPX_V_MOV_B32_DPPX, "v_mov_b32 dpp quad_perm") \
  X(87, OPX_CNDMASK64, "v_cndmask_b32 e64 sgpr mask") \
  X(88, OPX_ADD_SGPR, "v_add_u32 (sgpr src0)") \
  X(89, OPX_ADD_LIT, "v_add_u32 (literal)") \
  X(90, OPX_AND_LIT, "v_and_b32 (literal)") \
  X(91, OPX_CMP_U16, "v_cmp_lt_u16 vcc") \
  X(92, OPX_CMP64, "v_cmp_lt_u32 e64 sgpr dst") \
  X(93, OPX_MOV_SDWA, "v_mov_b32 sdwa byte insert") \
  X(94, OPX_OR_SDWA, "v_or_b32 sdwa")
#define NROWS 95

// 2-register-wide destination: separate kernel body
#define CHAIN4_PKF32                                                                                                 \
  asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\n"               \
               "v_pk_fma_f32 %3, %3, %4, %4\n"                                                                        \
               : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])                                                       \
               : "v"(q));

template <int K>
__global__ void bench(unsigned long long* out, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
  unsigned b = threadIdx.x | 0x00010001u, c = 0x00030005u;
  unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#define X(k, OP, name) if (K == k) CHAIN8(OP)
      ROWS(X)
      ROWS2(X)
#undef X
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x12345678u) out[threadIdx.x + 8] = s;
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[0] = t1 - t0;
}

__global__ void bench_pkf32(unsigned long long* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[4], q = {1.0001f, 0.9999f};
  for (int i = 0; i < 4; i++) p[i] = f2{(float)threadIdx.x + i, 1.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) CHAIN4_PKF32
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
  if (s == 1.2345f) out[threadIdx.x + 8] = 1;
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[0] = t1 - t0;
}

static hipDeviceProp_t g_prop;

template <class F>
void timeit(const char* name, unsigned long long* d, int k, int bpc, F launch) {
  const int iters = 4000;
  const int blocks = g_prop.multiProcessorCount * bpc, threads = 256 * k, w = k * bpc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(blocks, threads, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(blocks, threads, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long ticks = 0;
  hipMemcpy(&ticks, d, 8, hipMemcpyDeviceToHost);
  const double instsWave = (double)iters * 64, instsSimd = instsWave * w;
  const double clk = g_prop.clockRate * 1e3;
  printf("%-22s waves/SIMD %d (%d x %4d thr): %8.3f ms  event %.2f cyc/inst @%.0f MHz nominal | s_memtime %.2f ticks/inst/wave -> %.2f per SIMD slot\n",
         name, w, bpc, threads, ms, ms * 1e-3 * clk / instsSimd, clk / 1e6, (double)ticks / instsWave, (double)ticks / instsWave / w);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int K>
void run(const char* name, unsigned long long* d, int k, int bpc) {
  timeit(name, d, k, bpc, [&](int blocks, int threads, int iters) {
    hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  });
}

int main(int argc, char** argv) {
  hipGetDeviceProperties(&g_prop, 0);
  printf("# %s, %d CUs, clockRate %d kHz; s_memtime ticks: see the v_fma_f32 rows for its ratio to the event clock\n", g_prop.name,
         g_prop.multiProcessorCount, g_prop.clockRate);
  unsigned long long* d; hipMalloc(&d, 1 << 16);
  const int layouts[][2] = {{1, 1}, {2, 1}, {4, 1}, {4, 2}, {1, 2}, {1, 4}};
  const bool full = argc > 1 && !strcmp(argv[1], "--all-layouts");   // {k waves per SIMD per block, blocks per CU}
  for (auto& L : layouts) {
    const int k = L[0], bpc = L[1];
    if (!full && !((k == 1 && bpc == 1) || (k == 2 && bpc == 1) || (k == 4 && bpc == 1))) continue;
    const bool shortList = !(bpc == 1 || k == 4);   // the 64-thread-free alternative layouts: control rows + a few
#define X(kk, OP, name) if (!shortList || kk < 4 || kk == 8 || kk == 12 || kk == 14) run<kk>(name, d, k, bpc);
    ROWS(X)
    ROWS2(X)
#undef X
    timeit("v_pk_fma_f32", d, k, bpc, [&](int blocks, int threads, int iters) {
      hipLaunchKernelGGL(bench_pkf32, dim3(blocks), dim3(threads), 0, 0, d, iters);
    });
  }
  return 0;
}
