// VALU issue-rate microbenchmark for gfx950 (developer tool): how many cycles does a wave64 instruction of each kind
// occupy its SIMD?  One wave per SIMD slot x 8 waves, 8 independent dependency chains per wave, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAIN8(OP)                                                                                   \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                       \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(b), "v"(c));

#define OP_MAX_U32(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define OP_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_PK_MAX_U16(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define OP_PK_MIN_I16(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_SUB_U16C(i) "v_pk_sub_u16 %" #i ", %" #i ", %8 clamp\n"
#define OP_PK_MAD_I16(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %9 op_sel_hi:[0,1,1]\n"
#define OP_PK_LSHL(i) "v_pk_lshlrev_b16 %" #i ", 8, %" #i " op_sel_hi:[0,1]\n"
#define OP_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define OP_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define OP_MAX3(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN_U16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define OP_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_DPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP_PK_ADD_F16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define OP_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"

template <int K>
__global__ __launch_bounds__(64) void bench(unsigned* out, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
  unsigned b = threadIdx.x | 0x00010001u, c = 0x00030005u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (K == 0) CHAIN8(OP_MAX_U32)
      if (K == 1) CHAIN8(OP_ADD_U32)
      if (K == 2) CHAIN8(OP_PK_MAX_U16)
      if (K == 3) CHAIN8(OP_PK_MIN_I16)
      if (K == 4) CHAIN8(OP_PK_SUB_U16C)
      if (K == 5) CHAIN8(OP_PK_MAD_I16)
      if (K == 6) CHAIN8(OP_PK_LSHL)
      if (K == 7) CHAIN8(OP_ALIGNBYTE)
      if (K == 8) CHAIN8(OP_LSHL_OR)
      if (K == 9) CHAIN8(OP_BCNT)
      if (K == 10) CHAIN8(OP_MAX3)
      if (K == 11) CHAIN8(OP_MIN_U16)
      if (K == 12) CHAIN8(OP_DOT4)
      if (K == 13) CHAIN8(OP_CNDMASK)
      if (K == 14) CHAIN8(OP_MUL24)
      if (K == 15) CHAIN8(OP_MULLO)
      if (K == 16) CHAIN8(OP_DPP)
      if (K == 17) CHAIN8(OP_PK_ADD_F16)
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int K>
void run(const char* name, unsigned* d, int wavesPerSimd) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 4 * wavesPerSimd, iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(64), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)iters * 64 * wavesPerSimd;            // per SIMD
  const double clk = p.clockRate * 1e3;                               // Hz
  printf("%-16s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction at %.0f MHz (nominal)\n", name, wavesPerSimd, ms,
         ms * 1e-3 * clk / insts, clk / 1e6);
}

int main() {
  unsigned* d; hipMalloc(&d, 4096);
  for (int w : {1, 2, 4}) {
    run<0>("v_max_u32", d, w); run<1>("v_add_u32", d, w); run<2>("v_pk_max_u16", d, w); run<3>("v_pk_min_i16", d, w);
    run<4>("v_pk_sub_u16 clamp", d, w); run<5>("v_pk_mad_i16", d, w); run<6>("v_pk_lshlrev_b16", d, w); run<7>("v_alignbyte_b32", d, w);
    run<8>("v_lshl_or_b32", d, w); run<9>("v_bcnt_u32_b32", d, w); run<10>("v_max3_u32", d, w); run<11>("v_min_u16", d, w);
    run<12>("v_dot4_u32_u8", d, w); run<13>("v_cndmask_b32", d, w); run<14>("v_mul_u32_u24", d, w); run<15>("v_mul_lo_u32", d, w);
    run<16>("v_add_u32_dpp", d, w); run<17>("v_pk_add_f16", d, w);
  }
  return 0;
}
