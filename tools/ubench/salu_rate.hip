// Scalar-unit issue-rate microbenchmark for gfx950 (developer tool; companion of valu_rate.hip): how many cycles does a SCALAR instruction
// of each kind occupy, per SIMD and per CU, and does scalar work of one wave hide behind vector work of another?
//
// Why: k_fast_tasks issues 339 scalar instructions per cell-wave against 519 vector ones (profiles/r04j_pmc_summary.csv); the roofline
// priced the vector port only.  A CU has ONE scalar ALU for its four SIMDs, visited round-robin (one SIMD per cycle): at most one scalar
// instruction per SIMD per 4 cycles, the same cadence as the vector port -- if the two ports really issue side by side.
//
// Layouts as in valu_rate.hip: blocks of 256*k threads = k waves on each of the CU's 4 SIMDs, one block per CU, so waves per SIMD = k.
// Rows:
//   s_*            8 independent chains of one scalar opcode per wave (s_load rows: 8 loads in flight, then s_waitcnt lgkmcnt(0))
//   mix V|S        half of the waves of a SIMD run the v_and_b32 chain (a 2-cycle vector op: the worst case for port sharing), the other half the s_add_u32 chain; the
//                  time of each half alone is printed next to it: if the ports issue in parallel the mixed launch takes max(V, S),
//                  if they share an issue slot it takes V + S
//   one wave V+S   ONE instruction stream that alternates vector and scalar instructions (what a FAST wave's set-up looks like)
// Clocks: "event" = hipEvent time x nominal clock / instructions per SIMD; "s_memtime" = shader-clock ticks of one wave / its instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o salu_rate tools/ubench/salu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define SCHAIN8(OP)                                                                                                  \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                       \
               : "+s"(a[0]), "+s"(a[1]), "+s"(a[2]), "+s"(a[3]), "+s"(a[4]), "+s"(a[5]), "+s"(a[6]), "+s"(a[7])      \
               : "s"(b), "s"(c) : "scc");
#define VCHAIN8(OP)                                                                                                  \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                       \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])      \
               : "v"(vb));

#define S_ADD(i) "s_add_u32 %" #i ", %" #i ", %8\n"
#define S_ADDC(i) "s_addc_u32 %" #i ", %" #i ", %8\n"
#define S_AND(i) "s_and_b32 %" #i ", %" #i ", %8\n"
#define S_LSHL(i) "s_lshl_b32 %" #i ", %" #i ", 1\n"
#define S_MUL(i) "s_mul_i32 %" #i ", %" #i ", %8\n"
#define S_MULHI(i) "s_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define S_BFE(i) "s_bfe_u32 %" #i ", %" #i ", 0x100008\n"
#define S_MOV(i) "s_mov_b32 %" #i ", %8\n"
#define S_MOVK(i) "s_movk_i32 %" #i ", 0x1234\n"
#define S_MIN(i) "s_min_u32 %" #i ", %" #i ", %8\n"
#define S_LSHL_ADD(i) "s_lshl2_add_u32 %" #i ", %" #i ", %8\n"
#define S_CMP_CSEL(i) "s_cmp_lt_u32 %" #i ", %8\ns_cselect_b32 %" #i ", %" #i ", %9\n"
#define S_CMP(i) "s_cmp_lt_u32 %" #i ", %8\n"
#define S_ADDK(i) "s_addk_i32 %" #i ", 0x11\n"
#define S_BCNT(i) "s_bcnt1_i32_b32 %" #i ", %" #i "\n"
#define S_FF1(i) "s_ff1_i32_b32 %" #i ", %" #i "\n"
#define S_NOP(i) "s_nop 0\n"
#define S_WAIT(i) "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
#define V_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define V_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"

#define ROWS(X)                                                                                                     \
  X(0, S_ADD, "s_add_u32", 1) X(1, S_AND, "s_and_b32", 1) X(2, S_LSHL, "s_lshl_b32", 1) X(3, S_MUL, "s_mul_i32", 1)  \
  X(4, S_BFE, "s_bfe_u32", 1) X(5, S_MOV, "s_mov_b32", 1) X(6, S_MIN, "s_min_u32", 1)                                \
  X(7, S_CMP_CSEL, "s_cmp_lt_u32 + s_cselect_b32 (pair = 2 insts)", 2) X(8, S_CMP, "s_cmp_lt_u32", 1)               \
  X(9, S_ADDK, "s_addk_i32", 1) X(10, S_BCNT, "s_bcnt1_i32_b32", 1) X(11, S_FF1, "s_ff1_i32_b32", 1)                 \
  X(12, S_NOP, "s_nop 0", 1) X(13, S_WAIT, "s_waitcnt (nothing outstanding)", 1) X(14, S_MULHI, "s_mul_hi_u32", 1)  \
  X(15, S_MOVK, "s_movk_i32", 1) X(16, S_LSHL_ADD, "s_lshl2_add_u32", 1) X(17, S_ADDC, "s_addc_u32", 1)
#define NROWS 18

// MODE 0: every wave runs scalar row K.  MODE 1: every wave runs the vector chain VK (0: v_and_b32, 1: v_pk_max_u16).
// MODE 2: waves with an even index inside their SIMD run the vector chain, odd ones scalar row K (mix).  MODE 3: one stream, alternating.
template <int K, int MODE, int VK>
__global__ void bench(unsigned long long* out, int iters, const unsigned* tab) {
  unsigned a[8];
  unsigned v[8];
  for (int i = 0; i < 8; i++) { a[i] = __builtin_amdgcn_readfirstlane(blockIdx.x * 7 + i + 1); v[i] = threadIdx.x * 5 + i; }
  unsigned b = __builtin_amdgcn_readfirstlane(blockIdx.x | 0x10001u), c = __builtin_amdgcn_readfirstlane(3u);
  unsigned vb = threadIdx.x | 0x00010001u;
  const int waveInSimd = (threadIdx.x >> 6) >> 2;      // waves of a block are dealt round-robin over the 4 SIMDs
  const bool vecWave = MODE == 1 || (MODE == 2 && (waveInSimd & 1) == 0);
  unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 3) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        asm volatile("v_and_b32 %0, %0, %16\ns_add_u32 %8, %8, %17\nv_and_b32 %1, %1, %16\ns_add_u32 %9, %9, %17\n"
                     "v_and_b32 %2, %2, %16\ns_add_u32 %10, %10, %17\nv_and_b32 %3, %3, %16\ns_add_u32 %11, %11, %17\n"
                     "v_and_b32 %4, %4, %16\ns_add_u32 %12, %12, %17\nv_and_b32 %5, %5, %16\ns_add_u32 %13, %13, %17\n"
                     "v_and_b32 %6, %6, %16\ns_add_u32 %14, %14, %17\nv_and_b32 %7, %7, %16\ns_add_u32 %15, %15, %17\n"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                       "+s"(a[0]), "+s"(a[1]), "+s"(a[2]), "+s"(a[3]), "+s"(a[4]), "+s"(a[5]), "+s"(a[6]), "+s"(a[7])
                     : "v"(vb), "s"(b) : "scc");
    }
  } else if (vecWave) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (VK == 0) VCHAIN8(V_AND) else VCHAIN8(V_PKMAX)
      }
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
#define X(k, OP, name, n) if (K == k) SCHAIN8(OP)
        ROWS(X)
#undef X
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i] ^ v[i];
  if (s == 0x12345678u) out[threadIdx.x + 16] = s;
  // ticks of one vector wave ([0]) and one scalar wave ([1]) of the middle block
  if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) {
    if ((threadIdx.x >> 6) == 0) out[0] = t1 - t0;
    if ((threadIdx.x >> 6) == 4 || (blockDim.x == 256 && (threadIdx.x >> 6) == 0)) out[1] = t1 - t0;
  }
}

// s_load_dword: 8 loads in flight from a small table (scalar data cache hits), one wait per 8
__global__ void bench_sload(unsigned long long* out, int iters, const unsigned* tab) {
  unsigned acc = 0;
  const unsigned __attribute__((address_space(4)))* p = (const unsigned __attribute__((address_space(4)))*)(uintptr_t)tab;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      unsigned x0, x1, x2, x3, x4, x5, x6, x7;
      asm volatile("s_load_dword %0, %8, 0x0\ns_load_dword %1, %8, 0x40\ns_load_dword %2, %8, 0x80\ns_load_dword %3, %8, 0xc0\n"
                   "s_load_dword %4, %8, 0x100\ns_load_dword %5, %8, 0x140\ns_load_dword %6, %8, 0x180\ns_load_dword %7, %8, 0x1c0\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=s"(x0), "=s"(x1), "=s"(x2), "=s"(x3), "=s"(x4), "=s"(x5), "=s"(x6), "=s"(x7) : "s"(p) : "memory");
      acc ^= x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) out[threadIdx.x + 16] = acc;
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[0] = out[1] = t1 - t0;
}

static hipDeviceProp_t g_prop;
static const unsigned* g_tab;

struct Res { double ms, cycEvent, ticksV, ticksS; };

static int g_bpc = 1;   // blocks per CU (2 x 1024 threads = 8 waves per SIMD)

template <class F>
Res timeit(unsigned long long* d, int k, double instsPerWave, int wavesCounted, F launch) {
  const int iters = 3000;
  const int blocks = g_prop.multiProcessorCount * g_bpc, threads = 256 * k;
  wavesCounted *= g_bpc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(blocks, threads, 10);
  hipDeviceSynchronize();
  hipMemset(d, 0, 64);
  hipEventRecord(e0);
  launch(blocks, threads, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t[2] = {0, 0};
  hipMemcpy(t, d, 16, hipMemcpyDeviceToHost);
  hipEventDestroy(e0); hipEventDestroy(e1);
  const double clk = g_prop.clockRate * 1e3, insts = instsPerWave * iters;
  return {ms, ms * 1e-3 * clk / (insts * wavesCounted), (double)t[0] / insts, (double)t[1] / insts};
}

template <int K>
void run_row(const char* name, int n, unsigned long long* d, int k) {
  Res r = timeit(d, k, 64.0 * n, k, [&](int b, int t, int it) { hipLaunchKernelGGL((bench<K, 0, 0>), dim3(b), dim3(t), 0, 0, d, it, g_tab); });
  printf("%-48s waves/SIMD %d: %8.3f ms  event %.2f cyc/inst per SIMD (%.2f per CU) | s_memtime %.2f ticks/inst/wave\n", name, k * g_bpc, r.ms, r.cycEvent,
         r.cycEvent / 4, r.ticksS);
}

int main() {
  hipGetDeviceProperties(&g_prop, 0);
  printf("# %s, %d CUs, clockRate %d kHz (nominal; s_memtime ticks run at the 100 MHz reference: compare rows, not clocks)\n", g_prop.name,
         g_prop.multiProcessorCount, g_prop.clockRate);
  unsigned long long* d; hipMalloc(&d, 1 << 16);
  unsigned* tab; hipMalloc(&tab, 4096); hipMemset(tab, 1, 4096);
  g_tab = tab;
  for (int w : {1, 2, 4, 8}) {
    const int k = w == 8 ? 4 : w;     // 8 waves per SIMD = two 1024-thread blocks per CU
    g_bpc = w == 8 ? 2 : 1;
#define X(kk, OP, name, n) run_row<kk>(name, n, d, k);
    ROWS(X)
#undef X
    Res r = timeit(d, k, 64.0 + 8.0, k, [&](int b, int t, int it) { hipLaunchKernelGGL(bench_sload, dim3(b), dim3(t), 0, 0, d, it, g_tab); });
    printf("%-48s waves/SIMD %d: %8.3f ms  event %.2f cyc/inst per SIMD (8 s_load_dword + 1 s_waitcnt = 9 insts; %.1f cycles per group of 8 loads)\n",
           "s_load_dword x8 + s_waitcnt lgkmcnt(0)", k * g_bpc, r.ms, r.cycEvent, r.cycEvent * 9);
  }
  g_bpc = 1;
  // ---- do the vector and the scalar port issue side by side?  k waves per SIMD, half vector, half scalar
  for (int k : {2, 4}) {
    for (int vk = 0; vk < 2; vk++) {
      const char* vname = vk == 0 ? "v_and_b32 (2-cycle class)" : "v_pk_max_u16 (4-cycle class)";
      auto L = [&](auto kern) { return [=](int b, int t, int it) { hipLaunchKernelGGL(kern, dim3(b), dim3(t), 0, 0, d, it, g_tab); }; };
      Res v = vk == 0 ? timeit(d, k / 2, 64.0, k / 2, L(bench<0, 1, 0>)) : timeit(d, k / 2, 64.0, k / 2, L(bench<0, 1, 1>));
      Res s = timeit(d, k / 2, 64.0, k / 2, L(bench<0, 0, 0>));
      Res m = vk == 0 ? timeit(d, k, 64.0, k / 2, L(bench<0, 2, 0>)) : timeit(d, k, 64.0, k / 2, L(bench<0, 2, 1>));
      printf("mix: %d waves %s + %d waves s_add_u32 per SIMD: vector alone %.3f ms, scalar alone %.3f ms, together %.3f ms  -> %s (sum %.3f, max %.3f)\n",
             k / 2, vname, k / 2, v.ms, s.ms, m.ms, m.ms < 0.5 * (v.ms + s.ms + (v.ms > s.ms ? v.ms : s.ms)) ? "ports issue side by side" : "ports share the issue slot",
             v.ms + s.ms, v.ms > s.ms ? v.ms : s.ms);
    }
  }
  for (int k : {1, 2, 4}) {
    Res r = timeit(d, k, 64.0, k, [&](int b, int t, int it) { hipLaunchKernelGGL((bench<0, 3, 0>), dim3(b), dim3(t), 0, 0, d, it, g_tab); });
    printf("one stream alternating v_and_b32 / s_add_u32       waves/SIMD %d: %8.3f ms  event %.2f cyc/inst per SIMD (per instruction of either kind; a V + S pair = twice that)\n", k, r.ms, r.cycEvent);
  }
  return 0;
}
