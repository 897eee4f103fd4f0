// dispatch_rate.hip -- how fast does the chip refill wave slots with single-wave workgroups?  Each wave of the test kernel
// allocates `lds` bytes of LDS, keeps `vg` VGPRs alive, spins for about `cycles` shader cycles (s_memtime) and exits; the
// grid has `waves` workgroups of 64 threads.  If slots were refilled instantly the launch would take
// waves * cycles / (resident slots) -- the ratio of that to the measured duration is the achieved fraction of the residency
// the resources allow.   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_rate.hip -o tools/ubench/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int VG, int SG = 0>
__global__ __launch_bounds__(512) void k_spin(unsigned* out, int cycles, int touch) {
  extern __shared__ unsigned lds[];
  if (SG >= 1) asm volatile("" ::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
                            "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79");
  if (SG >= 2) asm volatile("" ::: "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
  if (SG >= 3) asm volatile("" ::: "s96", "s97", "s98", "s99", "s100", "s101");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned v[VG];
#pragma unroll
  for (int i = 0; i < VG; i++) v[i] = threadIdx.x * (i + 1);
  if (touch) lds[threadIdx.x] = v[0];
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) {
#pragma unroll
    for (int i = 0; i < VG; i++) v[i] = v[i] * 3u + 1u;
    __builtin_amdgcn_s_sleep(8);
  }
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < VG; i++) acc ^= v[i];
  if (acc == 0x12345u) out[0] = acc + (touch ? lds[threadIdx.x] : 0);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 203008;
  unsigned* out;
  hipMalloc(&out, 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  printf("%d single-wave workgroups; columns: LDS bytes / wave, spin cycles, duration us, ideal us at 8 and 7 waves per SIMD, achieved residency (waves per SIMD)\n", waves);
  auto run = [&](auto kern, const char* name, int wpb, int lds, int cyc) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(waves / wpb), dim3(64 * wpb), lds * wpb, 0, out, cyc, lds ? 1 : 0);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      best = std::min(best, ms);
    }
    const double us = best * 1e3, waveUs = cyc / 2400.0;
    printf("%-11s waves/WG %d  lds %5d  spin %6d cyc  %8.1f us   ideal(8/SIMD) %7.1f us   residency %.2f waves/SIMD\n", name, wpb, lds, cyc, us,
           waves * waveUs / 8192.0, waves * waveUs / us / 1024.0);
  };
  for (int cyc : {8000, 16000}) {
    run(k_spin<8>, "VG=8", 1, 0, cyc);
    run(k_spin<8>, "VG=8", 1, 4608, cyc);
    run(k_spin<8>, "VG=8", 1, 5632, cyc);
    run(k_spin<8>, "VG=8", 1, 6656, cyc);
    run(k_spin<8>, "VG=8", 4, 4608, cyc);
    run(k_spin<16>, "VG=16", 1, 4608, cyc);
    run(k_spin<24>, "VG=24", 1, 4608, cyc);
    run(k_spin<28>, "VG=28", 1, 4608, cyc);
    run(k_spin<32>, "VG=32", 1, 4608, cyc);
    run(k_spin<40>, "VG=40", 1, 4608, cyc);
    run(k_spin<24, 1>, "VG=24 S80", 1, 4608, cyc);
    run(k_spin<24, 2>, "VG=24 S96", 1, 4608, cyc);
    run(k_spin<24, 3>, "VG=24 S102", 1, 4608, cyc);
  }
  return 0;
}
