// orbfe_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB front end.
//
// All arithmetic that decides an output bit is integer, or float32 with contraction OFF
// (build flag -ffp-contract=off) so it rounds like the reference's separate mul/add on x86-64.
//
//   k_to_gray       cv::cvtColor RGB/BGR(A) -> gray, 8u fixed point (colour input only)
//   k_resize        cv::resize INTER_LINEAR 8u (level l <- level l-1), 11-bit fixed point, separable through LDS
//   k_resize_fixed  the same with the LDS geometry fixed at compile time (every level at scale 1.2): the kernel that runs
//   k_pyramid_cone  all levels in one launch for one- and two-frame calls (a block walks down the dependency cone of a
//                   top-level tile)
//   (k_fast_tasks   per reference FAST cell, one wave per cell pair: orbfe_fast.hip)
//   k_compact       ordered compaction of the per-cell slots into the reference's candidate order (1 lane/cell)
//   k_describe      per keypoint (1 wave): IC-angle (dot4), 7x7 fixed-point Gaussian of the 37x37 neighbourhood
//                   (dot4 rows, dot2 columns), steered BRIEF with __ballot packing
//   k_sincos        test hook for the device (cosf,sinf)
#include <mutex>
#include <hip/hip_runtime.h>

#include <type_traits>

#include "glibc_sincosf.h"
#include "orbfe_internal.h"

namespace orbfe {

// Index arithmetic of these kernels stays far below 2^23, so every product is a full-rate 24-bit
// multiply (v_mul_i32_i24 / v_mad_i32_i24) instead of the quarter-rate v_mul_lo_u32 / 64-bit mads.
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }
// v_mul_u32_u24 spelled out: the compiler rewrites __mul24 into a full 32-bit multiply (quarter rate) when it cannot
// prove the 24-bit range itself; callers guarantee 0 <= a, b < 2^24 and a*b < 2^32
__device__ __forceinline__ unsigned mulu24(unsigned a, unsigned b) {
  unsigned r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ------------------------------------------------------------------------------------------------
// Pyramid: level l <- bilinear(level l-1).  Reference: ComputePyramid, src/ORBextractor.cc:971-996
// (cv::resize call at :984); fixed-point semantics: SURVEY.md Appendix B.2.
//
// One 256-thread block produces a 64x64 output tile in two separable passes through LDS:
//   stage  the source footprint of the tile (about 79x79 bytes at scale 1.2), aligned dword loads;
//   rows   H[r][c] = (p[sx_c] * a0_c + p[sx_c + 1] * a1_c) >> 4 for every footprint row r and output column c
//          (u16, <= 32640): thread = one output column, its coefficient set in registers, 20 rows each;
//   cols   out = ((b0 * H[sy][c] >> 16) + (b1 * H[sy+1][c] >> 16) + 2) >> 2: thread = 4 columns x 4 rows, 64-bit LDS
//          reads of 4 columns, one dword store per row.
// The horizontal interpolation is done once per footprint row (79 per tile) instead of twice per output row (128).
// ------------------------------------------------------------------------------------------------
constexpr int kRzTile = 64;

__global__ __launch_bounds__(256) void k_resize(PyramidParams P, int level) {
  extern __shared__ __align__(16) uint8_t rz[];
  const LevelGeom& D = P.lv[level];
  const LevelGeom& S = P.lv[level - 1];
  const int f = P.frameBase + blockIdx.z;
  const int tx0 = blockIdx.x * kRzTile, ty0 = blockIdx.y * kRzTile;
  const int tx1 = min(tx0 + kRzTile, D.w) - 1, ty1 = min(ty0 + kRzTile, D.h) - 1;   // last column / row of the tile
  const uint8_t* src;
  long long sstride;
  if (level == 1) {
    src = level0_of(P, f);
    sstride = P.stride0;
  } else {
    src = P.slab + (long long)f * P.slabBytes + S.off;
    sstride = S.pitch;
  }
  const int tid = threadIdx.x;
  // coefficient sets are requested first, so that their latency overlaps the staging of the source tile:
  // the column set of the row pass (column tx0 + tid % 64) and the 4 row sets of the column pass
  const int hc = tid & 63;
  const int hx = min(tx0 + hc, D.w - 1);
  const int hsx = D.xofs[hx];
  const int hal = reinterpret_cast<const int*>(D.xalpha)[hx];   // (a0, a1) as two shorts
  const int cx = tx0 + (tid & 15) * 4, cy = ty0 + (tid >> 4) * 4;
  int syv[4], be[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int y = min(cy + i, D.h - 1);
    syv[i] = D.yofs[y];
    be[i] = reinterpret_cast<const int*>(D.ybeta)[y];
  }
  // source footprint of the tile
  const int rx0 = D.xofs[tx0], rx1 = min(D.xofs[tx1] + 1, S.w - 1);
  const int ry0 = min(max(D.yofs[ty0], 0), S.h - 1), ry1 = min(max(D.yofs[ty1] + 1, 0), S.h - 1);
  const int rw = rx1 - rx0 + 1, rh = ry1 - ry0 + 1;
  const int LP = D.rzPitch;
  const int istr = (int)sstride;
  const uint8_t* rbase = src + (long long)ry0 * sstride + rx0;
  const int a = (int)(reinterpret_cast<uintptr_t>(rbase) & 3);
  uint16_t* H = reinterpret_cast<uint16_t*>(rz + ((m24(LP, D.rzRows) + 15) & ~15));   // [rh][64]
  if ((sstride & 3) == 0) {
    // thread (c, r0) = (tid % 32, tid / 32) copies dword column c of rows r0, r0+8, ...: plain adds, no
    // per-element index arithmetic; 4 loads are issued before the first LDS write
    const int ndw = (a + rw + 3) >> 2;            // ~21 at scale 1.2; wider footprints take a second column pass
    const int r0 = tid >> 5;
    for (int c = tid & 31; c < ndw; c += 32) {
      const uint8_t* g = rbase - a + 4 * c + m24(r0, istr);
      uint8_t* l = rz + 4 * c + m24(r0, LP);
      const int gstep = 8 * istr, lstep = 8 * LP;
      for (int r = r0; r < rh; r += 32, g += 4 * gstep, l += 4 * lstep) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (r + 8 * u < rh) ? *reinterpret_cast<const uint32_t*>(g + u * gstep) : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (r + 8 * u < rh) *reinterpret_cast<uint32_t*>(l + u * lstep) = v[u];
      }
    }
  } else {
    const float rcp = 1.0f / (float)rw;
    const int total = rw * rh;
    for (int i = tid; i < total; i += 256) {
      const int y = (int)(((float)i + 0.5f) * rcp), x = i - m24(y, rw);
      rz[m24(y, LP) + a + x] = rbase[m24(y, istr) + x];
    }
  }
  __syncthreads();
  // ---- row pass ----
  {
    const int a0 = (short)hal, a1 = hal >> 16;
    const int o0 = a + hsx - rx0, o1 = a + min(hsx + 1, S.w - 1) - rx0;   // a1 == 0 whenever sx + 1 is out of range
    const uint8_t* p = rz + m24(tid >> 6, LP);
    uint16_t* h = H + (tid >> 6) * 64 + hc;
    for (int r = tid >> 6; r < rh; r += 4, p += 4 * LP, h += 4 * 64)
      *h = (uint16_t)((m24(p[o0], a0) + m24(p[o1], a1)) >> 4);
  }
  __syncthreads();
  // ---- column pass ----
  if (cx > tx1 || cy > ty1) return;
  uint8_t* dst = P.slab + (long long)f * P.slabBytes + D.off;
  const uint16_t* hcol = H + (cx - tx0);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int y = cy + j;
    if (y > ty1) break;
    const int sy = syv[j];
    const int sy0 = min(max(sy, 0), S.h - 1) - ry0, sy1 = min(max(sy + 1, 0), S.h - 1) - ry0;
    const unsigned b0 = (unsigned)(int)(short)be[j], b1 = (unsigned)(be[j] >> 16);
    const uint2 q0 = *reinterpret_cast<const uint2*>(hcol + sy0 * 64), q1 = *reinterpret_cast<const uint2*>(hcol + sy1 * 64);
    const unsigned h0[4] = {q0.x & 0xffffu, q0.x >> 16, q0.y & 0xffffu, q0.y >> 16};
    const unsigned h1[4] = {q1.x & 0xffffu, q1.x >> 16, q1.y & 0xffffu, q1.y >> 16};
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      // b <= 2048, h <= 32640: the products fit 27 bits
      const unsigned v = ((mulu24(b0, h0[i]) >> 16) + (mulu24(b1, h1[i]) >> 16) + 2) >> 2;
      packed |= v << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(dst + m24(y, D.pitch) + cx) = packed;  // pitch % 64 == 0: in bounds
  }
}

// k_resize with the LDS geometry fixed at compile time (pitch LP bytes, up to RH footprint rows): every LDS access
// of the row pass is base + immediate, so an H element costs 3 vector instructions (multiply, multiply-add, shift)
// instead of 9, and the column pass takes its 16-bit operands with SDWA selects instead of unpacking them.  Chosen by
// launch_pyramid for every level whose footprints fit (all levels at scale 1.2).
__device__ __forceinline__ unsigned mulu24_w0(unsigned b, unsigned hh) {   // b * (hh & 0xffff)
  unsigned r;
  asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(b), "v"(hh));
  return r;
}
__device__ __forceinline__ unsigned mulu24_w1(unsigned b, unsigned hh) {   // b * (hh >> 16)
  unsigned r;
  asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(b), "v"(hh));
  return r;
}

template <int WA, int WB>
__device__ __forceinline__ unsigned mulu24_hh(unsigned a, unsigned b) {   // (half WA of a) * (half WB of b)
  unsigned r;
  if (WA == 0 && WB == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
  if (WA == 0 && WB == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
  if (WA == 1 && WB == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
  if (WA == 1 && WB == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// scalar base + 32-bit lane byte offset: one address register, no 64-bit lane arithmetic
__device__ __forceinline__ int ld32(const void* base, unsigned byteOff) {
  return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + byteOff);
}

// a pointer every lane holds the same value of, moved to scalar registers
__device__ __forceinline__ const uint8_t* uniform_ptr(const uint8_t* p) {
  const uintptr_t v = reinterpret_cast<uintptr_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const uint8_t*>(((uintptr_t)hi << 32) | lo);
}

// both 16-bit halves: (x + y + 2) >> 2
__device__ __forceinline__ unsigned pk_round2(unsigned x, unsigned y) {
  unsigned t;
  asm("v_pk_add_u16 %0, %1, %2\n\tv_pk_add_u16 %0, %0, 2 op_sel_hi:[1,0]\n\tv_pk_lshrrev_b16 %0, 2, %0 op_sel_hi:[0,1]" : "=&v"(t) : "v"(x), "v"(y));
  return t;
}

// Launch shape: grid.x = 8 * chunk blocks per frame (chunk = ceil(tiles / 8)), grid.y = frames.  The hardware deals
// consecutive block ids to the 8 XCDs in turn, so block b runs on XCD b % 8; tile = (b % 8) * chunk + b / 8 hands every
// XCD a CONSECUTIVE run of the frame's tiles in row-major order (a horizontal band): the 128-byte lines two horizontally
// adjacent footprints share are fetched into one L2 once instead of into two L2s (profiles/r02n_pmc_summary.csv: the
// plain (x, y, frame) grid fetched 2.3x the level it reads).  tile -> (tx, ty) by a scalar multiply-high.
// Everything the kernel needs about its level as PLAIN kernel arguments (fixed offsets: one scalar load up front) instead of
// PyramidParams indexed by a run-time level -- that cost a chain of nine dependent scalar round trips (level record, frame
// pointer, footprint corners one by one) before the first byte of the tile was requested; now: arguments, then the footprint
// corners + coefficients + frame pointer together, then the tile (round 4).
struct ResizeLevel {
  int dw, dh, dpitch;        // destination level
  int sw, sh;                // source level
  long long sstride;         // source row stride (level 1: the frames' stride)
  long long doff, soff;      // byte offsets inside a frame's pyramid slab (soff unused when the source is level 0)
  const int* xofs;
  const short* xalpha;
  const unsigned* yofc;      // clamped source rows, packed (LevelGeom::yofc)
  const short* ybeta;
  const uint8_t* const* frame0;     // source is level 0: the frames' pointers (or the two inline ones)
  const uint8_t* frameInline[2];
  uint8_t* slab;
  long long slabBytes;
  int frameBase;
  int fromLevel0;
};

template <int LP, int RH>
__global__ __launch_bounds__(256) void k_resize_fixed(ResizeLevel R, int tilesX, int ntiles, unsigned rcpTilesX, int dma) {
  __shared__ __align__(16) uint8_t rz[LP * RH];
  __shared__ __align__(16) uint16_t H[RH * 64];
  struct { int w, h, pitch; const int* xofs; const short* xalpha; const unsigned* yofc; const short* ybeta; } D = {R.dw, R.dh, R.dpitch, R.xofs, R.xalpha, R.yofc, R.ybeta};
  struct { int w, h; } S = {R.sw, R.sh};
  const int f = R.frameBase + blockIdx.y;
  const int chunk = (ntiles + 7) >> 3;
  const int tileIx = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tileIx >= ntiles) return;
  const int tyI = (int)(((unsigned long long)(unsigned)tileIx * rcpTilesX) >> 32);   // tileIx / tilesX (exact: host-checked range)
  const int tx0 = (tileIx - tyI * tilesX) * kRzTile, ty0 = tyI * kRzTile;
  const int tx1 = min(tx0 + kRzTile, D.w) - 1, ty1 = min(ty0 + kRzTile, D.h) - 1;
  const int level = R.fromLevel0 ? 1 : 2;   // (only "is the source a caller-owned frame" matters below)
  // one round trip: the four footprint corners and the frame pointer (scalar), the lanes' coefficients (vector, below)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 fpw = {0u, 0u};
  if (R.fromLevel0) {
    if (R.frame0) {
      fpw = *reinterpret_cast<const u32x2 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(R.frame0 + f));
    } else {
      const unsigned long long v = reinterpret_cast<unsigned long long>((f & 1) ? R.frameInline[1] : R.frameInline[0]);
      fpw.x = (uint32_t)v;
      fpw.y = (uint32_t)(v >> 32);
    }
  }
  const int cxa = *reinterpret_cast<const int __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(D.xofs + tx0));
  const int cxb = *reinterpret_cast<const int __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(D.xofs + tx1));
  const unsigned cya = *reinterpret_cast<const unsigned __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(D.yofc + ty0));
  const unsigned cyb = *reinterpret_cast<const unsigned __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(D.yofc + ty1));
  const int tid = threadIdx.x;
  const int hc = tid & 63;
  const int hx = min(tx0 + hc, D.w - 1);
  const int hsx = ld32(D.xofs, (unsigned)hx << 2);
  const int hal = ld32(D.xalpha, (unsigned)hx << 2);
  const int cx = tx0 + (tid & 15) * 4, cy = ty0 + (tid >> 4) * 4;
  // the thread's four rows cy .. cy + 3 (cy % 4 == 0) as one 16-byte piece of each table (the host pads both to a multiple of four rows)
  const int4 syq = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(D.yofc) + ((unsigned)cy << 2));
  const int4 beq = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(D.ybeta) + ((unsigned)cy << 2));
  const int syv[4] = {syq.x, syq.y, syq.z, syq.w}, be[4] = {beq.x, beq.y, beq.z, beq.w};
  asm volatile("" ::"s"(cxa), "s"(cxb), "s"(cya), "s"(cyb), "s"(fpw.x), "s"(fpw.y));   // everything above is requested before the first wait
  const uint8_t* src;
  const long long sstride = R.sstride;
  if (R.fromLevel0) {
    src = reinterpret_cast<const uint8_t*>(((unsigned long long)fpw.y << 32) | fpw.x);
  } else {
    src = R.slab + (long long)f * R.slabBytes + R.soff;
  }
  const int rx0 = cxa, rx1 = min(cxb + 1, S.w - 1);
  const int ry0 = (int)(cya & 0xffffu), ry1 = (int)(cyb >> 16);   // first source row of the tile's first row, second of its last
  const int rw = rx1 - rx0 + 1, rh = ry1 - ry0 + 1;
  const int istr = (int)sstride;
  const uint8_t* rbase = uniform_ptr(src + (long long)ry0 * sstride + rx0);   // the same for the whole block
  const int a = (int)(reinterpret_cast<uintptr_t>(rbase) & 3);
  if ((sstride & 3) == 0 && dma) {
    // LDS-DMA staging (round 4): global_load_lds_dwordx4 moves 16 bytes per lane from memory straight into LDS -- lane L's
    // bytes land at the instruction's LDS base + 16 L, so with 6 lanes per 96-byte tile row one instruction of 60 lanes fills TEN
    // rows: the whole 80-row footprint is 8 instructions per BLOCK (2 per wave) instead of 10 loads + 10 ds_write per THREAD, no
    // staging registers, no per-element address arithmetic (the global side needs dword alignment only).  Rows past the footprint
    // are not fetched (nobody reads their H).  A 16-byte piece may reach up to 12 bytes past the footprint's last needed byte:
    // inside the row's pitch or the next row everywhere except on the LAST row of a caller-owned level-0 frame, which
    // therefore comes in by plain dword loads.
    static_assert(LP == 96 && RH % 10 == 0, "LDS-DMA staging layout");
    const uint8_t* gb = rbase - a;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the row group's base stays on the scalar unit
    const int lrow = (int)(((unsigned)lane * 43u) >> 8), lcol = lane - lrow * 6;   // lane / 6 for lane < 64
    const bool lastSpecial = level == 1 && ry1 == S.h - 1;
    const int rhDma = lastSpecial ? rh - 1 : rh;
    const bool on = lrow < 10 && 16 * lcol < a + rw;
    const unsigned voff = (unsigned)(m24(lrow, istr) + 16 * lcol);
#pragma unroll
    for (int u = 0; u < RH / 40; u++) {
      const int r = (wave * (RH / 40) + u) * 10;
      const uint8_t* gbr = gb + (long long)r * sstride;   // scalar base of the row group + the lane's constant 32-bit offset
      unsigned vo = voff;
      asm volatile("" : "+s"(gbr), "+v"(vo));             // (kept apart: together they become a 64-bit per-lane address)
      if (on && r + lrow < rhDma)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbr + vo),
                                         (__attribute__((address_space(3))) void*)(rz + r * LP), 16, 0, 0);
    }
    if (lastSpecial && tid < ((a + rw + 3) >> 2))
      *reinterpret_cast<uint32_t*>(rz + (rh - 1) * LP + 4 * tid) = (uint32_t)ld32(gb, (unsigned)(4 * tid + m24(rh - 1, istr)));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if ((sstride & 3) == 0) {
    // thread (c, r0) = (tid % 32, tid / 32) copies dword column c (LP / 4 <= 32 columns) of rows r0, r0 + 8, ...: all RH / 8
    // loads are issued before the first LDS write; rows past the footprint repeat its last row (never read back)
    static_assert(LP <= 128 && RH % 8 == 0, "staging layout");
    const int ndw = (a + rw + 3) >> 2;
    const int c = tid & 31, r0 = tid >> 5;
    if (c < ndw) {
      const uint8_t* gb = rbase - a;
      uint32_t v[RH / 8];
#pragma unroll
      for (int u = 0; u < RH / 8; u++) v[u] = (uint32_t)ld32(gb, (unsigned)(4 * c + m24(min(r0 + 8 * u, rh - 1), istr)));
      uint8_t* l = rz + 4 * c + r0 * LP;
#pragma unroll
      for (int u = 0; u < RH / 8; u++) *reinterpret_cast<uint32_t*>(l + u * 8 * LP) = v[u];
    }
  } else {
    const float rcp = 1.0f / (float)rw;
    const int total = rw * rh;
    for (int i = tid; i < total; i += 256) {
      const int y = (int)(((float)i + 0.5f) * rcp), x = i - m24(y, rw);
      rz[y * LP + a + x] = rbase[m24(y, istr) + x];
    }
  }
  __syncthreads();
  // ---- row pass: all RH rows, unconditionally (rows past the footprint hold stale bytes nobody reads the H of) ----
  {
    // H = (p0 a0 + p1 a1) >> 4 as the HIGH half of (p0 (a0 << 12) + p1 (a1 << 12)): the store takes the high half of the
    // register by itself (ds_write_b16_d16_hi), so an element is two vector instructions, not three.  a0 + a1 = 2048, both >= 0:
    // a << 12 <= 2^23 fits the 24-bit multiplier, the sum is at most 255 * 2048 * 4096 < 2^31.
    const unsigned a0 = (unsigned)(int)(short)hal << 12, a1 = (unsigned)(hal >> 16) << 12;
    const int o0 = a + hsx - rx0, o1 = a + min(hsx + 1, S.w - 1) - rx0;
    const uint8_t* p0 = rz + (tid >> 6) * LP + o0;
    const uint8_t* p1 = rz + (tid >> 6) * LP + o1;
    uint16_t* h = H + (tid >> 6) * 64 + hc;
#pragma unroll
    for (int i = 0; i < RH / 4; i++)
      h[i * 4 * 64] = (uint16_t)((__umul24(p0[i * 4 * LP], a0) + __umul24(p1[i * 4 * LP], a1)) >> 16);
  }
  __syncthreads();
  // ---- column pass ----
  if (cx > tx1 || cy > ty1) return;
  uint8_t* dst = R.slab + (long long)f * R.slabBytes + R.doff;   // scalar; the lane's byte offset is 32 bits
  unsigned doffs = (unsigned)(m24(cy, D.pitch) + cx);
  const unsigned hx2 = (unsigned)(cx - tx0) * 2u, ry0s = (unsigned)ry0 << 7;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int y = cy + j;
    if (y > ty1) break;
    // the row's two source rows come clamped and packed from the host's table (row0 | row1 << 16): H row = source row - ry0, 128 bytes each
    const unsigned r0 = ((unsigned)syv[j] & 0xffffu) << 7, r1 = ((unsigned)syv[j] >> 16) << 7;
    const uint8_t* hb = reinterpret_cast<const uint8_t*>(H) + hx2 - ry0s;
    const uint2 q0 = *reinterpret_cast<const uint2*>(hb + r0), q1 = *reinterpret_cast<const uint2*>(hb + r1);
    // b <= 2048, h <= 32640: the products fit 27 bits.  The weights are taken as the halves of their packed word and the H values as
    // the halves of theirs by SDWA selects; the high halves of two columns' products are gathered into one register (v_perm) and the
    // rounding runs on both at once: (x + y + 2) >> 2 in packed 16-bit arithmetic.
    const unsigned bw = (unsigned)be[j];
    const unsigned x01 = __builtin_amdgcn_perm(mulu24_hh<0, 1>(bw, q0.x), mulu24_hh<0, 0>(bw, q0.x), 0x07060302u);
    const unsigned y01 = __builtin_amdgcn_perm(mulu24_hh<1, 1>(bw, q1.x), mulu24_hh<1, 0>(bw, q1.x), 0x07060302u);
    const unsigned x23 = __builtin_amdgcn_perm(mulu24_hh<0, 1>(bw, q0.y), mulu24_hh<0, 0>(bw, q0.y), 0x07060302u);
    const unsigned y23 = __builtin_amdgcn_perm(mulu24_hh<1, 1>(bw, q1.y), mulu24_hh<1, 0>(bw, q1.y), 0x07060302u);
    const unsigned v01 = pk_round2(x01, y01), v23 = pk_round2(x23, y23);
    *reinterpret_cast<uint32_t*>(dst + doffs) = __builtin_amdgcn_perm(v23, v01, 0x06040200u);  // pitch % 64 == 0: in bounds
    doffs += (unsigned)D.pitch;
  }
}

// ------------------------------------------------------------------------------------------------
// Pyramid for small batches: all levels in one launch.  Seven dependent k_resize launches cost 6-9 us each for a
// single frame whatever the level's size; here one block owns a 32x32 tile of the TOP level and walks down the
// dependency cone: the region of level l-1 that region [x0..x1]x[y0..y1] of level l reads is
// [xofs[x0] .. xofs[x1]+1] x [yofs[y0] .. yofs[y1]+1] (extended to the level's edge at the image border, so that the
// union of all blocks' regions is the whole level).  The block stages its region of the base level in LDS and
// produces its region of every higher level from the previous one in LDS, writing each to the pyramid slab as it
// goes.  Regions of neighbouring blocks overlap by the interpolation halo, so about 1.5x the pixels of a level are
// computed (identical values, benign duplicate stores) -- irrelevant at batch size 1-2, where the chip is idle;
// larger batches keep the per-level kernels.  Arithmetic identical to k_resize (same tables, same two passes).
// ------------------------------------------------------------------------------------------------
constexpr int kConeThreads = 1024;
#ifdef ORBFE_EXPERIMENTS
// experiments build: s_memrealtime stamps (100 MHz) of thread 0 of the LAST tile's block of frame 0: entry, region + coefficients in LDS, then
// after every level (tools/cone_phases.py)
__device__ unsigned long long g_coneStamps[32];
#define CONE_STAMP(i)                                                                                                              \
  do {                                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2)                         \
      g_coneStamps[i] = __builtin_amdgcn_s_memrealtime();                                                                          \
  } while (0)
#else
#define CONE_STAMP(i) ((void)0)
#endif

__global__ __launch_bounds__(kConeThreads) void k_pyramid_cone(PyramidParams P, ConeParams C) {
  extern __shared__ __align__(16) uint8_t cone[];
  const int tid = threadIdx.x, f = P.frameBase + blockIdx.z;
  CONE_STAMP(0);
  // the block's ranges on every level: two 64-byte scalar loads, then registers (the level loops are fully unrolled)
  ConeRange rx[kMaxLevels], ry[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; l++) {
    rx[l] = C.regX[blockIdx.x * kMaxLevels + l];
    ry[l] = C.regY[blockIdx.y * kMaxLevels + l];
  }
  uint8_t* A = cone;
  uint8_t* B = cone + C.offB;
  uint16_t* H = reinterpret_cast<uint16_t*>(cone + C.offH);
  int2* coef = reinterpret_cast<int2*>(cone + C.offC);   // [level][x: coefLen | y: coefLen]
  // A region in LDS starts `pad` columns left of the region's first column, so that LDS column j and the global
  // byte it mirrors have the same alignment: rows are moved as whole dwords in both directions.
  ConeRange bx = rx[0], by = ry[0];
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++)
    if (l == C.base) { bx = rx[l]; by = ry[l]; }
  const uint8_t* src;
  int sstride;
  if (C.base == 0) {
    src = level0_of(P, f);
    sstride = (int)P.stride0;
  } else {
    src = P.slab + (long long)f * P.slabBytes + P.lv[C.base].off;
    sstride = P.lv[C.base].pitch;
  }
  src += (long long)by.lo * sstride + bx.lo;
  const bool dwordRows = (sstride & 3) == 0;
  int padA = dwordRows ? (int)(reinterpret_cast<uintptr_t>(src) & 3) : 0;
  int pitchA = (padA + (bx.hi - bx.lo + 1) + 3) & ~3;
  // ---- every global read of the block is issued here, before the first wait: the coefficient slices of all levels
  //      (thread t < 512: column t of every level, thread 512 + t: row t) and the base region ----
  {
    int cofs[kMaxLevels], cwgt[kMaxLevels];
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) {
      cofs[l] = 0; cwgt[l] = 0;
      if (l <= C.base || l > C.top) continue;
      const LevelGeom& D = P.lv[l];
      const int pad = rx[l].lo & 3;
      if (tid < 512) {
        const int c = tid - pad;
        if (c >= 0 && c <= rx[l].hi - rx[l].lo) {
          cofs[l] = D.xofs[rx[l].lo + c];
          cwgt[l] = reinterpret_cast<const int*>(D.xalpha)[rx[l].lo + c];
        }
      } else if (tid - 512 <= ry[l].hi - ry[l].lo) {
        cofs[l] = D.yofs[ry[l].lo + tid - 512];
        cwgt[l] = reinterpret_cast<const int*>(D.ybeta)[ry[l].lo + tid - 512];
      }
    }
    const int w = bx.hi - bx.lo + 1, h = by.hi - by.lo + 1;
    if (dwordRows) {
      const int ndw = pitchA >> 2;
      const int cw = ndw <= 32 ? 32 : ndw <= 64 ? 64 : 128;
      const int c = tid & (cw - 1), rstep = kConeThreads / cw;
      if (c < ndw) {
        const uint8_t* g = src - padA + 4 * c;
        for (int r = tid / cw; r < h; r += 4 * rstep) {
          uint32_t v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = r + u * rstep < h ? *reinterpret_cast<const uint32_t*>(g + m24(r + u * rstep, sstride)) : 0u;
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (r + u * rstep < h) reinterpret_cast<uint32_t*>(A)[m24(r + u * rstep, ndw) + c] = v[u];
        }
      }
    } else {
      const float rcp = 1.0f / (float)w;
      const int total = m24(w, h);
      for (int i = tid; i < total; i += kConeThreads) {
        const int r = (int)(((float)i + 0.5f) * rcp), c = i - m24(r, w);
        A[m24(r, pitchA) + c] = src[m24(r, sstride) + c];
      }
    }
    int prevPad = padA;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++) {
      if (l <= C.base || l > C.top) continue;
      const LevelGeom& S = P.lv[l - 1];
      const int dw = rx[l].hi - rx[l].lo + 1, dh = ry[l].hi - ry[l].lo + 1, pad = rx[l].lo & 3, pitch = (pad + dw + 3) & ~3;
      int2* cx = coef + (l * 2) * C.coefLen;
      int2* cy = cx + C.coefLen;
      if (tid < pitch) {
        int2 v = make_int2(0, 0);   // padding columns read offset 0 with weight 0
        const int c = tid - pad;
        if (c >= 0 && c < dw) {
          const int x = cofs[l];
          v.x = (x - rx[l - 1].lo + prevPad) | ((min(x + 1, S.w - 1) - rx[l - 1].lo + prevPad) << 16);   // a1 == 0 whenever x + 1 is out of range
          v.y = cwgt[l];
        }
        cx[tid] = v;
      } else if (tid >= 512 && tid - 512 < dh) {
        const int y = cofs[l];
        cy[tid - 512] = make_int2((min(max(y, 0), S.h - 1) - ry[l - 1].lo) | ((min(max(y + 1, 0), S.h - 1) - ry[l - 1].lo) << 16), cwgt[l]);
      }
      prevPad = pad;
    }
  }
  __syncthreads();
  CONE_STAMP(1);
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++) {
    if (l <= C.base || l > C.top) continue;
    const LevelGeom& D = P.lv[l];
    const ConeRange sy = ry[l - 1], dx = rx[l], dy = ry[l];
    const int sh = sy.hi - sy.lo + 1, dw = dx.hi - dx.lo + 1, dh = dy.hi - dy.lo + 1;
    const int pad = dx.lo & 3, pitch = (pad + dw + 3) & ~3;
    const int2* cx = coef + (l * 2) * C.coefLen;
    const int2* cy = cx + C.coefLen;
    // ---- row pass: thread = one (padded) output column, its coefficients in registers, a subset of the rows ----
    {
      const int cw = pitch <= 32 ? 32 : pitch <= 64 ? 64 : pitch <= 128 ? 128 : pitch <= 256 ? 256 : 512;
      const int j = tid & (cw - 1), r0 = tid / cw, rstep = kConeThreads / cw;
      if (j < pitch) {
        const int2 k = cx[j];
        const int a0 = (short)k.y, a1 = k.y >> 16;
        const uint8_t* p0 = A + (k.x & 0xffff);
        const uint8_t* p1 = A + (k.x >> 16);
        uint16_t* h = H + j;
        for (int r = r0; r < sh; r += 4 * rstep) {   // 4 rows in flight: the LDS reads of a batch precede its writes
          int v0[4], v1[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int rr = min(r + u * rstep, sh - 1);
            v0[u] = p0[m24(rr, pitchA)];
            v1[u] = p1[m24(rr, pitchA)];
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (r + u * rstep < sh) h[m24(r + u * rstep, pitch)] = (uint16_t)((m24(v0[u], a0) + m24(v1[u], a1)) >> 4);
        }
      }
    }
    __syncthreads();
    CONE_STAMP(16 + l);
    // ---- column pass: thread = 4 columns, a subset of the rows; dword stores to the next region buffer and the slab ----
    {
      uint8_t* dst = P.slab + (long long)f * P.slabBytes + D.off + (long long)dy.lo * D.pitch + (dx.lo - pad);
      const int ng = pitch >> 2;
      const int gw = ng <= 8 ? 8 : ng <= 16 ? 16 : ng <= 32 ? 32 : ng <= 64 ? 64 : 128;
      const int j = (tid & (gw - 1)) * 4, y0 = tid / gw, ystep = kConeThreads / gw;
      if (j < pitch) {
        const bool whole = j >= pad && j + 3 < pad + dw;
        for (int yb = y0; yb < dh; yb += 2 * ystep) {
          int2 k[2];
          uint2 q0[2], q1[2];
#pragma unroll
          for (int u = 0; u < 2; u++) k[u] = cy[min(yb + u * ystep, dh - 1)];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            q0[u] = *reinterpret_cast<const uint2*>(H + m24(k[u].x & 0xffff, pitch) + j);
            q1[u] = *reinterpret_cast<const uint2*>(H + m24(k[u].x >> 16, pitch) + j);
          }
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int y = yb + u * ystep;
            if (y >= dh) break;
            const unsigned b0 = (unsigned)(int)(short)k[u].y, b1 = (unsigned)(k[u].y >> 16);
            const unsigned h0[4] = {q0[u].x & 0xffffu, q0[u].x >> 16, q0[u].y & 0xffffu, q0[u].y >> 16};
            const unsigned h1[4] = {q1[u].x & 0xffffu, q1[u].x >> 16, q1[u].y & 0xffffu, q1[u].y >> 16};
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; e++)
              packed |= (((mulu24(b0, h0[e]) >> 16) + (mulu24(b1, h1[e]) >> 16) + 2) >> 2) << (8 * e);
            *reinterpret_cast<uint32_t*>(B + m24(y, pitch) + j) = packed;
            uint8_t* g = dst + m24(y, D.pitch) + j;
            if (whole) {
              *reinterpret_cast<uint32_t*>(g) = packed;
            } else {   // first / last group of the region: the neighbouring block owns the other columns
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (j + e >= pad && j + e < pad + dw) g[e] = (uint8_t)(packed >> (8 * e));
            }
          }
        }
      }
    }
    __syncthreads();
    CONE_STAMP(1 + l);
    uint8_t* t = A; A = B; B = t;
    padA = pad;
    pitchA = pitch;
  }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// ------------------------------------------------------------------------------------------------
// Ordered compaction of the per-cell candidate slots into the frame's candidate list (cells are numbered level-major,
// cell-row-major: exactly the order in which the reference appends to vToDistributeKeys, ORBextractor.cc:826-870).
// One wave per 64 consecutive cells, one LANE per cell.  A wave finds its first offset by itself -- the sum of the
// counts of all earlier cells, at most 100 coalesced dwords per lane at 1080p -- so no separate scan kernel (and no
// grid-wide dependency) is needed: one launch instead of two.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return (uint32_t)x;
}

__global__ __launch_bounds__(64) void k_compact(PyramidParams P) {
  const int cell0 = blockIdx.x * 64, lane = threadIdx.x, f = P.frameBase + blockIdx.y;
  const uint32_t* cnt = P.cellCount + (long long)f * P.ncells;
  const int cell = cell0 + lane;
  const bool valid = cell < P.ncells;
  // everything that does not depend on another load is requested up front (the data was written by another kernel,
  // possibly through another XCD's L2: every dependent round trip costs 1-2 us): the cell's slot address, its own
  // count, the counts of all earlier cells
  const uint32_t slotOff = valid ? P.cells[cell].slotOff : 0u;
  uint32_t n = valid ? cnt[cell] : 0u;
  uint32_t s = 0;
  for (int i = lane; i < cell0; i += 64 * 16) {   // 16 loads in flight per lane: the last wave of a 1080p frame needs 7 rounds
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = (i + 64 * u < cell0) ? cnt[i + 64 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
  }
  // the first 16 slots of the cell are fetched before its count is known (a cell owns at least 16 slots; surplus
  // entries are never stored)
  const uint32_t* slot = P.slots + (long long)f * P.slotsPerFrame + slotOff;
  uint32_t first[16];
#pragma unroll
  for (int u = 0; u < 16; u++) first[u] = slot[u];
  const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(s), 63);
  const uint32_t incl = wave_incl_scan(n);
  const uint32_t o = base + incl - n;
  uint32_t* ls = P.levelStart + (long long)f * (kMaxLevels + 1);
  if (valid) {
    for (int l = 0; l < P.nlevels; l++)
      if (P.lv[l].cellBase == cell) ls[l] = o;
    if (cell == P.ncells - 1) ls[P.nlevels] = o + n;
  }
  if (o >= (uint32_t)P.candCap) n = 0;
  else n = min(n, (uint32_t)P.candCap - o);
  uint32_t* dst = P.cand + (long long)f * P.candCap + o;
#pragma unroll
  for (int u = 0; u < 16; u++)
    if ((uint32_t)u < n) dst[u] = first[u];
  for (uint32_t i = 16; __any(i < n); i += 16) {
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = (i + u < n) ? slot[i + u] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (i + u < n) dst[i + u] = v[u];
  }
}

// The same compaction with LEVEL-LOCAL lists (one- and two-frame calls, whose levels run as two independent launch chains:
// level 0 needs no pyramid and runs beside the cone kernel and the other levels): level l's list starts at the fixed position
// lv[l].slotBase of the frame's candidate array (the slot array's own layout: it cannot overflow) and levelStart[f][l] receives
// its LENGTH.  A wave sums the counts of the earlier cells of ITS level only -- no launch looks at another chain's cells -- and
// requests them in ONE batch (40 per lane: 2 560 cells, more than a 1080p level has) together with its cell's first slots: the call
// waits for this kernel's single memory round trip, not for its throughput (k_compact's 16-wide batches were three dependent
// round trips for the last wave of level 0).  grid = (waves of the largest level in [l0, l1), l1 - l0, frames).
__global__ __launch_bounds__(64) void k_compact_local(PyramidParams P, int l0) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int level = l0 + blockIdx.y, lane = threadIdx.x, f = P.frameBase + blockIdx.z;
  const LevelGeom& L = P.lv[level];
  const int nc = L.nCols * L.nRows, cell0 = blockIdx.x * 64;
  if (cell0 >= nc) return;
  const uint32_t* cnt = P.cellCount + (long long)f * P.ncells + L.cellBase;
  const int cell = cell0 + lane;
  const bool valid = cell < nc;
  // a cell's slots start 16-byte aligned (slotCap and every slotBase are multiples of 4): its first 32 arrive as eight 16-byte
  // pieces, requested together with the counts -- a second round trip is left to cells with more than 32 keypoints
  const uint32_t* slot = P.slots + (long long)f * P.slotsPerFrame + L.slotBase + (long long)(valid ? cell : 0) * L.slotCap;
  constexpr int kB = 40, kF = 8;
  uint32_t n = valid ? cnt[cell] : 0u;
  uint32_t s = 0;
  u32x4 first[kF];
  {
    uint32_t v[kB];
#pragma unroll
    for (int u = 0; u < kB; u++) v[u] = (lane + 64 * u < cell0) ? cnt[lane + 64 * u] : 0u;
#pragma unroll
    for (int u = 0; u < kF; u++) first[u] = 4 * u < L.slotCap ? reinterpret_cast<const u32x4*>(slot)[u] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < kB; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
  }
  for (int i = lane + 64 * kB; i < cell0; i += 64 * 16) {   // levels of more than 2 560 cells (4K frames)
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = (i + 64 * u < cell0) ? cnt[i + 64 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
  }
  const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(s), 63);
  const uint32_t incl = wave_incl_scan(n);
  const uint32_t o = base + incl - n;
  if (valid && cell == nc - 1) P.levelStart[(long long)f * (kMaxLevels + 1) + level] = o + n;
  uint32_t* dst = P.cand + (long long)f * P.candCap + L.slotBase + o;
#pragma unroll
  for (int u = 0; u < kF; u++) {
    if ((uint32_t)(4 * u) < n) dst[4 * u] = first[u].x;
    if ((uint32_t)(4 * u + 1) < n) dst[4 * u + 1] = first[u].y;
    if ((uint32_t)(4 * u + 2) < n) dst[4 * u + 2] = first[u].z;
    if ((uint32_t)(4 * u + 3) < n) dst[4 * u + 3] = first[u].w;
  }
  for (uint32_t i = 4 * kF; __any(i < n); i += 16) {
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = (i + u < n) ? slot[i + u] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (i + u < n) dst[i + u] = v[u];
  }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor, one wave per selected keypoint.
// Reference: IC_Angle (src/ORBextractor.cc:86-113, on the UNBLURRED level), GaussianBlur 7x7 s=2
// reflect-101 (:949-950) and computeOrbDescriptor (:132-171, on the blurred level).
// Only the 37x37 blurred neighbourhood a descriptor can touch is produced (rotated offsets have
// |cvRound| <= 18), from the 43x43 raw neighbourhood, with the level's own reflect-101 border.
// Fixed-point blur: horizontal sum(k*p) in 16 bits, vertical (sum(k*h) + 32768) >> 16, k =
// [18,34,48,56,48,34,18].
// ------------------------------------------------------------------------------------------------
constexpr int8_t kPattern[1024] = {
#include "brief_pattern.inc"
};
// the 256 test pairs as floats (x1, y1, x2, y2): the rotation of computeOrbDescriptor is float arithmetic
struct PatternF { float v[1024]; };
constexpr PatternF make_pattern_f() {
  PatternF t{};
  for (int i = 0; i < 1024; i++) t.v[i] = (float)kPattern[i];
  return t;
}
__constant__ PatternF c_patternF = make_pattern_f();

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2, SURVEY.md B.4
  const float scale = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
              p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

constexpr int kRawW = 2 * kRawRad + 1;   // 43
constexpr int kBlurW = 2 * kBlurRad + 1; // 37
constexpr int kRawP = 48;    // raw pitch (bytes): alignment offset (<= 3) + 43 pixels
constexpr int kHPT = 50;     // pitch (u16) of the TRANSPOSED horizontal-pass image hbT[x][y]: y <= 45 is read, even, and
                             // 4 rows apart (the 4 outputs of one lane) land in different LDS banks
constexpr int kBPT = 40;     // pitch (bytes) of the TRANSPOSED blurred patch blT[x][y]

// IC-angle over the circular patch of radius 15 (rows v = -15..15, |u| <= umax[|v|], 749 pixels; umax from
// ORBextractor.cc:486-501 equals floor(sqrt(240 - v*v)), checked by the static_assert below).  The patch is
// walked as 31 rows x 8 dwords (u + 15 = 4k + j); per dword three byte-weight words feed v_dot4_u32_u8:
//   w1 = inside ? 1 : 0,   w2 = inside ? u + 15 : 0,   w3 = inside ? v + 15 : 0
// so that m10 = sum(w2 . I) - 15 sum(w1 . I) and m01 = sum(w3 . I) - 15 sum(w1 . I).
constexpr int kIcCount = 749;
constexpr int kIcItems = 31 * 8;
struct IcW { uint32_t w1, w2, w3, pad; };
struct IcTab { IcW e[256]; };   // 248 items + 8 of weight zero (row 31 of the walk: inside the 43-row raw patch)
constexpr IcTab make_ic_tab() {
  IcTab t{};
  const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int r = 0; r < 31; r++) {
    const int v = r - 15, d = um[v < 0 ? -v : v];
    for (int k = 0; k < 8; k++) {
      uint32_t w1 = 0, w2 = 0, w3 = 0;
      for (int j = 0; j < 4; j++) {
        const int u = 4 * k + j - 15;
        if (u >= -d && u <= d) {
          w1 |= 1u << (8 * j);
          w2 |= (uint32_t)(u + 15) << (8 * j);
          w3 |= (uint32_t)(v + 15) << (8 * j);
        }
      }
      t.e[r * 8 + k] = IcW{w1, w2, w3, 0};
    }
  }
  return t;
}
constexpr int ic_count() {
  const IcTab t = make_ic_tab();
  int n = 0;
  for (int i = 0; i < kIcItems; i++)
    for (int j = 0; j < 4; j++) n += (t.e[i].w1 >> (8 * j)) & 1;
  return n;
}
static_assert(ic_count() == kIcCount, "circular patch must have 749 pixels");
__constant__ IcTab c_icTab = make_ic_tab();

// The blur only where the descriptor can look.  A test location (x, y) of the pattern, |x|, |y| <= 13, radius <= 18.385
// ((-13, -13)), is rotated and rounded per coordinate (ORBextractor.cc:147-151): it lands on an integer point within 18.385 +
// sqrt(1/2) = 19.092 of the keypoint -- the 37 x 37 box minus its corners, 1 125 of 1 369 pixels.  Vertical-pass items (column x,
// 4 rows) that hold at least one such pixel: 308 of 370 -> 5 wave iterations instead of 6; horizontal-pass items (row r, 4
// columns) that feed at least one such pixel (blurred (x, yb) reads hb rows yb .. yb + 6): 368 of 430 -> 6 instead of 7.
// Outputs outside the disc are computed from whatever the skipped items left in LDS and never sampled.
constexpr int kDiscR2x4 = 1458;   // 4 * 19.092^2 = 1458.02: x^2 + y^2 <= 364 (integer points: 4 (x^2 + y^2) <= 1458)
constexpr bool in_disc(int x, int yb) { return 4 * ((x - kBlurRad) * (x - kBlurRad) + (yb - kBlurRad) * (yb - kBlurRad)) <= kDiscR2x4; }
struct BlurItems { uint16_t h[448]; uint16_t v[320]; int nh, nv; };
constexpr BlurItems make_blur_items() {
  BlurItems t{};
  // vertical: (x, yq) -> outputs (x, 4yq .. 4yq+3)
  bool vneed[kBlurW][10] = {};
  for (int x = 0; x < kBlurW; x++)
    for (int yq = 0; yq < 10; yq++)
      for (int j = 0; j < 4; j++)
        if (4 * yq + j < kBlurW && in_disc(x, 4 * yq + j)) vneed[x][yq] = true;
  for (int x = 0; x < kBlurW; x++)
    for (int yq = 0; yq < 10; yq++)
      if (vneed[x][yq]) t.v[t.nv++] = (uint16_t)(x << 4 | yq);
  // horizontal: (row r of the raw patch, g) -> hb columns 4g .. 4g+3 of row r; needed by the disc's pixels only
  bool hneed[kRawW][10] = {};
  for (int x = 0; x < kBlurW; x++)
    for (int yb = 0; yb < kBlurW; yb++)
      if (in_disc(x, yb))
        for (int r = yb; r < yb + 7; r++) hneed[r][x >> 2] = true;
  for (int r = 0; r < kRawW; r++)
    for (int g = 0; g < 10; g++)
      if (hneed[r][g]) t.h[t.nh++] = (uint16_t)(r << 4 | g);
  for (int i = t.nh; i < 448; i++) t.h[i] = t.h[0];
  for (int i = t.nv; i < 320; i++) t.v[i] = t.v[0];
  return t;
}
constexpr BlurItems kBlurItemsHost = make_blur_items();
static_assert(kBlurItemsHost.nh == 368 && kBlurItemsHost.nv == 308, "blur item lists: 6 and 5 wave iterations");
// The item lists as the kernel wants them: LDS byte offsets instead of coordinates, so that a lane decodes an item with one
// AND and one shift.  h[i]: low half = first raw byte the item reads (row y, dword 4g of the raw patch), high half = byte offset of
// its first output in the transposed u16 image; v[i]: low half = first byte of column x at row 4yq of that image, high half =
// byte offset of its output dword in the transposed blurred patch.  Both padded with their first item.
struct BlurCodes { uint32_t h[512], v[512]; };
constexpr BlurCodes make_blur_codes() {
  BlurCodes t{};
  const BlurItems b = make_blur_items();
  for (int i = 0; i < 512; i++) {
    const int hc = b.h[i < b.nh ? i : 0], y = hc >> 4, g = hc & 15;
    t.h[i] = (uint32_t)(y * kRawP + 4 * g) | ((uint32_t)((4 * g * kHPT + y) * 2) << 16);
    const int vc = b.v[i < b.nv ? i : 0], x = vc >> 4, yq = vc & 15;
    t.v[i] = (uint32_t)((x * kHPT + 4 * yq) * 2) | ((uint32_t)(x * kBPT + 4 * yq) << 16);
  }
  return t;
}
__constant__ BlurCodes c_blurCodes = make_blur_codes();


// The 8.8 Gaussian taps of cv::GaussianBlur(7x7, sigma 2) on CV_8U.  GAUSS = ORBFE_GAUSS_ED (0): [18,34,48,56,48,34,18], the
// error-diffused kernel of OpenCV >= 4.1.1 (sum 256).  GAUSS = ORBFE_GAUSS_ROUNDED (1): [18,34,49,55,49,34,18], every tap rounded on
// its own as OpenCV 4.0.0 - 4.1.0 did (sum 257: the horizontal sums still fit 16 bits -- 257 * 255 = 65 535 -- and the one place its
// saturating ufixedpoint arithmetic acts is the final cast: a rounded result of 256 or 257 becomes 255).
template <int GAUSS> constexpr unsigned blur_tap(int t) {
  constexpr unsigned ed[7] = {18, 34, 48, 56, 48, 34, 18}, rounded[7] = {18, 34, 49, 55, 49, 34, 18};
  return GAUSS == 1 ? rounded[t] : ed[t];
}
// ... laid over dword q of a row whose 7-tap window starts at byte s: byte b of the word weighs byte 4q + b, i.e. tap 4q + b - s (0
// outside the window)
template <int GAUSS> constexpr unsigned blur_tap_word(int s, int q) {
  unsigned w = 0;
  for (int b = 0; b < 4; b++) {
    const int t = 4 * q + b - s;
    if (t >= 0 && t <= 6) w |= blur_tap<GAUSS>(t) << (8 * b);
  }
  return w;
}

template <int S, int GAUSS>
__device__ __forceinline__ unsigned blur_row_out(const uint32_t (&d)[4]) {   // the 7-tap sum whose window starts at byte S of d
  constexpr unsigned w0 = blur_tap_word<GAUSS>(S, 0), w1 = blur_tap_word<GAUSS>(S, 1), w2 = blur_tap_word<GAUSS>(S, 2), w3 = blur_tap_word<GAUSS>(S, 3);
  unsigned acc = 0u;
  if constexpr (w0 != 0u) acc = __builtin_amdgcn_udot4(d[0], w0, acc, false);
  if constexpr (w1 != 0u) acc = __builtin_amdgcn_udot4(d[1], w1, acc, false);
  if constexpr (w2 != 0u) acc = __builtin_amdgcn_udot4(d[2], w2, acc, false);
  if constexpr (w3 != 0u) acc = __builtin_amdgcn_udot4(d[3], w3, acc, false);
  return acc;
}

// sum over the wave (wave-uniform result): DPP row rotations, then one value per row of 16 lanes
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);   // row_ror:8
  v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);   // row_ror:4
  v += __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false);   // row_ror:2
  v += __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false);   // row_ror:1
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// three sums at once: the row rotations as v_add_u32_dpp proper (the compiler's form of the function above is move + DPP move +
// add per step), the three chains interleaved so that each DPP read is two instructions behind the write it depends on (the wait
// states the hardware asks for between a vector write and a DPP read of the same register)
__device__ __forceinline__ void wave_sum3_i32(int& a, int& b, int& c) {
#define ORBFE_ROR3(n)                                                          \
  "v_add_u32_dpp %0, %0, %0 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"     \
  "v_add_u32_dpp %1, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"     \
  "v_add_u32_dpp %2, %2, %2 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" ORBFE_ROR3(8) ORBFE_ROR3(4) ORBFE_ROR3(2) ORBFE_ROR3(1) : "+v"(a), "+v"(b), "+v"(c));
#undef ORBFE_ROR3
  a = __builtin_amdgcn_readlane(a, 0) + __builtin_amdgcn_readlane(a, 16) + __builtin_amdgcn_readlane(a, 32) + __builtin_amdgcn_readlane(a, 48);
  b = __builtin_amdgcn_readlane(b, 0) + __builtin_amdgcn_readlane(b, 16) + __builtin_amdgcn_readlane(b, 32) + __builtin_amdgcn_readlane(b, 48);
  c = __builtin_amdgcn_readlane(c, 0) + __builtin_amdgcn_readlane(c, 16) + __builtin_amdgcn_readlane(c, 32) + __builtin_amdgcn_readlane(c, 48);
}

// Slot mode (GPU quadtree): `sel` is laid out [frame][selPerFrame] with per-level sub-regions; slot k is
// live iff its index inside its level region is below selCount[frame][level].  Dense mode (host
// quadtree): selCount == nullptr and every k < nsel is live.
struct SlotInfo {
  const uint32_t* selCount;  // [nframes][kMaxLevels] or nullptr
  int selPerFrame;
  int selOff[kMaxLevels + 1];
  int nlevels;
};

// What the kernel needs, as one compact argument block (328 bytes instead of PyramidParams + SlotInfo, 1.6 KB): its scalar fields arrive
// with the wave's first loads, the level record by one indexed load off the argument segment.
struct DescribeLevel { int w, h, pitch, pad; long long off, pad2; };   // 32 bytes: one s_load_dwordx8
struct DescribeArgs {
  DescribeLevel lv[kMaxLevels];
  const SelKp* sel;
  const uint32_t* selCount;          // slot mode: [nframes][kMaxLevels]; nullptr: dense mode
  float* angleOut;
  uint8_t* descOut;
  float* angleOut2;                  // optional second copy of both (batches: the first stays in HBM for the matching kernels, the second
  uint8_t* descOut2;                 // goes straight to the page-locked result arena: no copy command behind the batch); nullptr = none
  const uint8_t* const* frame0;
  const uint8_t* frameInline[2];
  long long stride0;
  uint8_t* slab;
  long long slabBytes;
  int nsel, selPerFrame, nlevels, frameBase, dma;
  int selOff[kMaxLevels + 1];
  int gauss;                         // host side only: which instantiation is launched
};

// WAVES = 1: one wave per keypoint (batches: throughput).  WAVES = 4: the same passes spread over a block of four waves
// (one- and two-frame calls: a keypoint's 760 dependent-ish instructions are the latency of the whole kernel there).
template <int WAVES, int GAUSS>
__global__ __launch_bounds__(64 * WAVES) void k_describe(DescribeArgs A) {
  constexpr int NT = 64 * WAVES;
  // Every table a lane will need depends on its lane number only: ALL of them are requested here, before the first wait of
  // the kernel, instead of one dependent round trip per pass iteration (4 + 6 + 5 of them for one wave) -- round 4.
  constexpr int NIC = (kIcItems + NT - 1) / NT, NH = (kBlurItemsHost.nh + NT - 1) / NT, NV = (kBlurItemsHost.nv + NT - 1) / NT;
  static_assert(NIC * NT <= 256 && NH * NT <= 512 && NV <= NH, "table padding");
  uint32_t icw[NIC][3], bh[NH], bv[NV];
#pragma unroll
  for (int it = 0; it < NIC; it++) {
    const IcW w = c_icTab.e[it * NT + threadIdx.x];
    icw[it][0] = w.w1; icw[it][1] = w.w2; icw[it][2] = w.w3;
  }
#pragma unroll
  for (int it = 0; it < NH; it++) bh[it] = c_blurCodes.h[it * NT + threadIdx.x];
#pragma unroll
  for (int it = 0; it < NV; it++) bv[it] = c_blurCodes.v[it * NT + threadIdx.x];
  __shared__ int icSum[3 * WAVES];
  __shared__ __align__(16) uint8_t raw[kRawW * kRawP + 16];   // + 16: the last row's 4-dword reads
  __shared__ __align__(16) uint16_t hbT[40 * kHPT];            // horizontal pass, transposed: hbT[x][y]
  // blurred 37x37 patch, transposed: blT[x][y].  It reuses the raw patch, which is dead once the horizontal pass
  // (and the barrier behind it) is through: LDS per wave decides how many waves of the kernels that run beside this
  // one (the next batches' FAST) fit on the CU.
  static_assert(kBlurW * kBPT <= kRawW * kRawP, "the blurred patch must fit into the raw patch");
  uint8_t* blT = raw;
  // XCD b % 8 works through a CONSECUTIVE eighth of the slots (frame-major, level by level): the keypoints of one
  // frame's level meet in one L2, where each 128-byte line of the level is fetched once however many 43-byte patch
  // rows touch it (the plain mapping spread neighbouring keypoints over all eight L2s: 3.1x the algorithmic bytes,
  // profiles/r02n_pmc_summary.csv)
  // A wave's life begins with memory round trips nothing can overlap: keep that chain SHORT (as k_fast_tasks does).  The scalar
  // fields of the argument block are requested by the first loads (pinned below: left alone the compiler fetches them where they are
  // used, a dependent wait each).  In slot mode the slot index alone gives frame and level, so the liveness count, the keypoint
  // record, the level-0 pointer (three scalar loads from memory) and the level record (off the argument segment) go out TOGETHER --
  // one round trip, then the patch request.  (Before round 4: ten dependent round trips; mid-round: three from memory and eight
  // from the argument segment.)
  asm volatile("" ::"s"(A.sel), "s"(A.selCount), "s"(A.angleOut), "s"(A.descOut), "s"(A.frame0), "s"(A.frameInline[0]), "s"(A.frameInline[1]),
               "s"(A.stride0), "s"(A.slab), "s"(A.slabBytes), "s"(A.nsel), "s"(A.selPerFrame), "s"(A.nlevels), "s"(A.frameBase), "s"(A.dma),
               "s"(A.selOff[1]), "s"(A.selOff[2]), "s"(A.selOff[3]), "s"(A.selOff[4]), "s"(A.selOff[5]), "s"(A.selOff[6]), "s"(A.selOff[7]));
  const int nsel = A.nsel, dma = A.dma;
  const SelKp* sel = A.sel;
  float* angleOut = A.angleOut;
  uint8_t* descOut = A.descOut;
  const int chunk = (nsel + 7) >> 3;
  const int k = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (k >= nsel) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  int level, f;
  u32x2 sw, fpw = {0u, 0u};
  u32x8 lw;
  auto frame_ptr_words = [&](int fa) -> u32x2 {   // the level-0 pointer of frame fa as two scalar words
    u32x2 r;
    if (A.frame0) {
      r = *reinterpret_cast<const u32x2 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(A.frame0 + fa));
    } else {
      const unsigned long long v = reinterpret_cast<unsigned long long>((fa & 1) ? A.frameInline[1] : A.frameInline[0]);
      r.x = (uint32_t)v;
      r.y = (uint32_t)(v >> 32);
    }
    return r;
  };
  if (A.selCount) {
    const int fr = k / A.selPerFrame, within = k - fr * A.selPerFrame;
    int l = 0, lbase = 0;
#pragma unroll
    for (int j = 1; j < kMaxLevels; j++) {   // selOff ascends: the last level region that starts at or before `within`
      const int o = A.selOff[j];
      if (j < A.nlevels && within >= o) { l = j; lbase = o; }
    }
    level = l;
    f = A.frameBase + fr;
    const uint32_t live = *reinterpret_cast<const uint32_t __attribute__((address_space(4)))*>(
        reinterpret_cast<uintptr_t>(A.selCount + (long long)f * kMaxLevels + l));
    sw = *reinterpret_cast<const u32x2 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(sel + k));
    fpw = frame_ptr_words(f);
    lw = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(ka + offsetof(DescribeArgs, lv) + 32 * (unsigned)l);
    asm volatile("" ::"s"(sw.x), "s"(live), "s"(fpw.x), "s"(lw[0]));   // all four requested before the test below
    if ((uint32_t)(within - lbase) >= live) return;
  } else {
    sw = *reinterpret_cast<const u32x2 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(sel + k));
    level = (int)(sw.y & 0xffu);
    f = (int)((sw.y >> 8) & 0xffffu);
    fpw = frame_ptr_words(f);
    lw = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(ka + offsetof(DescribeArgs, lv) + 32 * (unsigned)level);
  }
  const int cx = (int)(sw.x & 0xffffu), cy = (int)(sw.x >> 16);
  struct { int w, h; } L = {(int)lw[0], (int)lw[1]};
  const uint8_t* img;
  long long stride;
  if (level == 0) {
    img = reinterpret_cast<const uint8_t*>(((unsigned long long)fpw.y << 32) | fpw.x);
    stride = A.stride0;
  } else {
    img = A.slab + (long long)f * A.slabBytes + (long long)(((unsigned long long)lw[5] << 32) | lw[4]);
    stride = (long long)(int)lw[2];
  }
  // 43x43 raw patch -> LDS.  Interior keypoints (all but those within 21 px of the level border): aligned
  // dword loads, 9 in flight per lane, the patch keeps the byte alignment `pa` of its first pixel.
  // Border keypoints: byte loads with the level's own reflect-101 indexing (GaussianBlur's border).
  int pa = 0;
  const int istr = (int)stride;
  const bool interior = cx >= kRawRad && cy >= kRawRad && cx + kRawRad < L.w && cy + kRawRad < L.h && (stride & 3) == 0;
  // LDS-DMA (round 4, as the pyramid's staging): global_load_lds_dwordx4 moves 16 bytes per lane straight into LDS, lane L's
  // piece landing at the instruction's LDS base + 16 L -- with 3 pieces per 48-byte patch row one instruction of 63 lanes fills
  // 21 rows, the 43-row patch is 3 instructions per wave (1 for each of four waves) instead of 11 loads + 11 ds_write per lane,
  // no staging registers and no per-element address arithmetic.  A row's third piece reaches up to 5 bytes past the last pixel
  // the patch needs: inside the row's pitch or the next row everywhere except on the LAST row of a caller-owned level-0 frame,
  // whose keypoints keep the path through registers.
  if (interior && dma && (level != 0 || cy + kRawRad < L.h - 1)) {
    const uint8_t* rbase = uniform_ptr(img + m24(cy - kRawRad, istr) + (cx - kRawRad));   // the same for the whole block
    pa = (int)(reinterpret_cast<uintptr_t>(rbase) & 3);
    const uint8_t* gb = rbase - pa;
    constexpr int RPI = WAVES == 1 ? 21 : 11, NI = WAVES == 1 ? 3 : 1;   // rows per instruction, instructions per wave
    static_assert(RPI * NI * WAVES >= kRawW && 3 * RPI <= 64 && kRawP == 48, "LDS-DMA patch layout");
    const int lrow = (lane * 171) >> 9, lcol = lane - 3 * lrow;          // lane / 3 for lane < 64
    const unsigned voff = (unsigned)(m24(lrow, istr) + 16 * lcol);
#pragma unroll
    for (int u = 0; u < NI; u++) {
      const int r = (wave * NI + u) * RPI;
      if (lrow < RPI && r + lrow < kRawW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (voff + (unsigned)m24(r, istr))),   // scalar base + 32-bit lane offset
                                         (__attribute__((address_space(3))) void*)(raw + r * kRawP), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (interior) {
    const uint8_t* rbase = img + m24(cy - kRawRad, istr) + (cx - kRawRad);
    pa = (int)(reinterpret_cast<uintptr_t>(rbase) & 3);
    const int ndw = (pa + kRawW + 3) >> 2;  // 11 or 12
    // thread (c, r0) = (tid % 16, tid / 16) copies dword column c of rows r0, r0 + RS, ...: all loads in flight at once
    constexpr int RS = NT / 16, NL = (kRawW + RS - 1) / RS;   // 4 rows apart, 11 loads (one wave); 16 apart, 3 loads (four)
    const int c = tid & 15, r0 = tid >> 4;
    if (c < ndw) {
      const uint8_t* g = rbase - pa + 4 * c + m24(r0, istr);
      uint8_t* l = raw + 4 * c + m24(r0, kRawP);
      const int gstep = RS * istr;
      uint32_t v[NL];
#pragma unroll
      for (int u = 0; u < NL; u++) v[u] = (r0 + RS * u < kRawW) ? *reinterpret_cast<const uint32_t*>(g + u * gstep) : 0u;
#pragma unroll
      for (int u = 0; u < NL; u++)
        if (r0 + RS * u < kRawW) *reinterpret_cast<uint32_t*>(l + u * RS * kRawP) = v[u];
    }
  } else {
    for (int i0 = tid; i0 < kRawW * kRawW; i0 += NT * 8) {
      uint8_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * NT;
        v[u] = 0;
        if (i < kRawW * kRawW) {
          const int y = i / kRawW, x = i - y * kRawW;
          const int gx = reflect101(cx - kRawRad + x, L.w), gy = reflect101(cy - kRawRad + y, L.h);
          v[u] = img[m24(gy, istr) + gx];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * NT;
        if (i < kRawW * kRawW) {
          const int y = i / kRawW, x = i - y * kRawW;
          raw[y * kRawP + x] = v[u];
        }
      }
    }
  }
  __syncthreads();
  // raw pixel (x, y) of the 43x43 patch lives at raw[y * kRawP + pa + x]

  // IC-angle moments (ORBextractor.cc:86-113) of the UNBLURRED patch: 31 rows x 8 dwords, three v_dot4_u32_u8 each
  const int sIc = (pa + 6) & 3, qIc = (pa + 6) & ~3;   // patch column u = -15 is raw byte pa + 6 of a row
  int S1 = 0, S2 = 0, S3 = 0;
  // item it * NT + tid = (row it * NT / 8 + tid / 8, dword tid % 8): one address per lane, the iterations are constant offsets
  const uint8_t* icBase = raw + m24((tid >> 3) + (kRawRad - 15), kRawP) + qIc + 4 * (tid & 7);
#pragma unroll
  for (int it = 0; it < NIC; it++) {   // (items 248 .. 255 have weight zero)
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(icBase + it * (NT / 8) * kRawP);
    const uint32_t e = __builtin_amdgcn_alignbyte(rp[1], rp[0], sIc);
    S1 = (int)__builtin_amdgcn_udot4(e, icw[it][0], (unsigned)S1, false);
    S2 = (int)__builtin_amdgcn_udot4(e, icw[it][1], (unsigned)S2, false);
    S3 = (int)__builtin_amdgcn_udot4(e, icw[it][2], (unsigned)S3, false);
  }
  wave_sum3_i32(S1, S2, S3);
  if (WAVES > 1) {   // the waves' partial sums meet in LDS
    if (lane == 0) { icSum[3 * wave] = S1; icSum[3 * wave + 1] = S2; icSum[3 * wave + 2] = S3; }
    __syncthreads();
    S1 = 0; S2 = 0; S3 = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) { S1 += icSum[3 * w]; S2 += icSum[3 * w + 1]; S3 += icSum[3 * w + 2]; }
  }
  const int m10 = S2 - 15 * S1, m01 = S3 - 15 * S1;
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // GaussianBlur 7x7, sigma 2 (ORBextractor.cc:950) in OpenCV's 8.8 fixed point, kernel [18,34,48,56,48,34,18]/256:
  // exact integer sums with one rounding at the end, so the two passes commute.  Horizontal pass on the raw bytes with
  // v_dot4_u32_u8, 4 outputs per lane, written TRANSPOSED as u16 so that the vertical pass finds vertically adjacent
  // taps in one dword and runs on v_dot2_u32_u16 (4 outputs per lane, one dword store into the transposed patch).
  {
    // The byte alignment `pa` of the patch is the same for the whole block: one of four instantiations of the loop is taken by a
    // scalar branch, and in each the seven taps of output j sit at FIXED bytes pa + j .. pa + j + 6 of the item's four dwords -- the
    // tap weights are laid over those dwords as constants (zero elsewhere), so an item is 10 v_dot4 and nothing else: no funnel
    // shifts to align the data (3 + 6 of them before round 4's second half).
    auto hpass = [&](auto paTag) {
      constexpr int PA = decltype(paTag)::value;
#pragma unroll
      for (int it = 0; it < NH; it++) {
        if ((it + 1) * NT > kBlurItemsHost.nh && it * NT + tid >= kBlurItemsHost.nh) break;   // only the last iteration is partial
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(raw + (bh[it] & 0xffffu));
        const uint32_t d[4] = {rp[0], rp[1], rp[2], rp[3]};   // raw bytes 4g .. 4g+15 of row y; pixel x of the row is byte PA + x
        uint16_t* o = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(hbT) + (bh[it] >> 16));
        o[0] = (uint16_t)blur_row_out<PA + 0, GAUSS>(d);
        o[kHPT] = (uint16_t)blur_row_out<PA + 1, GAUSS>(d);
        o[2 * kHPT] = (uint16_t)blur_row_out<PA + 2, GAUSS>(d);
        o[3 * kHPT] = (uint16_t)blur_row_out<PA + 3, GAUSS>(d);
      }
    };
    switch (pa) {
      case 0: hpass(std::integral_constant<int, 0>{}); break;
      case 1: hpass(std::integral_constant<int, 1>{}); break;
      case 2: hpass(std::integral_constant<int, 2>{}); break;
      default: hpass(std::integral_constant<int, 3>{}); break;
    }
  }
  __syncthreads();
  {
    // (lo, hi) tap pairs for an output whose first tap is the LOW half of p0 (even) or the HIGH half (odd)
    constexpr uint16_t T0 = blur_tap<GAUSS>(0), T1 = blur_tap<GAUSS>(1), T2 = blur_tap<GAUSS>(2), T3 = blur_tap<GAUSS>(3);
    const u16x2 E0 = {T0, T1}, E1 = {T2, T3}, E2 = {T2, T1}, E3 = {T0, 0};
    const u16x2 O0 = {0, T0}, O1 = {T1, T2}, O2 = {T3, T2}, O3 = {T1, T0};
#pragma unroll
    for (int it = 0; it < NV; it++) {
      if ((it + 1) * NT > kBlurItemsHost.nv && it * NT + tid >= kBlurItemsHost.nv) break;
      const uint32_t* cp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(hbT) + (bv[it] & 0xffffu));
      u16x2 p[5];
#pragma unroll
      for (int t = 0; t < 5; t++) p[t] = as_u16x2(cp[t]);   // rows 4yq .. 4yq+9 of column x
      unsigned r4[4];   // (sum + 32768): the blurred pixel is byte 2 of each (ED taps: sum <= 255 * 65536)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        unsigned e = 32768u, o = 32768u;
        e = __builtin_amdgcn_udot2(p[h], E0, e, false);     o = __builtin_amdgcn_udot2(p[h], O0, o, false);
        e = __builtin_amdgcn_udot2(p[h + 1], E1, e, false); o = __builtin_amdgcn_udot2(p[h + 1], O1, o, false);
        e = __builtin_amdgcn_udot2(p[h + 2], E2, e, false); o = __builtin_amdgcn_udot2(p[h + 2], O2, o, false);
        e = __builtin_amdgcn_udot2(p[h + 3], E3, e, false); o = __builtin_amdgcn_udot2(p[h + 3], O3, o, false);
        if constexpr (GAUSS == 1) {   // saturate_cast<uchar>: (sum + 32768) >> 16 can be 256 or 257 with taps that add up to 257
          e = e < 0x00ffffffu ? e : 0x00ffffffu;
          o = o < 0x00ffffffu ? o : 0x00ffffffu;
        }
        r4[2 * h] = e;
        r4[2 * h + 1] = o;
      }
      // bytes 2 of the four sums -> one dword, by two byte permutes (v_perm_b32: selector 0..3 = second operand, 4..7 = first)
      const unsigned packed = __builtin_amdgcn_perm(r4[1], r4[0], 0x0c0c0602u) | __builtin_amdgcn_perm(r4[3], r4[2], 0x06020c0cu);
      *reinterpret_cast<uint32_t*>(blT + (bv[it] >> 16)) = packed;   // rows 37..39 are padding
    }
  }
  __syncthreads();

  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float a, b;
  // the angle is the same in every lane (wave sums): as a scalar, the range tests inside sincosf become uniform
  // branches and only the polynomials of the taken path are evaluated (they are double precision: half rate)
  const float rad = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, angle * factorPI)));
  sincosf_glibc(rad, &b, &a);  // a = cos, b = sin
  const uint8_t* center = blT + kBlurRad * kBPT + kBlurRad;   // center[x * kBPT + y]
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 AB = {a, b}, NBA = {-b, a}, kRoundMagic = {12582912.0f, 12582912.0f};
  unsigned long long words[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (WAVES == 4 && j != wave) continue;   // four waves: wave j tests bits 64 j .. 64 j + 63
    const int bit = j * 64 + lane;
    const float4 p = *reinterpret_cast<const float4*>(&c_patternF.v[4 * bit]);
    // (column, row) = (x a - y b, x b + y a) (ORBextractor.cc:147-151), both coordinates per packed-float instruction: the products
    // and the sum are the reference's own operations (x a + (-(y b)) is x a - y b exactly), unfused.  cvRound by the float adder:
    // t + 1.5 * 2^23 is rounded to an integer, ties to even, and that integer sits in the low mantissa bits (|t| <= 19).
    const f32x2 c1 = (f32x2){p.x, p.x} * AB + (f32x2){p.y, p.y} * NBA + kRoundMagic;
    const f32x2 c2 = (f32x2){p.z, p.z} * AB + (f32x2){p.w, p.w} * NBA + kRoundMagic;
    // low 24 bits of a rounded float = 0x400000 + n: (0x400000 + col) * kBPT + (0x4B400000 + row) - K = col * kBPT + row
    constexpr unsigned K = 0x400000u * (unsigned)kBPT + 0x4B400000u;
    // (elements copied to plain floats first: this compiler's __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
    const float c1x = c1.x, c1y = c1.y, c2x = c2.x, c2y = c2.y;
    const int i1 = (int)(__umul24(__float_as_uint(c1x), (unsigned)kBPT) + __float_as_uint(c1y) - K);
    const int i2 = (int)(__umul24(__float_as_uint(c2x), (unsigned)kBPT) + __float_as_uint(c2y) - K);
    const int t0 = center[i1], t1 = center[i2];
    words[j] = __ballot(t0 < t1);
  }
  uint8_t* descOut2 = A.descOut2;
  if (WAVES == 4) {
    if (lane == 0) {
      reinterpret_cast<unsigned long long*>(descOut + (long long)k * 32)[wave] = words[wave];
      if (descOut2) reinterpret_cast<unsigned long long*>(descOut2 + (long long)k * 32)[wave] = words[wave];
    }
  } else if (lane < 4) {
    reinterpret_cast<unsigned long long*>(descOut + (long long)k * 32)[lane] = words[lane];
    if (descOut2) reinterpret_cast<unsigned long long*>(descOut2 + (long long)k * 32)[lane] = words[lane];
  }
  if (tid == 0) {
    angleOut[k] = angle;
    if (A.angleOut2) A.angleOut2[k] = angle;
  }
}

__global__ void k_sincos(const float* deg, int n, float* c, float* s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  sincosf_glibc(deg[i] * factorPI, &s[i], &c[i]);
}

// ------------------------------------------------------------------------------------------------
// Colour -> gray for frames handed over as interleaved RGB / BGR / RGBA / BGRA (Tracking::GrabImageMonocular,
// Tracking.cc:96-109: cvtColor(RGB2GRAY / BGR2GRAY) before the Frame is built).  OpenCV's 8-bit path:
// gray = (src[0]*c0 + src[1]*c1 + src[2]*c2 + (1 << (shift-1))) >> shift, alpha ignored.  Four pixels per thread
// (3 dwords or one dwordx4 in, one dword out); unaligned sources and row tails take the byte path.
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ void __launch_bounds__(256) k_to_gray(const uint8_t* const* __restrict__ raw, long long rawStride,
                                                 const uint8_t* const* __restrict__ gray, long long grayPitch, int cols,
                                                 int c0, int c1, int c2, int shift, int aligned) {
  const int x = (int)(blockIdx.x * 256 + threadIdx.x) * 4;
  if (x >= cols) return;
  const int y = blockIdx.y, f = blockIdx.z;
  const uint8_t* src = raw[f] + (long long)y * rawStride + (long long)x * CH;
  uint8_t* dst = const_cast<uint8_t*>(gray[f]) + (long long)y * grayPitch + x;
  const int rnd = 1 << (shift - 1);
  if (aligned && x + 4 <= cols) {
    uint32_t w[CH];
    if (CH == 3) {
      const uint32_t* s32 = (const uint32_t*)src;
      w[0] = s32[0]; w[1] = s32[1]; w[2] = s32[2];
    } else {
      const uint4 q = *(const uint4*)src;
      w[0] = q.x; w[1] = q.y; w[2] = q.z; w[CH - 1] = q.w;
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t a, b, c;
      if (CH == 3) {   // bytes 3i, 3i+1, 3i+2 of the 12-byte group
        a = (w[(3 * i) >> 2] >> (((3 * i) & 3) * 8)) & 255u;
        b = (w[(3 * i + 1) >> 2] >> (((3 * i + 1) & 3) * 8)) & 255u;
        c = (w[(3 * i + 2) >> 2] >> (((3 * i + 2) & 3) * 8)) & 255u;
      } else {
        a = w[i] & 255u; b = (w[i] >> 8) & 255u; c = (w[i] >> 16) & 255u;
      }
      out |= ((uint32_t)(m24((int)a, c0) + m24((int)b, c1) + m24((int)c, c2) + rnd) >> shift) << (8 * i);
    }
    *(uint32_t*)dst = out;
  } else {
    for (int i = 0; i < 4 && x + i < cols; i++)
      dst[i] = (uint8_t)((m24(src[CH * i], c0) + m24(src[CH * i + 1], c1) + m24(src[CH * i + 2], c2) + rnd) >> shift);
  }
}

void launch_to_gray(const uint8_t* const* raw, long long rawStride, const uint8_t* const* gray, long long grayPitch, int rows,
                    int cols, int channels, const int coef[3], int shift, bool aligned, int nframes, hipStream_t st) {
  dim3 grid((cols + 1023) / 1024, rows, nframes);
  if (channels == 3)
    hipLaunchKernelGGL(k_to_gray<3>, grid, dim3(256), 0, st, raw, rawStride, gray, grayPitch, cols, coef[0], coef[1], coef[2],
                       shift, aligned ? 1 : 0);
  else
    hipLaunchKernelGGL(k_to_gray<4>, grid, dim3(256), 0, st, raw, rawStride, gray, grayPitch, cols, coef[0], coef[1], coef[2],
                       shift, aligned ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// Launchers (called by the host engine).
// ------------------------------------------------------------------------------------------------
// ORBFE_LDS_DMA=0 sends the pyramid footprints and the descriptor patches through registers instead of LDS-DMA: the route a
// frame's last row and strides that are not multiples of 4 take anyway, forced everywhere (tests/test_gpu_parity.py runs it in a
// process of its own).
static int lds_dma_enabled() {
  static const int v = [] { const char* e = getenv("ORBFE_LDS_DMA"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();
  return v;
}

// ---- k_ingest: a page-locked HOST frame -> its device copy, by a kernel on the compute stream ---------------------------------
// The one-frame call of Frame.cc:133 hands over a host cv::Mat.  A copy command on the upload lane followed by an event the compute
// stream waits for costs the copy (43 us for 2 MB) plus 18.5 us between the copy's end and the first kernel's start (two queues, one
// signal).  Here the kernels' own stream reads the frame over PCIe itself: 16 bytes per lane, four requests in flight per lane, every
// wave-instruction a contiguous kilobyte of a row; the next kernel starts behind it like behind any other kernel.
// V = bytes per lane and request (16: pointers, strides and the row length are multiples of 16; 4: multiples of 4; 1 otherwise).
template <int V, int U>
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ src, long long sstride, uint8_t* __restrict__ dst, long long dpitch,
                                                int rowBytes, int rows) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef typename std::conditional<V == 16, u32x4, typename std::conditional<V == 4, uint32_t, uint8_t>::type>::type T;
  const int perRow = (rowBytes + V - 1) / V;                       // V == 1 / 4 / 16: exact by the launcher's choice of V
  const int total = perRow * rows;
  const float rcpRow = 1.0f / (float)perRow;
  const int i0 = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
  // U independent requests per lane before the first store
  long long so[U], dof[U];
  bool ok[U];
  T v[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int i = i0 + u * step;
    ok[u] = i < total;
    const int ii = ok[u] ? i : 0;
    int yy = (int)(((float)ii + 0.5f) * rcpRow);
    int xx = ii - yy * perRow;
    if (xx < 0) { yy--; xx += perRow; }                             // the float quotient may be one off near a row boundary
    if (xx >= perRow) { yy++; xx -= perRow; }
    so[u] = (long long)yy * sstride + (long long)V * xx;
    dof[u] = (long long)yy * dpitch + (long long)V * xx;
  }
#pragma unroll
  for (int u = 0; u < U; u++)
    if (ok[u]) v[u] = __builtin_nontemporal_load(reinterpret_cast<const T*>(src + so[u]));
#pragma unroll
  for (int u = 0; u < U; u++)
    if (ok[u]) *reinterpret_cast<T*>(dst + dof[u]) = v[u];
}

// host (page-locked, device-readable) frame -> device copy on `st`
void launch_ingest(const uint8_t* src, long long sstride, uint8_t* dst, long long dpitch, int rowBytes, int rows, hipStream_t st) {
  const uintptr_t all = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)sstride | (uintptr_t)dpitch | (uintptr_t)rowBytes;
  const int V = (all & 15) == 0 ? 16 : (all & 3) == 0 ? 4 : 1;
  constexpr int U = 4;   // requests in flight per lane (1 / 2 / 4 / 8 measured: 0.157 / 0.156 / 0.149 / 0.149 ms per call)
  const long long total = (long long)((rowBytes + V - 1) / V) * rows;
  const int blocks = (int)std::max<long long>(1, (total + 256 * U - 1) / (256 * U));
#define ORBFE_INGEST(VV, UU) hipLaunchKernelGGL((k_ingest<VV, UU>), dim3(blocks), dim3(256), 0, st, src, sstride, dst, dpitch, rowBytes, rows)
  if (V == 16) ORBFE_INGEST(16, 4);
  else if (V == 4) ORBFE_INGEST(4, 4);
  else ORBFE_INGEST(1, 4);
#undef ORBFE_INGEST
}

int launch_pyramid(const PyramidParams& P, int nframes, hipStream_t st, const ConeParams* cone) {
  const int last = cone ? cone->base : P.nlevels - 1;   // levels built one launch each
  if (cone) {
    static std::mutex mu;            // see launch_qt3: check + set + record is one critical section
    static int attrBytes[64] = {};   // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    std::lock_guard<std::mutex> lk(mu);
    if (cone->ldsBytes > attrBytes[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyramid_cone), hipFuncAttributeMaxDynamicSharedMemorySize,
                              cone->ldsBytes) != hipSuccess)
        return 1;
      attrBytes[dev] = cone->ldsBytes;
    }
  }
  for (int l = 1; l <= last; l++) {
    dim3 grid((P.lv[l].w + kRzTile - 1) / kRzTile, (P.lv[l].h + kRzTile - 1) / kRzTile, nframes);
    // (tilesX == 1 -- a level at most 64 px wide -- takes the generic kernel: 2^32 / 1 + 1 wraps to rcp = 1 and the multiply-high
    // below would return row 0 for every tile; exact only for tilesX >= 2)
    if (P.lv[l].rzPitch <= 96 && P.lv[l].rzRows <= 80 && grid.x >= 2) {
      const int tilesX = (int)grid.x, ntiles = (int)(grid.x * grid.y);
      const unsigned rcp = (unsigned)(0x100000000ull / (unsigned)tilesX) + 1u;   // floor(t * rcp / 2^32) = t / tilesX for t * tilesX < 2^32
      if ((unsigned long long)ntiles * tilesX < (1ull << 31)) {
        const int dma = lds_dma_enabled();
        ResizeLevel R{};
        R.dw = P.lv[l].w; R.dh = P.lv[l].h; R.dpitch = P.lv[l].pitch;
        R.sw = P.lv[l - 1].w; R.sh = P.lv[l - 1].h;
        R.sstride = l == 1 ? P.stride0 : (long long)P.lv[l - 1].pitch;
        R.doff = P.lv[l].off; R.soff = P.lv[l - 1].off;
        R.xofs = P.lv[l].xofs; R.xalpha = P.lv[l].xalpha; R.yofc = P.lv[l].yofc; R.ybeta = P.lv[l].ybeta;
        R.frame0 = P.frame0; R.frameInline[0] = P.frameInline[0]; R.frameInline[1] = P.frameInline[1];
        R.slab = P.slab; R.slabBytes = P.slabBytes; R.frameBase = P.frameBase; R.fromLevel0 = l == 1;
        hipLaunchKernelGGL((k_resize_fixed<96, 80>), dim3(8 * ((ntiles + 7) / 8), nframes), dim3(256), 0, st, R, tilesX, ntiles, rcp, dma);
        continue;
      }
    }
    hipLaunchKernelGGL(k_resize, grid, dim3(256), (((size_t)P.lv[l].rzPitch * P.lv[l].rzRows + 15) & ~(size_t)15) + (size_t)P.lv[l].rzRows * 64 * 2, st,
                       P, l);
  }
  if (cone)
    hipLaunchKernelGGL(k_pyramid_cone, dim3(cone->tilesX, cone->tilesY, nframes), dim3(kConeThreads), cone->ldsBytes, st, P, *cone);
  return 0;
}

void launch_compact(const PyramidParams& P, int nframes, hipStream_t st) {
  hipLaunchKernelGGL(k_compact, dim3((P.ncells + 63) / 64, nframes), dim3(64), 0, st, P);
}
void launch_compact_local(const PyramidParams& P, int nframes, hipStream_t st, int level0, int level1) {
  if (level1 > P.nlevels) level1 = P.nlevels;
  int maxCells = 0;
  for (int l = level0; l < level1; l++) maxCells = std::max(maxCells, P.lv[l].nCols * P.lv[l].nRows);
  if (maxCells <= 0) return;
  hipLaunchKernelGGL(k_compact_local, dim3((maxCells + 63) / 64, level1 - level0, nframes), dim3(64), 0, st, P, level0);
}

static DescribeArgs describe_args(const PyramidParams& P, const SelKp* sel, int nsel, float* angle, uint8_t* desc) {
  DescribeArgs A{};
  for (int l = 0; l < kMaxLevels; l++) {
    A.lv[l].w = P.lv[l].w; A.lv[l].h = P.lv[l].h; A.lv[l].pitch = P.lv[l].pitch; A.lv[l].off = P.lv[l].off;
  }
  A.sel = sel; A.selCount = nullptr; A.angleOut = angle; A.descOut = desc; A.angleOut2 = nullptr; A.descOut2 = nullptr;
  A.frame0 = P.frame0; A.frameInline[0] = P.frameInline[0]; A.frameInline[1] = P.frameInline[1];
  A.stride0 = P.stride0; A.slab = P.slab; A.slabBytes = P.slabBytes;
  A.nsel = nsel; A.selPerFrame = 1; A.nlevels = P.nlevels; A.frameBase = P.frameBase; A.dma = lds_dma_enabled();
  A.gauss = P.gaussVariant;
  return A;
}

void launch_describe(const PyramidParams& P, const SelKp* sel, int nsel, float* angle, uint8_t* desc,
                     hipStream_t st) {
  if (nsel <= 0) return;
  const DescribeArgs A = describe_args(P, sel, nsel, angle, desc);
  if (A.gauss == 1) hipLaunchKernelGGL((k_describe<1, 1>), dim3(8 * ((nsel + 7) / 8)), dim3(64), 0, st, A);
  else hipLaunchKernelGGL((k_describe<1, 0>), dim3(8 * ((nsel + 7) / 8)), dim3(64), 0, st, A);
}

// sel/angle/desc point at the first slot of frame P.frameBase; nslots = nframes * selPerFrame
void launch_describe_slots(const PyramidParams& P, const SelKp* sel, int nslots, float* angle, uint8_t* desc,
                           const uint32_t* selCount, int selPerFrame, const int* selOff, hipStream_t st, bool fourWaves, float* angle2,
                           uint8_t* desc2) {
  if (nslots <= 0) return;
  DescribeArgs A = describe_args(P, sel, nslots, angle, desc);
  A.angleOut2 = angle2;
  A.descOut2 = desc2;
  A.selCount = selCount;
  A.selPerFrame = selPerFrame;
  for (int l = 0; l <= P.nlevels; l++) A.selOff[l] = selOff[l];
  const dim3 grid(8 * ((nslots + 7) / 8));
  if (fourWaves) {
    if (A.gauss == 1) hipLaunchKernelGGL((k_describe<4, 1>), grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_describe<4, 0>), grid, dim3(256), 0, st, A);
  } else {
    if (A.gauss == 1) hipLaunchKernelGGL((k_describe<1, 1>), grid, dim3(64), 0, st, A);
    else hipLaunchKernelGGL((k_describe<1, 0>), grid, dim3(64), 0, st, A);
  }
}

// One wave that does nothing for `ticks` periods of the 100 MHz reference clock: the stream runner's probe of how many of its HIP
// streams the runtime lets run side by side (streams folded onto one hardware queue serialise).
__global__ void k_spin(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); n++; }
  if (sink && n == 0xffffffffu) *sink = n;
}
void launch_spin(unsigned long long ticks, hipStream_t st) {
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, (unsigned*)nullptr);
}

void launch_sincos(const float* deg, int n, float* c, float* s, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, st, deg, n, c, s);
}

}  // namespace orbfe

#ifdef ORBFE_EXPERIMENTS
extern "C" int orbfe_exp_cone_stamps(unsigned long long* out /* [32] */) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(orbfe::g_coneStamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif
