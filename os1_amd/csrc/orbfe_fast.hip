// orbfe_fast.hip -- FAST-9/16 per reference cell, the dominant kernel of the extractor (gfx950, wave64).
//
// Reference: ComputeKeyPointsOctTree, src/ORBextractor.cc:797-870 (cv::FAST calls at :848,:854); cv::FAST
// semantics: SURVEY.md Appendix B.1.
//
// Formulation (DESIGN.md "FAST"): with S(p) = max(max_arc min(v-ring), max_arc min(ring-v)) a pixel is a corner at
// threshold t iff S(p) > t and its OpenCV score is S(p)-1, independent of t.  A cell's cv::FAST(t, nms) output is
// therefore {p in emit region : S(p) > t and s(p) > s(q) for the 8 neighbours q inside the SAME cell's emit region},
// s = S-1 where S > t and 0 elsewhere.  The emit region of cell (i,j) is [19+j*wCell, min(19+(j+1)*wCell, w-19)) x
// [19+i*hCell, ...); the regions tile the level exactly.
//
// Work decomposition: ONE WAVE PER TASK, a task being one cell or two horizontally adjacent cells of one cell row
// (FastTask; the union's ROI is one contiguous tile of at most 70 x (hCell+6) bytes).  Two cells per wave fill the
// 64 lanes of the score stage (a 31x31 cell leaves about 91 candidates = 1.4 wave iterations; two cells need 3 instead
// of 4) and halve the per-wave fixed cost, at less LDS per cell than one cell per wave.  All stages keep row-major
// order over the union, which restricted to either cell is that cell's row-major order = cv::FAST's emission order:
//   stage 1  every emit pixel, 16 per lane (one row, 16 adjacent columns): compass pre-test (a 9-arc always contains
//            two ADJACENT compass points of one polarity) on packed 16-bit lanes -> DPP-scan-compacted queue
//   stage 2  queue: S = max over both polarities of max_arc min9, both polarities per packed op; S > t keeps the
//            entry (compacted in place) and writes the score S-1 to the score tile
//   stage 3  queue: 3x3 NMS against the score tile (neighbours across the boundary between the two cells count 0),
//            ballot-ordered emission into each cell's slots
// Two passes at most, like the reference (:846-856): cv::FAST at iniThFAST; only a cell left without a keypoint (after
// NMS) is emitted again from a pass at minThFAST.
// Blocks are remapped so that the blocks an XCD receives (b, b+8, b+16, ...) are CONSECUTIVE tasks: neighbouring cells
// share ROI halos and cache lines in that XCD's L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "orbfe_internal.h"

namespace orbfe {

namespace {
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }
}  // namespace

// integer min / max of three packed 16-bit values whose bit patterns are normal positive half floats (see stage 2)
__device__ __forceinline__ unsigned pk_min3_h(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ unsigned pk_max3_h(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// LDS traffic of one wave needs no s_barrier (a block is one wave; LDS operations of a wave execute in order): a
// compiler-level memory fence plus the LDS counter is enough.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// NPX = pixels per lane in the pre-test (8 or 16)
// PAIRS = false: every task is a single cell (the default task table); the second cell's bookkeeping compiles away
// ABL = 0 is the product, and the only instantiation of a default build.  A build with -DORBFE_EXPERIMENTS (make EXPERIMENTS=1)
// also instantiates the measurement modes ORBFE_FAST_ABLATE selects (tools/fast_ablation.sh, tools/fast_phases.py): the kernel stops
// after its set-up + ROI load (1), after the pre-test and its compaction (2), after the score stage (3) -- every cell then reports no
// candidate; 4 = the product with s_memtime stamps between its phases, one record per wave (orbfe_debug_fast_stamps); 5 = the product
// without the second pass at minThFAST (what that pass costs)
// What the kernel needs of PyramidParams, as a compact argument block of its own (136 bytes): the whole of it arrives with the
// first scalar load of a wave; out of the 1.4 KB PyramidParams the fields came in three dependent groups, the last one behind
// the task fetch.
struct FastArgs {
  const FastTask* tasks;
  const uint8_t* const* frame0;
  const uint8_t* frameInline[2];
  long long stride0;
  uint8_t* slab;
  long long slabBytes;
  uint32_t* cellCount;
  uint32_t* slots;
  long long slotsPerFrame;
  const uint8_t* zeros;
  int ncells, ntasks, iniTh, minTh, frameBase;
};
static FastArgs fast_args(const PyramidParams& P) {
  FastArgs A{};
  A.tasks = P.tasks; A.frame0 = P.frame0; A.frameInline[0] = P.frameInline[0]; A.frameInline[1] = P.frameInline[1];
  A.stride0 = P.stride0; A.slab = P.slab; A.slabBytes = P.slabBytes; A.cellCount = P.cellCount; A.slots = P.slots;
  A.slotsPerFrame = P.slotsPerFrame; A.zeros = P.zeros; A.ncells = P.ncells; A.ntasks = P.ntasks; A.iniTh = P.iniTh; A.minTh = P.minTh;
  A.frameBase = P.frameBase;
  return A;
}

__device__ uint32_t* g_fastStampBuf;   // [waves of the launch][8]
// LEAN (round 5): the prologue for the case every batch launch is in -- LDS-DMA with 16 bytes per lane, row pitches that are
// multiples of 4, the zero line present, and every per-frame offset (frame * slab bytes, frame * slots, frame * cells) below 4 GiB,
// all checked on the HOST (launch_fast) instead of by every wave: no staging-mode branches, the LDS carve and the rows-per-instruction
// quotient come precomputed in the task record (FastTask::geo), pointer arithmetic is 32-bit offsets on 64-bit bases.  The set-up was
// 242 of the wave's 339 scalar instructions (profiles/r04_fast_ablation.txt); the scalar port is a shared resource of the CU
// (profiles/r05_salu_rate.txt) and those instructions sit in front of the wave's first memory request.
template <int NPX, bool PAIRS, int ABL = 0, bool LEAN = false>
// (amdgpu_num_sgpr(96): the kernel asks for 105 scalar registers by itself, which caps a SIMD at 6 waves; 94 with 3 values parked in
// a vector register allow 7 -- +1 % in the pipeline, 80 / 88 measured the same)
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void k_fast_tasks(FastArgs P, int t0, int nt) {
  extern __shared__ __align__(16) uint8_t lds[];
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (ABL == 4) stamp[0] = __builtin_amdgcn_s_memtime();
  // this launch works on tasks [t0, t0 + nt): the levels of one LDS class (launch_fast)
  // the whole argument block is requested by the wave's first scalar loads, before anything waits (left alone the compiler
  // fetches t0 / nt, tests the bound below, and only then asks for the rest)
  asm volatile("" ::"s"(P.tasks), "s"(P.frame0), "s"(P.frameInline[0]), "s"(P.frameInline[1]), "s"(P.stride0), "s"(P.slab), "s"(P.slabBytes),
               "s"(P.cellCount), "s"(P.slots), "s"(P.slotsPerFrame), "s"(P.zeros), "s"(P.ncells), "s"(P.iniTh), "s"(P.minTh), "s"(P.frameBase),
               "s"(t0), "s"(nt));
  const int chunk = (nt + 7) >> 3;
  const int tloc = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tloc >= nt) return;
  const int tix = t0 + tloc;
  const int f = P.frameBase + blockIdx.y;
  const int lane = threadIdx.x;
  // A wave's life begins with memory round trips it cannot overlap with anything: keep that chain SHORT.  The task record
  // (32 bytes: cell, slots AND the level's geometry) and the level-0 pointer of the frame come by two SCALAR loads issued
  // together -- one round trip through the scalar cache -- and then the ROI loads go out.  (Before round 4: the task by vector
  // loads because of its sub-dword fields, a second vector round trip for the rest of it, the level table by a dependent
  // scalar load, the frame pointer by a dependent flat load: eight dependent round trips, 2-3 us of a 7 us wave.)
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x8 tw = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(P.tasks + tix));
  u32x2 fpw;
  if (P.frame0) {
    fpw = *reinterpret_cast<const u32x2 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(P.frame0 + f));
  } else {   // one- and two-frame calls carry the level-0 pointers in the kernel arguments
    const unsigned long long v = reinterpret_cast<unsigned long long>((f & 1) ? P.frameInline[1] : P.frameInline[0]);
    fpw.x = (uint32_t)v;
    fpw.y = (uint32_t)(v >> 32);
  }
  const int ex0 = (int)(tw[0] & 0xffffu), ey0 = (int)(tw[0] >> 16);
  const int ew0 = (int)(tw[1] & 0xffu), ew1 = PAIRS ? (int)((tw[1] >> 8) & 0xffu) : 0, eh = (int)((tw[1] >> 16) & 0xffu), level = (int)(tw[1] >> 24);
  const uint32_t cell0 = tw[2], slotOff0 = tw[3], roiOff = tw[4], pitchL = tw[5];
  const int fastW = (int)(tw[6] & 0xffu), hCell = (int)((tw[6] >> 8) & 0xffu), slotCap = (int)(tw[6] >> 16);
  uint32_t* cnt;
  if constexpr (LEAN) cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.cellCount) + 4u * ((uint32_t)f * (uint32_t)P.ncells + cell0));
  else cnt = P.cellCount + (long long)f * P.ncells + cell0;
  if (ew0 == 0) {
    if (lane == 0) cnt[0] = 0;
    return;
  }
  const int W2 = ew0 + ew1;   // emit width of the task
  long long stride;
  const uint8_t* roi;
  if constexpr (LEAN) {
    // 32-bit offsets (the host has checked that they fit) on the two possible 64-bit bases, selected without a branch
    const bool l0 = level == 0;
    const uint32_t st32 = l0 ? (uint32_t)P.stride0 : pitchL;
    const uint32_t off = l0 ? (uint32_t)(ey0 - 3) * st32 + (uint32_t)(ex0 - 3) : (uint32_t)f * (uint32_t)P.slabBytes + roiOff;
    const uint8_t* base = l0 ? reinterpret_cast<const uint8_t*>(((unsigned long long)fpw.y << 32) | fpw.x) : P.slab;
    stride = (long long)st32;
    roi = base + off;
  } else if (level == 0) {
    stride = P.stride0;
    roi = reinterpret_cast<const uint8_t*>(((unsigned long long)fpw.y << 32) | fpw.x) + (long long)(ey0 - 3) * stride + (ex0 - 3);
  } else {
    stride = (long long)pitchL;
    roi = P.slab + (long long)f * P.slabBytes + roiOff;
  }
  // LDS carve (level-uniform): ROI tile, score tile with a zero ring, queue (y<<8|x)
  // (the tile pitch is a multiple of 16 -- the ROI arrives in 16-byte LDS-DMA pieces -- that holds alignment offset + widest ROI row)
  // LEAN: the carve comes with the task (FastTask::geo = pieces per tile row | rows per instruction << 3 | score tile offset / 16 << 8 |
  // queue offset / 16 << 18, written by fast_task_geo() below from the same formulas)
  const uint32_t geo = tw[7];
  const int TP = LEAN ? (int)(geo & 7u) << 4 : ((fastW + 6 + 3 + 15) & ~15);
  const int SP = fastW + 2;
  uint8_t* tile = lds;
  const int scOff = LEAN ? (int)((geo >> 8) & 0x3ffu) << 4 : (TP * (hCell + 6) + 15) & ~15;   // 16-byte aligned: it is cleared by 16-byte LDS-DMA pieces
  uint8_t* sc = tile + scOff;
  uint16_t* queue = reinterpret_cast<uint16_t*>(lds + (LEAN ? (int)((geo >> 18) & 0x7ffu) << 4 : ((scOff + SP * (hCell + 2) + 15) & ~15)));

  const int rw = W2 + 6, rh = eh + 6;
  const int istr = (int)stride;
  // ROI -> LDS by LDS-DMA where the row pitch is a multiple of 4 (every level of the pyramid slab; level 0 when the caller's
  // stride allows it), byte by byte otherwise.  `a` is the byte offset of the ROI inside its first dword.
  const int a = (int)(reinterpret_cast<uintptr_t>(roi) & 3);
  if constexpr (ABL == 4) { asm volatile("" ::"s"(a), "s"(istr), "s"(TP) : "memory"); stamp[1] = __builtin_amdgcn_s_memtime(); }
  if constexpr (LEAN) {
    // the staging below with everything wave-uniform taken from the task: pieces per row, rows per instruction
    const int ppr = (int)(geo & 7u), rpi = (int)((geo >> 3) & 31u);
    const float rn = __builtin_amdgcn_rcpf((float)ppr);
    const int lrow = (int)(((float)lane + 0.5f) * rn), lcol = lane - m24(lrow, ppr);
    const bool on = lrow < rpi && 16 * lcol < a + rw;
    const unsigned voff = (unsigned)(m24(lrow, istr) + 16 * lcol);
    const uint8_t* gp = roi - a;
    const uint32_t gstep = (uint32_t)rpi * (uint32_t)istr;
    const int lstep = rpi * TP;
    uint8_t* lp = tile;
#pragma nounroll
    for (int left = rh; left > 0; left -= rpi, gp += gstep, lp += lstep) {
      unsigned vo = voff;
      asm volatile("" : "+s"(gp), "+v"(vo));
      if (on && lrow < left)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + vo),
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
    }
  } else if ((stride & 3) == 0) {
    // 16 bytes per lane (global_load_lds_dwordx4): TP / 16 pieces per tile row, 64 / pieces rows per instruction -- the 37-row tile
    // of a single cell is TWO instructions (seven with one dword per lane); the global side needs dword alignment only, a row's last
    // piece may reach up to 11 bytes past the ROI: inside the level's row (an emit region stays 16 pixels off the border)
    const int ppr = TP >> 4;
    const float rn = __builtin_amdgcn_rcpf((float)ppr);
    const int rpi = 64 / ppr;
    const int lrow = (int)(((float)lane + 0.5f) * rn), lcol = lane - m24(lrow, ppr);
    const bool on = lrow < rpi && 16 * lcol < a + rw;
    const unsigned voff = (unsigned)(m24(lrow, istr) + 16 * lcol);
    const uint8_t* gp = roi - a;
    const long long gstep = (long long)rpi * stride;
    const int lstep = m24(rpi, TP);
    uint8_t* lp = tile;
    for (int left = rh; left > 0; left -= rpi, gp += gstep, lp += lstep) {
      unsigned vo = voff;
      asm volatile("" : "+s"(gp), "+v"(vo));
      if (on && lrow < left)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + vo),
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
    }
  } else {
    const float rcpRw = 1.0f / (float)rw;
    const int total = rw * rh;
    for (int i0 = lane; i0 < total; i0 += 64 * 8) {
      uint8_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * 64;
        v[u] = 0;
        if (i < total) {
          const int y = (int)(((float)i + 0.5f) * rcpRw), x = i - m24(y, rw);
          v[u] = roi[m24(y, istr) + x];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * 64;
        if (i < total) {
          const int y = (int)(((float)i + 0.5f) * rcpRw), x = i - m24(y, rw);
          tile[m24(y, TP) + a + x] = v[u];
        }
      }
    }
  }
  tile += a;  // ROI pixel (x, y) lives at tile[y * TP + x]
  if constexpr (LEAN) {
    // (as below; the loop is kept a loop: unrolled eight times by the compiler it was thirty scalar instructions of preamble for a
    // trip count of one or two)
    const int pieces = (SP * (eh + 2) + 15) >> 4;
#pragma nounroll
    for (int p0 = 0; p0 < pieces; p0 += 64)
      if (lane < pieces - p0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)P.zeros,
                                         (__attribute__((address_space(3))) void*)(sc + 16 * p0), 16, 0, 0);
  } else {
    // the score tile is cleared by LDS-DMA too: every lane fetches the same 16 zero bytes (one cache line for the whole chip) and
    // lane L's copy lands at base + 16 L -- two instructions for the 1.1 KB tile, no vector instruction, no ds_write.  (The last
    // piece may run up to 15 bytes into the queue, which stage 1 writes later.)
    const int pieces = (SP * (eh + 2) + 15) >> 4;
    for (int p0 = 0; p0 < pieces; p0 += 64)
      if (lane < pieces - p0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)P.zeros,
                                         (__attribute__((address_space(3))) void*)(sc + 16 * p0), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA writes are tracked by vmcnt)
  wave_lds_fence();
  if constexpr (ABL == 4) stamp[2] = __builtin_amdgcn_s_memtime();
  if constexpr (ABL == 1) {   // keep the ROI load alive: fold the tile into a word nobody reads as a candidate
    unsigned acc = 0;
    for (int i = lane; i < (TP * rh) >> 2; i += 64) acc ^= reinterpret_cast<const uint32_t*>(tile - a)[i];
    if (acc == 0x9e3779b9u && lane == 63) cnt[0] = 0;
    if (lane == 0) cnt[0] = 0;
    return;
  }

  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t* slot0;
  if constexpr (LEAN) slot0 = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.slots) + 4u * ((uint32_t)f * (uint32_t)P.slotsPerFrame + slotOff0));
  else slot0 = P.slots + (long long)f * P.slotsPerFrame + slotOff0;
  uint32_t* slot1 = slot0 + slotCap;
  int base0 = 0, base1 = 0;
  bool emit0 = true, emit1 = ew1 > 0;
  for (int pass = 0; pass < 2; pass++) {
    const int tlo = pass ? P.minTh : P.iniTh;
    // ---- stage 1: compass pre-test, NPX horizontally adjacent pixels per lane --------------------------------------
    // Lane item i = (row y, group g): pixels x = NPX*g .. NPX*g + NPX-1.  The centre-row bytes and the bytes of rows
    // y-3 / y+3 come from aligned LDS dwords and one funnel shift per 4-pixel window (the byte alignment `a` of the ROI
    // is uniform for the task, so it is a template constant of the loop body).
    int nq = 0;
    constexpr int GS = NPX == 16 ? 4 : 3, NS = NPX / 4;   // log2(NPX), 4-pixel windows per lane
    const int G = (W2 + NPX - 1) >> GS, nItems = G * eh;
    // item i+64 = (y + stepY, g + stepG) with one carry.  64 / G from a byte table for G = 1..4 (all there is with 16 pixels per lane
    // and tasks at most 64 pixels wide): the generic quotient is a 25-instruction reciprocal sequence per wave and pass
    const int stepY = NPX == 16 ? (int)((0x10152040u >> ((G - 1) << 3)) & 0xffu) : 64 / G, stepG = 64 - stepY * G;
    auto stage1 = [&](auto aTag) {
      constexpr int A = decltype(aTag)::value;
      // The test is CONSERVATIVE (a superset of "two adjacent compass points of one polarity", stage 2 decides
      // exactly) and runs on the packed bytes as they come out of LDS: a 16-bit lane holds pixels (2j, 2j+1) as
      // (low byte, high byte).  min/max of such lanes is exact in the high byte, so for the odd pixels
      //   bright: min(max(r0,r8), max(r4,r12)) > v+t     dark: max(min(r0,r8), min(r4,r12)) < v-t
      // are evaluated with v_pk_min/max_u16 and saturating add/sub against (v +- t) << 8; the junk low byte can only
      // turn an exact tie into a pass.  The even pixels take the same path after a packed shift left by 8 (exact).
      const u16x2 T2 = as_u16x2((unsigned)tlo * 0x01000100u);
      // fo / fe: bit 0 and bit 16 = odd pixels (1, 3) / even pixels (0, 2) of the 4-pixel windows (L = left, C = centre, ...)
      auto test4 = [&](uint32_t L4, uint32_t C4, uint32_t R4, uint32_t U4, uint32_t D4, unsigned& fe, unsigned& fo) {
        unsigned flag[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {   // h = 0: odd pixels (high bytes), h = 1: even pixels (shifted up)
          const u16x2 v2 = h ? as_u16x2(C4) << 8 : as_u16x2(C4);
          const u16x2 r0 = h ? as_u16x2(D4) << 8 : as_u16x2(D4), r4 = h ? as_u16x2(R4) << 8 : as_u16x2(R4);
          const u16x2 r8 = h ? as_u16x2(U4) << 8 : as_u16x2(U4), r12 = h ? as_u16x2(L4) << 8 : as_u16x2(L4);
          const u16x2 mx = __builtin_elementwise_min(__builtin_elementwise_max(r0, r8), __builtin_elementwise_max(r4, r12));
          const u16x2 mn = __builtin_elementwise_max(__builtin_elementwise_min(r0, r8), __builtin_elementwise_min(r4, r12));
          // brighter by more than t or darker by more than t: max(mx - v, v - mn) > t, all saturating (one instruction fewer than
          // comparing mx with v + t and mn with v - t separately; the junk low bytes still only turn a tie into a pass)
          const u16x2 dd = __builtin_elementwise_max(__builtin_elementwise_sub_sat(mx, v2), __builtin_elementwise_sub_sat(v2, mn));
          const unsigned e = as_u32(__builtin_elementwise_sub_sat(dd, T2));
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(flag[h]) : "v"(e), "v"(0x00010001u));   // 1 per passing 16-bit lane
        }
        fo = flag[0];
        fe = flag[1];
      };
      int y = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)G)), g = lane - m24(y, G);   // lane / G (see the ROI load)
      int ro = m24(y, TP) + (g << GS);               // byte offset of (row y, column NPX*g) in the tile
      const int roStep = m24(stepY, TP) + (stepG << GS), roCarry = TP - (G << GS);
      const uint8_t* t0 = tile - a;   // (A == a: written with the run-time value so that the four alignment variants share their row bases instead of each hoisting its own set out of the pass loop)
      for (int i0 = 0; i0 < nItems; i0 += 64) {
        unsigned m = 0;   // bit k = pixel NPX*g + k passes
        if (i0 + lane < nItems) {
          // aligned dwords of (row y+3, tile column NPX*g): byte windows of a row: left = bytes [A, A+NPX),
          // centre / up / down = [A+3, A+3+NPX), right = [A+6, A+6+NPX)
          constexpr int c0 = (A + 3) >> 2, cs = (A + 3) & 3, q0 = (A + 6) >> 2, qs = (A + 6) & 3;
          const uint32_t* cw = reinterpret_cast<const uint32_t*>(t0 + 3 * TP + ro);
          uint32_t w[NS + 3];
#pragma unroll
          for (int k = 0; k < NS + 3; k++) w[k] = cw[k];
          const uint32_t* uw = reinterpret_cast<const uint32_t*>(t0 + ro) + c0;               // row y-3 (+3 halo)
          const uint32_t* dw = reinterpret_cast<const uint32_t*>(t0 + 6 * TP + ro) + c0;      // row y+3
          uint32_t u[NS + 1], d[NS + 1];
#pragma unroll
          for (int k = 0; k < NS + 1; k++) { u[k] = uw[k]; d[k] = dw[k]; }
          unsigned E = 0, O = 0;
#pragma unroll
          for (int j = 0; j < NS; j++) {
            const uint32_t Lj = __builtin_amdgcn_alignbyte(w[j + 1], w[j], A);
            const uint32_t Cj = __builtin_amdgcn_alignbyte(w[c0 + j + 1], w[c0 + j], cs);
            const uint32_t Rj = __builtin_amdgcn_alignbyte(w[q0 + j + 1], w[q0 + j], qs);
            const uint32_t Uj = __builtin_amdgcn_alignbyte(u[j + 1], u[j], cs);
            const uint32_t Dj = __builtin_amdgcn_alignbyte(d[j + 1], d[j], cs);
            unsigned fe, fo;
            test4(Lj, Cj, Rj, Uj, Dj, fe, fo);
            E |= fe << (4 * j);   // bit 4j = pixel 4j, bit 16+4j = pixel 4j+2
            O |= fo << (4 * j);   // bit 4j = pixel 4j+1, bit 16+4j = pixel 4j+3
          }
          const unsigned T = E | (O << 1);
          const int rem = W2 - (g << GS);   // pixels of this group inside the emit width (>= 1)
          m = (T | (T >> 14)) & ((1u << min(rem, NPX)) - 1u);
        }
        // ordered compaction: inclusive wave scan of the per-lane counts (DPP, 6 adds)
        const int c = __popc(m);
        int incl = c;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);   // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);   // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
        const int pos = nq + incl - c;
        const unsigned e = (unsigned)((y << 8) | (g << GS));
        // Ordered emission of the lane's set bits, three vector instructions per bit position instead of five: the mask is
        // bit-reversed once, so that `rev + rev` shifts the next bit out as the carry -- v_add_co writes it straight into VCC, which
        // masks the store and the advance of the lane's write address; the entry counts up unconditionally.
        {
          unsigned rev = __builtin_bitreverse32(m) >> (16 - NPX), ent = e;   // bit k of m -> bit 31 - k ... (NPX = 8: bits 0..7 -> 31..24)
          rev <<= (16 - NPX);
          unsigned addr = (unsigned)(uintptr_t)(queue + pos);   // LDS byte address
          unsigned long long saved;
          asm volatile(
              "s_mov_b64 %3, exec\n\t"
              ".rept %4\n\t"
              "v_add_co_u32 %0, vcc, %0, %0\n\t"
              "s_and_b64 exec, %3, vcc\n\t"
              "ds_write_b16 %1, %2\n\t"
              "v_add_u32 %1, 2, %1\n\t"
              "s_mov_b64 exec, %3\n\t"
              "v_add_u32 %2, 1, %2\n\t"
              ".endr"
              : "+v"(rev), "+v"(addr), "+v"(ent), "=&s"(saved)
              : "n"(NPX)
              : "vcc", "scc", "memory");   // (s_and_b64 writes SCC)
        }
        nq += __builtin_amdgcn_readlane(incl, 63);
        if (i0 + 64 >= nItems) break;   // (a single cell is one iteration: its bookkeeping for the next would be ten wasted instructions)
        y += stepY;
        g += stepG;
        ro += roStep;
        if (g >= G) { g -= G; y++; ro += roCarry; }
      }
    };
    switch (a) {
      case 0: stage1(std::integral_constant<int, 0>{}); break;
      case 1: stage1(std::integral_constant<int, 1>{}); break;
      case 2: stage1(std::integral_constant<int, 2>{}); break;
      default: stage1(std::integral_constant<int, 3>{}); break;
    }
    wave_lds_fence();
    if constexpr (ABL == 4) { if (pass == 0) stamp[3] = __builtin_amdgcn_s_memtime(); }
    if constexpr (ABL == 2) {
      unsigned acc = 0;
      for (int i = lane; i < nq; i += 64) acc ^= queue[i];
      if (lane == 0) cnt[0] = (acc == 0xffffffffu) ? 1u : 0u;   // (never 1: queue entries are 16-bit)
      return;
    }
    // ---- stage 2: score of every stage-1 survivor; survivors of the arc test stay in the queue --------------------
    // S = max(max_arc min(v - ring), max_arc min(ring - v)) decides both "is a corner at tlo" (S > tlo) and the
    // OpenCV score (S - 1).  Both polarities are evaluated at once: each register holds (v - r, r - v) as two signed
    // 16-bit lanes and the arc minima / their maximum run on v_pk_min_i16 / v_pk_max_i16.
    int nq2 = 0;
    for (int i0 = 0; i0 < nq; i0 += 64) {
      const int i = i0 + lane;
      bool ok = false;
      unsigned e = 0;
      if (i < nq) {
        e = queue[i];
        const int y = e >> 8, x = e & 0xff;
        const uint8_t* c = tile + m24(y + 3, TP) + (x + 3);
        const unsigned vv = c[0];
        // xr[k] = (B + v - r_k, B + r_k - v) in the two 16-bit lanes by ONE packed multiply-add per ring pixel:
        // r_k (low half, used by both lanes) * (-1, +1) + (B + v, B - v).  The bias B = 2048 makes every value the bit pattern of
        // a NORMAL, positive half float (0x0701 .. 0x08ff), and positive floats order like their bit patterns: the minima and
        // maxima below are integer minima and maxima taken by gfx950's THREE-input packed half-float instructions
        // (v_pk_minimum3_f16 / v_pk_maximum3_f16; there is no packed integer min3).
        constexpr unsigned B = 2048u;
        unsigned xr[16];
        const unsigned VV = (B + vv) | ((B - vv) << 16);
#define RING(k, off) asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(xr[k]) : "v"((unsigned)c[off]), "v"(0x0001ffffu), "v"(VV));
        RING(0, 3 * TP);       RING(1, 3 * TP + 1);   RING(2, 2 * TP + 2);    RING(3, TP + 3);
        RING(4, 3);            RING(5, -TP + 3);      RING(6, -2 * TP + 2);   RING(7, -3 * TP + 1);
        RING(8, -3 * TP);      RING(9, -3 * TP - 1);  RING(10, -2 * TP - 2);  RING(11, -TP - 3);
        RING(12, -3);          RING(13, TP - 3);      RING(14, 2 * TP - 2);   RING(15, 3 * TP - 1);
#undef RING
        // min over every arc of 9 consecutive ring pixels, both polarities per instruction: t[k] = min of pixels k .. k+2,
        // arc[k] = min(t[k], t[k+3], t[k+6]) (indices mod 16): 16 + 16 three-input minima, then 8 three-input maxima
        unsigned t[16], arc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = pk_min3_h(xr[k], xr[(k + 1) & 15], xr[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) arc[k] = pk_min3_h(t[k], t[(k + 3) & 15], t[(k + 6) & 15]);
        const unsigned m0 = pk_max3_h(arc[0], arc[1], arc[2]), m1 = pk_max3_h(arc[3], arc[4], arc[5]), m2 = pk_max3_h(arc[6], arc[7], arc[8]);
        const unsigned m3 = pk_max3_h(arc[9], arc[10], arc[11]), m4 = pk_max3_h(arc[12], arc[13], arc[14]);
        const unsigned best = pk_max3_h(pk_max3_h(m0, m1, m2), pk_max3_h(m3, m4, arc[15]), arc[15]);
        const int S = (int)max(best & 0xffffu, best >> 16) - (int)B;
        ok = S > tlo;
        if (ok) sc[m24(y + 1, SP) + (x + 1)] = (uint8_t)(S - 1);  // tlo <= S-1 <= 254
      }
      // the ballot consumes every lane's queue read, so the in-place writes below cannot overtake them
      const unsigned long long mk = __ballot(ok);
      if (ok) queue[nq2 + __popcll(mk & below)] = (uint16_t)e;
      nq2 += __popcll(mk);
    }
    wave_lds_fence();
    if constexpr (ABL == 4) { if (pass == 0) stamp[4] = __builtin_amdgcn_s_memtime(); }
    if constexpr (ABL == 3) {
      unsigned acc = 0;
      for (int i = lane; i < nq2; i += 64) acc ^= queue[i] ^ sc[i];
      if (lane == 0) cnt[0] = (acc == 0xffffffffu) ? 1u : 0u;
      return;
    }
    // ---- stage 3: NMS inside each cell's emit region and ordered emission (every survivor has score >= tlo) --------
    for (int i0 = 0; i0 < nq2; i0 += 64) {
      const int i = i0 + lane;
      int keep = 0, y = 0, x = 0;
      if (i < nq2) {
        const unsigned e = queue[i];
        y = e >> 8;
        x = e & 0xff;
        const uint8_t* q = sc + m24(y + 1, SP) + (x + 1);
        const int sv = q[0];
        // neighbours on the other side of the boundary between the two cells count 0 (each cell is its own cv::FAST ROI)
        const int lm = (ew1 && x == ew0) ? 0 : 0xff, rm = (ew1 && x == ew0 - 1) ? 0 : 0xff;
        // strictly greater than all eight neighbours = greater than their maximum (three-input maxima: 4 instructions for 8 values)
        const int nmax = max(max(max(q[-1] & lm, q[-SP - 1] & lm), q[SP - 1] & lm), max(max(max(q[1] & rm, q[-SP + 1] & rm), q[SP + 1] & rm), max((int)q[-SP], (int)q[SP])));
        if (sv > nmax) keep = sv;   // (sv > 0 follows: nmax >= 0)
      }
      const bool in1 = x >= ew0;
      const uint32_t rec = (uint32_t)(ex0 + x) | ((uint32_t)(ey0 + y) << 12) | ((uint32_t)keep << 24);
      if (emit0) {
        const unsigned long long mk = __ballot(keep > 0 && !in1);
        if (keep > 0 && !in1) slot0[base0 + __popcll(mk & below)] = rec;
        base0 += __popcll(mk);
      }
      if (emit1) {
        const unsigned long long mk = __ballot(keep > 0 && in1);
        if (keep > 0 && in1) slot1[base1 + __popcll(mk & below)] = rec;
        base1 += __popcll(mk);
      }
    }
    if (pass == 1 || P.minTh == P.iniTh) break;
    if constexpr (ABL == 5) break;   // measurement: what the second pass costs
    // ORBextractor.cc:850-856: a cell whose first cv::FAST call returned nothing is detected again at minThFAST
    emit0 = base0 == 0;
    emit1 = ew1 > 0 && base1 == 0;
    if (!emit0 && !emit1) break;
    wave_lds_fence();   // the queue is rebuilt by the second pass
  }
  if (lane == 0) {
    cnt[0] = (uint32_t)base0;
    if (ew1) cnt[1] = (uint32_t)base1;
  }
  if constexpr (ABL == 4) {
    stamp[5] = __builtin_amdgcn_s_memtime();
    // one record per wave, plain stores (same-address atomics from 200 000 waves back up the memory pipeline and inflate the
    // very latencies being measured)
    if (lane == 0 && g_fastStampBuf) {
      // indexed by the TASK (unique across the LDS-class launches of a batch), not by the block: blocks are dealt XCD-consecutively,
      // so blockIdx.x runs up to 8 * ceil(nt / 8) - 1 and the tail of one class used to land on the first records of the next
      uint32_t* rec = g_fastStampBuf + 8ull * ((unsigned long long)blockIdx.y * (P.ntasks + 64) + tix);
      for (int k = 0; k < 5; k++) rec[k] = (uint32_t)(stamp[k + 1] - stamp[k]);
      rec[5] = 1u;
    }
  }
}

#ifdef ORBFE_EXPERIMENTS
// measurement only (ORBFE_FAST_ABLATE=4): per-wave phase cycles of the LAST launch, summed on the host: entry -> geometry
// known, -> ROI in LDS, -> pre-test done, -> scores done, -> end; [5] = waves.  ONE buffer per process (g_fastStampBuf is a
// device symbol): meant for a single extractor on a single device, as tools/fast_phases.py uses it.
static uint32_t* s_stampBuf = nullptr;
static size_t s_stampWaves = 0;
int fast_stamps(unsigned long long out[8], int reset) {
  for (int k = 0; k < 8; k++) out[k] = 0;
  if (!s_stampBuf || !s_stampWaves) return 0;
  std::vector<uint32_t> h(8 * s_stampWaves);
  if (hipMemcpy(h.data(), s_stampBuf, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (size_t w = 0; w < s_stampWaves; w++)
    if (h[8 * w + 5]) {
      for (int k = 0; k < 5; k++) out[k] += h[8 * w + k];
      out[5]++;
    }
  if (reset && hipMemset(s_stampBuf, 0, h.size() * 4) != hipSuccess) return 1;
  return 0;
}
static void stamp_buffer_for(size_t waves) {
  if (waves > s_stampWaves) {
    if (s_stampBuf) (void)hipFree(s_stampBuf);
    s_stampBuf = nullptr;
    if (hipMalloc((void**)&s_stampBuf, waves * 32) != hipSuccess) { s_stampWaves = 0; return; }
    s_stampWaves = waves;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fastStampBuf), &s_stampBuf, sizeof s_stampBuf);
  }
  (void)hipMemset(s_stampBuf, 0, s_stampWaves * 32);
}
#else
int fast_stamps(unsigned long long out[8], int) {   // (a default build has no stamped instantiation)
  for (int k = 0; k < 8; k++) out[k] = 0;
  return 1;
}
#endif

static size_t fast_lds_bytes_level(const LevelGeom& L) {
  const size_t TP = (size_t)((L.fastW + 6 + 3 + 15) & ~15);
  const size_t scOff = (TP * (L.hCell + 6) + 15) & ~(size_t)15;
  const size_t b = ((scOff + (size_t)(L.fastW + 2) * (L.hCell + 2) + 15) & ~(size_t)15) +
                   2 * (size_t)L.fastW * L.hCell + 64;  // tile + score tile + u16 queue + slack for the group over-read
  return (b + 15) & ~(size_t)15;
}
// FastTask::geo of a level (the LEAN prologue's LDS carve and staging constants); 0: not representable -> the level's launches
// take the generic prologue
uint32_t fast_task_geo(int fastW, int hCell) {
  const uint32_t TP = (uint32_t)((fastW + 6 + 3 + 15) & ~15), ppr = TP >> 4;
  if (ppr == 0 || ppr > 7) return 0u;
  const uint32_t rpi = 64u / ppr;
  const uint32_t scOff = (TP * (uint32_t)(hCell + 6) + 15u) & ~15u;
  const uint32_t qOff = (scOff + (uint32_t)(fastW + 2) * (uint32_t)(hCell + 2) + 15u) & ~15u;
  if (rpi > 31u || (scOff >> 4) > 0x3ffu || (qOff >> 4) > 0x7ffu) return 0u;
  return ppr | (rpi << 3) | ((scOff >> 4) << 8) | ((qOff >> 4) << 18);
}

void launch_fast(const PyramidParams& P, int nframes, hipStream_t st) {
  // 16 pixels per lane in the pre-test: one wave iteration covers a 31 x 31 cell (8 pixels per lane take two and are
  // equal in time, 9.9 us per 1080p frame, with more instructions)
  const FastArgs FA = fast_args(P);
  const int level0 = 0, level1 = P.nlevels;
  bool pairs = false;
  for (int l = 0; l < P.nlevels; l++) pairs = pairs || P.lv[l].fastW != P.lv[l].wCell;
  // LDS CLASSES (round 4).  LDS is handed out in 1 KB granules and a CU has 160 of them: a one-wave workgroup that asks for up
  // to 5 120 bytes leaves room for 32 waves per CU (8 per SIMD), one that asks for 5 121 .. 6 144 for 26 (6.5 per SIMD) --
  // measured with tools/ubench/dispatch_rate.hip: 7.35 / 5.85 / 4.96 resident waves per SIMD at 4 608 / 5 632 / 6 656 bytes.
  // At 1080p the seven lower levels have 31 x 30..32 cells (4.4 - 4.7 KB per wave) and only the top level's 32 x 34 cells need
  // 5.2 KB: sizing every wave for those 2 % of the cells cost the other 98 % a fifth of their residency, and a FAST wave spends
  // 40 % of its life waiting for its ROI -- residency is what hides that.  So consecutive levels that fit 5 KB go out as one
  // launch with 5 KB of LDS, the others as launches of their own.
  constexpr size_t kLdsFull = 5120;
  // ORBFE_FAST_LEAN=0 forces the generic prologue (the one odd strides and unusual cell sizes take) on every launch: tests/test_gpu_parity.py
  static const int leanEnv = [] { const char* e = getenv("ORBFE_FAST_LEAN"); return e ? atoi(e) : 1; }();
#ifdef ORBFE_EXPERIMENTS
  static const int ablate = [] { const char* e = ORBFE_EXP_ENV("ORBFE_FAST_ABLATE"); return e ? atoi(e) : 0; }();   // measurement only
#endif
  int l = level0;
  while (l < level1) {
    int e = l + 1;
    size_t need = fast_lds_bytes_level(P.lv[l]);
    if (nframes > 2) {   // (a one- or two-frame call is latency-bound: one launch less is worth more than residency there)
      const bool small = need <= kLdsFull;
      while (e < level1 && (fast_lds_bytes_level(P.lv[e]) <= kLdsFull) == small) { need = std::max(need, fast_lds_bytes_level(P.lv[e])); e++; }
    } else {
      while (e < level1) { need = std::max(need, fast_lds_bytes_level(P.lv[e])); e++; }
    }
    const int t0 = P.taskStart[l], nt = P.taskStart[e] - t0;
    l = e;
    if (nt <= 0) continue;
    const dim3 grid(8 * ((nt + 7) / 8), nframes);
    // the LEAN prologue's preconditions, checked once per launch instead of by every wave (level 0: the ROI offset
    // (ey0 - 3) * stride0 + ex0 - 3 of the last cell row must fit 32 bits)
    const unsigned long long frames = (unsigned long long)P.frameBase + (unsigned long long)nframes;
    const bool lean = leanEnv && P.fastLean && P.zeros && (P.stride0 & 3) == 0 && P.stride0 > 0 && P.stride0 < (1ll << 31) &&
                      frames * (unsigned long long)P.slabBytes < (1ull << 32) && frames * (unsigned long long)P.slotsPerFrame * 4ull < (1ull << 32) &&
                      frames * (unsigned long long)P.ncells * 4ull < (1ull << 32) &&
                      (unsigned long long)P.lv[0].h * (unsigned long long)P.stride0 < (1ull << 32);
#ifdef ORBFE_EXPERIMENTS
    if (ablate && !pairs) {
      if (ablate == 4 && t0 == 0) stamp_buffer_for((size_t)(P.ntasks + 64) * nframes);
#define ORBFE_FAST_ABL(A)                                                                                              \
  do {                                                                                                                 \
    if (lean) hipLaunchKernelGGL((k_fast_tasks<16, false, A, true>), grid, dim3(64), need, st, FA, t0, nt);            \
    else hipLaunchKernelGGL((k_fast_tasks<16, false, A, false>), grid, dim3(64), need, st, FA, t0, nt);                \
  } while (0)
      switch (ablate) {
        case 1: ORBFE_FAST_ABL(1); break;
        case 2: ORBFE_FAST_ABL(2); break;
        case 3: ORBFE_FAST_ABL(3); break;
        case 4: ORBFE_FAST_ABL(4); break;
        default: ORBFE_FAST_ABL(5); break;
      }
#undef ORBFE_FAST_ABL
      continue;
    }
#endif
    if (pairs) hipLaunchKernelGGL((k_fast_tasks<16, true, 0, false>), grid, dim3(64), need, st, FA, t0, nt);
    else if (lean) hipLaunchKernelGGL((k_fast_tasks<16, false, 0, true>), grid, dim3(64), need, st, FA, t0, nt);
    else hipLaunchKernelGGL((k_fast_tasks<16, false, 0, false>), grid, dim3(64), need, st, FA, t0, nt);
  }
}

}  // namespace orbfe
