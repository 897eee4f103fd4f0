// orbfe_internal.h -- structures shared by the host engine and the HIP kernels.
#pragma once
#include <cstddef>
#include <cstdint>

namespace orbfe {

constexpr int kMaxLevels = 12;
constexpr int kEdge = 19;        // EDGE_THRESHOLD  (reference src/ORBextractor.cc:79)
constexpr int kBorder = 16;      // minBorderX/Y = EDGE_THRESHOLD-3 (ORBextractor.cc:807-808)
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE (ORBextractor.cc:76)
constexpr int kBlurRad = 18;     // largest |cvRound| of a rotated rBRIEF offset
constexpr int kRawRad = kBlurRad + 3;  // raw pixels needed to blur the 37x37 neighbourhood

// Geometry of one pyramid level; identical for every frame of a batch.
struct LevelGeom {
  int w, h;                // level size in pixels
  int pitch;               // row pitch in bytes inside the pyramid slab (levels >= 1)
  long long off;           // byte offset of the level inside one frame's pyramid slab (levels >= 1)
  int nCols, nRows;        // FAST cell grid (ORBextractor.cc:820-821)
  int wCell, hCell;        // cell size (ORBextractor.cc:822-823)
  int cellBase;            // index of this level's first cell in the per-frame cell arrays
  int slotCap;             // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2) (strict-local-max bound)
  long long slotBase;      // first slot (u32 units) of this level in the per-frame slot array
  // bilinear tables for producing THIS level from level-1 (device pointers; unused for level 0)
  const int* xofs;         // [w]   source column
  const short* xalpha;     // [2*w] 11-bit weights (a0,a1)
  const int* yofs;         // [h]   source row (unclamped)
  const short* ybeta;      // [2*h] 11-bit weights (b0,b1)
  const unsigned* yofc;    // [h]   the two source rows of a destination row, clamped to the source: row0 | row1 << 16 (k_resize_fixed)
  int rzPitch, rzRows;     // LDS pitch / rows of the largest 64x64-tile source footprint (k_resize)
  int fastW;               // widest emit region of a FAST task on this level (2 * wCell when cells are paired, else wCell)
};

// Per-cell geometry, precomputed on the host so a cell's wave needs one 16-byte load instead of a scalar
// search over the level table.
struct CellInfo {
  uint16_t ex0, ey0;     // first emit pixel (level coordinates)
  int8_t ew, eh;         // emit size (<= 0: the cell emits nothing)
  uint8_t level, pad;
  uint32_t slotOff;      // first slot of the cell inside a frame's slot array (u32 units)
  uint32_t local;        // cell index inside its level
};

// One wave of k_fast_tasks: one FAST cell, or two horizontally adjacent cells of the same cell row whose union is at
// most 64 pixels wide (the second cell starts where the first ends, so the union's ROI is one contiguous tile).
struct FastTask {
  uint16_t ex0, ey0;     // first emit pixel of cell 0 (level coordinates)
  uint8_t ew0, ew1;      // emit widths of cell 0 / cell 1 (ew1 = 0: single cell; ew0 = 0: the cell emits nothing)
  uint8_t eh, level;
  uint32_t cell0;        // index of cell 0 in the per-frame cell arrays (cell 1 = cell0 + 1)
  uint32_t slotOff0;     // first slot of cell 0 (cell 1: + the level's slotCap)
  // the level's geometry as far as the FAST wave needs it, so that ONE 32-byte scalar load gives a wave everything it needs
  // to issue its ROI loads (the level table in the kernel arguments cost a second, dependent scalar round trip per wave):
  uint32_t roiOff;       // levels >= 1: byte offset of ROI pixel (ex0 - 3, ey0 - 3) inside one frame's pyramid slab
  uint32_t pitch;        // levels >= 1: row pitch of the level in the slab (level 0: the caller's stride, PyramidParams::stride0)
  uint8_t fastW, hCell;  // LevelGeom::fastW / hCell of the level (tile pitch and LDS carve)
  uint16_t slotCap;      // LevelGeom::slotCap
  uint32_t geo;          // the LEAN prologue's LDS carve and staging constants of the level (fast_task_geo(); 0: generic prologue only)
};
static_assert(sizeof(FastTask) == 32, "one s_load_dwordx8 per FAST wave");

struct PyramidParams {
  LevelGeom lv[kMaxLevels];
  int nlevels;
  int ncells;                       // cells per frame (all levels)
  long long slabBytes;              // pyramid slab bytes per frame (levels >= 1)
  long long slotsPerFrame;          // u32 slots per frame
  long long candCap;                // candidate capacity per frame (u32 entries)
  const uint8_t* const* frame0;     // [nframes] level-0 pointers (device memory); nullptr: use frameInline
  const uint8_t* frameInline[2];    // level-0 pointers of a one- or two-frame call, carried in the kernel arguments
                                    // (saves the table's H2D copy in front of a latency-bound call)
  long long stride0;                // level-0 row stride in bytes
  uint8_t* slab;                    // [nframes][slabBytes]
  uint32_t* cellCount;              // [nframes][ncells]
  uint32_t* slots;                  // [nframes][slotsPerFrame]
  uint32_t* cand;                   // [nframes][candCap]  packed x | y<<12 | score<<24 (level coords)
  uint32_t* levelStart;             // [nframes][kMaxLevels+1]  first candidate of every level (packed list); level-local lists: the levels' COUNTS
  const CellInfo* cells;            // [ncells]
  const FastTask* tasks;            // [ntasks] work items of k_fast_tasks, level-major
  int ntasks;
  int taskStart[kMaxLevels + 1];    // first task of each level (host side of launch_fast: LDS classes)
  const uint8_t* zeros;             // 256 zero bytes in device memory: source of the LDS-DMA that clears a FAST wave's score tile
  int iniTh, minTh;
  int frameBase;                    // first frame of this launch (sub-batch pipelining)
  int gaussVariant;                 // ORBFE_GAUSS_ED / ORBFE_GAUSS_ROUNDED: which GaussianBlur k_describe reproduces
  int fastLean;                     // every task carries FastTask::geo: k_fast_tasks may take its LEAN prologue (launch_fast decides per launch)
};

// Small batches build the pyramid in ONE launch (k_pyramid_cone): a block owns a tile of the top level and computes
// the part of every level below it that the tile depends on.  Host-computed per tile column / tile row.
struct ConeRange { int lo, hi; };   // inclusive
struct ConeParams {
  int base, top;                    // the cone starts from level `base` (already in memory) and produces base+1 .. top
  int tile, tilesX, tilesY;         // tile edge on level `top`, tiles per row / column
  int offB, offH, offC, coefLen;    // LDS byte offsets: second region buffer, row-pass intermediate, coefficient slices
  int ldsBytes;
  const ConeRange* regX;            // [tilesX][kMaxLevels] columns of level l a tile column needs
  const ConeRange* regY;            // [tilesY][kMaxLevels]
};

// Measurement switches of the lab notebook (DESIGN_NOTES.md) exist only in a build with -DORBFE_EXPERIMENTS (make EXPERIMENTS=1);
// a default build reads none of them: every environment variable the shipped library looks at is listed in DESIGN.md s4 and
// enumerated by tests/test_switches.py.
#ifdef ORBFE_EXPERIMENTS
#define ORBFE_EXP_ENV(name) getenv(name)
#else
#define ORBFE_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif
#ifdef __HIPCC__
// Level-0 pointer of frame f, in SCALAR registers (f is uniform for a block): a lane-held base would turn every load
// of the frame into 64-bit vector address arithmetic.
__device__ __forceinline__ const uint8_t* level0_of(const PyramidParams& P, int f) {
  const uint8_t* p = P.frame0 ? P.frame0[f] : P.frameInline[f & 1];
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
}
#endif

// One selected keypoint handed to the orientation + descriptor kernel.
struct SelKp {
  uint32_t xy;     // x | y << 16 (level coordinates)
  uint32_t lf;     // level | frame << 8
};

// ---- view of one frame of an extractor's last collected batch, for building a device-resident orbfe_frame from it
// (orbfe_frame.hip): the result arena's slots as the kernels left them -- in device memory, or in page-locked host
// memory (device-readable) when the latency route wrote the results there.
struct ExtractView {
  const SelKp* sel;            // first slot of the frame
  const float* angle;
  const uint8_t* desc;
  int selOff[kMaxLevels + 1];  // first slot of every level inside the frame's slot region
  int count[kMaxLevels];       // keypoints per level (host copy of selCount)
  float sf[kMaxLevels];        // mvScaleFactor
  int nlevels, n, device;
  void* stream;                // hipStream_t the arena is ordered on
};

// ---- GPU-resident SearchForInitialization over consecutive frames (orbfe_sfi.hip) ------------------------
struct SfiParams {
  // result arena of the batch (device): level-0 slots are the first region of every frame
  const SelKp* sel;
  const float* angle;
  const uint8_t* desc;
  const uint32_t* selCount;     // [nframes][kMaxLevels]
  int selPerFrame, n0cap, frameBase;
  // predecessor of the batch's first frame (level-0 data of the previous batch's last frame)
  const SelKp* carrySel;
  const float* carryAngle;
  const uint8_t* carryDesc;
  const uint32_t* carryCount;   // device word; > n0cap (0xffffffff) = no predecessor
  float minX, minY, invW, invH; // Frame grid (Frame.cc:98-99)
  float window, nnratio;
  int checkOri;
  // scratch
  uint16_t* order;              // [nframes][n0cap] level-0 keypoints of a frame in GetFeaturesInArea order
  int* orderCount;              // [nframes]
  uint32_t* pool;               // [nframes][n0cap][n0cap] candidate entries  index | distance << 16
  uint32_t* pcount;             // [nframes][n0cap]
  // outputs (inside the result arena)
  int32_t* matches12;           // [nframes][n0cap] vnMatches12 restricted to level-0 queries
  int32_t* nmatches;            // [nframes]
};

// ---- GPU quadtree (orbfe_quadtree.hip) -------------------------------------------------------------
constexpr int kQtNodeCap = 2048;  // live nodes <= N + 3  =>  N <= 2044 per level

struct QtParams {
  const uint32_t* cand;        // [nframes][candCap] x | y<<12 | score<<24 (level coordinates)
  const uint32_t* levelStart;  // [nframes][kMaxLevels+1]
  long long candCap;           // < 2^24 (the final pick packs a candidate index into 24 bits)
  int nlevels, frameBase;
  int levW[kMaxLevels], levH[kMaxLevels], nfeat[kMaxLevels];
  int selOff[kMaxLevels];      // first slot of the level inside a frame's selection region
  int selPerFrame;
  uint16_t* own;               // [nframes][candCap] node id * 4 + quadrant per candidate
  SelKp* sel;                  // [nframes][selPerFrame]
  uint32_t* selCount;          // [nframes][kMaxLevels]
  // optional second copy of the two outputs in page-locked host memory (latency-bound calls: the result needs no
  // device-to-host copy command after the last kernel); nullptr = none
  SelKp* selHost;
  uint32_t* selCountHost;
  int jump;                    // != 0: the first passes of a dense level (every node divides) in one sweep (orbfe_quadtree.hip)
  // != 0: level-local candidate lists (k_compact_local: one- and two-frame calls) -- level l's list starts at candBase[l] inside a
  // frame's candidate array and levelStart[f][l] holds its LENGTH; 0: one packed list per frame, levelStart = prefix offsets (k_compact)
  int levelLocal;
  long long candBase[kMaxLevels];
};


}  // namespace orbfe
