// orb_shim.hpp -- header-only C++ host side above the C ABI (include/orbfe.h).
//
// Generic (duck-typed) so it compiles with or without OpenCV: the reference's own types
// (cv::KeyPoint, cv::Mat, ORB_SLAM2::Frame, ORB_SLAM2::MapPoint) are template parameters and are
// only touched through the member names the reference uses.  include/orbfe/ORBextractor.h binds the
// extractor part to the exact ORB_SLAM2::ORBextractor signature when OpenCV headers are present;
// INTEGRATION.md shows the three-line bodies that replace the hot ORBmatcher functions.
//
// Error behaviour mirrors the reference: no exceptions for data conditions (empty image => outputs
// untouched, matchers return the match count); a failing GPU call throws std::runtime_error with
// orbfe_last_error() because the reference has no channel to report it and continuing would
// silently corrupt tracking.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../orbfe.h"

namespace orbfe {

inline void check(int rc) {
  if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error());
}

namespace detail {
// ORBFE_DEVICE (default 0): the GPU the cv-typed facades use -- extractor AND matcher (include/orbfe/ORBextractor.h,
// ORBmatcher.h), so that an integrated build on GPU != 0 extracts and searches on the same device.
inline int defaultDevice() {
  static const int d = [] { const char* e = std::getenv("ORBFE_DEVICE"); return e ? std::atoi(e) : 0; }();
  return d;
}

// 128-bit content fingerprint (two independently seeded multiply-xorshift lanes over 8-byte words, four interleaved
// chains each for throughput: ~10 bytes per cycle, a 2 000-keypoint frame's 120 KB in 3-4 us).  Used as IDENTITY of a
// frame's searchable content -- not cryptographic, but 2^-128-ish against accidental equality of distinct frames.
struct Hash128 {
  uint64_t a = 0x243F6A8885A308D3ull, b = 0x13198A2E03707344ull;
  bool operator==(const Hash128& o) const { return a == o.a && b == o.b; }
  bool operator!=(const Hash128& o) const { return !(*this == o); }
};
inline uint64_t mix64(uint64_t h, uint64_t v, uint64_t k) {
  h = (h ^ v) * k;
  return h ^ (h >> 29);
}
inline void hashBytes(Hash128& H, const void* p, size_t bytes) {
  const uint64_t K1 = 0x9E3779B97F4A7C15ull, K2 = 0xC2B2AE3D27D4EB4Full;
  const unsigned char* c = static_cast<const unsigned char*>(p);
  uint64_t a[4] = {H.a, H.a ^ K2, H.a + K1, ~H.a}, b[4] = {H.b, H.b ^ K1, H.b + K2, ~H.b};
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    std::memcpy(w, c + i, 32);
    for (int l = 0; l < 4; l++) { a[l] = mix64(a[l], w[l], K1); b[l] = mix64(b[l], w[l] + (uint64_t)l, K2); }
  }
  uint64_t tail[4] = {0, 0, 0, 0};
  if (i < bytes) std::memcpy(tail, c + i, bytes - i);
  for (int l = 0; l < 4; l++) { a[l] = mix64(a[l], tail[l] ^ bytes, K1); b[l] = mix64(b[l], tail[l] + bytes, K2); }
  H.a = mix64(mix64(mix64(a[0], a[1], K2), a[2], K2), a[3], K2);
  H.b = mix64(mix64(mix64(b[0], b[1], K1), b[2], K1), b[3], K1);
}
// the descriptor rows of a CV_8U n x 32 matrix (any step)
template <class MatT>
inline void hashDescriptorRows(Hash128& H, const MatT& m, int n) {
  if (n <= 0) return;
  if ((size_t)m.step == 32) { hashBytes(H, m.data, (size_t)n * 32); return; }
  for (int i = 0; i < n; i++) hashBytes(H, m.data + (size_t)i * m.step, 32);
}
}  // namespace detail

// ORBextractor (reference include/ORBextractor.h:155-373) minus the cv:: types.
//
// Besides extracting, an Extractor remembers WHAT it extracted last (count + fingerprint of the keypoints and descriptor
// rows it returned) while those results still sit in its device arena: the first search of the Frame built from them
// (Frame.cc:100-111: ExtractORB, UndistortKeyPoints, AssignFeaturesToGrid -- no other use of the extractor in between)
// then takes keypoints and descriptors where the kernels left them (MatcherContext::resident ->
// orbfe_frame_create_from_extract) instead of uploading mvKeysUn / mDescriptors again.
class Extractor {
 public:
  Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0) : device_(device) {
    check(orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_));
    const int n = orbfe_extractor_levels(h_);
    sf_.resize(n); isf_.resize(n); s2_.resize(n); is2_.resize(n);
    check(orbfe_extractor_scale_tables(h_, sf_.data(), isf_.data(), s2_.data(), is2_.data()));
    cap_ = orbfe_extractor_max_keypoints(h_);
    std::lock_guard<std::mutex> g(registryMutex());
    registry().push_back(this);
  }
  ~Extractor() {
    {
      std::lock_guard<std::mutex> g(registryMutex());
      auto& r = registry();
      r.erase(std::remove(r.begin(), r.end(), this), r.end());
    }
    std::lock_guard<std::mutex> g(mu_);
    orbfe_extractor_destroy(h_);
  }
  Extractor(const Extractor&) = delete;
  Extractor& operator=(const Extractor&) = delete;

  // Which cv::GaussianBlur the descriptors are sampled from (ORBextractor.cc:949-950): ORBFE_GAUSS_ED = OpenCV >= 4.1.1 (default),
  // ORBFE_GAUSS_ROUNDED = OpenCV 4.0.0 - 4.1.0.  The cv-typed facade calls this with the variant of the OpenCV it is compiled against.
  void SetBlurVariant(int variant) {
    std::lock_guard<std::mutex> g(mu_);
    check(orbfe_extractor_set_blur_variant(h_, variant));
  }

  // operator() core: KeyPointT must have cv::KeyPoint's 28-byte layout.
  template <class KeyPointT>
  void extract(const uint8_t* gray, int rows, int cols, size_t step, std::vector<KeyPointT>& keypoints,
               std::vector<uint8_t>& descriptors) {
    static_assert(sizeof(KeyPointT) == sizeof(OrbfeKeyPoint), "KeyPointT must match cv::KeyPoint's layout");
    if (!gray || rows == 0 || cols == 0) return;  // reference: silent return, outputs untouched
    std::lock_guard<std::mutex> g(mu_);
    lastValid_ = false;
    cap_ = std::max(cap_, orbfe_extractor_max_keypoints_for_size(h_, rows, cols));   // strips wider than 4.5 : 1
    kp_.resize(cap_);
    desc_.resize((size_t)cap_ * 32);
    int n = 0;
    check(orbfe_extract(h_, gray, rows, cols, step, kp_.data(), desc_.data(), cap_, &n));
    keypoints.clear();
    keypoints.resize(n);
    if (n) std::memcpy(static_cast<void*>(keypoints.data()), kp_.data(), (size_t)n * sizeof(OrbfeKeyPoint));
    descriptors.assign(desc_.begin(), desc_.begin() + (size_t)n * 32);
    if (trackLast_ && n > 0) {
      lastN_ = n;
      lastKeys_ = detail::Hash128();
      lastDesc_ = detail::Hash128();
      detail::hashBytes(lastKeys_, kp_.data(), (size_t)n * sizeof(OrbfeKeyPoint));
      detail::hashBytes(lastDesc_, desc_.data(), (size_t)n * 32);
      lastValid_ = true;
    }
  }

  int GetLevels() const { return (int)sf_.size(); }
  float GetScaleFactor() const { return orbfe_extractor_scale_factor(h_); }
  std::vector<float> GetScaleFactors() const { return sf_; }
  std::vector<float> GetInverseScaleFactors() const { return isf_; }
  std::vector<float> GetScaleSigmaSquares() const { return s2_; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return is2_; }
  orbfe_extractor* handle() const { return h_; }
  int capacity() const { return cap_; }
  int device() const { return device_; }
  // false: do not fingerprint the outputs (a caller that never searches Frames through a MatcherContext saves the 3-4 us)
  void setTrackLastOutput(bool on) { std::lock_guard<std::mutex> g(mu_); trackLast_ = on; lastValid_ = false; }

  // A resident frame straight from the arena of whichever live Extractor on `device` produced exactly these outputs last
  // (n keypoints with fingerprint `keys` of the cv::KeyPoint records as returned, `desc` of the rows); xy_un = the
  // undistorted coordinates [2n] when they differ from the extracted ones, else nullptr.  nullptr if there is none.
  static orbfe_frame* residentFromLastExtract(int device, int n, const detail::Hash128& keys, const detail::Hash128& desc,
                                              const float bounds[4], const float* xy_un) {
    std::lock_guard<std::mutex> g(registryMutex());
    for (Extractor* e : registry()) {
      if (e->device_ != device) continue;
      std::lock_guard<std::mutex> ge(e->mu_);
      if (!e->lastValid_ || e->lastN_ != n || e->lastKeys_ != keys || e->lastDesc_ != desc) continue;
      orbfe_frame* f = nullptr;
      if (orbfe_frame_create_from_extract(e->h_, 0, bounds, xy_un, &f) == ORBFE_OK && f && orbfe_frame_size(f) == n) return f;
      if (f) orbfe_frame_destroy(f);
    }
    return nullptr;
  }

 private:
  static std::mutex& registryMutex() { static std::mutex m; return m; }
  static std::vector<Extractor*>& registry() { static std::vector<Extractor*> r; return r; }
  orbfe_extractor* h_ = nullptr;
  int cap_ = 0, device_ = 0;
  std::vector<float> sf_, isf_, s2_, is2_;
  std::vector<OrbfeKeyPoint> kp_;
  std::vector<uint8_t> desc_;
  std::mutex mu_;                       // extract() vs. a search thread taking the arena's content
  bool trackLast_ = true, lastValid_ = false;
  int lastN_ = 0;
  detail::Hash128 lastKeys_, lastDesc_;
};

// One GPU matcher context per thread that runs searches (Tracking constructs ORBmatcher objects on
// the stack per use, Tracking.cc:383,596,818; the context is the long-lived part).
//
// The context also keeps the DEVICE-RESIDENT copies of the frames it has searched (orbfe_frame): the reference builds a
// frame's grid once (Frame.cc:111, 114-129) and reuses it in every search of that frame -- 2-3 per tracked frame
// (Tracking.cc:608, 614, 824), dozens per keyframe (Fuse, SearchBySim3, relocalisation).
//
// IDENTITY OF A CACHED FRAME = ITS CONTENT.  (kind, mnId, N) is NOT an identity in the reference: Tracking::Reset() sets
// Frame::nNextId and KeyFrame::nNextId back to 0 (Tracking.cc:1159-1160) -- routine after a failed monocular
// initialisation -- and Osmap's map load re-creates KeyFrames with the ids stored in the file (Osmap.cpp:586); N sits in
// 2000-2010 at nFeatures 2000, so a recycled id with an equal N is likely.  An entry is therefore keyed by the keypoint
// count, the image bounds (they position the grid) and a 128-bit fingerprint of ALL of mvKeysUn and of ALL descriptor rows,
// recomputed at every lookup (3-4 us against >= 45 us for the search it precedes).  A Frame, its copies (Frame.cc:39-62)
// and the KeyFrame made from it (KeyFrame.cc:37-60) share content and therefore one resident copy; two different frames
// never do, whatever their ids.  orbfe_resident_invalidate() (Tracking::Reset, map load) additionally empties every
// context's cache at its next lookup -- that frees memory; correctness does not depend on it.
// First use uploads the features once (or takes them from the extractor's arena, Extractor::residentFromLastExtract);
// later searches upload only their queries.  Least-recently-used entries are dropped beyond `capacity` (default 48, at
// least the 2 most recent are always kept: a search uses up to two frames; 0 disables the cache and every search takes the
// host-array call form).
class MatcherContext {
 public:
  explicit MatcherContext(int device = 0, size_t frame_cache_capacity = 48) : device_(device), cap_(frame_cache_capacity) {
    check(orbfe_matcher_create(device, &m_));
    epoch_ = orbfe_resident_epoch();
  }
  ~MatcherContext() {
    invalidate();
    if (table_.pending) orbfe_matcher_synchronize(m_);
    if (table_.rows) orbfe_device_free(device_, table_.rows);
    if (table_.mirror) orbfe_host_free(table_.mirror);
    orbfe_matcher_destroy(m_);
    for (auto& s : scratch_)
      if (s.p) orbfe_host_free(s.p);
  }
  // Page-locked scratch array `slot` of at least n elements, zero-filled on request: what the per-frame searches marshal their
  // MapPoint snapshots into.  The search kernels read page-locked arrays in place (no copy inside the library), and the
  // arrays are reused from call to call (no allocation per search).  Valid until the next request for the same slot.
  template <class T>
  T* scratch(int slot, size_t n, bool zero) {
    Scratch& s = scratch_[slot];
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (bytes > s.bytes) {
      if (s.p) orbfe_host_free(s.p);
      s.p = nullptr; s.bytes = 0;
      check(orbfe_host_alloc(bytes + bytes / 2, &s.p));
      s.bytes = bytes + bytes / 2;
    }
    if (zero) std::memset(s.p, 0, bytes);
    return static_cast<T*>(s.p);
  }
  MatcherContext(const MatcherContext&) = delete;
  MatcherContext& operator=(const MatcherContext&) = delete;
  orbfe_matcher* get() const { return m_; }
  int device() const { return device_; }
  void setFrameCacheCapacity(size_t c) { cap_ = c; trim(); }
  size_t residentFrames() const { return cache_.size(); }
  size_t residentUploads() const { return uploads_; }          // frames whose features crossed PCIe (orbfe_frame_create)
  size_t residentFromExtract() const { return fromExtract_; }  // frames taken from an extractor's arena (no upload)
  size_t residentHits() const { return hits_; }
  // drop every cached frame (this context).  Tracking::Reset() / a map load call orbfe_resident_invalidate() instead,
  // which reaches the contexts of all threads.
  void invalidate() {
    for (auto& e : cache_) orbfe_frame_destroy(e.frame);
    cache_.clear();
  }

  // the resident copy of F (a Frame: kind 0, a KeyFrame: kind 1 -- informational, identity is content), created on first
  // use; nullptr when the cache is off
  template <class FrameLike>
  orbfe_frame* resident(const FrameLike& F, int /*kind*/) {
    if (cap_ == 0) return nullptr;
    const unsigned long long ep = orbfe_resident_epoch();
    if (ep != epoch_) { invalidate(); epoch_ = ep; }
    static_assert(sizeof(F.mvKeysUn[0]) == sizeof(OrbfeKeyPoint), "mvKeysUn must hold cv::KeyPoint-layout records");
    const int n = (int)F.mvKeysUn.size();
    const float b[4] = {(float)F.mnMinX, (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY};
    detail::Hash128 hk, hd;
    detail::hashBytes(hk, F.mvKeysUn.data(), (size_t)n * sizeof(OrbfeKeyPoint));
    detail::hashDescriptorRows(hd, F.mDescriptors, n);
    for (auto it = cache_.begin(); it != cache_.end(); ++it)
      if (it->n == n && it->keys == hk && it->desc == hd && std::memcmp(it->bounds, b, sizeof b) == 0) {
        if (it != cache_.begin()) {   // most recently used first
          Entry e = *it;
          cache_.erase(it);
          cache_.insert(cache_.begin(), e);
        }
        hits_++;
        return cache_.front().frame;
      }
    Entry e;
    e.n = n; e.keys = hk; e.desc = hd;
    std::memcpy(e.bounds, b, sizeof b);
    if (n > 0) e.frame = fromExtractor(F, n, hd, b);
    if (e.frame) fromExtract_++;
    else {
      std::vector<uint8_t> tmp;
      const uint8_t* desc = nullptr;
      if (n) {
        if ((size_t)F.mDescriptors.step == 32) desc = F.mDescriptors.data;
        else {
          tmp.resize((size_t)n * 32);
          for (int i = 0; i < n; i++) std::memcpy(&tmp[(size_t)i * 32], F.mDescriptors.data + (size_t)i * F.mDescriptors.step, 32);
          desc = tmp.data();
        }
      }
      check(orbfe_frame_create(m_, reinterpret_cast<const OrbfeKeyPoint*>(F.mvKeysUn.data()), desc, n, b, &e.frame));
      uploads_++;
    }
    cache_.insert(cache_.begin(), e);
    trim();
    return cache_.front().frame;
  }

  // ---- the local map's descriptors on the device.  Tracking::SearchLocalPoints (Tracking.cc:818-824) sends the same few
  // thousand MapPoints frame after frame; their 32-byte descriptors need not cross PCIe every time.  The context keeps a
  // table of rows keyed by the MapPoint's ADDRESS, in two copies: a page-locked host MIRROR (always complete and current)
  // and the DEVICE rows.  Per search the shim asks for each in-view MapPoint's row:
  //   rowFor(pMP, bytes)   compares GetDescriptor()'s bytes with the mirror row of that MapPoint.  Equal: returns the row
  //                        number -- the kernel reads the device copy.  Different (the descriptor was recomputed,
  //                        MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:227-292; a new MapPoint; a new object at
  //                        a recycled address): the mirror row is rewritten and the number comes back with bit 31 set --
  //                        the kernel reads THAT row from the mirror, over PCIe, exactly what a search without the table
  //                        does for every row.  The comparison is of the BYTES, so the device copy can never be stale.
  //   tableCommit()        after the search: rows that changed go to the device asynchronously on the matcher's stream
  //                        (runs of neighbouring rows as one copy), ready for the next frame.
  // The local map's order may change from frame to frame (Tracking::UpdateLocalPoints rebuilds the vector): rows belong to
  // MapPoints, not to positions.  Rows of MapPoints that are gone are reclaimed by starting over when the table holds more
  // than 4x the rows one search uses (at least 16 384).
  void tableBegin(size_t nQueries) {
    DescTable& t = table_;
    if (t.pending) { check(orbfe_matcher_synchronize(m_)); t.pending = false; }
    if (t.slotOf.size() > std::max<size_t>(16384, 4 * std::max(nQueries, t.lastQueries))) {
      t.slotOf.clear(); t.used = 0;
      t.dirty.clear();                                     // the rows start over: nothing of the old assignment is owed to the device
      std::fill(t.notOnDevice.begin(), t.notOnDevice.end(), (uint8_t)0);
    }
    t.lastQueries = nQueries;
    // t.dirty is NOT cleared here: rows a previous snapshot rewrote in the mirror and never sent (its search threw before
    // tableCommit) are still owed to the device; until then rowFor hands them out as mirror rows (notOnDevice)
    if (t.cap == 0) growTable(1);
  }
  int32_t rowFor(const void* pMP, const uint8_t* bytes) {
    DescTable& t = table_;
    auto it = t.slotOf.find(pMP);
    uint32_t slot;
    if (it == t.slotOf.end()) {
      slot = (uint32_t)t.used;
      if (t.used + 1 > t.cap) growTable(t.used + 1);
      t.used++;
      t.slotOf.emplace(pMP, slot);
    } else {
      slot = it->second;
      if (std::memcmp(t.mirror + (size_t)slot * 32, bytes, 32) == 0) {
        // equal bytes in the MIRROR say nothing about the device row while an upload of this row has not been enqueued yet: the same
        // MapPoint twice in one vpMapPoints (the reference tolerates that), or a row of a snapshot whose search never committed
        if (t.notOnDevice[slot]) return (int32_t)(slot | 0x80000000u);
        t.rowsFromDevice++;
        return (int32_t)slot;
      }
    }
    std::memcpy(t.mirror + (size_t)slot * 32, bytes, 32);
    if (!t.notOnDevice[slot]) { t.notOnDevice[slot] = 1; t.dirty.push_back(slot); }
    return (int32_t)(slot | 0x80000000u);
  }
  const uint8_t* tableDevice() const { return table_.rows; }
  const uint8_t* tableMirror() const { return table_.mirror; }
  size_t tableRowCapacity() const { return table_.cap; }
  void tableCommit() {
    DescTable& t = table_;
    if (t.dirty.empty()) { t.cleanSearches++; return; }
    std::sort(t.dirty.begin(), t.dirty.end());
    size_t i = 0;
    while (i < t.dirty.size()) {   // runs of changed rows at most 32 rows apart travel as one copy
      size_t j = i;
      while (j + 1 < t.dirty.size() && t.dirty[j + 1] - t.dirty[j] <= 32) j++;
      const size_t lo = t.dirty[i], hi = (size_t)t.dirty[j] + 1;
      t.pending = true;
      if (orbfe_matcher_upload_async(m_, t.rows + lo * 32, t.mirror + lo * 32, (hi - lo) * 32) != ORBFE_OK) {
        t.dirty.erase(t.dirty.begin(), t.dirty.begin() + (std::ptrdiff_t)i);   // rows i.. stay owed (and stay mirror rows)
        check(ORBFE_ERR_HIP);
      }
      for (size_t k = i; k <= j; k++) t.notOnDevice[t.dirty[k]] = 0;   // enqueued on the matcher's stream: in order before the next search
      t.copies++;
      i = j + 1;
    }
    t.rowsChanged += t.dirty.size();
    t.dirty.clear();
  }
  size_t tableRowsChanged() const { return table_.rowsChanged; }        // rows that went to the device (first sight or new bytes)
  size_t tableCleanSearches() const { return table_.cleanSearches; }    // searches that read every descriptor from device memory
  size_t tableRowsFromDevice() const { return table_.rowsFromDevice; }  // descriptor reads served by the device copy
  size_t tableRows() const { return table_.used; }

 private:
  struct Entry { int n = 0; detail::Hash128 keys, desc; float bounds[4] = {0, 0, 0, 0}; orbfe_frame* frame = nullptr; };
  // the extractor route: F.mvKeys are the records some live Extractor of this device returned last (same count, same
  // bytes, same descriptor rows) -> its arena still holds them
  template <class FrameLike>
  orbfe_frame* fromExtractor(const FrameLike& F, int n, const detail::Hash128& hd, const float b[4]) {
    if ((int)F.mvKeys.size() != n) return nullptr;
    detail::Hash128 hraw;
    detail::hashBytes(hraw, F.mvKeys.data(), (size_t)n * sizeof(OrbfeKeyPoint));
    // undistorted coordinates only when they differ from the extracted ones (Frame.cc:288-292: no distortion => mvKeysUn = mvKeys)
    const bool same = std::memcmp(static_cast<const void*>(F.mvKeys.data()), static_cast<const void*>(F.mvKeysUn.data()),
                                  (size_t)n * sizeof(OrbfeKeyPoint)) == 0;
    std::vector<float> xy;
    if (!same) {
      xy.resize((size_t)n * 2);
      for (int i = 0; i < n; i++) std::memcpy(&xy[2 * (size_t)i], &F.mvKeysUn[i], 8);   // pt = the first two floats
      // everything but pt must be what the extractor returned (UndistortKeyPoints copies the record and replaces pt)
      for (int i = 0; i < n; i++)
        if (std::memcmp(reinterpret_cast<const char*>(&F.mvKeys[i]) + 8, reinterpret_cast<const char*>(&F.mvKeysUn[i]) + 8,
                        sizeof(OrbfeKeyPoint) - 8) != 0)
          return nullptr;
    }
    return Extractor::residentFromLastExtract(device_, n, hraw, hd, b, same ? nullptr : xy.data());
  }
  void trim() {
    const size_t keep = cap_ == 0 ? 0 : std::max<size_t>(cap_, 2);   // a search uses up to two frames: never evict those
    while (cache_.size() > keep) {
      orbfe_frame_destroy(cache_.back().frame);
      cache_.pop_back();
    }
  }
  orbfe_matcher* m_ = nullptr;
  int device_ = 0;
  size_t cap_ = 48, uploads_ = 0, fromExtract_ = 0, hits_ = 0;
  unsigned long long epoch_ = 0;
  std::vector<Entry> cache_;
  struct Scratch { void* p = nullptr; size_t bytes = 0; };
  Scratch scratch_[8];
  struct DescTable {
    uint8_t *rows = nullptr, *mirror = nullptr;   // device rows; page-locked host mirror (complete and current)
    size_t cap = 0, used = 0, lastQueries = 0;
    std::unordered_map<const void*, uint32_t> slotOf;   // MapPoint address -> row
    std::vector<uint32_t> dirty;                  // rows rewritten in the mirror whose upload has not been enqueued yet (each once)
    std::vector<uint8_t> notOnDevice;             // [cap] 1: the row is in `dirty` -- rowFor hands it out as a mirror row whatever the bytes
    size_t rowsChanged = 0, rowsFromDevice = 0, cleanSearches = 0, copies = 0;
    bool pending = false;                         // an upload from the mirror may still be in flight
  };
  void growTable(size_t need) {
    DescTable& t = table_;
    const size_t cap = std::max<size_t>(4096, need + need / 2);
    void *h = nullptr, *d = nullptr;
    check(orbfe_host_alloc(cap * 32, &h));
    if (orbfe_device_malloc(device_, cap * 32, &d) != ORBFE_OK) { orbfe_host_free(h); check(ORBFE_ERR_HIP); }
    if (t.cap) {   // (no upload is in flight here: tableBegin has waited, tableCommit comes after the search)
      std::memcpy(h, t.mirror, t.cap * 32);
      check(orbfe_device_upload(device_, d, t.mirror, t.cap * 32));
      orbfe_device_free(device_, t.rows);
      orbfe_host_free(t.mirror);
    }
    t.mirror = static_cast<uint8_t*>(h);
    t.rows = static_cast<uint8_t*>(d);
    t.cap = cap;
    t.notOnDevice.resize(cap, 0);
  }
  DescTable table_;
};

namespace detail {
// rows of a CV_8U N x 32 descriptor matrix -> contiguous bytes (cv::Mat has public data/step/rows)
template <class MatT>
inline const uint8_t* packedDescriptors(const MatT& m, int n, std::vector<uint8_t>& tmp) {
  if (n == 0) return nullptr;
  if ((size_t)m.step == 32) return m.data;
  tmp.resize((size_t)n * 32);
  for (int i = 0; i < n; i++) std::memcpy(&tmp[(size_t)i * 32], m.data + (size_t)i * m.step, 32);
  return tmp.data();
}

// The descriptor rows of F for the bag-of-words searches: its resident copy's rows in device memory when the context's cache
// is on (nothing is copied then), else the packed host rows.
template <class FrameLike>
inline const uint8_t* descriptorRows(MatcherContext& ctx, const FrameLike& F, int kind, int n, std::vector<uint8_t>& tmp) {
  if (n > 0)
    if (orbfe_frame* rf = ctx.resident(F, kind))
      if (const uint8_t* rows = orbfe_frame_descriptors_device(rf)) return rows;
  return packedDescriptors(F.mDescriptors, n, tmp);
}
template <class FrameT>
inline void frameBounds(const FrameT& F, float b[4]) {
  b[0] = F.mnMinX; b[1] = F.mnMaxX; b[2] = F.mnMinY; b[3] = F.mnMaxY;
}
}  // namespace detail

// int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)   (ORBmatcher.cc:1605-1621)
template <class MatT>
inline int DescriptorDistance(const MatT& a, const MatT& b) { return orbfe_hamming(a.data, b.data); }

// int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
//                                         vector<int>& vnMatches12, int windowSize)   (ORBmatcher.cc:400-515)
template <class FrameT, class Point2fT>
inline int SearchForInitialization(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, FrameT& F1,
                                   FrameT& F2, std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12,
                                   int windowSize) {
  static_assert(sizeof(Point2fT) == 8, "Point2fT must be two packed floats (cv::Point2f)");
  const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
  vnMatches12.assign(n1, -1);
  std::vector<uint8_t> t1, t2;
  float b[4];
  detail::frameBounds(F2, b);
  int nmatches = 0;
  check(orbfe_search_for_initialization(
      ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(F1.mvKeysUn.data()),
      detail::packedDescriptors(F1.mDescriptors, n1, t1), n1,
      reinterpret_cast<const OrbfeKeyPoint*>(F2.mvKeysUn.data()), detail::packedDescriptors(F2.mDescriptors, n2, t2),
      n2, b, reinterpret_cast<float*>(vbPrevMatched.data()), vnMatches12.data(), windowSize, mfNNratio,
      mbCheckOrientation ? 1 : 0, &nmatches));
  return nmatches;
}

// int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
// (ORBmatcher.cc:45-124).  MapPoint fields are snapshotted through the reference's own accessors
// (isBad(), GetDescriptor(), Observations() take the MapPoint mutexes, MapPoint.cc:126-129,294-298)
// BEFORE the GPU call; assignments are written back to F.mvpMapPoints afterwards.
template <class FrameT, class MapPointT>
inline int SearchByProjection(MatcherContext& ctx, float mfNNratio, FrameT& F,
                              const std::vector<MapPointT*>& vpMapPoints, const float th) {
  const int n = (int)F.mvKeysUn.size(), nmp = (int)vpMapPoints.size();
  // snapshots go into the context's page-locked arrays (flags and levels zeroed: an absent MapPoint is "not in view", level 0)
  uint8_t* occ = ctx.scratch<uint8_t>(0, n, true);
  uint8_t* flags = ctx.scratch<uint8_t>(1, nmp, true);
  float* xy = ctx.scratch<float>(3, (size_t)nmp * 2, false);
  float* vcos = ctx.scratch<float>(4, nmp, false);
  int32_t* lvl = ctx.scratch<int32_t>(5, nmp, true);
  int32_t* assigned = ctx.scratch<int32_t>(6, n, false);
  float b[4];
  detail::frameBounds(F, b);
  orbfe_frame* rf = ctx.resident(F, 0);
  // With a resident frame the MapPoints' descriptors come from the context's device table (rows keyed by MapPoint, only
  // changed rows cross PCIe); the host-array call form takes plain rows.
  int32_t* drow = rf ? ctx.scratch<int32_t>(2, nmp, true) : nullptr;
  uint8_t* mdesc = rf ? nullptr : ctx.scratch<uint8_t>(2, (size_t)nmp * 32, false);
  if (rf) ctx.tableBegin((size_t)nmp);
  std::vector<uint8_t> tmp;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  for (int i = 0; i < n; i++)
    if (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) occ[i] = 1;
  for (int i = 0; i < nmp; i++) {
    MapPointT* p = vpMapPoints[i];
    xy[2 * i] = 0.f; xy[2 * i + 1] = 0.f; vcos[i] = 0.f;
    if (!p->mbTrackInView) continue;
    if (p->isBad()) { flags[i] = ORBFE_MP_IN_VIEW | ORBFE_MP_BAD; continue; }
    flags[i] = ORBFE_MP_IN_VIEW | (p->plCandidato ? ORBFE_MP_CANDIDATO : 0) |
               (p->Observations() > 0 ? ORBFE_MP_OBSERVED : 0);
    xy[2 * i] = p->mTrackProjX;
    xy[2 * i + 1] = p->mTrackProjY;
    lvl[i] = p->mnTrackScaleLevel;
    vcos[i] = p->mTrackViewCos;
    const auto d = p->GetDescriptor();
    if (rf) drow[i] = ctx.rowFor(p, d.data);
    else std::memcpy(&mdesc[(size_t)i * 32], d.data, 32);
  }
  int nmatches = 0;
  if (rf) {
    check(orbfe_search_by_projection_frame_rows(ctx.get(), rf, F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), occ, xy, lvl,
                                                vcos, flags, ctx.tableDevice(), ctx.tableMirror(), drow, (int)ctx.tableRowCapacity(), nmp, th,
                                                mfNNratio, assigned, &nmatches));
    ctx.tableCommit();
  } else
    check(orbfe_search_by_projection(ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(F.mvKeysUn.data()),
                                     detail::packedDescriptors(F.mDescriptors, n, tmp), n, b, F.mvScaleFactors.data(),
                                     (int)F.mvScaleFactors.size(), occ, xy, lvl, vcos, flags, mdesc, nmp, th, mfNNratio,
                                     assigned, &nmatches));
  for (int i = 0; i < n; i++)
    if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------
// Pose-driven searches.  The reference projects MapPoints with cv::Mat expressions; what those expressions compute
// is OpenCV arithmetic.  RestatedOps restates it (OpenCV 4.x core/src/matmul.simd.hpp small-matrix gemm path,
// matrix_expressions.cpp, norm.cpp; recalled) so that this header needs no OpenCV; a build that
// has OpenCV passes CvOps (include/orbfe/ORBmatcher.h), whose members ARE the reference's expressions.
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
struct RestatedOps {
  // alpha * A(3x3) * b(3x1) + beta * c   -- `Rcw*x3Dw+tcw`, `-sR21*t12` (gemm, flags 0, len 3: float dot, double epilogue)
  static void gemm3(const float A[9], const float b[3], double alpha, const float* c, double beta, float d[3]) {
    for (int i = 0; i < 3; i++) {
      const float t = A[3 * i] * b[0] + A[3 * i + 1] * b[1] + A[3 * i + 2] * b[2];
      d[i] = (float)((double)t * alpha + (double)(c ? c[i] : 0.f) * beta);
    }
  }
  // alpha * A.t() * b   -- `-Rcw.t()*tcw` (gemm with GEMM_1_T: generic path, double accumulation)
  static void gemmT3(const float A[9], const float b[3], double alpha, float d[3]) {
    for (int i = 0; i < 3; i++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += (double)A[3 * k + i] * (double)b[k];
      d[i] = (float)(s * alpha);
    }
  }
  static double norm3(const float v[3]) {   // cv::norm(v)
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)v[k] * (double)v[k];
    return std::sqrt(s);
  }
  static double dot3(const float a[3], const float b[3]) {   // a.dot(b)
    double r = 0;
    for (int k = 0; k < 3; k++) r += (double)a[k] * (double)b[k];
    return r;
  }
  static void scale(const float* M, int n, double s, float* out) {   // `s*M` (MatExpr scale -> convertTo with a float factor)
    const float f = (float)s;
    for (int i = 0; i < n; i++) out[i] = M[i] * f;
  }
  static void divide(const float* M, int n, double s, float* out) { scale(M, n, 1.0 / s, out); }   // `M/s` = M * (1./s)
};

template <class MatT>
inline void poseRt(const MatT& T, float R[9], float t[3]) {   // T.rowRange(0,3).colRange(0,3), T.rowRange(0,3).col(3)
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[3 * r + c] = T.template at<float>(r, c);
    t[r] = T.template at<float>(r, 3);
  }
}
template <class MatT>
inline void vec3(const MatT& m, float v[3]) {
  for (int r = 0; r < 3; r++) v[r] = m.template at<float>(r, 0);
}
template <class MatT>
inline void mat33(const MatT& m, float v[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) v[3 * r + c] = m.template at<float>(r, c);
}
// Scw -> Rcw, tcw, Ow   (ORBmatcher.cc:293-298, 949-954)
template <class Ops, class MatT>
inline void decomposeScw(const MatT& Scw, float Rcw[9], float tcw[3], float Ow[3]) {
  float sR[9], st[3];
  poseRt(Scw, sR, st);
  const float scw = (float)std::sqrt(Ops::dot3(sR, sR));
  Ops::divide(sR, 9, (double)scw, Rcw);   // sRcw/scw
  Ops::divide(st, 3, (double)scw, tcw);   // Scw.rowRange(0,3).col(3)/scw
  Ops::gemmT3(Rcw, tcw, -1.0, Ow);
}

// one windowed search of MapPoints projected into a KeyFrame (the loops of SearchByProjection(KF, Scw), Fuse x2, SearchBySim3)
struct ProjectedSources {
  std::vector<float> uv, radius;
  std::vector<int32_t> level, bestIdx, bestDist;
  std::vector<uint8_t> valid, desc;
  explicit ProjectedSources(size_t n) : uv(2 * n, 0.f), radius(n, 0.f), level(n, 0), bestIdx(n, -1), bestDist(n, -1), valid(n, 0), desc(32 * n, 0) {}
};
template <class KeyFrameT>
inline int searchProjected(MatcherContext& ctx, KeyFrameT* pKF, ProjectedSources& S, const uint8_t* kp_skip, int claim,
                           bool chi2, int max_dist) {
  const int n = (int)pKF->mvKeysUn.size(), ns = (int)S.valid.size();
  std::vector<uint8_t> tmp;
  float b[4];
  frameBounds(*pKF, b);
  int nm = 0;
  if (orbfe_frame* rf = ctx.resident(*pKF, 1))
    check(orbfe_search_projected_frame(ctx.get(), rf, ns, S.uv.data(), S.radius.data(), S.level.data(), S.valid.data(), S.desc.data(),
                                       kp_skip, claim, chi2 ? pKF->mvInvLevelSigma2.data() : nullptr,
                                       chi2 ? (int)pKF->mvInvLevelSigma2.size() : 0, 5.99, max_dist, S.bestIdx.data(),
                                       S.bestDist.data(), &nm));
  else
    check(orbfe_search_projected(ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(pKF->mvKeysUn.data()),
                                 packedDescriptors(pKF->mDescriptors, n, tmp), n, b, ns, S.uv.data(), S.radius.data(),
                                 S.level.data(), S.valid.data(), S.desc.data(), kp_skip, claim,
                                 chi2 ? pKF->mvInvLevelSigma2.data() : nullptr, chi2 ? (int)pKF->mvInvLevelSigma2.size() : 0, 5.99,
                                 max_dist, S.bestIdx.data(), S.bestDist.data(), &nm));
  return nm;
}
// the part of the KeyFrame-side loops between GetWorldPos and GetFeaturesInArea (:322-355, 826-873, 972-1013): fills
// source i and returns whether the point survives the depth / image / distance / viewing-angle tests
template <class Ops, class KeyFrameT, class MapPointT>
inline bool projectIntoKeyFrame(KeyFrameT* pKF, MapPointT* pMP, const float Rcw[9], const float tcw[3], const float Ow[3],
                                float th, bool invzDouble, ProjectedSources& S, size_t i) {
  float p3Dw[3], p3Dc[3];
  vec3(pMP->GetWorldPos(), p3Dw);
  Ops::gemm3(Rcw, p3Dw, 1.0, tcw, 1.0, p3Dc);
  if (p3Dc[2] < 0.0f) return false;
  const float invz = invzDouble ? (float)(1.0 / p3Dc[2]) : 1 / p3Dc[2];
  const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
  const float u = pKF->fx * x + pKF->cx, v = pKF->fy * y + pKF->cy;
  if (!pKF->IsInImage(u, v)) return false;
  const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
  const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
  const float dist3D = (float)Ops::norm3(PO);
  if (dist3D < minDistance || dist3D > maxDistance) return false;
  float Pn[3];
  vec3(pMP->GetNormal(), Pn);
  if (Ops::dot3(PO, Pn) < 0.5 * dist3D) return false;
  const int nPredictedLevel = pMP->PredictScale(dist3D, pKF->mfLogScaleFactor);
  S.uv[2 * i] = u; S.uv[2 * i + 1] = v;
  S.level[i] = nPredictedLevel;
  S.radius[i] = th * pKF->mvScaleFactors[nPredictedLevel];
  const auto d = pMP->GetDescriptor();
  std::memcpy(&S.desc[32 * i], d.data, 32);
  S.valid[i] = 1;
  return true;
}
}  // namespace detail

// int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th)   (ORBmatcher.cc:1292-1423)
template <class Ops = detail::RestatedOps, class FrameT>
inline int SearchByProjection(MatcherContext& ctx, bool mbCheckOrientation, FrameT& CurrentFrame, const FrameT& LastFrame,
                              const float th) {
  float Rcw[9], tcw[3];
  detail::poseRt(CurrentFrame.mTcw, Rcw, tcw);
  const int n = (int)CurrentFrame.mvKeysUn.size(), ns = (int)LastFrame.N;
  // the context's page-locked arrays (see SearchByProjection(F, MapPoints) above); valid / flags / levels zeroed
  float* uv = ctx.scratch<float>(3, (size_t)ns * 2, true);
  float* ang = ctx.scratch<float>(4, ns, true);
  int32_t* lvl = ctx.scratch<int32_t>(5, ns, true);
  int32_t* assigned = ctx.scratch<int32_t>(6, n, false);
  uint8_t* valid = ctx.scratch<uint8_t>(7, ns, true);
  uint8_t* flags = ctx.scratch<uint8_t>(1, ns, true);
  uint8_t* sdesc = ctx.scratch<uint8_t>(2, (size_t)ns * 32, false);
  uint8_t* occ = ctx.scratch<uint8_t>(0, n, true);
  std::vector<uint8_t> tmp;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  for (int i = 0; i < n; i++)
    if (CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0) occ[i] = 1;   // :1364-1366
  for (int i = 0; i < ns; i++) {
    auto* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    float x3Dw[3], x3Dc[3];
    detail::vec3(pMP->GetWorldPos(), x3Dw);
    Ops::gemm3(Rcw, x3Dw, 1.0, tcw, 1.0, x3Dc);                       // Rcw*x3Dw+tcw
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
    const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
    if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
    if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
    uv[2 * i] = u; uv[2 * i + 1] = v;
    lvl[i] = LastFrame.mvKeys[i].octave;                              // nLastOctave
    ang[i] = LastFrame.mvKeysUn[i].angle;
    flags[i] = pMP->Observations() > 0 ? ORBFE_MP_OBSERVED : 0;
    const auto d = pMP->GetDescriptor();
    std::memcpy(&sdesc[(size_t)i * 32], d.data, 32);
    valid[i] = 1;
  }
  float b[4];
  detail::frameBounds(CurrentFrame, b);
  int nmatches = 0;
  if (orbfe_frame* rf = ctx.resident(CurrentFrame, 0))
    check(orbfe_search_by_projection_uv_frame(ctx.get(), rf, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                              occ, uv, lvl, ang, flags, valid, sdesc,
                                              ns, th, /*TH_HIGH*/ 100, /*skip_any_occupied*/ 0, mbCheckOrientation ? 1 : 0,
                                              assigned, &nmatches));
  else
    check(orbfe_search_by_projection_uv(ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(CurrentFrame.mvKeysUn.data()),
                                        detail::packedDescriptors(CurrentFrame.mDescriptors, n, tmp), n, b,
                                        CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(), occ,
                                        uv, lvl, ang, flags, valid, sdesc, ns, th,
                                        /*TH_HIGH*/ 100, /*skip_any_occupied*/ 0, mbCheckOrientation ? 1 : 0, assigned,
                                        &nmatches));
  for (int i = 0; i < n; i++) {
    if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[assigned[i]];
    else if (assigned[i] == -2) CurrentFrame.mvpMapPoints[i] = nullptr;    // rotation check, :1409-1419
  }
  return nmatches;
}

// int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound,
//                                    const float th, const int ORBdist)   (ORBmatcher.cc:1425-1552)
template <class Ops = detail::RestatedOps, class FrameT, class KeyFrameT, class SetT>
inline int SearchByProjection(MatcherContext& ctx, bool mbCheckOrientation, FrameT& CurrentFrame, KeyFrameT* pKF,
                              const SetT& sAlreadyFound, const float th, const int ORBdist) {
  float Rcw[9], tcw[3], Ow[3];
  detail::poseRt(CurrentFrame.mTcw, Rcw, tcw);
  Ops::gemmT3(Rcw, tcw, -1.0, Ow);                                       // -Rcw.t()*tcw
  const auto vpMPs = pKF->GetMapPointMatches();
  const int n = (int)CurrentFrame.mvKeysUn.size(), ns = (int)vpMPs.size();
  std::vector<float> uv((size_t)ns * 2, 0.f), ang(ns, 0.f);
  std::vector<int32_t> lvl(ns, 0), assigned(n > 0 ? n : 1, -1);
  std::vector<uint8_t> valid(ns, 0), flags(ns, 0), sdesc((size_t)ns * 32, 0), occ(n > 0 ? n : 1, 0), tmp;
  for (int i = 0; i < n; i++)
    if (CurrentFrame.mvpMapPoints[i]) occ[i] = 1;                         // :1493-1494
  for (int i = 0; i < ns; i++) {
    auto* pMP = vpMPs[i];
    if (!pMP) continue;
    if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
    float x3Dw[3], x3Dc[3];
    detail::vec3(pMP->GetWorldPos(), x3Dw);
    Ops::gemm3(Rcw, x3Dw, 1.0, tcw, 1.0, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
    const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
    if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
    if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
    const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
    const float dist3D = (float)Ops::norm3(PO);
    const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    uv[2 * i] = u; uv[2 * i + 1] = v;
    lvl[i] = pMP->PredictScale(dist3D, CurrentFrame.mfLogScaleFactor);
    ang[i] = pKF->mvKeysUn[i].angle;
    const auto d = pMP->GetDescriptor();
    std::memcpy(&sdesc[(size_t)i * 32], d.data, 32);
    valid[i] = 1;
  }
  float b[4];
  detail::frameBounds(CurrentFrame, b);
  int nmatches = 0;
  if (orbfe_frame* rf = ctx.resident(CurrentFrame, 0))
    check(orbfe_search_by_projection_uv_frame(ctx.get(), rf, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                              occ.data(), uv.data(), lvl.data(), ang.data(), flags.data(), valid.data(), sdesc.data(),
                                              ns, th, ORBdist, /*skip_any_occupied*/ 1, mbCheckOrientation ? 1 : 0, assigned.data(),
                                              &nmatches));
  else
    check(orbfe_search_by_projection_uv(ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(CurrentFrame.mvKeysUn.data()),
                                        detail::packedDescriptors(CurrentFrame.mDescriptors, n, tmp), n, b,
                                        CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(), occ.data(),
                                        uv.data(), lvl.data(), ang.data(), flags.data(), valid.data(), sdesc.data(), ns, th, ORBdist,
                                        /*skip_any_occupied*/ 1, mbCheckOrientation ? 1 : 0, assigned.data(), &nmatches));
  for (int i = 0; i < n; i++) {
    if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[assigned[i]];
    else if (assigned[i] == -2) CurrentFrame.mvpMapPoints[i] = nullptr;
  }
  return nmatches;
}

// int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
//                                    vector<MapPoint*>& vpMatched, int th)   (ORBmatcher.cc:285-398)
template <class Ops = detail::RestatedOps, class KeyFrameT, class MatT, class MapPointT>
inline int SearchByProjection(MatcherContext& ctx, KeyFrameT* pKF, const MatT& Scw, const std::vector<MapPointT*>& vpPoints,
                              std::vector<MapPointT*>& vpMatched, int th) {
  float Rcw[9], tcw[3], Ow[3];
  detail::decomposeScw<Ops>(Scw, Rcw, tcw, Ow);
  std::vector<MapPointT*> found(vpMatched.begin(), vpMatched.end());     // spAlreadyFound (:301-302)
  std::sort(found.begin(), found.end());
  const size_t ns = vpPoints.size();
  detail::ProjectedSources S(ns);
  for (size_t i = 0; i < ns; i++) {
    MapPointT* pMP = vpPoints[i];
    if (pMP->isBad() || std::binary_search(found.begin(), found.end(), pMP)) continue;
    detail::projectIntoKeyFrame<Ops>(pKF, pMP, Rcw, tcw, Ow, (float)th, /*invz = 1/z in float*/ false, S, i);
  }
  std::vector<uint8_t> skip(vpMatched.size() ? vpMatched.size() : 1, 0);
  for (size_t i = 0; i < vpMatched.size(); i++) skip[i] = vpMatched[i] ? 1 : 0;   // :366-367
  const int nmatches = detail::searchProjected(ctx, pKF, S, skip.data(), /*claim*/ 1, /*chi2*/ false, /*TH_LOW*/ 50);
  for (size_t i = 0; i < ns; i++)
    if (S.bestIdx[i] >= 0) vpMatched[S.bestIdx[i]] = vpPoints[i];
  return nmatches;
}

// int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th)   (ORBmatcher.cc:806-939).
// Which keypoint a MapPoint selects depends only on the keyframe's keypoints and the point itself, so all windows are
// searched in one GPU call; the reference's loop is then replayed IN ORDER with its live checks (isBad / IsInKeyFrame
// change as earlier points are replaced or added).  vpMapPoints holds distinct points, as the reference's callers
// guarantee (LocalMapping::SearchInNeighbors marks candidates with mnFuseCandidateForKF).
template <class Ops = detail::RestatedOps, class KeyFrameT, class MapPointT>
inline int Fuse(MatcherContext& ctx, KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th) {
  float Rcw[9], tcw[3], Ow[3];
  detail::mat33(pKF->GetRotation(), Rcw);
  detail::vec3(pKF->GetTranslation(), tcw);
  detail::vec3(pKF->GetCameraCenter(), Ow);
  const size_t ns = vpMapPoints.size();
  detail::ProjectedSources S(ns);
  for (size_t i = 0; i < ns; i++)
    if (vpMapPoints[i]) detail::projectIntoKeyFrame<Ops>(pKF, vpMapPoints[i], Rcw, tcw, Ow, th, false, S, i);
  detail::searchProjected(ctx, pKF, S, nullptr, 0, /*chi2 gate :896-903*/ true, /*TH_LOW*/ 50);
  int nFused = 0;
  for (size_t i = 0; i < ns; i++) {
    MapPointT* pMP = vpMapPoints[i];
    if (!pMP) continue;
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
    if (!S.valid[i] || S.bestIdx[i] < 0) continue;
    const int bestIdx = S.bestIdx[i];
    MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, bestIdx);
      pKF->AddMapPoint(pMP, bestIdx);
    }
    nFused++;
  }
  return nFused;
}

// int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th,
//                      vector<MapPoint*>& vpReplacePoint)   (ORBmatcher.cc:941-1064)
template <class Ops = detail::RestatedOps, class KeyFrameT, class MatT, class MapPointT>
inline int Fuse(MatcherContext& ctx, KeyFrameT* pKF, const MatT& Scw, const std::vector<MapPointT*>& vpPoints, float th,
                std::vector<MapPointT*>& vpReplacePoint) {
  float Rcw[9], tcw[3], Ow[3];
  detail::decomposeScw<Ops>(Scw, Rcw, tcw, Ow);
  const auto spAlreadyFound = pKF->GetMapPoints();
  const size_t ns = vpPoints.size();
  detail::ProjectedSources S(ns);
  for (size_t i = 0; i < ns; i++) {
    MapPointT* pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    detail::projectIntoKeyFrame<Ops>(pKF, pMP, Rcw, tcw, Ow, th, /*invz = 1.0/z*/ true, S, i);
  }
  detail::searchProjected(ctx, pKF, S, nullptr, 0, false, /*TH_LOW*/ 50);
  int nFused = 0;
  for (size_t i = 0; i < ns; i++) {
    if (!S.valid[i] || S.bestIdx[i] < 0) continue;
    MapPointT* pMP = vpPoints[i];
    const int bestIdx = S.bestIdx[i];
    MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);     // live: an earlier point of this call may have been added here
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, bestIdx);
      pKF->AddMapPoint(pMP, bestIdx);
    }
    nFused++;
  }
  return nFused;
}

// int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const float& s12,
//                              const cv::Mat& R12, const cv::Mat& t12, const float th)   (ORBmatcher.cc:1066-1290)
template <class Ops = detail::RestatedOps, class KeyFrameT, class MatT, class MapPointT>
inline int SearchBySim3(MatcherContext& ctx, KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12,
                        const float& s12, const MatT& R12m, const MatT& t12m, const float th) {
  const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
  float R1w[9], t1w[3], R2w[9], t2w[3], R12[9], t12[3], sR12[9], sR21[9], R12t[9], t21[3];
  detail::mat33(pKF1->GetRotation(), R1w); detail::vec3(pKF1->GetTranslation(), t1w);
  detail::mat33(pKF2->GetRotation(), R2w); detail::vec3(pKF2->GetTranslation(), t2w);
  detail::mat33(R12m, R12); detail::vec3(t12m, t12);
  Ops::scale(R12, 9, (double)s12, sR12);                                  // s12*R12
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R12t[3 * r + c] = R12[3 * c + r];
  Ops::scale(R12t, 9, 1.0 / (double)s12, sR21);                           // (1.0/s12)*R12.t()
  Ops::gemm3(sR21, t12, -1.0, nullptr, 0.0, t21);                         // -sR21*t12
  const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
  std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
  for (int i = 0; i < N1; i++) {
    MapPointT* pMP = vpMatches12[i];
    if (pMP) {
      vbAlreadyMatched1[i] = true;
      const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
      if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
    }
  }
  // one direction: points of `from` (camera pose Rfw, tfw) through (sR, t) into `to`
  auto direction = [&](const std::vector<MapPointT*>& pts, const std::vector<bool>& already, const float* Rfw, const float* tfw,
                       const float* sR, const float* tt, KeyFrameT* to, std::vector<int>& vnMatch) {
    detail::ProjectedSources S(pts.size());
    for (size_t i = 0; i < pts.size(); i++) {
      MapPointT* pMP = pts[i];
      if (!pMP || already[i]) continue;
      if (pMP->isBad()) continue;
      float p3Dw[3], pa[3], pb[3];
      detail::vec3(pMP->GetWorldPos(), p3Dw);
      Ops::gemm3(Rfw, p3Dw, 1.0, tfw, 1.0, pa);
      Ops::gemm3(sR, pa, 1.0, tt, 1.0, pb);
      if (pb[2] < 0.0) continue;
      const float invz = (float)(1.0 / pb[2]);
      const float x = pb[0] * invz, y = pb[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;                       // pKF1's intrinsics in both directions (:1069-1072)
      if (!to->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      const float dist3D = (float)Ops::norm3(pb);
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, to->mfLogScaleFactor);
      S.uv[2 * i] = u; S.uv[2 * i + 1] = v;
      S.level[i] = nPredictedLevel;
      S.radius[i] = th * to->mvScaleFactors[nPredictedLevel];
      const auto d = pMP->GetDescriptor();
      std::memcpy(&S.desc[32 * i], d.data, 32);
      S.valid[i] = 1;
    }
    detail::searchProjected(ctx, to, S, nullptr, 0, false, /*TH_HIGH*/ 100);
    vnMatch.assign(S.bestIdx.begin(), S.bestIdx.end());
  };
  std::vector<int> vnMatch1, vnMatch2;
  direction(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1);
  direction(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      const int idx1 = vnMatch2[idx2];
      if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
    }
  }
  return nFound;
}

// void Frame::UndistortKeyPoints()   (Frame.cc:286-320) and void Frame::ComputeImageBounds(const cv::Mat& imLeft)
// (:322-353), both camera models (camaraModo 0 = pinhole with mDistCoef through cv::undistortPoints, 1 = os1's
// equidistant fisheye).  K = {fx, fy, cx, cy}; dist = mDistCoef (4, 5 or 8 floats; ignored for mode 1).
template <class KeyPointT>
inline void UndistortKeyPoints(const std::vector<KeyPointT>& mvKeys, std::vector<KeyPointT>& mvKeysUn, int camaraModo, float fx,
                               float fy, float cx, float cy, const float* dist, int ndist) {
  if (camaraModo == 0 && (ndist == 0 || dist[0] == 0.0)) { mvKeysUn = mvKeys; return; }
  static_assert(sizeof(KeyPointT) == sizeof(OrbfeKeyPoint), "KeyPointT must match cv::KeyPoint's layout");
  const int N = (int)mvKeys.size();
  std::vector<float> mat((size_t)N * 2);
  for (int i = 0; i < N; i++) {
    OrbfeKeyPoint k;
    std::memcpy(&k, &mvKeys[i], sizeof k);
    mat[2 * i] = k.x; mat[2 * i + 1] = k.y;
  }
  if (camaraModo) check(orbfe_undistort_equidistant(mat.data(), N, fx, fy, cx, cy));
  else check(orbfe_undistort_pinhole(mat.data(), N, fx, fy, cx, cy, dist, ndist));
  mvKeysUn.resize(N);
  for (int i = 0; i < N; i++) {
    KeyPointT kp = mvKeys[i];
    std::memcpy(static_cast<void*>(&kp), &mat[2 * i], 2 * sizeof(float));   // kp.pt = (x, y): the first two floats of cv::KeyPoint
    mvKeysUn[i] = kp;
  }
}
inline void ComputeImageBounds(int cols, int rows, int camaraModo, float fx, float fy, float cx, float cy, const float* dist,
                               int ndist, float& mnMinX, float& mnMaxX, float& mnMinY, float& mnMaxY) {
  float b[4];
  check(orbfe_compute_image_bounds(cols, rows, camaraModo, fx, fy, cx, cy, dist, ndist, b));
  mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
}

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as far as the path uses it: loaded from the
// fork's binary vocabulary file (TemplatedVocabulary.h:1563-1640), resident in HBM.
class Vocabulary {
 public:
  explicit Vocabulary(int device = 0) : device_(device) {}
  ~Vocabulary() { orbfe_vocabulary_destroy(v_); }
  Vocabulary(const Vocabulary&) = delete;
  Vocabulary& operator=(const Vocabulary&) = delete;
  // bool loadFromBinaryFile(const std::string& filename)
  bool loadFromBinaryFile(const std::string& filename) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> image;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) image.insert(image.end(), buf, buf + got);
    std::fclose(f);
    orbfe_vocabulary* nv = nullptr;
    if (orbfe_vocabulary_create_from_image(device_, image.data(), image.size(), &nv) != ORBFE_OK) return false;
    orbfe_vocabulary_destroy(v_);
    v_ = nv;
    return true;
  }
  bool empty() const { return v_ == nullptr; }
  orbfe_vocabulary* get() const { return v_; }

  // void transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup) const
  // (TemplatedVocabulary.h:1136-1204) on the rows of a CV_8U N x 32 matrix.  BowVectorT / FeatureVectorT are the
  // DBoW2 map types (std::map<WordId, WordValue>, std::map<NodeId, std::vector<unsigned int>>).
  template <class MatT, class BowVectorT, class FeatureVectorT>
  void transform(const MatT& descriptors, int n, BowVectorT& v, FeatureVectorT& fv, int levelsup) {
    v.clear();
    fv.clear();
    if (!v_ || n == 0) return;
    std::vector<uint8_t> tmp;
    ids_.resize(n); vals_.resize(n); nodes_.resize(n); offs_.resize(n + 1); feats_.resize(n);
    int nw = 0, nn = 0;
    check(orbfe_bow_transform(v_, detail::packedDescriptors(descriptors, n, tmp), n, 0, levelsup, ids_.data(),
                              vals_.data(), &nw, nodes_.data(), offs_.data(), feats_.data(), &nn, nullptr, nullptr));
    for (int i = 0; i < nw; i++) v.insert(v.end(), typename BowVectorT::value_type(ids_[i], vals_[i]));
    for (int i = 0; i < nn; i++) {
      auto it = fv.insert(fv.end(), typename FeatureVectorT::value_type(nodes_[i], typename FeatureVectorT::mapped_type()));
      it->second.assign(feats_.begin() + offs_[i], feats_.begin() + offs_[i + 1]);
    }
  }

 private:
  int device_;
  orbfe_vocabulary* v_ = nullptr;
  std::vector<uint32_t> ids_, nodes_, offs_, feats_;
  std::vector<double> vals_;
};

// void Frame::ComputeBoW()   (Frame.cc:277-284); KeyFrame::ComputeBoW (KeyFrame.cc) is the same call.
template <class FrameT>
inline void ComputeBoW(Vocabulary& voc, FrameT& F) {
  if (F.mBowVec.empty()) voc.transform(F.mDescriptors, (int)F.mDescriptors.rows, F.mBowVec, F.mFeatVec, 4);
}

namespace detail {
template <class FeatureVectorT>
inline void flattenFeatureVector(const FeatureVectorT& fv, std::vector<uint32_t>& nodes, std::vector<uint32_t>& offs,
                                 std::vector<uint32_t>& feats) {
  nodes.clear(); offs.clear(); feats.clear();
  for (const auto& e : fv) {
    nodes.push_back((uint32_t)e.first);
    offs.push_back((uint32_t)feats.size());
    feats.insert(feats.end(), e.second.begin(), e.second.end());
  }
  offs.push_back((uint32_t)feats.size());
}
template <class KeyPointT>
inline void angles(const std::vector<KeyPointT>& k, std::vector<float>& a) {
  a.resize(k.size());
  for (size_t i = 0; i < k.size(); i++) a[i] = k[i].angle;
}
template <class MapPointT>
inline void validFlags(const std::vector<MapPointT*>& mps, std::vector<uint8_t>& valid) {
  valid.assign(mps.size(), 0);
  for (size_t i = 0; i < mps.size(); i++)
    if (mps[i] && !mps[i]->isBad()) valid[i] = 1;   // ORBmatcher.cc:191-196 / 553-557
}
}  // namespace detail

// int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  (ORBmatcher.cc:154-283)
template <class KeyFrameT, class FrameT, class MapPointT>
inline int SearchByBoW(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, KeyFrameT* pKF, FrameT& F,
                       std::vector<MapPointT*>& vpMapPointMatches) {
  const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
  const int n1 = (int)vpMapPointsKF.size(), n2 = (int)F.N;
  vpMapPointMatches.assign(n2, static_cast<MapPointT*>(nullptr));
  std::vector<uint8_t> valid1, t1, t2;
  std::vector<float> a1, a2;
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::validFlags(vpMapPointsKF, valid1);
  detail::angles(pKF->mvKeysUn, a1);
  detail::angles(F.mvKeys, a2);
  detail::flattenFeatureVector(pKF->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(F.mFeatVec, n2v, o2v, f2v);
  std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
  int nmatches = 0;
  check(orbfe_search_by_bow(ctx.get(), detail::descriptorRows(ctx, *pKF, 1, n1, t1), a1.data(), valid1.data(), n1,
                            n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
                            detail::descriptorRows(ctx, F, 0, n2, t2), a2.data(), nullptr, n2, n2v.data(),
                            o2v.data(), f2v.data(), (int)n2v.size(), mfNNratio, mbCheckOrientation ? 1 : 0, 0, m12.data(),
                            &nmatches));
  for (int i = 0; i < n1; i++)
    if (m12[i] >= 0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
  return nmatches;
}

// The SearchByBoW loop of Tracking::Relocalization (Tracking.cc:1005-1030) as ONE GPU submission: every candidate keyframe
// against the current frame.  vvpMapPointMatches[k] and the returned counts are what the per-keyframe function above gives;
// keyframes for which `skip[k]` is set (pKF->isBad(), :1010-1011) are left out (count 0, matches untouched).  Optional: an
// integration that keeps Tracking.cc unchanged simply keeps calling the per-keyframe member.
template <class KeyFrameT, class FrameT, class MapPointT>
inline std::vector<int> SearchByBoW(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, const std::vector<KeyFrameT*>& vpKFs,
                                    const std::vector<bool>& skip, FrameT& F, std::vector<std::vector<MapPointT*> >& vvpMapPointMatches) {
  const int K = (int)vpKFs.size(), n2 = (int)F.N;
  std::vector<int> counts(K, 0);
  vvpMapPointMatches.resize(K);
  struct Side { std::vector<MapPointT*> mps; std::vector<uint8_t> valid, tmp; std::vector<float> ang; std::vector<uint32_t> nodes, offs, feats;
                std::vector<int32_t> m12; const uint8_t* desc = nullptr; };
  std::vector<Side> S(K);
  std::vector<const uint8_t*> d1, v1;
  std::vector<const float*> a1;
  std::vector<const uint32_t*> fn, fo, ff;
  std::vector<int> n1, nf, which;
  std::vector<int32_t*> out;
  for (int k = 0; k < K; k++) {
    if (k < (int)skip.size() && skip[k]) continue;
    Side& s = S[k];
    s.mps = vpKFs[k]->GetMapPointMatches();
    const int n = (int)s.mps.size();
    detail::validFlags(s.mps, s.valid);
    detail::angles(vpKFs[k]->mvKeysUn, s.ang);
    detail::flattenFeatureVector(vpKFs[k]->mFeatVec, s.nodes, s.offs, s.feats);
    s.desc = detail::packedDescriptors(vpKFs[k]->mDescriptors, n, s.tmp);
    s.m12.assign(n > 0 ? n : 1, -1);
    d1.push_back(s.desc); v1.push_back(s.valid.data()); a1.push_back(s.ang.data()); n1.push_back(n);
    fn.push_back(s.nodes.data()); fo.push_back(s.offs.data()); ff.push_back(s.feats.data()); nf.push_back((int)s.nodes.size());
    out.push_back(s.m12.data());
    which.push_back(k);
    vvpMapPointMatches[k].assign(n2, static_cast<MapPointT*>(nullptr));
  }
  if (which.empty()) return counts;
  std::vector<uint8_t> t2;
  std::vector<float> a2;
  std::vector<uint32_t> n2v, o2v, f2v;
  detail::angles(F.mvKeys, a2);
  detail::flattenFeatureVector(F.mFeatVec, n2v, o2v, f2v);
  std::vector<int> nm(which.size(), 0);
  check(orbfe_search_by_bow_batch(ctx.get(), (int)which.size(), d1.data(), a1.data(), v1.data(), n1.data(), fn.data(), fo.data(), ff.data(),
                                  nf.data(), detail::packedDescriptors(F.mDescriptors, n2, t2), a2.data(), nullptr, n2, n2v.data(), o2v.data(),
                                  f2v.data(), (int)n2v.size(), mfNNratio, mbCheckOrientation ? 1 : 0, 0, out.data(), nm.data()));
  for (size_t j = 0; j < which.size(); j++) {
    const int k = which[j];
    counts[k] = nm[j];
    for (int i = 0; i < n1[j]; i++)
      if (S[k].m12[i] >= 0) vvpMapPointMatches[k][S[k].m12[i]] = S[k].mps[i];
  }
  return counts;
}

// int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  (ORBmatcher.cc:517-650)
template <class KeyFrameT, class MapPointT>
inline int SearchByBoW(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, KeyFrameT* pKF1, KeyFrameT* pKF2,
                       std::vector<MapPointT*>& vpMatches12) {
  const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
  vpMatches12.assign(n1, static_cast<MapPointT*>(nullptr));
  std::vector<uint8_t> valid1, valid2, t1, t2;
  std::vector<float> a1, a2;
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::validFlags(vpMapPoints1, valid1);
  detail::validFlags(vpMapPoints2, valid2);
  detail::angles(pKF1->mvKeysUn, a1);
  detail::angles(pKF2->mvKeysUn, a2);
  detail::flattenFeatureVector(pKF1->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(pKF2->mFeatVec, n2v, o2v, f2v);
  std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
  if (valid2.empty()) valid2.push_back(0);
  int nmatches = 0;
  check(orbfe_search_by_bow(ctx.get(), detail::descriptorRows(ctx, *pKF1, 1, n1, t1), a1.data(), valid1.data(), n1,
                            n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
                            detail::descriptorRows(ctx, *pKF2, 1, n2, t2), a2.data(), valid2.data(), n2,
                            n2v.data(), o2v.data(), f2v.data(), (int)n2v.size(), mfNNratio, mbCheckOrientation ? 1 : 0, 1,
                            m12.data(), &nmatches));
  for (int i = 0; i < n1; i++)
    if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
  return nmatches;
}

// int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
//                                        vector<pair<size_t,size_t>>& vMatchedPairs)   (ORBmatcher.cc:652-804).
// The body keeps the reference's epipole lines (:657-666) and passes (ex, ey); F12 is read through .at<float>(r,c).
template <class KeyFrameT, class MatT>
inline int SearchForTriangulation(MatcherContext& ctx, bool mbCheckOrientation, KeyFrameT* pKF1, KeyFrameT* pKF2,
                                  const MatT& F12, float ex, float ey,
                                  std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
  const int n1 = (int)pKF1->N, n2 = (int)pKF2->N;
  std::vector<uint8_t> has1(n1 > 0 ? n1 : 1, 0), has2(n2 > 0 ? n2 : 1, 0), t1, t2;
  for (int i = 0; i < n1; i++) has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;   // ORBmatcher.cc:704-709
  for (int i = 0; i < n2; i++) has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;   // :723-727
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::flattenFeatureVector(pKF1->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(pKF2->mFeatVec, n2v, o2v, f2v);
  float F[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) F[3 * r + c] = F12.template at<float>(r, c);
  std::vector<int32_t> pairs((size_t)(n1 > 0 ? n1 : 1) * 2, -1);
  int nmatches = 0;
  check(orbfe_search_for_triangulation(
      ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(pKF1->mvKeysUn.data()), detail::packedDescriptors(pKF1->mDescriptors, n1, t1),
      has1.data(), n1, n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
      reinterpret_cast<const OrbfeKeyPoint*>(pKF2->mvKeysUn.data()), detail::packedDescriptors(pKF2->mDescriptors, n2, t2),
      has2.data(), n2, n2v.data(), o2v.data(), f2v.data(), (int)n2v.size(), F, ex, ey, pKF2->mvScaleFactors.data(),
      pKF2->mvLevelSigma2.data(), (int)pKF2->mvScaleFactors.size(), mbCheckOrientation ? 1 : 0, pairs.data(), &nmatches));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(nmatches);
  for (int i = 0; i < nmatches; i++) vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
  return nmatches;
}

}  // namespace orbfe
