// orbfe_stream.cpp -- native frame-stream runner on top of the extractor / matcher C ABI.
//
// Role in the reference: the per-frame front end is driven by the Tracking thread, one frame at a
// time (System::TrackMonocular -> Frame::Frame -> ORBextractor, then ORBmatcher).
// For throughput over a stream (BASELINE.json config 4: independent streams, one per GPU) this
// runner keeps the GPU fed without any per-frame host orchestration in the caller's language:
//   push(batch of frames)  ->  [extract thread]  async submit on one of `depth` extractor handles,
//                              collect in order
//                          ->  [match thread]    SearchForInitialization of every frame against its
//                              predecessor in the stream (one batched GPU submission)
//   pop()                  ->  results of the oldest finished batch, in push order.
// Everything the GPU computes goes through the same entry points a single-frame caller uses
// (orbfe_extract_batch_submit/_collect, orbfe_search_for_initialization_batch), so results are
// identical to calling those one by one.
#include <pthread.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"

namespace orbfe {
void set_err(const char* fmt, ...);
}
int orbfe_concurrent_streams(orbfe_extractor* const* hs, int n);   // orbfe_extractor.hip
using orbfe::set_err;

namespace {
struct Slot {
  std::vector<const uint8_t*> frames;
  int rows = 0, cols = 0;
  size_t stride = 0;
  int onDevice = 1;
  std::vector<OrbfeKeyPoint> kps;
  std::vector<uint8_t> desc;
  std::vector<int> n;
  std::vector<int32_t> m12;
  std::vector<int> nm;
  std::vector<float> prevxy;
  std::vector<uint32_t> bowLeaf, bowNode;   // [batch][cap], with a vocabulary set
  // predecessor of frame 0 (last frame of the previous batch), copied at collect time
  std::vector<OrbfeKeyPoint> prevKps;
  std::vector<uint8_t> prevDesc;
  int prevN = -1;
  long long seq = 0;
  bool done = false;
  int status = ORBFE_OK;
  std::string err;
  // match parameters of the stream at push time (orbfe_stream_set_matching only works on an idle stream, but the
  // snapshot keeps submit and collect of one batch on the same variant by construction)
  int window = 0, checkOri = 1;
  float nnratio = 0.9f;
  float bounds[4] = {0, 0, 0, 0};
};
}  // namespace

struct orbfe_stream {
  int batch = 0, depth = 0, cap = 0, device = 0;
  int inFlight = 0;   // batches on the GPU at a time: depth, or fewer when the process has fewer hardware queues than that (orbfe_stream_create)
  int window = 100, checkOri = 1;
  float nnratio = 0.9f;
  float bounds[4] = {0, 0, 0, 0};
  std::vector<orbfe_extractor*> ext;
  std::vector<orbfe_matcher*> matchers;   // one per match worker (host-side matching path)
  orbfe_sfi_chain* chain = nullptr;       // GPU-resident matching path (default)
  bool gpuMatch = true;
  bool isolated = false;                  // orbfe_stream_set_isolated_batches: frame 0 of a batch has no predecessor
  std::vector<Slot> slots;
  void growSlots(int nslots) {     // caller holds no batch in flight (or is the constructor)
    const int old = (int)slots.size();
    if (nslots <= old) return;
    slots.resize(nslots);
    for (int i = old; i < nslots; i++) {
      Slot& sl = slots[i];
      sl.frames.resize(batch);
      sl.kps.resize((size_t)batch * cap);
      sl.desc.resize((size_t)batch * cap * 32);
      sl.n.assign(batch, 0);
      sl.m12.assign((size_t)batch * cap, -1);
      sl.nm.assign(batch, 0);
      sl.prevxy.resize((size_t)batch * cap * 2);
      freeQ.push_back(i);
    }
  }

  std::mutex mu;
  std::condition_variable cv;
  std::deque<int> freeQ, extractQ, matchQ, doneQ;
  int popped = -1;  // slot handed to the caller by the last pop (returned to freeQ on the next pop)
  bool stop = false;
  std::thread tExtract;
  std::vector<std::thread> tMatch;
  long long pushSeq = 0, popSeq = 0;
  int channels = 1;   // bytes per pixel of the pushed frames (orbfe_stream_set_input_format)
  bool bow = false;   // orbfe_stream_set_vocabulary

  // last frame of the previous batch (the predecessor of frame 0 of the next one)
  std::vector<OrbfeKeyPoint> lastKps;
  std::vector<uint8_t> lastDesc;
  int lastN = -1;

  // busy time of the two workers (ms) and batches done, for orbfe_stream_stats
  double busySubmit = 0, busyCollect = 0, busyMatch = 0;
  long long nBatches = 0;
  static double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  void extractLoop() {
    std::deque<std::pair<int, int>> inflight;  // (slot, extractor)
    int nextExt = 0;
    for (;;) {
      int job = -1;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !inflight.empty() || !extractQ.empty(); });
        // on stop, batches that were queued but never submitted are dropped (their frames may already be gone);
        // only batches the GPU is working on are waited for
        if (stop) extractQ.clear();
        if (stop && inflight.empty()) return;
        if (!extractQ.empty() && (int)inflight.size() < inFlight) {
          job = extractQ.front();
          extractQ.pop_front();
        }
      }
      if (job >= 0) {
        Slot& s = slots[job];
        orbfe_extractor* h = ext[nextExt];
        const double ta = nowMs();
        if (gpuMatch && s.window > 0)
          s.status = orbfe_extract_batch_submit_matched(h, chain, batch, s.frames.data(), s.onDevice, s.rows, s.cols, s.stride,
                                                        s.bounds, s.window, s.nnratio, s.checkOri);
        else
          s.status = orbfe_extract_batch_submit(h, batch, s.frames.data(), s.onDevice, s.rows, s.cols, s.stride);
        {
          std::lock_guard<std::mutex> lk(mu);
          busySubmit += nowMs() - ta;
        }
        if (s.status != ORBFE_OK) s.err = orbfe_last_error();
        inflight.emplace_back(job, nextExt);
        nextExt = (nextExt + 1) % (int)ext.size();
        continue;
      }
      if (inflight.empty()) continue;
      const int slot = inflight.front().first, e = inflight.front().second;
      inflight.pop_front();
      Slot& s = slots[slot];
      if (s.status == ORBFE_OK) {
        const double ta = nowMs();
        if (gpuMatch && s.window > 0)
          s.status = orbfe_extract_batch_collect_matched(ext[e], s.kps.data(), s.desc.data(), cap, s.n.data(), s.m12.data(),
                                                         s.nm.data());
        else
          s.status = orbfe_extract_batch_collect(ext[e], s.kps.data(), s.desc.data(), cap, s.n.data());
        if (s.status == ORBFE_OK && bow) {   // raw (leaf, node) pairs of the batch; vectors are assembled on demand
          s.bowLeaf.resize((size_t)batch * cap);
          s.bowNode.resize((size_t)batch * cap);
          for (int f = 0; f < batch && s.status == ORBFE_OK; f++) {
            int nb = 0;
            s.status = orbfe_extract_bow_raw(ext[e], f, s.bowLeaf.data() + (size_t)f * cap, s.bowNode.data() + (size_t)f * cap, cap, &nb);
          }
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          busyCollect += nowMs() - ta;
        }
        if (s.status != ORBFE_OK) s.err = orbfe_last_error();
      }
      if (s.status == ORBFE_OK) {
        s.prevN = isolated ? -1 : lastN;
        if (s.prevN >= 0) { s.prevKps = lastKps; s.prevDesc = lastDesc; }
        lastN = s.n[batch - 1];
        lastKps.assign(s.kps.begin() + (size_t)(batch - 1) * cap, s.kps.begin() + (size_t)(batch - 1) * cap + lastN);
        lastDesc.assign(s.desc.begin() + (size_t)(batch - 1) * cap * 32, s.desc.begin() + ((size_t)(batch - 1) * cap + lastN) * 32);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (gpuMatch || s.window <= 0) {   // matches (if any) came back with the batch: done
          nBatches++;
          s.done = true;
        } else {
          matchQ.push_back(slot);
        }
      }
      cv.notify_all();
    }
  }

  void matchLoop(int worker) {
    orbfe_matcher* matcher = matchers[worker];
    std::vector<const OrbfeKeyPoint*> k1, k2;
    std::vector<const uint8_t*> d1, d2;
    std::vector<int> n1, n2;
    std::vector<float*> prev;
    std::vector<int32_t*> m12;
    std::vector<int> nm;
    for (;;) {
      int slot = -1;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !matchQ.empty(); });
        if (matchQ.empty()) {
          if (stop) return;
          continue;
        }
        slot = matchQ.front();
        matchQ.pop_front();
      }
      Slot& s = slots[slot];
      const double tm0 = nowMs();
      if (s.status == ORBFE_OK && s.window > 0 && !gpuMatch) {
        const int window = s.window, checkOri = s.checkOri;
        const float nnratio = s.nnratio;
        const float* bounds = s.bounds;
        // pairs (predecessor, frame): Tracking::MonocularInitialization style, vbPrevMatched := F1 keypoints
        k1.clear(); k2.clear(); d1.clear(); d2.clear(); n1.clear(); n2.clear(); prev.clear(); m12.clear();
        std::vector<int> frameOfPair;
        for (int i = 0; i < batch; i++) {
          const OrbfeKeyPoint* pk;
          const uint8_t* pd;
          int pn;
          if (i == 0) {
            if (s.prevN < 0) {  // very first frame of the stream: no predecessor
              s.nm[0] = 0;
              std::fill(s.m12.begin(), s.m12.begin() + cap, -1);
              continue;
            }
            pk = s.prevKps.data(); pd = s.prevDesc.data(); pn = s.prevN;
          } else {
            pk = s.kps.data() + (size_t)(i - 1) * cap; pd = s.desc.data() + (size_t)(i - 1) * cap * 32; pn = s.n[i - 1];
          }
          float* pxy = s.prevxy.data() + (size_t)i * cap * 2;
          for (int j = 0; j < pn; j++) { pxy[2 * j] = pk[j].x; pxy[2 * j + 1] = pk[j].y; }
          k1.push_back(pk); d1.push_back(pd); n1.push_back(pn);
          k2.push_back(s.kps.data() + (size_t)i * cap); d2.push_back(s.desc.data() + (size_t)i * cap * 32); n2.push_back(s.n[i]);
          prev.push_back(pxy);
          m12.push_back(s.m12.data() + (size_t)i * cap);
          frameOfPair.push_back(i);
        }
        nm.assign(k1.size(), 0);
        if (!k1.empty()) {
          s.status = orbfe_search_for_initialization_batch(matcher, (int)k1.size(), k1.data(), d1.data(), n1.data(), k2.data(),
                                                           d2.data(), n2.data(), bounds, prev.data(), m12.data(), window,
                                                           nnratio, checkOri, nm.data());
          if (s.status != ORBFE_OK) s.err = orbfe_last_error();
          for (size_t p = 0; p < frameOfPair.size(); p++) s.nm[frameOfPair[p]] = nm[p];
        }
      }
      const double tmBusy = nowMs() - tm0;
      {
        std::lock_guard<std::mutex> lk(mu);
        busyMatch += tmBusy;
        nBatches++;
        s.done = true;
      }
      cv.notify_all();
    }
  }
};

extern "C" {

int orbfe_stream_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id,
                        int batch, int depth, orbfe_stream** out) {
  if (!out || batch < 1 || depth < 1 || depth > 8) { set_err("invalid stream parameters"); return ORBFE_ERR_INVALID; }
  *out = nullptr;
  orbfe_stream* s = new orbfe_stream();
  s->batch = batch;
  s->depth = depth;
  s->device = device_id;
  // depth handles for the batches on the GPU (a spare one for the batch being assembled on the host measured slower: DESIGN_NOTES.md E.5)
  const int nhandles = depth;
  for (int d = 0; d < nhandles; d++) {
    orbfe_extractor* h = nullptr;
    int rc = orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id, &h);
    if (rc == ORBFE_OK && !getenv("ORBFE_POLL_WAIT_US")) rc = orbfe_extractor_set_wait_mode(h, 50);   // the runner pipelines: do not burn a core on waiting
    if (rc != ORBFE_OK) {
      for (auto* e : s->ext) orbfe_extractor_destroy(e);
      delete s;
      return rc;
    }
    s->ext.push_back(h);
  }
  if (const char* hv = getenv("ORBFE_STREAM_HOST_MATCH")) s->gpuMatch = atoi(hv) == 0;
  // host-side matching path only: SearchForInitialization workers, each with its own matcher (and HIP stream; streams
  // are not created unless used, because the runtime folds them onto a handful of hardware queues)
  int nMatch = 0;
  if (!s->gpuMatch) {
    nMatch = 2;
  }
  for (int w = 0; w < nMatch; w++) {
    orbfe_matcher* mm = nullptr;
    int rc = orbfe_matcher_create(device_id, &mm);
    if (rc != ORBFE_OK) {
      for (auto* e : s->ext) orbfe_extractor_destroy(e);
      for (auto* q : s->matchers) orbfe_matcher_destroy(q);
      delete s;
      return rc;
    }
    s->matchers.push_back(mm);
  }
  {
    int rc = orbfe_sfi_chain_create(s->ext[0], &s->chain);
    if (rc != ORBFE_OK) {
      for (auto* e : s->ext) orbfe_extractor_destroy(e);
      for (auto* q : s->matchers) orbfe_matcher_destroy(q);
      delete s;
      return rc;
    }
  }
  // `depth` batches in flight run on `depth` HIP streams, and the HIP runtime folds a process's streams onto GPU_MAX_HW_QUEUES hardware
  // queues (4 unless the variable said otherwise BEFORE the runtime started -- a library cannot set it for the process it is loaded
  // into); two streams on one queue serialise, and a fourth batch that waits behind the first on its queue costs more than it
  // overlaps (87.9 k frames/s at depth 4 on 4 queues against 89.8 k at depth 3; 93.0 k at depth 4 on 8 queues: gpurun r06j,
  // DESIGN.md s6).  So the runner MEASURES how many of its streams run side by side and keeps that many batches in flight.
  s->inFlight = depth;
  if (depth > 1) {
    const int k = orbfe_concurrent_streams(s->ext.data(), depth);
    if (k < 0) {
      for (auto* e : s->ext) orbfe_extractor_destroy(e);
      for (auto* q : s->matchers) orbfe_matcher_destroy(q);
      orbfe_sfi_chain_destroy(s->chain);
      delete s;
      return k;
    }
    s->inFlight = k;
    while ((int)s->ext.size() > k) {   // the handles whose streams share a queue are not used (nor their memory kept)
      orbfe_extractor_destroy(s->ext.back());
      s->ext.pop_back();
    }
    static bool said = false;
    if (k < depth && !said && !getenv("ORBFE_QUIET")) {
      said = true;
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      fprintf(stderr, "liborbfe: orbfe_stream_create(depth = %d): only %d of the runner's streams run side by side in this process (GPU_MAX_HW_QUEUES=%s), "
                      "so %d batches are kept in flight; export GPU_MAX_HW_QUEUES=8 before the process starts for the last 3 %%\n",
              depth, k, q ? q : "unset: 4", k);
    }
  }
  s->cap = orbfe_extractor_max_keypoints(s->ext[0]);
  // result slots: `depth` batches on the GPU, one in the caller's hands, a few queued in front of the worker.  A caller
  // whose own thread can be held up for milliseconds (bench.py's Python driver) asks for a deeper queue with
  // orbfe_stream_set_queue_slots; a slot is host memory only (4 MB at 1080p / 2000 features / 32 frames).
  s->growSlots(depth + 4);
  s->tExtract = std::thread([s] { pthread_setname_np(pthread_self(), "orbfe-runner"); s->extractLoop(); });
  for (int w = 0; w < nMatch; w++) s->tMatch.emplace_back([s, w] { pthread_setname_np(pthread_self(), "orbfe-match"); s->matchLoop(w); });
  *out = s;
  return ORBFE_OK;
}

void orbfe_stream_destroy(orbfe_stream* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stop = true;
  }
  s->cv.notify_all();
  if (s->tExtract.joinable()) s->tExtract.join();
  for (auto& t : s->tMatch) if (t.joinable()) t.join();
  for (auto* e : s->ext) orbfe_extractor_destroy(e);
  for (auto* q : s->matchers) orbfe_matcher_destroy(q);
  orbfe_sfi_chain_destroy(s->chain);
  delete s;
}

int orbfe_stream_set_matching(orbfe_stream* s, const float bounds[4], int window_size, float nnratio,
                              int check_orientation) {
  if (!s || (window_size > 0 && !bounds)) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  if (bounds) memcpy(s->bounds, bounds, sizeof s->bounds);
  s->window = window_size;
  s->nnratio = nnratio;
  s->checkOri = check_orientation;
  return ORBFE_OK;
}

int orbfe_stream_set_isolated_batches(orbfe_stream* s, int isolated) {
  if (!s) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  s->isolated = isolated != 0;
  s->lastN = -1;
  return orbfe_sfi_chain_set_isolated(s->chain, isolated);
}

int orbfe_stream_set_input_format(orbfe_stream* s, int format, int gray_variant) {
  if (!s) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  for (orbfe_extractor* e : s->ext) {
    const int rc = orbfe_extractor_set_input_format(e, format, gray_variant);
    if (rc) return rc;
  }
  s->channels = format == ORBFE_INPUT_GRAY8 ? 1 : (format == ORBFE_INPUT_RGB8 || format == ORBFE_INPUT_BGR8) ? 3 : 4;
  return ORBFE_OK;
}

int orbfe_stream_set_blur_variant(orbfe_stream* s, int variant) {
  if (!s) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  for (orbfe_extractor* e : s->ext) {
    const int rc = orbfe_extractor_set_blur_variant(e, variant);
    if (rc) return rc;
  }
  return ORBFE_OK;
}

int orbfe_stream_set_vocabulary(orbfe_stream* s, orbfe_vocabulary* v, int levelsup) {
  if (!s) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  for (orbfe_extractor* e : s->ext) {
    const int rc = orbfe_extractor_set_vocabulary(e, v, levelsup);
    if (rc) return rc;
  }
  s->bow = v != nullptr;
  return ORBFE_OK;
}

int orbfe_stream_bow_raw(orbfe_stream* s, int frame, const uint32_t** leaf_node, const uint32_t** level_node, int* n) {
  if (!s || !leaf_node || !level_node || !n || frame < 0 || frame >= s->batch) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  if (s->popped < 0 || !s->bow || s->slots[s->popped].bowLeaf.empty()) {
    set_err("no popped batch with bag-of-words results");
    return ORBFE_ERR_INVALID;
  }
  const Slot& sl = s->slots[s->popped];
  *leaf_node = sl.bowLeaf.data() + (size_t)frame * s->cap;
  *level_node = sl.bowNode.data() + (size_t)frame * s->cap;
  *n = sl.n[frame];
  return ORBFE_OK;
}

int orbfe_stream_capacity(const orbfe_stream* s) { return s ? s->cap : 0; }
int orbfe_stream_batches_in_flight(const orbfe_stream* s) { return s ? s->inFlight : 0; }

int orbfe_stream_set_queue_slots(orbfe_stream* s, int nslots) {
  if (!s) { set_err("stream is NULL"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->popSeq != s->pushSeq) { set_err("orbfe_stream_set_queue_slots: batches in flight"); return ORBFE_ERR_INVALID; }
  if (nslots < s->depth + 2 || nslots > 256) { set_err("orbfe_stream_set_queue_slots: need depth+2 <= nslots <= 256"); return ORBFE_ERR_INVALID; }
  s->growSlots(nslots);
  return ORBFE_OK;
}
int orbfe_stream_queue_slots(const orbfe_stream* s) { return s ? (int)s->slots.size() : 0; }

int orbfe_stream_push(orbfe_stream* s, const uint8_t* const* gray, int in_device_memory, int rows, int cols,
                      size_t stride_bytes) {
  if (!s || !gray || rows <= 0 || cols <= 0 || stride_bytes < (size_t)cols * s->channels) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  // a geometry whose levels return more keypoints than the slots hold (strips wider than 4.5 : 1: every root node of a level
  // yields keypoints, ORBextractor.cc:620-700): grow every slot -- only while the pipeline is empty, the row stride changes
  const int need = orbfe_extractor_max_keypoints_for_size(s->ext[0], rows, cols);
  if (need > s->cap) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->popSeq != s->pushSeq) {
      set_err("frames of %dx%d need %d keypoint slots per frame, the runner holds %d: pop every pushed batch before changing the frame size", cols, rows, need, s->cap);
      return ORBFE_ERR_INVALID;
    }
    s->cap = need;
    for (Slot& sl : s->slots) {
      sl.kps.resize((size_t)s->batch * need);
      sl.desc.resize((size_t)s->batch * need * 32);
      sl.m12.assign((size_t)s->batch * need, -1);
      sl.prevxy.resize((size_t)s->batch * need * 2);
    }
  }
  int slot;
  {
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return !s->freeQ.empty(); });
    slot = s->freeQ.front();
    s->freeQ.pop_front();
  }
  Slot& sl = s->slots[slot];
  for (int i = 0; i < s->batch; i++) sl.frames[i] = gray[i];
  sl.rows = rows; sl.cols = cols; sl.stride = stride_bytes; sl.onDevice = in_device_memory;
  sl.status = ORBFE_OK;
  sl.done = false;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    sl.window = s->window; sl.checkOri = s->checkOri; sl.nnratio = s->nnratio;
    memcpy(sl.bounds, s->bounds, sizeof sl.bounds);
    sl.seq = s->pushSeq++;
    s->extractQ.push_back(slot);
  }
  s->cv.notify_all();
  return ORBFE_OK;
}

int orbfe_stream_pop(orbfe_stream* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps,
                     const int32_t** matches12, const int** nmatches) {
  if (!s) { set_err("stream is NULL"); return ORBFE_ERR_INVALID; }
  int slot;
  {
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->popped >= 0) {
      s->freeQ.push_back(s->popped);
      s->popped = -1;
      s->cv.notify_all();
    }
    if (s->popSeq == s->pushSeq) { set_err("no batch outstanding (every pushed batch has been popped)"); return ORBFE_ERR_INVALID; }
    auto ready = [&]() -> int {
      for (size_t i = 0; i < s->slots.size(); i++)
        if (s->slots[i].done && s->slots[i].seq == s->popSeq) return (int)i;
      return -1;
    };
    s->cv.wait(lk, [&] { return ready() >= 0; });
    slot = ready();
    s->slots[slot].done = false;
    s->popSeq++;
    s->popped = slot;
  }
  Slot& sl = s->slots[slot];
  if (kps) *kps = sl.kps.data();
  if (desc) *desc = sl.desc.data();
  if (n_kps) *n_kps = sl.n.data();
  if (matches12) *matches12 = sl.m12.data();
  if (nmatches) *nmatches = sl.nm.data();
  if (sl.status != ORBFE_OK) set_err("%s", sl.err.c_str());
  return sl.status;
}

// pop WITHOUT the "valid until the next pop" rule: the result stays where it is until orbfe_stream_release(s, *ticket) -- any number of
// results may be held, each keeps one result slot of the runner busy (orbfe_stream_queue_slots).  For a consumer that hands results on
// to another thread (orbfe_stream_multi's finishers) without copying them.
int orbfe_stream_pop_hold(orbfe_stream* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps, const int32_t** matches12,
                          const int** nmatches, int* ticket) {
  if (!s || !ticket) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  int slot;
  {
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->popSeq == s->pushSeq) { set_err("no batch outstanding (every pushed batch has been popped)"); return ORBFE_ERR_INVALID; }
    auto ready = [&]() -> int {
      for (size_t i = 0; i < s->slots.size(); i++)
        if (s->slots[i].done && s->slots[i].seq == s->popSeq) return (int)i;
      return -1;
    };
    s->cv.wait(lk, [&] { return ready() >= 0; });
    slot = ready();
    s->slots[slot].done = false;
    s->popSeq++;
  }
  *ticket = slot;
  Slot& sl = s->slots[slot];
  if (kps) *kps = sl.kps.data();
  if (desc) *desc = sl.desc.data();
  if (n_kps) *n_kps = sl.n.data();
  if (matches12) *matches12 = sl.m12.data();
  if (nmatches) *nmatches = sl.nm.data();
  if (sl.status != ORBFE_OK) set_err("%s", sl.err.c_str());
  return sl.status;
}

int orbfe_stream_release(orbfe_stream* s, int ticket) {
  if (!s || ticket < 0) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    if (ticket >= (int)s->slots.size()) { set_err("not a ticket of this runner"); return ORBFE_ERR_INVALID; }
    s->freeQ.push_back(ticket);
  }
  s->cv.notify_all();
  return ORBFE_OK;
}

int orbfe_stream_stats(orbfe_stream* s, double out[4], int reset) {
  if (!s || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(s->mu);
  out[0] = s->busySubmit; out[1] = s->busyCollect; out[2] = s->busyMatch; out[3] = (double)s->nBatches;
  if (reset) { s->busySubmit = s->busyCollect = s->busyMatch = 0; s->nBatches = 0; }
  return ORBFE_OK;
}

int orbfe_stream_kernel_ms(orbfe_stream* s, double out_ms[5], long long* batches, long long* frames, int reset) {
  if (!s) { set_err("stream is NULL"); return ORBFE_ERR_INVALID; }
  double acc[5] = {0, 0, 0, 0, 0};
  long long b = 0, f = 0;
  for (auto* e : s->ext) {
    double ms[5];
    long long bb = 0, ff = 0;
    orbfe_debug_kernel_ms(e, ms, &bb, &ff, reset);
    for (int i = 0; i < 5; i++) acc[i] += ms[i];
    b += bb;
    f += ff;
  }
  if (out_ms) for (int i = 0; i < 5; i++) out_ms[i] = acc[i];
  if (batches) *batches = b;
  if (frames) *frames = f;
  return ORBFE_OK;
}

}  // extern "C"
