// orbfe_stream_multi.cpp -- ONE camera stream over several GPUs (include/orbfe.h, orbfe_stream_multi_*).
//
// SURVEY.md s8(e) names two ways the path shards: stream g -> GPU g (orbfe_stream, one per process and device) and, for a single
// stream, "round-robin frames over GPUs with an in-order completion queue" -- the shape the reference itself has: one Video thread,
// one Tracking thread, frames strictly in order (/root/reference/src/main.cc:113-141, System.cc:115-152).  This runner is that second
// shape.  Batch k of the stream goes to device k mod n -- each device has an ordinary single-device runner (orbfe_stream, `depth`
// batches in flight) whose batches stand alone -- and results are handed out strictly in push order whatever order the devices
// finish in (InOrderGate).  There is NO collective and no device-to-device traffic; the one piece of state that crosses a batch
// boundary, the predecessor of a batch's first frame for SearchForInitialization (ORBmatcher.cc:400-515, Tracking.cc:355-357), is the
// last frame of the previous batch -- which lives on ANOTHER device -- and takes a host bounce: the finisher thread of the device that
// popped batch k - 1 publishes that frame's keypoints and descriptors, the finisher of batch k's device matches its first frame
// against them with the host-array search (orbfe_search_for_initialization: the same kernels) and writes row 0 of the batch's match
// vectors.  63 of a 64-frame batch's pairs never leave their GPU.  A finisher never waits for the consumer: it holds a finished batch's
// result slot (orbfe_stream_pop_hold) and goes on to its device's next batch; the consumer gives the slot back when it lets go.
#include <pthread.h>

#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"
#include "inorder_gate.h"

namespace orbfe {
void set_err(const char* fmt, ...);
}
using orbfe::set_err;

struct orbfe_stream_multi {
  int ndev = 0, batch = 0;
  std::vector<int> devices;
  std::vector<orbfe_stream*> sub;        // one single-device runner per entry of device_ids, batches isolated
  std::vector<orbfe_matcher*> matcher;   // the boundary pairs of a device's batches
  std::vector<std::thread> finisher;
  std::unique_ptr<orbfe::InOrderGate> gate;

  std::mutex mu;
  std::condition_variable cv;
  long long pushSeq = 0, popSeq = 0;
  bool stop = false;
  int window = 100, checkOri = 1;
  float nnratio = 0.9f;
  float bounds[4] = {0, 0, 0, 0};

  // what finisher d hands to the consumer (valid from publish(seq) until the consumer lets go of seq)
  struct Done {
    const OrbfeKeyPoint* kps = nullptr;
    const uint8_t* desc = nullptr;
    const int* n = nullptr;
    const int32_t* m12 = nullptr;
    const int* nm = nullptr;
    int status = ORBFE_OK;
    int ticket = -1;            // orbfe_stream_pop_hold's: the consumer releases the result slot when it lets go of the batch
    std::string err;
  };
  static constexpr int kRing = 256;   // >= result slots of a device runner (orbfe_stream_set_queue_slots allows at most 256)
  std::vector<Done> done;             // [ndev][kRing]: a device's finished batches wait here for their turn
  Done& doneOf(long long seq) { return done[(size_t)(seq % ndev) * kRing + (size_t)((seq / ndev) % kRing)]; }
  long long heldSeq = -1;             // the batch in the consumer's hands
  // the last frame of the batches a device has popped: two generations per device, so that with one device batch k reads what
  // batch k - 1 left while it writes its own
  struct Tail {
    std::vector<OrbfeKeyPoint> kps;
    std::vector<uint8_t> desc;
    int n = -1;
    long long seq = -1;
  };
  std::vector<Tail> tail;   // [ndev][2]
  std::vector<long long> tailRead;   // [ndev]: the latest batch of the device whose last frame the next batch's finisher has taken
  Tail& tailOf(long long seq) { return tail[(size_t)(seq % ndev) * 2 + (size_t)((seq / ndev) & 1)]; }

  void finish(int d) {
    std::vector<float> prevxy;
    std::vector<int32_t> row;
    Tail prev;
    for (long long k = d;; k += ndev) {
      int window_k, checkOri_k;
      float nnratio_k, bounds_k[4];
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || pushSeq > k; });
        if (pushSeq <= k) return;   // stopped with nothing of ours outstanding
        window_k = window; checkOri_k = checkOri; nnratio_k = nnratio;
        memcpy(bounds_k, bounds, sizeof bounds_k);
      }
      Done r;
      // (held, not "valid until the next pop": this thread goes on to the device's next batch while the consumer still reads this one)
      r.status = orbfe_stream_pop_hold(sub[d], &r.kps, &r.desc, &r.n, &r.m12, &r.nm, &r.ticket);
      if (r.status != ORBFE_OK) r.err = orbfe_last_error();
      const int stride = orbfe_stream_capacity(sub[d]);   // keypoint slots per frame of the result arrays (grows only while the runner is idle)
      const bool matching = window_k > 0;
      // 1. the next batch's predecessor, before anything else: its finisher may be waiting for it.  Two generations per device: this one
      //    overwrites batch k - 2 n's, which the finisher of batch k - 2 n + 1 must have read (a device may run several batches ahead of
      //    its neighbour)
      {   // (published and taken whether or not this phase of the stream matches: the hand-over stays in step when matching is switched on)
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || k < 2 * (long long)ndev || tailRead[(size_t)d] >= k - 2 * (long long)ndev; });
        if (stop && !(k < 2 * (long long)ndev || tailRead[(size_t)d] >= k - 2 * (long long)ndev)) return;
        Tail& t = tailOf(k);
        t.n = -1;
        if (r.status == ORBFE_OK) {
          const int last = batch - 1, ln = r.n[last];
          t.kps.assign(r.kps + (size_t)last * stride, r.kps + (size_t)last * stride + ln);
          t.desc.assign(r.desc + (size_t)last * stride * 32, r.desc + ((size_t)last * stride + ln) * 32);
          t.n = ln;
        }
        t.seq = k;
      }
      cv.notify_all();
      // 2. this batch's first frame against ITS predecessor (batch k - 1's last frame, popped on another device)
      if (k > 0) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || tailOf(k - 1).seq == k - 1; });
          if (tailOf(k - 1).seq != k - 1) return;
          if (matching) prev = tailOf(k - 1);   // (a copy: the search runs outside the lock)
          tailRead[(size_t)((k - 1) % ndev)] = k - 1;
        }
        cv.notify_all();
        if (matching && r.status == ORBFE_OK && prev.n >= 0) {
          // the runner owns the slot memory behind the pointers orbfe_stream_pop_hold returned: row 0 of the match vectors is written in place
          int32_t* m12 = const_cast<int32_t*>(r.m12);
          int* nm = const_cast<int*>(r.nm);
          prevxy.resize((size_t)(prev.n > 0 ? prev.n : 1) * 2);
          for (int j = 0; j < prev.n; j++) { prevxy[2 * j] = prev.kps[j].x; prevxy[2 * j + 1] = prev.kps[j].y; }   // vbPrevMatched := F1's keypoints (Tracking.cc:355-357)
          row.assign((size_t)(prev.n > 0 ? prev.n : 1), -1);
          int nmatch = 0;
          const int rc = orbfe_search_for_initialization(matcher[d], prev.kps.data(), prev.desc.data(), prev.n, r.kps, r.desc, r.n[0], bounds_k,
                                                         prevxy.data(), row.data(), window_k, nnratio_k, checkOri_k, &nmatch);
          if (rc != ORBFE_OK) {
            r.status = rc;
            r.err = orbfe_last_error();
          } else {
            const int m = prev.n < stride ? prev.n : stride;
            memcpy(m12, row.data(), sizeof(int32_t) * (size_t)m);
            for (int i = m; i < stride; i++) m12[i] = -1;
            nm[0] = nmatch;
          }
        }
      }
      doneOf(k) = r;
      gate->publish(k);   // 3. the consumer releases the runner's result slot when it lets go of the batch (orbfe_stream_multi_pop)
    }
  }
};

// the consumer lets go of the batch it holds: its runner gets the result slot back
static void release_held_batch(orbfe_stream_multi* s) {
  if (s->heldSeq >= 0) {
    const orbfe_stream_multi::Done& r = s->doneOf(s->heldSeq);
    if (r.ticket >= 0) (void)orbfe_stream_release(s->sub[(size_t)(s->heldSeq % s->ndev)], r.ticket);
    s->heldSeq = -1;
  }
  s->gate->release_held();
}

extern "C" {

int orbfe_stream_multi_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, const int* device_ids,
                              int n_devices, int batch, int depth, orbfe_stream_multi** out) {
  if (!out || !device_ids || n_devices < 1 || n_devices > 64 || batch < 1 || depth < 1 || depth > 8) {
    set_err("invalid multi-device stream parameters");
    return ORBFE_ERR_INVALID;
  }
  *out = nullptr;
  std::unique_ptr<orbfe_stream_multi> s(new orbfe_stream_multi());
  s->ndev = n_devices;
  s->batch = batch;
  s->devices.assign(device_ids, device_ids + n_devices);
  s->done.resize((size_t)n_devices * orbfe_stream_multi::kRing);
  s->tail.resize((size_t)n_devices * 2);
  s->tailRead.assign((size_t)n_devices, -1);
  s->gate.reset(new orbfe::InOrderGate(n_devices));
  int rc = ORBFE_OK;
  std::string err;
  for (int d = 0; d < n_devices && rc == ORBFE_OK; d++) {
    // NUMA: a device's runner is created from a thread bound to the CPUs of the device's node, so that the runner's worker thread
    // (which inherits the mask) and the page-locked arenas it allocates sit next to their GPU (the expected limiter of
    // a multi-GPU node is the host side); orbfe_bind_thread_to_device never widens a mask, hence one helper thread per device
    std::thread helper([&, d] {
      (void)orbfe_bind_thread_to_device(device_ids[d]);
      orbfe_stream* sub = nullptr;
      int r = orbfe_stream_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_ids[d], batch, depth, &sub);
      if (r == ORBFE_OK) {
        s->sub.push_back(sub);
        r = orbfe_stream_set_isolated_batches(sub, 1);
      }
      if (r == ORBFE_OK) {
        orbfe_matcher* m = nullptr;
        r = orbfe_matcher_create(device_ids[d], &m);
        if (r == ORBFE_OK) s->matcher.push_back(m);
      }
      if (r != ORBFE_OK) err = orbfe_last_error();   // (the message is thread-local: carry it to the caller's thread)
      rc = r;
    });
    helper.join();
  }
  if (rc != ORBFE_OK) {
    set_err("%s", err.c_str());
    for (auto* q : s->sub) orbfe_stream_destroy(q);
    for (auto* m : s->matcher) orbfe_matcher_destroy(m);
    return rc;
  }
  orbfe_stream_multi* p = s.release();
  for (int d = 0; d < n_devices; d++)
    p->finisher.emplace_back([p, d] {
      pthread_setname_np(pthread_self(), "orbfe-finish");
      (void)orbfe_bind_thread_to_device(p->devices[(size_t)d]);
      p->finish(d);
    });
  *out = p;
  return ORBFE_OK;
}

void orbfe_stream_multi_destroy(orbfe_stream_multi* s) {
  if (!s) return;
  // batches that were pushed and never popped are dropped: a finisher blocked in its sub-runner's pop returns when the GPU is done
  // with the batch, one parked behind a published batch returns when the gate closes
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stop = true;
  }
  s->cv.notify_all();
  release_held_batch(s);
  s->gate->close();
  for (auto& t : s->finisher) if (t.joinable()) t.join();
  for (auto* q : s->sub) orbfe_stream_destroy(q);
  for (auto* m : s->matcher) orbfe_matcher_destroy(m);
  delete s;
}

int orbfe_stream_multi_devices(const orbfe_stream_multi* s) { return s ? s->ndev : 0; }
int orbfe_stream_multi_capacity(const orbfe_stream_multi* s) { return s && !s->sub.empty() ? orbfe_stream_capacity(s->sub[0]) : 0; }
int orbfe_stream_multi_device_of_next_push(const orbfe_stream_multi* s) {
  if (!s) return -1;
  return s->devices[(size_t)(s->pushSeq % s->ndev)];
}

int orbfe_stream_multi_set_matching(orbfe_stream_multi* s, const float bounds[4], int window_size, float nnratio, int check_orientation) {
  if (!s || (window_size > 0 && !bounds)) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  }
  release_held_batch(s);   // (the batch the caller still holds)
  for (auto* q : s->sub) {
    const int rc = orbfe_stream_set_matching(q, bounds, window_size, nnratio, check_orientation);
    if (rc != ORBFE_OK) return rc;
  }
  std::lock_guard<std::mutex> lk(s->mu);
  if (bounds) memcpy(s->bounds, bounds, sizeof s->bounds);
  s->window = window_size;
  s->nnratio = nnratio;
  s->checkOri = check_orientation;
  return ORBFE_OK;
}

int orbfe_stream_multi_set_blur_variant(orbfe_stream_multi* s, int variant) {
  if (!s) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->pushSeq != s->popSeq) { set_err("batches are still in flight"); return ORBFE_ERR_INVALID; }
  }
  release_held_batch(s);
  for (auto* q : s->sub) {
    const int rc = orbfe_stream_set_blur_variant(q, variant);
    if (rc != ORBFE_OK) return rc;
  }
  return ORBFE_OK;
}

int orbfe_stream_multi_push(orbfe_stream_multi* s, const uint8_t* const* gray, int in_device_memory, int rows, int cols, size_t stride_bytes) {
  if (!s || !gray) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  long long k;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    k = s->pushSeq;
  }
  const int rc = orbfe_stream_push(s->sub[(size_t)(k % s->ndev)], gray, in_device_memory, rows, cols, stride_bytes);
  if (rc != ORBFE_OK) return rc;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->pushSeq = k + 1;
  }
  s->cv.notify_all();
  return ORBFE_OK;
}

int orbfe_stream_multi_pop(orbfe_stream_multi* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps, const int32_t** matches12,
                           const int** nmatches) {
  if (!s) { set_err("stream is NULL"); return ORBFE_ERR_INVALID; }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->popSeq == s->pushSeq) { set_err("no batch outstanding (every pushed batch has been popped)"); return ORBFE_ERR_INVALID; }
  }
  release_held_batch(s);
  const long long k = s->gate->take();
  if (k < 0) { set_err("the stream is shutting down"); return ORBFE_ERR_INVALID; }
  s->heldSeq = k;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->popSeq = k + 1;
  }
  const orbfe_stream_multi::Done& r = s->doneOf(k);
  if (kps) *kps = r.kps;
  if (desc) *desc = r.desc;
  if (n_kps) *n_kps = r.n;
  if (matches12) *matches12 = r.m12;
  if (nmatches) *nmatches = r.nm;
  if (r.status != ORBFE_OK) set_err("%s", r.err.c_str());
  return r.status;
}

}  // extern "C"
