// inorder_gate.h -- in-order hand-over of results that complete out of order (host only, no HIP).
//
// SURVEY.md s8(e), the single-stream shape: "round-robin frames over GPUs with an in-order completion queue".  The reference has
// ONE camera stream and ONE consumer (main.cc:113-141 feeds System::TrackMonocular frame by frame, System.cc:115-152): whatever
// order the devices finish in, the consumer must see batch k before batch k + 1.
//
// `lanes` producers (one per device runner) own the sequence numbers d, d + lanes, d + 2 lanes, ... .  A producer publishes a
// sequence number when its result is complete (in order within its lane; it may run ahead of the consumer) and either waits until the
// consumer has let go of it before it recycles the storage behind it (wait_released) or leaves the recycling to the consumer; the consumer
// takes sequence numbers strictly in order, and taking k + 1 is what lets go of k.
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

namespace orbfe {

class InOrderGate {
 public:
  explicit InOrderGate(int lanes) : lanes_(lanes < 1 ? 1 : lanes), ready_(lanes_, -1) {}

  int lanes() const { return lanes_; }

  // producer of lane seq % lanes: result `seq` is complete
  void publish(long long seq) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      ready_[(size_t)(seq % lanes_)] = seq;
    }
    cv_.notify_all();
  }

  // producer: block until the consumer has let go of `seq`; false = the gate was closed first
  bool wait_released(long long seq) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return closed_ || released_ >= seq; });
    return released_ >= seq;
  }

  // consumer: lets go of what it holds, then blocks until the next sequence number is published; -1 = closed
  long long take() {
    std::unique_lock<std::mutex> lk(mu_);
    if (held_ >= 0) {
      released_ = held_;
      held_ = -1;
      cv_.notify_all();
    }
    const long long want = next_;
    // (a lane publishes its sequence numbers in order, so "the lane has reached `want`" is >=: a producer may run ahead of the consumer)
    cv_.wait(lk, [&] { return closed_ || ready_[(size_t)(want % lanes_)] >= want; });
    if (ready_[(size_t)(want % lanes_)] < want) return -1;
    held_ = want;
    next_ = want + 1;
    return want;
  }

  // consumer: lets go of what it holds without taking anything (before shutting down, or before re-configuring an idle pipeline)
  void release_held() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (held_ >= 0) { released_ = held_; held_ = -1; }
    }
    cv_.notify_all();
  }

  long long next() const {
    std::lock_guard<std::mutex> lk(mu_);
    return next_;
  }

  void close() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      closed_ = true;
    }
    cv_.notify_all();
  }

 private:
  const int lanes_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::vector<long long> ready_;   // per lane: the sequence number whose result is complete (-1: none yet)
  long long next_ = 0;             // the sequence number the consumer takes next
  long long held_ = -1;            // ... and the one it holds
  long long released_ = -1;        // everything up to here has been let go of
  bool closed_ = false;
};

}  // namespace orbfe
