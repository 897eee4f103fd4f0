// host_pool.h -- a small persistent worker pool for the host-side, per-(frame, level) quadtree
// tasks of a batch (they are independent; inside one level the algorithm is serial).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace orbfe {

class HostPool {
 public:
  explicit HostPool(int nthreads) : n_(nthreads < 1 ? 1 : nthreads) {
    for (int i = 1; i < n_; i++) workers_.emplace_back([this, i] { loop(i); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int size() const { return n_; }

  // Runs fn(task, worker) for task in [0, ntasks); the calling thread is worker 0.
  void parallelFor(int ntasks, const std::function<void(int, int)>& fn) {
    if (ntasks <= 0) return;
    if (n_ == 1 || ntasks == 1) {
      for (int t = 0; t < ntasks; t++) fn(t, 0);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      ntasks_ = ntasks;
      next_.store(0);
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    work(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work(int wid) {
    for (;;) {
      const int t = next_.fetch_add(1);
      if (t >= ntasks_) break;
      (*fn_)(t, wid);
    }
  }
  void loop(int wid) {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      work(wid);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int, int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int ntasks_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
};

}  // namespace orbfe
