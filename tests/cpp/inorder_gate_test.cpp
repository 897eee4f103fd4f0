// inorder_gate_test.cpp -- the in-order completion queue of the multi-device stream runner (os1_amd/csrc/inorder_gate.h) under SHUFFLED
// completion: `lanes` producers finish their batches after random delays (later sequence numbers routinely before earlier ones), one
// consumer takes them.  Checked: the consumer sees 0, 1, 2, ... ; a producer never recycles the storage behind a sequence number
// before the consumer has let go of it; closing the gate releases everybody.  Host only (g++ -pthread), no GPU.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "inorder_gate.h"

int main(int argc, char** argv) {
  const int lanes = argc > 1 ? atoi(argv[1]) : 4, total = argc > 2 ? atoi(argv[2]) : 400;
  orbfe::InOrderGate gate(lanes);
  std::vector<long long> storage(lanes, -1);          // what lane d currently exposes (the "slot" behind its published batch)
  std::atomic<long long> pushed{0};
  std::atomic<int> bad{0};
  std::atomic<long long> outOfOrderCompletions{0}, lastCompleted{-1};
  std::vector<std::thread> prod;
  for (int d = 0; d < lanes; d++)
    prod.emplace_back([&, d] {
      std::mt19937 rng(1234 + d);
      for (long long k = d; k < total; k += lanes) {
        while (pushed.load() <= k) std::this_thread::yield();
        std::this_thread::sleep_for(std::chrono::microseconds(rng() % 400));   // the "GPU": lanes finish in any order
        if (lastCompleted.exchange(k) > k) outOfOrderCompletions++;
        storage[d] = k;
        gate.publish(k);
        if (!gate.wait_released(k)) return;
        storage[d] = -2;                                                         // recycled: nobody may be looking at it now
      }
    });
  std::mt19937 rng(99);
  long long got = 0;
  for (long long k = 0; k < total; k++) {
    pushed = std::min<long long>(total, k + 2 * lanes);                           // the caller keeps a few batches in flight
    const long long s = gate.take();
    if (s != k) bad++;
    if (storage[s % lanes] != s) bad++;                                            // still the batch we were handed
    std::this_thread::sleep_for(std::chrono::microseconds(rng() % 60));           // the consumer works on it ...
    if (storage[s % lanes] != s) bad++;                                            // ... and it is still there
    got++;
  }
  gate.release_held();
  gate.close();
  for (auto& t : prod) t.join();
  // a second gate: closing while producers wait and the consumer takes releases everybody
  orbfe::InOrderGate g2(2);
  std::thread waiter([&] { if (g2.take() != -1) bad++; });
  std::thread parked([&] { g2.publish(1); if (g2.wait_released(1)) bad++; });
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  g2.close();
  waiter.join();
  parked.join();
  printf("taken %lld of %d in order, %lld completions arrived out of order, bad %d\n", got, total, outOfOrderCompletions.load(), bad.load());
  return bad.load() == 0 && got == total && (lanes == 1 || outOfOrderCompletions.load() > 0) ? 0 : 1;
}
