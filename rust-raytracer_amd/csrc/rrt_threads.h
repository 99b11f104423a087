// Host threads that never let an exception escape: the C ABI promises return codes, and an
// exception that unwinds past a joinable std::thread ends the process.
#pragma once
#include <atomic>
#include <system_error>
#include <thread>
#include <utility>

namespace rrt {

// Test hook (rrt_debug_fail_threads_after): thread creations left before every further one is
// refused as if the system had run out of threads; negative = never.
extern std::atomic<int> g_thread_creations_allowed;

inline bool thread_creation_refused() {
  int left = g_thread_creations_allowed.load(std::memory_order_relaxed);
  while (left >= 0) {
    if (left == 0) return true;
    if (g_thread_creations_allowed.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) return false;
  }
  return false;
}

// Starts fn() on a new thread.  False when no thread could be had: the caller runs fn() itself.
// fn must not throw (wrap its body; see guarded()).
template <class Fn>
bool try_spawn(std::thread* t, const Fn& fn) {
  try {
    if (thread_creation_refused()) return false;
    *t = std::thread(fn);
    return true;
  } catch (...) {  // std::system_error (no more threads), std::bad_alloc
    return false;
  }
}

// Joins what is still joinable when the scope is left, normally or by an exception.
struct JoinGuard {
  std::thread* threads;
  int n;
  ~JoinGuard() {
    for (int i = 0; i < n; ++i)
      if (threads[i].joinable()) threads[i].join();
  }
};

// body() with every exception turned into a flag the owner of the threads looks at afterwards.
template <class Body>
void guarded(std::atomic<bool>* failed, const Body& body) {
  try {
    body();
  } catch (...) {
    failed->store(true, std::memory_order_relaxed);
  }
}

}  // namespace rrt
