// TEST-ONLY shim: blocking bounded FIFO between exactly one producer thread and one
// consumer thread (every stream in kernel/Top.cpp has a single writer and a single reader).
// Stream<T, depth> derives from Stream<T> (= Stream<T, 0>) WITHOUT adding members, so arrays of
// Stream<T, depth> may be handed to functions declared with Stream<T>[] / Stream<T>& exactly as
// kernel/Top.cpp:69-110 does.
#pragma once
#include <atomic>
#include <chrono>
#include <iostream>  // hlslib's Stream.h provides it transitively (kernel/Memory.cpp:384 uses std::cout)
#include <cstddef>
#include <memory>
#include <string>
#include <thread>
namespace hlslib {

enum class Storage { Unspecified, BRAM, LUTRAM, SRL };

template <typename T, unsigned depth = 0, Storage storage = Storage::Unspecified>
class Stream;

template <typename T>
class Stream<T, 0, Storage::Unspecified> {
 public:
  // A larger software FIFO than the declared hardware depth cannot introduce a deadlock into
  // a process network that is deadlock-free at the declared depth; it only reduces blocking.
  static constexpr std::size_t kMinCapacity = 1024;

  Stream() : Stream(nullptr, 0) {}
  explicit Stream(char const *name) : Stream(name, 0) {}
  Stream(char const *name, std::size_t cap) : name_(name ? name : "") { Reset(cap); }
  Stream(Stream const &) = delete;
  Stream &operator=(Stream const &) = delete;

  void set_name(char const *name) { name_ = name ? name : ""; }
  std::string const &name() const { return name_; }

  void Push(T const &v) {
    const std::size_t t = tail_.load(std::memory_order_relaxed);
    Wait([&] { return t - head_.load(std::memory_order_acquire) < cap_; });
    buf_[t % cap_] = v;
    tail_.store(t + 1, std::memory_order_release);
  }
  T Pop() {
    const std::size_t h = head_.load(std::memory_order_relaxed);
    Wait([&] { return tail_.load(std::memory_order_acquire) != h; });
    T v = buf_[h % cap_];
    head_.store(h + 1, std::memory_order_release);
    return v;
  }
  // hlslib spells these too
  void WriteBlocking(T const &v) { Push(v); }
  T ReadBlocking() { return Pop(); }
  bool IsEmpty() const { return tail_.load() == head_.load(); }
  std::size_t Size() const { return tail_.load() - head_.load(); }

 protected:
  void Reset(std::size_t cap) {
    cap_ = cap < kMinCapacity ? kMinCapacity : cap;
    buf_.reset(new T[cap_]);
  }

 private:
  template <typename Pred>
  static void Wait(Pred ready) {
    for (unsigned spin = 0; !ready(); ++spin) {
      if (spin < 64) continue;
      if (spin < 4096) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  std::string name_;
  std::size_t cap_ = 0;
  std::unique_ptr<T[]> buf_;
  alignas(64) std::atomic<std::size_t> head_{0};
  alignas(64) std::atomic<std::size_t> tail_{0};
};

template <typename T, unsigned depth, Storage storage>
class Stream : public Stream<T, 0, Storage::Unspecified> {
  using Base = Stream<T, 0, Storage::Unspecified>;

 public:
  Stream() : Base(nullptr, depth) {}
  explicit Stream(char const *name) : Base(name, depth) {}
};

}  // namespace hlslib
