// TEST-ONLY shim: HLS dataflow processes become host threads, joined at FINALIZE
// (what kernel/Top.cpp:67-116 relies on when HLSLIB_SYNTHESIS is undefined).
#pragma once
#include <thread>
#include <tuple>
#include <utility>
#include <vector>
namespace hlslib {
class _DataflowContext {
  std::vector<std::thread> threads_;

 public:
  // Arguments are converted to the callee's declared parameter types up front:
  // reference parameters bind to the caller's objects (Stream<T, d> -> Stream<T>&),
  // arrays decay to pointers, scalars are copied.
  template <typename Ret, typename... Params, typename... Passed>
  void AddFunction(Ret (*func)(Params...), Passed &&...passed) {
    std::tuple<Params...> bound(std::forward<Passed>(passed)...);
    threads_.emplace_back([func, bound]() mutable { std::apply(func, bound); });
  }
  void Join() {
    for (auto &t : threads_) t.join();
    threads_.clear();
  }
  ~_DataflowContext() { Join(); }
};
}  // namespace hlslib
#define HLSLIB_DATAFLOW_INIT() ::hlslib::_DataflowContext __hlslib_dataflow_context
#define HLSLIB_DATAFLOW_FUNCTION(func, ...) __hlslib_dataflow_context.AddFunction(func, __VA_ARGS__)
#define HLSLIB_DATAFLOW_FINALIZE() __hlslib_dataflow_context.Join()
