/* Shim for dmlc-core's <dmlc/thread_local.h> (empty submodule). */
#pragma once
namespace dmlc {
template <typename T>
class ThreadLocalStore {
 public:
  static T* Get() {
    static thread_local T inst;
    return &inst;
  }
};
}  // namespace dmlc
