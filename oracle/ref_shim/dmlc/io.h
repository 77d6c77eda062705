/* Shim for dmlc-core's <dmlc/io.h> (empty submodule in the reference checkout): the
 * abstract byte stream the reference's NDArray / CSRMatrix Save/Load members are declared
 * against.  The oracle never serialises anything; only the declarations are needed. */
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "./logging.h"

#define DMLC_IO_NO_ENDIAN_SWAP 1
#define DMLC_DECLARE_TRAITS(Trait, Type, Value) \
  template <>                                   \
  struct Trait<Type> {                          \
    static const bool value = Value;            \
  }

namespace dmlc {
template <typename T>
struct has_saveload {
  static const bool value = false;
};
template <typename T>
inline T* BeginPtr(std::vector<T>& vec) {
  return vec.empty() ? nullptr : &vec[0];
}
template <typename T>
inline const T* BeginPtr(const std::vector<T>& vec) {
  return vec.empty() ? nullptr : &vec[0];
}
inline void ByteSwap(void* data, size_t elem_bytes, size_t num_elems) {
  unsigned char* p = static_cast<unsigned char*>(data);
  for (size_t i = 0; i < num_elems; ++i, p += elem_bytes)
    for (size_t a = 0, b = elem_bytes - 1; a < b; ++a, --b) {
      unsigned char t = p[a];
      p[a] = p[b];
      p[b] = t;
    }
}

class Stream {
 public:
  virtual size_t Read(void* ptr, size_t size) = 0;
  virtual void Write(const void* ptr, size_t size) = 0;
  virtual ~Stream() {}
  template <typename T>
  inline void Write(const T& data);
  template <typename T>
  inline bool Read(T* out_data);
  template <typename T>
  inline void WriteArray(const T* data, size_t num_elems) {
    for (size_t i = 0; i < num_elems; ++i) this->Write<T>(data[i]);
  }
  template <typename T>
  inline bool ReadArray(T* data, size_t num_elems) {
    for (size_t i = 0; i < num_elems; ++i)
      if (!this->Read<T>(data + i)) return false;
    return true;
  }
};
class SeekStream : public Stream {
 public:
  virtual void Seek(size_t pos) = 0;
  virtual size_t Tell() = 0;
};
}  // namespace dmlc
#include "./serializer.h"
