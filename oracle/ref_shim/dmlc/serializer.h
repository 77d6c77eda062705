/* Shim for dmlc-core's <dmlc/serializer.h> (empty submodule): primary Handler template
 * (raw bytes for PODs, length-prefixed for std::vector / std::string) so the reference's
 * partial specialisations in include/dgl/runtime/serializer.h have something to specialise. */
#pragma once
#include <cstdint>
#include <string>
#include <type_traits>
#include <vector>

#include "./io.h"

namespace dmlc {
namespace serializer {
template <typename T>
struct Handler {
  static void Write(Stream* strm, const T& data) {
    if constexpr (std::is_trivially_copyable<T>::value) {
      strm->Write(static_cast<const void*>(&data), sizeof(T));
    } else {
      data.Save(strm);
    }
  }
  static bool Read(Stream* strm, T* data) {
    if constexpr (std::is_trivially_copyable<T>::value) {
      return strm->Read(static_cast<void*>(data), sizeof(T)) == sizeof(T);
    } else {
      return data->Load(strm);
    }
  }
};
template <typename T>
struct Handler<std::vector<T>> {
  static void Write(Stream* strm, const std::vector<T>& v) {
    uint64_t n = v.size();
    strm->Write(static_cast<const void*>(&n), sizeof(n));
    for (const T& x : v) Handler<T>::Write(strm, x);
  }
  static bool Read(Stream* strm, std::vector<T>* v) {
    uint64_t n;
    if (strm->Read(static_cast<void*>(&n), sizeof(n)) != sizeof(n)) return false;
    v->resize(n);
    for (T& x : *v)
      if (!Handler<T>::Read(strm, &x)) return false;
    return true;
  }
};
template <>
struct Handler<std::string> {
  static void Write(Stream* strm, const std::string& s) {
    uint64_t n = s.size();
    strm->Write(static_cast<const void*>(&n), sizeof(n));
    if (n) strm->Write(static_cast<const void*>(s.data()), n);
  }
  static bool Read(Stream* strm, std::string* s) {
    uint64_t n;
    if (strm->Read(static_cast<void*>(&n), sizeof(n)) != sizeof(n)) return false;
    s->resize(n);
    return n == 0 || strm->Read(static_cast<void*>(&(*s)[0]), n) == n;
  }
};
}  // namespace serializer

template <typename T>
inline void Stream::Write(const T& data) {
  serializer::Handler<T>::Write(this, data);
}
template <typename T>
inline bool Stream::Read(T* out_data) {
  return serializer::Handler<T>::Read(this, out_data);
}
}  // namespace dmlc
