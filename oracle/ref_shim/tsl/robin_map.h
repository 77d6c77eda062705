// oracle/ref_shim — TEST INFRASTRUCTURE.  Stand-in for the un-vendored third_party/tsl_robin_map
// submodule (empty in the checkout): the reference's CPU array utilities only need an
// associative container with the std::unordered_map interface (src/array/cpu/array_utils.h).
#pragma once
#include <unordered_map>
#include <unordered_set>
namespace tsl {
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
using robin_map = std::unordered_map<K, V, H, E>;
template <class K, class H = std::hash<K>, class E = std::equal_to<K>>
using robin_set = std::unordered_set<K, H, E>;
}  // namespace tsl
