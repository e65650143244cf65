// rocprim.hpp -- host stand-in for the one rocPRIM call ba.hip makes (test infrastructure, see ../hip/hip_runtime.h)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
// stable LSD radix sort of (key, value) pairs on the key bits [begin_bit, end_bit): with a null temporary buffer only its size is returned
template <class K, class V>
inline hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n, unsigned begin_bit,
                                   unsigned end_bit, hipStream_t = nullptr, bool = false) {
  if (!tmp) {
    bytes = 256;
    return hipSuccess;
  }
  const unsigned long long mask = end_bit - begin_bit >= 64 ? ~0ull : ((1ull << (end_bit - begin_bit)) - 1ull);
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
    return (((unsigned long long)keys_in[a] >> begin_bit) & mask) < (((unsigned long long)keys_in[b] >> begin_bit) & mask);
  });
  std::vector<K> k(n);
  std::vector<V> v(n);
  for (size_t i = 0; i < n; i++) {
    k[i] = keys_in[idx[i]];
    v[i] = vals_in[idx[i]];
  }
  for (size_t i = 0; i < n; i++) {
    keys_out[i] = k[i];
    vals_out[i] = v[i];
  }
  return hipSuccess;
}
}  // namespace rocprim
