// Host-side replica of numpy's LEGACY global generator for the negative-sampling collate (data/dataloader.py:154-198 calls
// np.random.choice(items_in_split, n, replace=True) == items_in_split[np.random.randint(0, len, n)] on the global
// RandomState). No device code: the function advances an MT19937 state handed over by the caller
// (np.random.get_state() -> key[624], pos) exactly like numpy's `RandomState.randint(0, high, size=n)` for high - 1 < 2^32:
//   mask = smallest 2^k - 1 >= high - 1;  every value: do { v = genrand_uint32() & mask; } while (v > high - 1)
// (numpy/random/src/distributions/distributions.c: buffered_bounded_masked_uint32 with use_masked = true; mt19937 from
// numpy/random/src/mt19937/mt19937.c). numpy spends ~4.5 ns per value here (per-value function calls through the bit
// generator interface); the tight loop below runs at ~1.5 ns. Pinned against numpy by tests/test_host_cpu.py.
#include <stdint.h>
#include "common.h"

#define MT_N 624
#define MT_M 397

static inline void mt19937_regen(uint32_t* mt) {
  int i;
  uint32_t y;
  for (i = 0; i < MT_N - MT_M; ++i) {
    y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
    mt[i] = mt[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  for (; i < MT_N - 1; ++i) {
    y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
    mt[i] = mt[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
}

// key: 624 state words (updated in place), pos: in/out position in [0, 624]; out: n int64 values in [0, high).
extern "C" int sbr_host_mt19937_randint(uint32_t* key, int* pos, long high, long n, long* out) {
  SBR_REQUIRE(key && pos && (out || n == 0), "sbr_host_mt19937_randint: null operand");
  SBR_REQUIRE(high >= 1 && high - 1 <= 0xFFFFFFFFL, "sbr_host_mt19937_randint: high=%ld outside [1, 2^32]", high);
  SBR_REQUIRE(*pos >= 0 && *pos <= MT_N, "sbr_host_mt19937_randint: bad state position %d", *pos);
  const uint32_t rng = (uint32_t)(high - 1);
  if (rng == 0) {                       // numpy draws nothing for a one-value range
    for (long i = 0; i < n; ++i) out[i] = 0;
    return SBR_OK;
  }
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  // Branch-free acceptance: every word is tempered, masked and stored at out[j]; j advances only when the value is accepted
  // (a rejected value is overwritten by the next one). The rejection test of the textbook loop mispredicts on ~25 % of the
  // words for high = 50,000.
  int p = *pos;
  long j = 0;
  while (j < n) {
    if (p == MT_N) { mt19937_regen(key); p = 0; }
    while (p < MT_N && j < n) {
      uint32_t y = key[p++];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      const uint32_t v = y & mask;
      out[j] = (long)v;
      j += (v <= rng);
    }
  }
  *pos = p;
  return SBR_OK;
}


// Host twin of csr_contains_kernel for the small redraw rounds of the collate (a few dozen pairs: a GPU round trip costs more
// than the search): out[q] = items[q] in row users[q] of the sorted interaction CSR (data/dataloader.py:184-191).
extern "C" int sbr_host_csr_contains(const long* indptr, const int* indices, const long* users, const long* items, long n,
                                     unsigned char* out) {
  SBR_REQUIRE(indptr && indices && users && items && out, "sbr_host_csr_contains: null operand");
  for (long q = 0; q < n; ++q) {
    long lo = indptr[users[q]], hi = indptr[users[q] + 1];
    const long v = items[q];
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (indices[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[q] = (lo < indptr[users[q] + 1] && indices[lo] == v) ? 1 : 0;
  }
  return SBR_OK;
}


// ---- the whole default collate for small batches in one call -----------------------------------------------------------------
// NegativeSamplingDataLoader._neg_sampling_collate_fn (data/dataloader.py:154-198) for batches whose B * n_neg slots are few
// enough that the host search beats a GPU round trip (the reference's default batch, 256 x 10 slots): the Python version spends
// most of its time in ~20 small numpy calls. Same draws in the same order: all slots (slot s belongs to user s % B), then
// rounds that redraw only the colliding slots in ascending slot order, each value one masked-rejection draw of the MT19937
// stream (np.random.choice(arr, m) == arr[np.random.randint(0, len(arr), m)], drawn sequentially).
struct MtStream {
  uint32_t* key;
  int p;
  uint32_t rng, mask;
  inline long next() {
    if (rng == 0) return 0;                                  // numpy draws nothing for a one-value range
    for (;;) {
      if (p == MT_N) { mt19937_regen(key); p = 0; }
      uint32_t y = key[p++];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      const uint32_t v = y & mask;
      if (v <= rng) return (long)v;
    }
  }
};

static inline bool csr_has(const long* indptr, const int* indices, long u, long v) {
  long lo = indptr[u], hi = indptr[u + 1];
  const long end = hi;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (indices[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo < end && indices[lo] == v;
}

// users, pos_items: [B]; items_in_split: [n_cand] or NULL for the identity; (indptr, indices): sorted CSR of the split's
// interactions; out_items: [B, 1 + n_neg] (column 0 = the positive); values, todo: scratch of B * n_neg longs each.
extern "C" int sbr_host_recbole_collate(uint32_t* key, int* pos, const long* users, const long* pos_items, long B, int n_neg,
                                        long n_cand, const long* items_in_split, const long* indptr, const int* indices,
                                        long* out_items, long* values, long* todo) {
  SBR_REQUIRE(key && pos && users && pos_items && indptr && indices && out_items && values && todo,
              "sbr_host_recbole_collate: null operand");
  SBR_REQUIRE(n_cand >= 1 && n_cand - 1 <= 0xFFFFFFFFL, "sbr_host_recbole_collate: n_cand=%ld outside [1, 2^32]", n_cand);
  SBR_REQUIRE(*pos >= 0 && *pos <= MT_N, "sbr_host_recbole_collate: bad state position %d", *pos);
  MtStream g;
  g.key = key;
  g.p = *pos;
  g.rng = (uint32_t)(n_cand - 1);
  g.mask = g.rng;
  g.mask |= g.mask >> 1; g.mask |= g.mask >> 2; g.mask |= g.mask >> 4; g.mask |= g.mask >> 8; g.mask |= g.mask >> 16;
  const long total = B * n_neg;
  for (long s = 0; s < total; ++s) {
    const long r = g.next();
    values[s] = items_in_split ? items_in_split[r] : r;
  }
  long m = 0;
  for (long s = 0; s < total; ++s)
    if (csr_has(indptr, indices, users[s % B], values[s])) todo[m++] = s;
  while (m > 0) {
    for (long q = 0; q < m; ++q) {                           // one randint(n_cand, m) call: m sequential draws
      const long r = g.next();
      values[todo[q]] = items_in_split ? items_in_split[r] : r;
    }
    long m2 = 0;
    for (long q = 0; q < m; ++q) {
      const long s = todo[q];
      if (csr_has(indptr, indices, users[s % B], values[s])) todo[m2++] = s;
    }
    m = m2;
  }
  for (long b = 0; b < B; ++b) {
    long* o = out_items + b * (1 + n_neg);
    o[0] = pos_items[b];
    for (int j = 0; j < n_neg; ++j) o[1 + j] = values[(long)j * B + b];
  }
  *pos = g.p;
  return SBR_OK;
}


// items[b, 0] = pos_items[b]; items[b, 1 + j] = values[j * B + b] — the layout step of the collate (data/dataloader.py:192:
// value_ids.reshape(n_neg, -1).T next to the positive column). numpy's strided transpose copy of 8192 x 10 int64 takes ~85 us.
extern "C" int sbr_host_assemble_items(const long* pos_items, const long* values, long B, int n_neg, long* out_items) {
  SBR_REQUIRE(pos_items && values && out_items, "sbr_host_assemble_items: null operand");
  const long w = 1 + n_neg;
  for (long b = 0; b < B; ++b) out_items[b * w] = pos_items[b];
  for (int j = 0; j < n_neg; ++j) {
    const long* v = values + (long)j * B;
    for (long b = 0; b < B; ++b) out_items[b * w + 1 + j] = v[b];
  }
  return SBR_OK;
}
