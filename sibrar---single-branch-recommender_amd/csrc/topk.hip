// Full-catalogue evaluation helpers (eval/eval.py:205-222): exclusion mask, per-user top-k and ranking metrics.
//   sbr_mask_scores   out[b, excl(u_b)] = -inf          (eval.py:219-220, CSR rows instead of a dense bool matrix)
//   sbr_topk_rows     exact top-k of every score row, sorted by (score desc, index asc)  (torch.topk, eval.py:320)
//   sbr_rank_metrics  NDCG / recall / precision @k from the top-k indices and the CSR labels (eval/metrics.py:4-105)
// Integer / ordering work: results are exact (no tolerance); the only freedom is the order of exactly tied scores,
// which torch.topk leaves unspecified and which is fixed here to "lower index first".
#include "common.h"

__global__ void mask_scores_kernel(float* __restrict__ S, long ld, const long* __restrict__ u_idx,
                                   const long* __restrict__ indptr, const int* __restrict__ indices, long Bu, int item_offset, int n_cols) {
  const long b = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= Bu) return;
  const long u = u_idx ? u_idx[b] : b;
  for (long q = indptr[u] + (threadIdx.x & 63); q < indptr[u + 1]; q += 64) {
    const unsigned int c = (unsigned int)(indices[q] - item_offset);      // columns outside [item_offset, item_offset + n_cols): another shard's
    if (c < (unsigned int)n_cols) S[b * ld + c] = -INFINITY;
  }
}

extern "C" int sbr_mask_scores(float* scores, long ld, const long* u_idx, const long* excl_indptr, const int* excl_indices,
                               long Bu, void* stream) {
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(scores && excl_indptr && excl_indices, "sbr_mask_scores: null operand");
  mask_scores_kernel<<<sbr_cdiv(Bu, 4), 256, 0, (hipStream_t)stream>>>(scores, ld, u_idx, excl_indptr, excl_indices, Bu, 0, 0x7FFFFFFF);
  SBR_CHECK_LAUNCH("sbr_mask_scores");
  return SBR_OK;
}

// the same for a score matrix that holds the item columns [item_offset, item_offset + n_cols) only (item-sharded evaluation)
extern "C" int sbr_mask_scores_shard(float* scores, long ld, const long* u_idx, const long* excl_indptr, const int* excl_indices,
                                     long Bu, int item_offset, int n_cols, void* stream) {
  if (Bu == 0 || n_cols == 0) return SBR_OK;
  SBR_REQUIRE(scores && excl_indptr && excl_indices && item_offset >= 0 && n_cols > 0, "sbr_mask_scores_shard: bad operand");
  mask_scores_kernel<<<sbr_cdiv(Bu, 4), 256, 0, (hipStream_t)stream>>>(scores, ld, u_idx, excl_indptr, excl_indices, Bu, item_offset, n_cols);
  SBR_CHECK_LAUNCH("sbr_mask_scores_shard");
  return SBR_OK;
}

// order-preserving map float -> uint32 (larger float <=> larger key; -inf is the smallest non-NaN key)
__device__ __forceinline__ unsigned int f2key(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

#define TOPK_MAX 256

// Exact selection by radix: 4-pass 8-bit radix select of the k-th largest key (six reads of the row in total), ordered
// collection, bitonic sort. One 256-thread workgroup per row. Used directly for short rows and as the fallback of the sampled
// kernel below.
__device__ __forceinline__ void topk_row_radix(const float* __restrict__ S, long ld, int I, int k, int kpad,
                                               float* __restrict__ out_val, int* __restrict__ out_idx) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long cand[TOPK_MAX];
  __shared__ unsigned int s_prefix, s_need, s_cnt, s_wave[4], s_taken;
  const float* row = S + blockIdx.x * ld;
  const int t = threadIdx.x;

  if (t == 0) { s_prefix = 0; s_need = (unsigned)k; }
  unsigned int prefix = 0, need = (unsigned)k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[t] = 0;
    __syncthreads();
    const unsigned int hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = t; i < I; i += 256) {
      const unsigned int key = f2key(row[i]);
      if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned int acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= need) break;
        acc += hist[d];
      }
      s_prefix = prefix | ((unsigned)d << shift);
      s_need = need - acc;              // how many keys with this digit (and the same higher digits) are still needed
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  // prefix is now the key of the k-th largest element; `need` of the elements equal to it belong to the top-k
  const unsigned int T = prefix;
  if (t == 0) { s_cnt = 0; s_taken = 0; }
  for (int i = t; i < kpad; i += 256) cand[i] = 0ull;       // padding sorts last
  __syncthreads();
  for (int i = t; i < I; i += 256) {
    const unsigned int key = f2key(row[i]);
    if (key > T) {
      const unsigned int pos = atomicAdd(&s_cnt, 1u);
      cand[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
  }
  __syncthreads();
  const unsigned int base = s_cnt;       // == k - need
  // ties at the threshold: the `need` lowest indices, found by an ordered sweep
  const int lane = t & 63, w = t >> 6;
  for (int i0 = 0; i0 < I; i0 += 256) {
    const int i = i0 + t;
    const bool eq = i < I && f2key(row[i]) == T;
    const unsigned long long bal = __ballot(eq);
    if (lane == 0) s_wave[w] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned int before = s_taken;
    for (int q = 0; q < w; ++q) before += s_wave[q];
    const unsigned int mine = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (eq && mine < need) cand[base + mine] = ((unsigned long long)T << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    __syncthreads();
    if (t == 0) s_taken += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    if (s_taken >= need) break;
  }
  __syncthreads();
  // bitonic sort, descending, on kpad (power of two <= 256) composite keys
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = t;
      if (i < kpad) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = (i & size) == 0;
          const unsigned long long a = cand[i], b = cand[j];
          if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (t < k) {
    const unsigned long long c = cand[t];
    out_val[blockIdx.x * (long)k + t] = key2f((unsigned int)(c >> 32));
    out_idx[blockIdx.x * (long)k + t] = (int)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFull));
  }
}


__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ S, long ld, int I, int k, int kpad,
                                                        float* __restrict__ out_val, int* __restrict__ out_idx) {
  topk_row_radix(S, ld, I, k, kpad, out_val, out_idx);
}

// Long rows (the [users, items] score matrix of the fp32 evaluation path is read from HBM, 200 KB per row at 50k items): the
// radix select above reads a row six times. Here a 1/16 sample of the row (kept in LDS) yields a threshold T0 whose rank in the
// whole row is ~4k +- 20 %; ONE full read then collects every element >= T0 (a few hundred) into an LDS buffer, which is
// sorted exactly (score descending, index ascending — the same composite key as everywhere). The result is exact whenever
// k <= #collected <= TOPK_CAND, which the kernel checks; otherwise (heavy ties, adversarial rows) the row takes the radix path.
// The sample is one 64-byte line (16 floats) out of every 16 lines: an element-strided sample (every 16th float) touches every
// line of the row, i.e. costs a second full read of the row from HBM (measured: 2.1 TB/s of useful traffic for the kernel).
#define TOPK_SAMPLE_STRIDE 16
#define TOPK_SAMPLE_MAX 4096
#define TOPK_CAND 2048

__global__ __launch_bounds__(256) void topk_rows_sampled_kernel(const float* __restrict__ S, long ld, int I, int k, int kpad,
                                                                float* __restrict__ out_val, int* __restrict__ out_idx) {
  __shared__ unsigned int samp[TOPK_SAMPLE_MAX];
  __shared__ unsigned long long cand[TOPK_CAND];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_need, s_cnt;
  const float* row = S + blockIdx.x * ld;
  const int t = threadIdx.x;
  constexpr int LINE = 16, SPAN = LINE * TOPK_SAMPLE_STRIDE;                 // floats per sampled line / per group of lines
  const int tail = I % SPAN;
  const int ns = (I / SPAN) * LINE + (tail < LINE ? tail : LINE);            // <= TOPK_SAMPLE_MAX (checked by the launcher)
  for (int j = t; j < ns; j += 256) samp[j] = f2key(row[(long)(j / LINE) * SPAN + (j % LINE)]);
  // rank of T0 in the sample: m-th largest, m ~ 4k / stride (at least 4): its rank in the row is ~ m * stride
  int m = (4 * k + TOPK_SAMPLE_STRIDE - 1) / TOPK_SAMPLE_STRIDE;
  m = m < 4 ? 4 : m;
  m = m > ns ? ns : m;
  unsigned int prefix = 0, need = (unsigned)m;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[t] = 0;
    __syncthreads();
    const unsigned int hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int j = t; j < ns; j += 256) {
      const unsigned int key = samp[j];
      if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned int acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= need) break;
        acc += hist[d];
      }
      s_prefix = prefix | ((unsigned)d << shift);
      s_need = need - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  const unsigned int T0 = prefix;                    // m-th largest sample key
  if (t == 0) s_cnt = 0;
  __syncthreads();
  // one pass over the row: everything >= T0 (float4 loads when the row is 16-byte aligned). Eight loads are issued before the
  // first one is consumed: with a single outstanding 16-byte load per thread the pass runs at memory LATENCY — 1024 resident
  // workgroups x 4 KB per ~2 us round trip = 2.1 TB/s, which is what the first version measured.
  const bool vec = ((((uintptr_t)row) & 15) == 0);
  const int I4 = vec ? (I >> 2) : 0;
  auto take4 = [&](const float4& v, int q) {
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned int key = f2key(e[c]);
      if (key >= T0) {
        const unsigned int pos = atomicAdd(&s_cnt, 1u);
        if (pos < TOPK_CAND) cand[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(4 * q + c));
      }
    }
  };
  constexpr int UNR = 8;
  const float4* row4 = reinterpret_cast<const float4*>(row);
  int q = t;
  for (; q + (UNR - 1) * 256 < I4; q += UNR * 256) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = row4[q + u * 256];
#pragma unroll
    for (int u = 0; u < UNR; ++u) take4(v[u], q + u * 256);
  }
  for (; q < I4; q += 256) take4(row4[q], q);
  for (int i = 4 * I4 + t; i < I; i += 256) {
    const unsigned int key = f2key(row[i]);
    if (key >= T0) {
      const unsigned int pos = atomicAdd(&s_cnt, 1u);
      if (pos < TOPK_CAND) cand[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
  }
  __syncthreads();
  const unsigned int n = s_cnt;
  if (n < (unsigned)k || n > TOPK_CAND) {            // uniform per workgroup: exact fallback
    __syncthreads();
    topk_row_radix(S, ld, I, k, kpad, out_val, out_idx);
    return;
  }
  // bitonic sort (descending) of the candidates, padded with zeros to a power of two >= n
  int np = 2;
  while (np < (int)n) np <<= 1;
  for (int i = (int)n + t; i < np; i += 256) cand[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= np; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < np; i += 256) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = (i & size) == 0;
          const unsigned long long a = cand[i], b = cand[j];
          if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (t < k) {
    const unsigned long long c = cand[t];
    out_val[blockIdx.x * (long)k + t] = key2f((unsigned int)(c >> 32));
    out_idx[blockIdx.x * (long)k + t] = (int)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFull));
  }
}

extern "C" int sbr_topk_rows(const float* scores, long ld, long Bu, int I, int k, float* out_val, int* out_idx, void* stream) {
  SBR_REQUIRE(k >= 1 && k <= TOPK_MAX, "sbr_topk_rows: k=%d outside [1, %d]", k, TOPK_MAX);
  SBR_REQUIRE(k <= I, "sbr_topk_rows: k=%d larger than the row length %d", k, I);
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(scores && out_val && out_idx, "sbr_topk_rows: null operand");
  int kpad = 2;
  while (kpad < k) kpad <<= 1;
  // long rows: one-read sampled selection; short rows (or k close to the candidate capacity): radix select
  if (I >= 8192 && I <= TOPK_SAMPLE_MAX * TOPK_SAMPLE_STRIDE && 8 * k <= TOPK_CAND)
    topk_rows_sampled_kernel<<<(unsigned)Bu, 256, 0, (hipStream_t)stream>>>(scores, ld, I, k, kpad, out_val, out_idx);
  else
    topk_rows_kernel<<<(unsigned)Bu, 256, 0, (hipStream_t)stream>>>(scores, ld, I, k, kpad, out_val, out_idx);
  SBR_CHECK_LAUNCH("sbr_topk_rows");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Exact merge of W per-shard top-k lists (item-sharded scoring, SURVEY 8(e): every rank scores its item shard, the lists are
// all-gathered): out[b] = the k best of the W * k entries (score desc, item index asc — the rule every top-k kernel here
// uses; idx < 0 marks an empty slot). One wave per user; every entry is ranked against all others by counting over
// v_readlane broadcasts (W * k <= 256: up to four entries per lane), the entry of rank j writes output position j.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ vals, const int* __restrict__ idxs, int W,
                                                         long Bu, int k, float* __restrict__ out_val, int* __restrict__ out_idx) {
  const long b = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= Bu) return;
  const int lane = threadIdx.x & 63;
  const int n = W * k;                                        // <= 256
  unsigned long long e[4];
  int rank[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = lane + 64 * q;
    e[q] = 0ull;
    rank[q] = 0;
    if (j < n) {
      const int w = j / k, r = j - w * k;
      const long src = ((long)w * Bu + b) * k + r;
      const int id = idxs[src];
      if (id >= 0) e[q] = ((unsigned long long)f2key(vals[src]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)id);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hi = (int)(e[q] >> 32), lo = (int)e[q];
    const int m = n - 64 * q < 64 ? n - 64 * q : 64;
    for (int j = 0; j < m; ++j) {
      const unsigned long long kj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(hi, j) << 32) |
                                    (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(lo, j);
#pragma unroll
      for (int p = 0; p < 4; ++p) rank[p] += kj > e[p];
    }
  }
  // empty entries (composite 0) rank behind every real one but tie with each other: they fill the remaining positions below
  int n_real = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) n_real += __popcll(__ballot(e[q] != 0ull));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (e[q] != 0ull && rank[q] < k) {
      out_val[b * k + rank[q]] = key2f((unsigned int)(e[q] >> 32));
      out_idx[b * k + rank[q]] = (int)(0xFFFFFFFFu - (unsigned int)(e[q] & 0xFFFFFFFFull));
    }
  }
  for (int j = n_real + lane; j < k; j += 64) {
    out_val[b * k + j] = -INFINITY;
    out_idx[b * k + j] = -1;
  }
}

extern "C" int sbr_merge_topk(const float* vals, const int* idxs, int W, long Bu, int k, float* out_val, int* out_idx, void* stream) {
  SBR_REQUIRE(W >= 1 && k >= 1 && (long)W * k <= 256, "sbr_merge_topk: W * k = %ld outside [1, 256]", (long)W * k);
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(vals && idxs && out_val && out_idx, "sbr_merge_topk: null operand");
  merge_topk_kernel<<<sbr_cdiv(Bu, 4), 256, 0, (hipStream_t)stream>>>(vals, idxs, W, Bu, k, out_val, out_idx);
  SBR_CHECK_LAUNCH("sbr_merge_topk");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ranking metrics with binary relevance (eval/metrics.py): for each user b and each cutoff ks[q]
//   hits   = #{r < k : topk[b, r] in labels(u_b)}
//   recall = hits / n_pos (0 when n_pos == 0), precision = hits / k,
//   ndcg   = min(1, sum_{r<k, hit} 1/log2(r+2) / sum_{r<min(k,n_pos)} 1/log2(r+2))   (0 when n_pos == 0)
// labels: CSR over user ids, column ids sorted ascending within a row. out: [3, n_ks, Bu] (ndcg, recall, precision).
// ---------------------------------------------------------------------------------------------------------------
#define METRIC_MAX_KS 8
struct KList { int n; int k[METRIC_MAX_KS]; };

__global__ void rank_metrics_kernel(const int* __restrict__ topk, int kmax, const long* __restrict__ u_idx,
                                    const long* __restrict__ indptr, const int* __restrict__ indices, long Bu, KList ks,
                                    float* __restrict__ out) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (b >= Bu) return;
  const long u = u_idx ? u_idx[b] : b;
  const long beg = indptr[u], end = indptr[u + 1];
  const int npos = (int)(end - beg);
  float dcg = 0.f, idcg = 0.f;
  int hits = 0, q = 0;
  for (int r = 0; r < kmax && q < ks.n; ++r) {
    const int item = topk[b * kmax + r];
    long lo = beg, hi = end;
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (indices[mid] < item) lo = mid + 1; else hi = mid;
    }
    const float disc = 1.f / log2f((float)(r + 2));
    if (lo < end && indices[lo] == item) { ++hits; dcg += disc; }
    if (r < npos) idcg += disc;
    while (q < ks.n && ks.k[q] == r + 1) {
      const float nd = npos > 0 ? fminf(dcg / idcg, 1.f) : 0.f;
      out[(0 * ks.n + q) * Bu + b] = nd;
      out[(1 * ks.n + q) * Bu + b] = npos > 0 ? (float)hits / (float)npos : 0.f;
      out[(2 * ks.n + q) * Bu + b] = (float)hits / (float)ks.k[q];
      ++q;
    }
  }
}

extern "C" int sbr_rank_metrics(const int* topk_idx, int kmax, const long* u_idx, const long* label_indptr,
                                const int* label_indices, long Bu, const int* ks, int n_ks, float* out, void* stream) {
  SBR_REQUIRE(n_ks >= 1 && n_ks <= METRIC_MAX_KS, "sbr_rank_metrics: n_ks=%d outside [1, %d]", n_ks, METRIC_MAX_KS);
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(topk_idx && label_indptr && label_indices && ks && out, "sbr_rank_metrics: null operand");
  KList kl;
  kl.n = n_ks;
  for (int i = 0; i < n_ks; ++i) {
    SBR_REQUIRE(ks[i] >= 1 && ks[i] <= kmax && (i == 0 || ks[i] > ks[i - 1]), "sbr_rank_metrics: ks must be ascending and <= kmax");
    kl.k[i] = ks[i];
  }
  rank_metrics_kernel<<<sbr_cdiv(Bu, 256), 256, 0, (hipStream_t)stream>>>(topk_idx, kmax, u_idx, label_indptr, label_indices, Bu, kl, out);
  SBR_CHECK_LAUNCH("sbr_rank_metrics");
  return SBR_OK;
}
