// Fused full-catalogue scorer for evaluation (eval/eval.py:205-222 without ever writing the [users, items] score matrix):
//   scores = U[Bu, D] x I[I_s, D]^T on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate),
//   out[b, excl(u_b)] = -inf (eval.py:219-220) applied to the few values that matter,
//   running exact top-k per user kept on chip; output sorted by (score desc, item index asc).
//
// Geometry: one workgroup = 8 wavefronts = 256 users, one workgroup per CU. Wave w owns users [32w, 32w+32) of the block for
// the whole kernel and keeps their fp16 rows as MFMA A-fragments in registers (D/16 x 4 VGPRs). The item matrix is streamed
// once per workgroup through a ring of XOR-swizzled 64-item LDS tiles filled by LDS-DMA (global_load_lds_dwordx4, swizzle on
// the per-lane SOURCE address): up to NS-1 tiles are in flight behind a counted s_waitcnt vmcnt and one raw s_barrier per
// tile, so the L2 -> LDS latency is hidden behind the MFMAs of the preceding tiles. Every wave multiplies its 32 users by the
// 64 items of a tile (2 x D/16 MFMAs).
//
// Top-k: each accumulator value is compared with its row's current k-th best score held in a register (16 v_cmp per 32x32
// tile, OR-reduced to one branch). Only the rare survivors are tested against the tile's exclusion bit mask (each lane walks
// the sorted exclusion CSR row of one user in step with the tiles) and appended to the row's candidate buffer in LDS. A full
// buffer is compacted by its owning wave (rank by counting), which raises the row's threshold. Rows are owned by exactly one
// wave, so the top-k state needs no cross-wave synchronisation.
#include "common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ST_TILE 64        // items per LDS tile
#define ST_WAVES 8
#define ST_ROWS (ST_WAVES * 32)
#define ST_THREADS (ST_WAVES * 64)

__device__ __forceinline__ unsigned int st_f2key(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float st_key2f(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct TopkState {
  unsigned long long* buf;   // [rows][cap] composite keys (score key << 32 | ~item)
  int* cnt;                  // [rows]
  int cap, k;
};

// all 64 lanes of the owning wave: keep the k best of row r's buffer, sorted; returns the new threshold
__device__ __forceinline__ float st_compact(const TopkState& st, int r, int lane) {
  unsigned long long* b = st.buf + r * st.cap;
  int n = st.cnt[r];
  n = n < st.cap ? n : st.cap;
  const unsigned long long mine = lane < n ? b[lane] : 0ull;
  int rank = 0;
  for (int j = 0; j < n; ++j) rank += (b[j] > mine);
  if (lane < n && rank < st.k) b[rank] = mine;
  const int kept = n < st.k ? n : st.k;
  if (lane == 0) st.cnt[r] = kept;
  const unsigned long long who = __ballot(lane < n && rank == st.k - 1);
  float thr = -INFINITY;
  if (who) thr = st_key2f((unsigned int)(__shfl(mine, __ffsll((long long)who) - 1, 64) >> 32));
  return thr;
}

// overflow path of one accumulator register step: some lanes could not append because their row's buffer is full.
// Compacts those rows (raising their thresholds) and retries until every pending candidate is stored or beaten.
// Returns the (possibly raised) threshold of this lane's row.
__device__ __noinline__ float st_overflow(TopkState st, float v, bool pending, int row, unsigned long long key, float thr,
                                          int lane) {
  for (;;) {
    unsigned long long ov = __ballot(pending);
    if (!ov) break;
    while (ov) {
      const int src = __ffsll((long long)ov) - 1;
      const int r = __shfl(row, src, 64);
      const float nt = st_compact(st, r, lane);
      if (row == r) {
        thr = nt;
        if (pending && !(v > thr)) pending = false;
      }
      ov &= ~__ballot(row == r);
    }
    if (pending) {
      const int pos = atomicAdd(&st.cnt[row], 1);
      if (pos < st.cap) { st.buf[row * st.cap + pos] = key; pending = false; }
    }
  }
  return thr;
}

template <int N>
__device__ __forceinline__ void st_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KS, int NS>   // KS = D / 16; NS = LDS ring slots (NS - 1 tiles in flight)
__global__ __launch_bounds__(ST_THREADS, 2) void score_topk_f16_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const long* __restrict__ u_idx,
    const long* __restrict__ excl_indptr, const int* __restrict__ excl_indices, int item_offset, int k, int cap,
    float* __restrict__ out_val, int* __restrict__ out_idx, int dbg) {
  constexpr int D = KS * 16;
  constexpr int ROWB = D * 2;                              // bytes per item row
  constexpr int TILEB = ST_TILE * ROWB;                    // bytes per LDS tile
  constexpr int CPR = D / 8;                               // 16-byte chunks per item row
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);        // XOR swizzle mask over the chunks of a row
  constexpr int PER_W = (ST_TILE * CPR) / (ST_WAVES * 64); // 1-KiB LDS-DMA instructions per wave and tile
  constexpr int PF = NS - 1;                               // tiles in flight
  static_assert(PER_W >= 1 && (ST_TILE * CPR) % (ST_WAVES * 64) == 0, "tile must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TopkState st;
  st.buf = reinterpret_cast<unsigned long long*>(smem + NS * TILEB);
  st.cnt = reinterpret_cast<int*>(smem + NS * TILEB + (size_t)ST_ROWS * cap * 8);
  st.cap = cap;
  st.k = k;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long row0 = (long)blockIdx.x * ST_ROWS;

  if (lane < 32) st.cnt[wave * 32 + lane] = 0;          // each wave initialises the rows it owns

  // A fragments: user row (32*wave + l31), k = 16*s + 8*half + j
  f16x8 afrag[KS];
  const long my_row = row0 + wave * 32 + l31;            // the user row this lane loads and whose exclusion list it walks
  {
    const long ur = my_row < Bu ? my_row : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) afrag[s] = src[2 * s + half];
  }
  // exclusion cursor of row l31 of this wave (eval/eval.py:219-220): the user's sorted CSR row is consumed in step with the
  // item tiles; per tile a 64-bit mask of the excluded columns is kept in a register (both lane halves hold a copy).
  long ecur = 0, eend = 0;
  int enext = 0x7FFFFFFF;
  const bool row_valid = my_row < Bu;
  if (row_valid && excl_indptr) {
    const long u = u_idx ? u_idx[my_row] : my_row;
    long lo = excl_indptr[u];
    eend = excl_indptr[u + 1];
    long hi = eend;
    while (lo < hi) {                                      // first entry >= item_offset (item-sharded catalogues)
      const long mid = (lo + hi) >> 1;
      if (excl_indices[mid] < item_offset) lo = mid + 1; else hi = mid;
    }
    ecur = lo;
    if (ecur < eend) enext = excl_indices[ecur];
  }
  float thr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) thr[r] = -INFINITY;

  // LDS-DMA fill of one tile: wave w issues PER_W instructions, each writing 64 consecutive 16-byte chunk positions
  // (1 KiB) of the slot; chunk position (row i, cp) receives source chunk cp ^ (i & SWZ) of item row j0 + i.
  auto issue_tile = [&](int tile_idx) {
    const int j0 = tile_idx * ST_TILE;
    unsigned char* slot = smem + (tile_idx % NS) * TILEB;
#pragma unroll
    for (int q = 0; q < PER_W; ++q) {
      const int cb = wave * PER_W + q;                     // 1-KiB block of the tile (wave-uniform)
      const int P = cb * 64 + lane;
      const int i = P / CPR, cp = P % CPR;
      int gi = j0 + i;
      gi = gi < I ? gi : I - 1;                            // clamp: values of padded columns are never used
      const _Float16* src = It + (long)gi * D + ((cp ^ (i & SWZ)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + cb * 1024), 16, 0, 0);
    }
  };

  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < n_tiles) issue_tile(p);

  for (int tl = 0; tl < n_tiles; ++tl) {
    // tile tl has landed once at most (PF-1)*PER_W younger LDS-DMA instructions of this wave are outstanding
    if (tl + PF - 1 < n_tiles) st_wait_vmcnt<(PF - 1) * PER_W>();
    else st_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // every wave's part of tile tl is in LDS; tile tl-1 fully consumed
    if (tl + PF < n_tiles) issue_tile(tl + PF);            // refill the slot that tile tl-1 occupied
    const unsigned char* cur = smem + (tl % NS) * TILEB;

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        const int i = nj * 32 + l31;                       // item row within the tile
        const int c = 2 * s + half;                        // 16-byte chunk: k = 16 s + 8 half .. +7
        const f16x8 b = *reinterpret_cast<const f16x8*>(cur + i * ROWB + ((c ^ (i & SWZ)) << 4));
        acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[s], b, acc[nj], 0, 0, 0);
      }
    }
    // exclusion mask of this tile for row l31
    const int j0 = tl * ST_TILE;
    const int gbase = item_offset + j0;
    unsigned long long rmask = row_valid ? 0ull : ~0ull;
    while (enext < gbase + ST_TILE) {
      rmask |= 1ull << (enext - gbase);
      ++ecur;
      enext = ecur < eend ? excl_indices[ecur] : 0x7FFFFFFF;
    }
    if (dbg == 1) { asm volatile("" ::"v"(acc[0]), "v"(acc[1])); continue; }
    // epilogue: threshold filter; survivors are appended to their row's buffer
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int item = j0 + nj * 32 + l31;
      const bool in_range = item < I;
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) any |= __ballot(acc[nj][r] > thr[r]);
      if (dbg == 2) { if (any) asm volatile("s_nop 0"); continue; }
      if (!any) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[nj][r];
        bool cand = in_range && (v > thr[r]);
        if (__ballot(cand)) {
          const int rw = (r & 3) + 8 * (r >> 2) + 4 * half;            // row within the wave's 32
          const unsigned long long m = __shfl(rmask, rw, 64);
          cand = cand && !((m >> (nj * 32 + l31)) & 1ull);
          const int lrow = wave * 32 + rw;
          const unsigned long long key = ((unsigned long long)st_f2key(v) << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (unsigned)(item_offset + item));
          bool pending = cand;
          if (pending) {
            const int pos = atomicAdd(&st.cnt[lrow], 1);
            if (pos < cap) { st.buf[lrow * cap + pos] = key; pending = false; }
          }
          if (__ballot(pending)) thr[r] = st_overflow(st, v, pending, lrow, key, thr[r], lane);
        }
      }
    }
  }

  // final compaction + output: wave-owned rows
  for (int q = 0; q < 32; ++q) {
    const int lrow = wave * 32 + q;
    const long ur = row0 + lrow;
    if (ur >= Bu) break;
    st_compact(st, lrow, lane);
    const int n = st.cnt[lrow];
    if (lane < k) {
      float val = -INFINITY;
      int idx = -1;
      if (lane < n) {
        const unsigned long long c = st.buf[lrow * cap + lane];
        val = st_key2f((unsigned int)(c >> 32));
        idx = (int)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFull));
      }
      out_val[ur * k + lane] = val;
      out_idx[ur * k + lane] = idx;
    }
  }
}

static int st_cap(int k) { int c = 2 * k; if (c < k + 16) c = k + 16; if (c > 64) c = 64; return c; }

extern "C" long sbr_score_topk_f16_workspace(long Bu, int I, int k) { (void)Bu; (void)I; (void)k; return 0; }

template <int KS, int NS>
static int st_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                     int item_offset, int k, float* out_val, int* out_idx, hipStream_t s) {
  const int cap = st_cap(k);
  const size_t lds = (size_t)NS * ST_TILE * KS * 32 + (size_t)ST_ROWS * cap * 8 + ST_ROWS * 4;
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  if (hipFuncSetAttribute((const void*)score_topk_f16_kernel<KS, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  score_topk_f16_kernel<KS, NS><<<sbr_cdiv(Bu, ST_ROWS), ST_THREADS, lds, s>>>(
      (const _Float16*)U, (const _Float16*)It, Bu, I, u_idx, eptr, eidx, item_offset, k, cap, out_val, out_idx,
      getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  return SBR_OK;
}

extern "C" int sbr_score_topk_f16(const void* U_f16, const void* I_f16, int D, long Bu, int I, const long* u_idx,
                                  const long* excl_indptr, const int* excl_indices, int item_offset, int k, float* out_val,
                                  int* out_idx, void* workspace, long workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  SBR_REQUIRE(k >= 1 && k <= 32, "sbr_score_topk_f16: k=%d outside [1, 32] (use sbr_gemm_f32 + sbr_topk_rows)", k);
  SBR_REQUIRE(I >= 1, "sbr_score_topk_f16: empty catalogue");
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(U_f16 && I_f16 && out_val && out_idx, "sbr_score_topk_f16: null operand");
  SBR_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr), "sbr_score_topk_f16: exclusion CSR must be given whole or not at all");
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 64: return st_launch<4, 4>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, s);
    case 128: return st_launch<8, 4>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, s);
    case 256: return st_launch<16, 2>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, s);
    default:
      sbr_set_error("sbr_score_topk_f16: D=%d not supported (64, 128, 256)", D);
      return SBR_ERR_ARG;
  }
}

__global__ void cast_f16_kernel(const float* __restrict__ X, _Float16* __restrict__ Y, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) Y[e] = (_Float16)X[e];
}

extern "C" int sbr_cast_f32_to_f16(const float* X, void* Y_f16, long n, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(X && Y_f16, "sbr_cast_f32_to_f16: null operand");
  int blocks = sbr_cdiv(n, 256);
  if (blocks > 8192) blocks = 8192;
  cast_f16_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(X, (_Float16*)Y_f16, n);
  SBR_CHECK_LAUNCH("sbr_cast_f32_to_f16");
  return SBR_OK;
}
