// Fused full-catalogue scorer for evaluation (eval/eval.py:205-222 without ever writing the [users, items] score matrix):
//   scores = U[Bu, D] x I[I_s, D]^T on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate),
//   out[b, excl(u_b)] = -inf (eval.py:219-220) applied to the few values that matter,
//   running exact top-k per user kept on chip; output sorted by (score desc, item index asc).
//
// Geometry: one workgroup per CU = 7 consumer wavefronts + 1 loader wavefront. Consumer wave w owns users [32w, 32w+32) of the
// workgroup's 224-user block for the whole kernel and keeps their fp16 rows as MFMA A-fragments in registers (D/16 x 4 VGPRs).
// The loader wave streams the item matrix once per workgroup through a ring of XOR-swizzled 64-item LDS tiles with LDS-DMA
// (global_load_lds_dwordx4, swizzle on the per-lane SOURCE address) and hands tiles over through two LDS counters per slot
// (FULL: published tile number, FREE: consumers done). There is NO workgroup barrier in the main loop: a consumer that is
// busy with top-k bookkeeping only delays the others once the whole ring is used up.
//
// Top-k: each accumulator value is compared with its row's current k-th best score held in a register (16 v_cmp per 32x32
// tile whose ballots stay in SGPRs, OR-reduced to one branch). The rare survivors are tested against the tile's exclusion
// bits (each lower-half lane walks the sorted exclusion CSR row of one user in step with the tiles; look-ahead entries arrive
// by 4-byte LDS-DMA so that the loop contains no ordinary global load) and appended to the row's candidate buffer in LDS at
// positions derived from the wave ballot (no atomics). A full buffer is compacted by its owning wave — during the stream by
// SELECTION of the k best (st_select: bitwise binary search of the k-th key over ballot counts), in the final pass by ranking
// (st_compact: counting over LDS broadcasts, sorted output) — which raises the row's threshold. Rows are owned by exactly one
// wave: the top-k state needs no cross-wave synchronisation. This kernel serves D = 256; D <= 128 takes the wide kernel below.
#include "score_topk_common.h"


#define ST_WAVES 7                       // consumer waves
#define ST_ROWS (ST_WAVES * 32)          // users per workgroup
#define ST_THREADS ((ST_WAVES + 1) * 64) // + loader wave


struct TopkState {
  lds_u64* buf;              // [rows][cap] composite keys (score key << 32 | ~item); explicit LDS address space: 32-bit
                             // addresses and plain ds_read/ds_write also inside the non-inlined overflow path
  int cap, k;
};


// all 64 lanes of the owning wave: keep the k best of the first n (wave-uniform) entries of row r's buffer, sorted;
// returns the new threshold (-inf while fewer than k entries exist)
__device__ __forceinline__ float st_compact(const TopkState& st, int r, int n_any, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_any);     // wave-uniform: scalar loop control below
  lds_u64* b = st.buf + r * st.cap;
  st_wave_fence();
  const unsigned long long mine = lane < n ? b[lane] : 0ull;
  // rank by counting: the n keys are read back as LDS broadcasts (uniform address), eight reads in flight per step
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 8) {
    unsigned long long kj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) kj[q] = b[j0 + q < n ? j0 + q : n - 1];
#pragma unroll
    for (int q = 0; q < 8; ++q) rank += (j0 + q < n) && (kj[q] > mine);
  }
  st_wave_fence();
  if (lane < n && rank < st.k) b[rank] = mine;
  st_wave_fence();
  const unsigned long long who = __ballot(lane < n && rank == st.k - 1);
  float thr = -INFINITY;
  if (who) thr = st_key2f((unsigned int)__builtin_amdgcn_readlane((int)(mine >> 32), __ffsll((long long)who) - 1));
  return thr;
}

// Compaction during the stream does not need the survivors sorted: keep the k best of the first n (<= 64, wave-uniform) entries
// of row r's buffer at positions [0, k), in buffer order, and return the new threshold. The k-th largest score key is found by
// a bitwise binary search over ballot counts (32 steps of v_cmp + s_bcnt1; ties at that key are resolved on the item half of
// the composite the same way) — about a sixth of the cycles of the ranking in st_compact, which the final pass still uses.
__device__ __forceinline__ float st_select(const TopkState& st, int r, int n_any, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_any);
  lds_u64* b = st.buf + r * st.cap;
  st_wave_fence();
  const unsigned long long mine = lane < n ? b[lane] : 0ull;
  if (n < st.k) return -INFINITY;
  const unsigned int h = (unsigned int)(mine >> 32);
  unsigned int T = 0u;                                       // real keys are > 0, empty lanes hold 0
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    T = __popcll(__ballot(h >= trial)) >= st.k ? trial : T;
  }
  unsigned long long C = (unsigned long long)T << 32;
  if (__popcll(__ballot(h >= T)) != st.k) {                  // several entries share the k-th key: smallest item indices stay
    const int need = st.k - __popcll(__ballot(h > T));
    const unsigned int l = (unsigned int)mine;
    unsigned int Lw = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = Lw | (1u << bit);
      Lw = __popcll(__ballot(h == T && l >= trial)) >= need ? trial : Lw;
    }
    C |= (unsigned long long)Lw;
  }
  const bool keep = mine >= C;                               // exactly k lanes (composites are unique and non-zero)
  const unsigned long long kb = __ballot(keep);
  const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)kb, 0u));
  st_wave_fence();
  if (keep) b[pos] = mine;
  st_wave_fence();
  return st_key2f(T);
}

// overflow path of one accumulator register step: some lanes could not append because their row's buffer is full.
// Compacts those rows (raising their thresholds) and retries until every pending candidate is stored or beaten.
// thr / cnt are the lane's register copies of its row's threshold and fill count (identical in the 32 lanes that share the
// row); the updated pair is returned. (The ABI makes a called function wait for vmcnt(0): harmless here because the
// consumer waves have no tile DMA of their own in flight — only the rare exclusion look-ahead.)
struct RowState { float thr; int cnt; };

__device__ __noinline__ RowState st_overflow(TopkState st, float v, bool pending, int row, unsigned long long key, float thr,
                                             int cnt, int lane) {
  for (int guard = 0; guard < 4096; ++guard) {             // bounded: every round stores or drops >= 1 candidate
    const unsigned long long ov = __ballot(pending);
    if (!ov) break;
    const int src = __ffsll((long long)ov) - 1;
    const int r = __shfl(row, src, 64);
    int n = __shfl(cnt, src, 64);                          // fill count of row r (register copy of the source lane)
    n = n < st.cap ? n : st.cap;
    const float nt = st_select(st, r, n, lane);
    const int kept = n < st.k ? n : st.k;
    const bool mine = row == r;
    if (mine) {
      thr = nt;
      cnt = kept;
      if (pending && !(v > thr)) pending = false;
    }
    // retry the still-pending candidates of row r: positions by ballot order
    const unsigned long long pb = __ballot(mine && pending);
    if (mine) {
      if (pending) {
        const int pos = cnt + __popcll(pb & ((1ull << lane) - 1ull));
        if (pos < st.cap) { st.buf[r * st.cap + pos] = key; pending = false; }
      }
      cnt += __popcll(pb);
    }
  }
  RowState out;
  out.thr = thr;
  out.cnt = cnt;
  return out;
}


template <int KS, int NS, int ST_TILE, int DBG>   // KS = D / 16; NS = LDS ring slots; ST_TILE = items per LDS tile (32 | 64); DBG: ablations
__global__ __launch_bounds__(ST_THREADS, 2) void score_topk_f16_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const long* __restrict__ u_idx,
    const long* __restrict__ excl_indptr, const int* __restrict__ excl_indices, int item_offset, int k, int cap,
    float* __restrict__ out_val, int* __restrict__ out_idx, unsigned long long* __restrict__ dbgbuf) {
  constexpr int D = KS * 16;
  constexpr int NJ = ST_TILE / 32;                         // 32-column accumulator tiles per item tile
  constexpr int ROWB = D * 2;                              // bytes per item row
  constexpr int TILEB = ST_TILE * ROWB;                    // bytes per LDS tile
  constexpr int CPR = D / 8;                               // 16-byte chunks per item row
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);        // XOR swizzle mask over the chunks of a row
  constexpr int PER_T = (ST_TILE * CPR) / 64;              // 1-KiB LDS-DMA instructions per tile (all issued by the loader)
  constexpr int LFL = NS > 2 ? NS - 2 : 1;                 // tiles the loader keeps in flight behind the one it waits for
  static_assert(LFL * PER_T <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TopkState st;
  st.buf = (lds_u64*)(smem + NS * TILEB);
  st.cap = cap;
  st.k = k;
  unsigned int* exw = reinterpret_cast<unsigned int*>(smem + NS * TILEB + (size_t)ST_ROWS * cap * 8);  // [ST_WAVES*64] exclusion bits
  int* enx = reinterpret_cast<int*>(exw + ST_WAVES * 64);                                               // [ST_WAVES*64] look-ahead entries
  lds_int* full_lds = (lds_int*)(enx + ST_WAVES * 64);     // [NS] published tile number + 1 of each slot
  lds_int* free_lds = full_lds + NS;                       // [NS] number of consumer waves done with the slot (monotonic)
  lds_int* enx_lds = (lds_int*)enx;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long row0 = (long)blockIdx.x * ST_ROWS;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();                                         // the only workgroup barrier of the kernel

  if (wave == ST_WAVES) {
    // ---------------------------------------------- loader wave ------------------------------------------------------
    // chunk position (row i, cp) of a tile receives source chunk cp ^ (i & SWZ) of item row j0 + i
    for (int tile = 0; tile < n_tiles; ++tile) {
      const int slot = tile % NS;
      if (tile >= NS) {
        const int need = ST_WAVES * (tile / NS);
        while (st_peek(free_lds + slot) < need) __builtin_amdgcn_s_sleep(1);
      }
      const int j0 = tile * ST_TILE;
      unsigned char* dst = smem + slot * TILEB;
#pragma unroll
      for (int q = 0; q < PER_T; ++q) {
        const int P = q * 64 + lane;
        const int i = P / CPR, cp = P % CPR;
        int gi = j0 + i;
        gi = gi < I ? gi : I - 1;                          // clamp: values of padded columns are never used
        const _Float16* src = It + (long)gi * D + ((cp ^ (i & SWZ)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      if (tile >= LFL) {                                   // tile - LFL has landed once only LFL tiles remain outstanding
        st_wait_vmcnt<LFL * PER_T>();
        st_wave_fence();
        *(volatile lds_int*)(full_lds + (tile - LFL) % NS) = tile - LFL + 1;
      }
    }
    st_wait_vmcnt<0>();
    st_wave_fence();
    for (int tile = (n_tiles > LFL ? n_tiles - LFL : 0); tile < n_tiles; ++tile)
      *(volatile lds_int*)(full_lds + tile % NS) = tile + 1;
    return;
  }

  // ------------------------------------------------ consumer waves ------------------------------------------------------
  // A fragments: user row (32*wave + l31), k = 16*s + 8*half + j
  f16x8 afrag[KS];
  const long my_row = row0 + wave * 32 + l31;            // the user row this lane loads and whose exclusion list it walks
  {
    const long ur = my_row < Bu ? my_row : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) afrag[s] = src[2 * s + half];
  }
  exw[t] = 0u;
  // exclusion cursor of row l31 of this wave (eval/eval.py:219-220): lane l31 of the lower half walks user row l31's sorted
  // CSR row in step with the item tiles. e0 = next excluded item (register), the one after it sits in LDS (enx[t]) where
  // it is delivered by a 4-byte LDS-DMA: inside the main loop there is NO ordinary global load (hipcc would otherwise
  // insert s_waitcnt vmcnt(0) at every iteration).
  long eidx = 0, eend = 0;              // CSR position of the entry held in enx[t]; end of the row
  int e0 = 0x7FFFFFFF;
  {
    int e1 = 0x7FFFFFFF;
    if (half == 0 && my_row < Bu && excl_indptr) {
      const long u = u_idx ? u_idx[my_row] : my_row;
      long lo = excl_indptr[u];
      eend = excl_indptr[u + 1];
      long hi = eend;
      while (lo < hi) {                                    // first entry >= item_offset (item-sharded catalogues)
        const long mid = (lo + hi) >> 1;
        if (excl_indices[mid] < item_offset) lo = mid + 1; else hi = mid;
      }
      if (lo < eend) e0 = excl_indices[lo];
      if (lo + 1 < eend) e1 = excl_indices[lo + 1];
      eidx = lo + 1;
    }
    enx[t] = e1;
  }
  bool e_pending = false;               // wave-uniform: an exclusion look-ahead DMA of this wave may still be in flight
  // per-row state replicated in the 32 lanes that see the row's accumulators: threshold and buffer fill count of the rows
  // (r & 3) + 8 * (r >> 2) + 4 * half, r = 0..15
  float thr[16];
  int fill[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { thr[r] = -INFINITY; fill[r] = 0; }

  unsigned long long t_wait = 0, t_evt = 0, t_ovf = 0, n_evt = 0, n_ovf = 0;
  const unsigned long long t_begin = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int tl = 0; tl < n_tiles; ++tl) {
    const int slot = tl % NS;
    const unsigned long long tw0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    while (st_peek(full_lds + slot) != tl + 1) __builtin_amdgcn_s_sleep(1);
    st_wave_fence();
    if constexpr (DBG == 4) t_wait += __builtin_amdgcn_s_memtime() - tw0;
    const unsigned char* cur = smem + slot * TILEB;

    f32x16 acc[2];     // NJ of them are used (a template-dependent array bound makes hipcc drop the kernel's host stub)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nj][r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        const int i = nj * 32 + l31;                       // item row within the tile
        const int c = 2 * s + half;                        // 16-byte chunk: k = 16 s + 8 half .. +7
        const f16x8 b = *reinterpret_cast<const f16x8*>(cur + i * ROWB + ((c ^ (i & SWZ)) << 4));
        acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[s], b, acc[nj], 0, 0, 0);
      }
    }
    // the tile's LDS reads have been consumed by the MFMAs above: hand the slot back to the loader
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(acc[0]), "v"(acc[NJ - 1]) : "memory");
    if (lane == 0) atomicAdd((int*)(free_lds + slot), 1);

    // exclusions of this tile (eval.py:219-220): an excluded (row, column) pair is delivered as ONE bit to the lane that
    // holds that accumulator: lane (col & 31) + 32 * ((row >> 2) & 1), bit (col >> 5) * 16 + (row & 3) + 4 * (row >> 3).
    const int j0 = tl * ST_TILE;
    const int gbase = item_offset + j0;
    bool wrote_ex = false;
    for (int round = 0;; ++round) {
      const bool take = e0 < gbase + ST_TILE;              // only lower-half lanes of valid rows ever hold a finite e0
      if (!__ballot(take)) break;
      // the look-ahead entry in LDS must have landed before it is shifted in
      if (e_pending) st_wait_vmcnt<0>();
      e_pending = false;
      wrote_ex = true;
      if (take) {
        const int col = e0 - gbase;
        const int tgt = wave * 64 + (col & 31) + 32 * ((l31 >> 2) & 1);
        atomicOr(&exw[tgt], 1u << ((col >> 5) * 16 + (l31 & 3) + 4 * (l31 >> 3)));
        st_wave_fence();
        e0 = enx_lds[t];                                   // ds_read_b32 (explicit LDS address space)
        ++eidx;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // e0 is in its register before the slot is refilled
        if (eidx < eend) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(excl_indices + eidx),
                                           (__attribute__((address_space(3))) void*)(enx + wave * 64), 4, 0, 0);
        } else {
          enx_lds[t] = 0x7FFFFFFF;
          st_wave_fence();
        }
      }
      e_pending = true;
    }
    if constexpr (DBG == 1) { asm volatile("" ::"v"(acc[0]), "v"(acc[NJ - 1])); continue; }
    // epilogue: threshold filter. Per 32x32 accumulator tile: 16 v_cmp whose ballots stay in SGPRs, OR-reduced to one branch.
    // Survivors are appended to their row's buffer at positions derived from the ballot (v_mbcnt prefix count; no LDS
    // atomics, no round trip: the append is a fire-and-forget ds_write).
    const bool have_ex = __ballot(wrote_ex) != 0ull;
    unsigned int ex = 0u;
    bool ex_loaded = false;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      unsigned long long br[16];
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { br[r] = __ballot(acc[nj][r] > thr[r]); any |= br[r]; }
      if constexpr (DBG == 2) { if (any) asm volatile("s_nop 0"); continue; }
      if (!any) continue;
      if (have_ex && !ex_loaded) { st_wave_fence(); ex = exw[t]; ex_loaded = true; }
      const int item = j0 + nj * 32 + l31;
      const bool in_range = item < I;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // the per-row registers are read into scalars, updated, and written back unconditionally: array elements are
        // never assigned inside a branch (keeps thr[] / fill[] in fixed registers without whole-array copies)
        float thr_r = thr[r];
        int fill_r = fill[r];
        if (br[r]) {                                                     // SGPR test: no VALU work on the common path
          const unsigned long long te0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
          const float v = acc[nj][r];
          // re-check against the current threshold (an earlier step of this tile may have raised it) and the exclusions
          const bool cand = in_range && (v > thr_r) && !((ex >> (nj * 16 + r)) & 1u);
          const unsigned long long bal = __ballot(cand);
          const unsigned int bal_lo = (unsigned int)bal, bal_hi = (unsigned int)(bal >> 32);
          const int n_lo = __popc(bal_lo), n_hi = __popc(bal_hi);        // SALU
          const int below = (int)__builtin_amdgcn_mbcnt_hi(bal_hi, __builtin_amdgcn_mbcnt_lo(bal_lo, 0u));
          const int pos = fill_r + (half ? below - n_lo : below);       // rank among the candidates of MY row (my half)
          const int lrow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const unsigned long long key = ((unsigned long long)st_f2key(v) << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (unsigned)(item_offset + item));
          bool pending = cand;
          if (cand && pos < cap) { st.buf[lrow * cap + pos] = key; pending = false; }
          fill_r += half ? n_hi : n_lo;
          if (__ballot(pending)) {
            const unsigned long long to0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
            const RowState rs = st_overflow(st, v, pending, lrow, key, thr_r, fill_r, lane);
            thr_r = rs.thr;
            fill_r = rs.cnt;
            if constexpr (DBG == 4) { t_ovf += __builtin_amdgcn_s_memtime() - to0; ++n_ovf; }
          }
          if constexpr (DBG == 4) { t_evt += __builtin_amdgcn_s_memtime() - te0; ++n_evt; }
        }
        thr[r] = thr_r;
        fill[r] = fill_r;
      }
    }
    if (have_ex) { exw[t] = 0u; st_wave_fence(); }
  }

  if constexpr (DBG == 4) {
    if (lane == 0 && dbgbuf) {
      unsigned long long* d = dbgbuf + ((long)blockIdx.x * ST_WAVES + wave) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_evt; d[3] = t_ovf; d[4] = n_evt; d[5] = n_ovf;
    }
  }
  // final compaction + output of the wave-owned rows: row (r & 3) + 8 * (r >> 2) + 4 * h has its fill count in register
  // slot r of the lanes of half h
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int lrow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long ur = row0 + lrow;
      if (ur >= Bu) continue;
      int n = __shfl(fill[r], 32 * h, 64);
      n = n < cap ? n : cap;
      st_compact(st, lrow, n, lane);
      st_wave_fence();
      if (n > k) n = k;
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
}

// =====================================================================================================================
// Wide kernel (D <= 128): 64 users per consumer wave, candidate buffers in global memory.
//
// In the kernel above a wave owns 32 users, so every MFMA (32 cycles) needs a fresh 1 KiB B fragment from LDS — four SIMDs
// ask for 128 B / cycle, the whole LDS bandwidth of the CU (MFMA-only ablation: 37 % of the fp16 peak) — and 100k users are
// 1.75 rounds of 224-user workgroups. Here a wave keeps TWO 32-user A fragment sets in registers and feeds each B fragment
// to two MFMAs (half the LDS traffic per flop), and a workgroup covers 448 users (100k users = 224 workgroups = one round).
// 448 candidate buffers do not fit in LDS, so they live in a global workspace (S3_CAP entries of 8 B per user, touched only
// by the owning wave, L2-resident). Hot path per accumulator register: v_cmp against the row's threshold; survivors are
// appended with fire-and-forget global stores at ballot-derived positions. There is no overflow handling on the hot path:
// before the MFMAs of a tile every row is guaranteed 64 free entries (one tile can add at most 64 candidates to a row).
// That guarantee is kept by a cold maintenance step at the top of the tile loop (taken when a lane's high-water mark says
// some row is above S3_CAP - 64): the owning wave compacts those rows — 64 lanes load the <= 128 entries, rank them by
// counting over v_readlane broadcasts (no LDS traffic), store the k best back sorted — and raises their thresholds.
// Every lane walks the exclusion CSR row of one of the wave's 64 users (two bit words per lane). Loader wave, LDS-DMA ring,
// XOR swizzle and the ordering rule (score desc, item index asc) are those described at the top of this file.
// =====================================================================================================================
#define S3_WAVES 7
#define S3_ROWS (S3_WAVES * 64)
#define S3_THREADS ((S3_WAVES + 1) * 64)
#define S3_CAP 128                       // candidate buffer entries per user (global workspace)

// All 64 lanes of the owning wave: keep the k best of the first n (<= 128, wave-uniform) entries of the global row buffer b
// (lane l holds entries l and l + 64) at b[0..min(n, k)), UNSORTED, and return the new threshold (-inf while fewer than k
// entries exist). Selection, not sorting: the k-th largest score key is found by a bitwise binary search over ballot counts
// (32 steps of v_cmp + s_bcnt1, ~600 cycles; a full ranking by counting costs ~10 k), ties at that key are resolved on the item
// half of the composite the same way (rare). e / keep: the lane's two entries and whether they survived.
__device__ __forceinline__ float s3_select(unsigned long long* b, int n_any, int k, int lane, unsigned long long e[2],
                                           bool keep[2]) {
  const int n = __builtin_amdgcn_readfirstlane(n_any);
  // The buffer is written and read by this wave only: its stores and loads pass through the same CU's vector memory path in
  // order, so a workgroup-scope fence (compiler ordering + store drain; no L2 write-back — an agent-scope release costs
  // ~18 us here because the XCDs' L2s are not coherent with each other) is all the synchronisation the reload needs.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  e[0] = lane < n ? b[lane] : 0ull;
  e[1] = lane + 64 < n ? b[lane + 64] : 0ull;
  keep[0] = lane < n;
  keep[1] = lane + 64 < n;
  if (n < k) return -INFINITY;                              // nothing to drop: entries stay where they are
  const unsigned int h0 = (unsigned int)(e[0] >> 32), h1 = (unsigned int)(e[1] >> 32);
  unsigned int T = 0u;                                       // k-th largest score key (real keys are > 0, empties are 0)
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    const int cnt = __popcll(__ballot(h0 >= trial)) + __popcll(__ballot(h1 >= trial));
    T = cnt >= k ? trial : T;
  }
  unsigned long long C = (unsigned long long)T << 32;
  const int c_ge = __popcll(__ballot(h0 >= T)) + __popcll(__ballot(h1 >= T));
  if (c_ge != k) {
    // several entries share the k-th key: of those, the ones with the largest low words (= smallest item indices) stay
    const int need = k - (__popcll(__ballot(h0 > T)) + __popcll(__ballot(h1 > T)));
    const unsigned int l0 = (unsigned int)e[0], l1 = (unsigned int)e[1];
    unsigned int Lw = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = Lw | (1u << bit);
      const int cnt = __popcll(__ballot(h0 == T && l0 >= trial)) + __popcll(__ballot(h1 == T && l1 >= trial));
      Lw = cnt >= need ? trial : Lw;
    }
    C |= (unsigned long long)Lw;
  }
  keep[0] = e[0] >= C;
  keep[1] = e[1] >= C;
  const unsigned long long b0 = __ballot(keep[0]), b1 = __ballot(keep[1]);
  const int p0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b0, 0u));
  const int p1 = __popcll(b0) + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b1, 0u));
  if (keep[0]) b[p0] = e[0];
  if (keep[1]) b[p1] = e[1];
  return st_key2f(T);
}

template <int KS, int NS, int NJ, int DBG>   // KS = D / 16; NS = LDS ring slots; NJ = 32-item accumulator tiles per LDS tile; DBG: ablations
__global__ __launch_bounds__(S3_THREADS, 2) void score_topk_f16_wide_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const long* __restrict__ u_idx,
    const long* __restrict__ excl_indptr, const int* __restrict__ excl_indices, int item_offset, int k,
    float* __restrict__ out_val, int* __restrict__ out_idx, unsigned long long* __restrict__ gbuf,
    unsigned long long* __restrict__ dbgbuf) {
  constexpr int D = KS * 16;
  constexpr int ST_TILE = 32 * NJ;
  constexpr int LIMIT = S3_CAP - ST_TILE;                  // fill above which a row is compacted before the next tile
  constexpr int ROWB = D * 2;
  constexpr int TILEB = ST_TILE * ROWB;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_T = (ST_TILE * CPR) / 64;
  constexpr int LFL0 = NS > 2 ? NS - 2 : 1;
  // tiles in flight behind the awaited one: the vmcnt field has 6 bits (D = 128: 3 tiles of 16 DMA instructions; measured with
  // 2 instead of 3: the MFMA-only ablation does not move, 1.27 vs 1.34 ms — the loader's depth is not what bounds the main loop)
  constexpr int LFL = LFL0 * PER_T <= 63 ? LFL0 : 63 / PER_T;
  static_assert(LFL >= 1 && LFL * PER_T <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* exw = reinterpret_cast<unsigned int*>(smem + NS * TILEB);                            // [S3_WAVES*64][2]
  int* enx = reinterpret_cast<int*>(exw + S3_WAVES * 64 * 2);                                        // [S3_WAVES*64]
  int* tab_fill = enx + S3_WAVES * 64;                   // [S3_WAVES*64] maintenance step: fill count / new threshold per row
  float* tab_thr = reinterpret_cast<float*>(tab_fill + S3_WAVES * 64);
  lds_int* full_lds = (lds_int*)(tab_thr + S3_WAVES * 64);
  lds_int* free_lds = full_lds + NS;
  lds_int* enx_lds = (lds_int*)enx;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long row0 = (long)blockIdx.x * S3_ROWS;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();                                         // the only workgroup barrier of the kernel

  if (wave == S3_WAVES) {
    // ---------------------------------------------- loader wave ------------------------------------------------------
    for (int tile = 0; tile < n_tiles; ++tile) {
      const int slot = tile % NS;
      if (tile >= NS) {
        const int need = S3_WAVES * (tile / NS);
        while (st_peek(free_lds + slot) < need) __builtin_amdgcn_s_sleep(1);
      }
      const int j0 = tile * ST_TILE;
      unsigned char* dst = smem + slot * TILEB;
#pragma unroll
      for (int q = 0; q < PER_T; ++q) {
        const int P = q * 64 + lane;
        const int i = P / CPR, cp = P % CPR;
        int gi = j0 + i;
        gi = gi < I ? gi : I - 1;
        const _Float16* src = It + (long)gi * D + ((cp ^ (i & SWZ)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      if (tile >= LFL) {
        st_wait_vmcnt<LFL * PER_T>();
        st_wave_fence();
        *(volatile lds_int*)(full_lds + (tile - LFL) % NS) = tile - LFL + 1;
      }
    }
    st_wait_vmcnt<0>();
    st_wave_fence();
    for (int tile = (n_tiles > LFL ? n_tiles - LFL : 0); tile < n_tiles; ++tile)
      *(volatile lds_int*)(full_lds + tile % NS) = tile + 1;
    return;
  }

  // ------------------------------------------------ consumer waves ------------------------------------------------------
  // A fragments of the wave's two 32-user tiles: user row 64 * wave + 32 * mt + l31, k = 16 s + 8 half + j
  f16x8 afrag[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long r = row0 + wave * 64 + mt * 32 + l31;
    const long ur = r < Bu ? r : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) afrag[mt][s] = src[2 * s + half];
  }
  unsigned long long* gb = gbuf + row0 * S3_CAP;         // candidate buffers of this workgroup's rows
  exw[2 * t] = 0u;
  exw[2 * t + 1] = 0u;
  // exclusion cursor: lane L walks the sorted CSR row of user 64 * wave + L in step with the item tiles (see the kernel above)
  const long my_row = row0 + wave * 64 + lane;
  long eidx = 0, eend = 0;
  int e0 = 0x7FFFFFFF;
  {
    int e1 = 0x7FFFFFFF;
    if (my_row < Bu && excl_indptr) {
      const long u = u_idx ? u_idx[my_row] : my_row;
      long lo = excl_indptr[u];
      eend = excl_indptr[u + 1];
      long hi = eend;
      while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (excl_indices[mid] < item_offset) lo = mid + 1; else hi = mid;
      }
      if (lo < eend) e0 = excl_indices[lo];
      if (lo + 1 < eend) e1 = excl_indices[lo + 1];
      eidx = lo + 1;
    }
    enx[t] = e1;
  }
  st_wave_fence();
  bool e_pending = false;
  // thresholds of the rows whose accumulators this lane holds: tile mt, register r <-> row 32 mt + (r & 3) + 8 (r >> 2) + 4 half
  float thr[2][16];
  unsigned int fillp[2][4];             // fill counts of the same rows, 8 bits each (<= S3_CAP + 63 < 256): register r in byte r & 3 of word r >> 2
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) thr[mt][r] = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) fillp[mt][q] = 0u;
  }
  int hw = 0;                           // the largest fill count among this lane's rows
  int* wfill = tab_fill + wave * 64;
  float* wthr = tab_thr + wave * 64;
  unsigned long long* wgb = gb + (long)wave * 64 * S3_CAP;            // wave-uniform
  const unsigned int hoff = (unsigned int)half * (4u * S3_CAP * 8u);

  unsigned long long t_wait = 0, t_evt = 0, n_evt = 0, n_ins = 0, n_cand = 0, t_cmp = 0;
  const unsigned long long t_begin = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int tl = 0; tl < n_tiles; ++tl) {
    const int slot = tl % NS;
    if (__ballot(hw > LIMIT)) {
      // ---- maintenance (cold): compact the rows above LIMIT so that this tile's appends cannot overflow
      const unsigned long long tm0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wfill[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = (int)((fillp[mt][r >> 2] >> (8 * (r & 3))) & 0xFFu);
      }
      st_wave_fence();
      const int myfill = wfill[lane];
      unsigned long long need = __ballot(myfill > LIMIT);
      const unsigned long long done = need;
      while (need) {
        const int row = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        const int n = __builtin_amdgcn_readlane(myfill, row);
        unsigned long long e[2];
        bool kp[2];
        const float nt = s3_select(wgb + (long)row * S3_CAP, n, k, lane, e, kp);
        if (lane == 0) { wthr[row] = nt; wfill[row] = n < k ? n : k; }
        if constexpr (DBG == 4) ++n_ins;
      }
      st_wave_fence();
      hw = 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fillp[mt][q] = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const bool d = (done >> row) & 1ull;
          const float nt = wthr[row];
          const int nf = wfill[row];                                   // unchanged rows read back what was written above
          thr[mt][r] = d ? nt : thr[mt][r];
          fillp[mt][r >> 2] |= (unsigned int)nf << (8 * (r & 3));
          hw = nf > hw ? nf : hw;
        }
      }
      if constexpr (DBG == 4) t_wait += 0 * tm0, t_cmp += __builtin_amdgcn_s_memtime() - tm0;
    }
    const unsigned long long tw0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    while (st_peek(full_lds + slot) != tl + 1) __builtin_amdgcn_s_sleep(1);
    st_wave_fence();
    if constexpr (DBG == 4) t_wait += __builtin_amdgcn_s_memtime() - tw0;
    const unsigned char* cur = smem + slot * TILEB;

    f32x16 acc[2][2];     // [2][NJ] used (a template-dependent array bound makes hipcc drop the host stub)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nj][r] = 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        const int i = nj * 32 + l31;
        const int c = 2 * s + half;
        const f16x8 b = *reinterpret_cast<const f16x8*>(cur + i * ROWB + ((c ^ (i & SWZ)) << 4));
        acc[0][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[0][s], b, acc[0][nj], 0, 0, 0);
        acc[1][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[1][s], b, acc[1][nj], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(acc[0][0]), "v"(acc[0][NJ - 1]), "v"(acc[1][0]), "v"(acc[1][NJ - 1]) : "memory");
    if (lane == 0) atomicAdd((int*)(free_lds + slot), 1);

    // exclusions of this tile: one bit per excluded (row, column) for the lane that holds that accumulator — lane
    // (col & 31) + 32 ((row >> 2) & 1), word mt = row >> 5, bit (col >> 5) * 16 + (row & 3) + 4 ((row & 31) >> 3)
    const int j0 = tl * ST_TILE;
    const int gbase = item_offset + j0;
    bool wrote_ex = false;
    for (int round = 0;; ++round) {
      const bool take = e0 < gbase + ST_TILE;
      if (!__ballot(take)) break;
      if (e_pending) st_wait_vmcnt<0>();
      e_pending = false;
      wrote_ex = true;
      if (take) {
        const int col = e0 - gbase;
        const int tgt = wave * 64 + (col & 31) + 32 * ((l31 >> 2) & 1);
        atomicOr(&exw[2 * tgt + half], 1u << ((col >> 5) * 16 + (l31 & 3) + 4 * (l31 >> 3)));
        st_wave_fence();
        e0 = enx_lds[t];
        ++eidx;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (eidx < eend) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(excl_indices + eidx),
                                           (__attribute__((address_space(3))) void*)(enx + wave * 64), 4, 0, 0);
        } else {
          enx_lds[t] = 0x7FFFFFFF;
          st_wave_fence();
        }
      }
      e_pending = true;
    }
    if constexpr (DBG == 1) {
      asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][NJ - 1]), "v"(acc[1][0]), "v"(acc[1][NJ - 1]));
      continue;
    }
    const bool have_ex = __ballot(wrote_ex) != 0ull;
    unsigned int ex[2] = {0u, 0u};
    bool ex_loaded = false;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        unsigned long long br[16];
        unsigned long long any = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { br[r] = __ballot(acc[mt][nj][r] > thr[mt][r]); any |= br[r]; }
        if constexpr (DBG == 2) { if (any) asm volatile("s_nop 0"); continue; }
        if (!any) continue;
        if (have_ex && !ex_loaded) { st_wave_fence(); ex[0] = exw[2 * t]; ex[1] = exw[2 * t + 1]; ex_loaded = true; }
        const bool in_range = j0 + nj * 32 + l31 < I;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float thr_r = thr[mt][r];
          if (br[r]) {                                                   // SGPR test: no VALU work on the common path
            const unsigned long long te0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
            const float v = acc[mt][nj][r];
            const bool cand = in_range && (v > thr_r) && !((ex[mt] >> (nj * 16 + r)) & 1u);
            const unsigned long long bal = __ballot(cand);
            const unsigned int bal_lo = (unsigned int)bal, bal_hi = (unsigned int)(bal >> 32);
            const int n_lo = __popc(bal_lo), n_hi = __popc(bal_hi);
            const int below = (int)__builtin_amdgcn_mbcnt_hi(bal_hi, __builtin_amdgcn_mbcnt_lo(bal_lo, 0u));
            int fill_r = (int)((fillp[mt][r >> 2] >> (8 * (r & 3))) & 0xFFu);
            const int pos = fill_r + (half ? below - n_lo : below);      // < S3_CAP: the row had 64 free entries at tile start
            const unsigned long long key = ((unsigned long long)st_f2key(v) << 32) |
                                           (unsigned long long)(0xFFFFFFFFu - (unsigned)(item_offset + j0 + nj * 32 + l31));
            // row mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half: uniform base + 32-bit lane offset (no per-row 64-bit pointers)
            const unsigned int off = hoff + ((unsigned int)pos << 3) + (unsigned int)((mt * 32 + (r & 3) + 8 * (r >> 2)) * S3_CAP * 8);
            if (cand) *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(wgb) + off) = key;
            const int add = half ? n_hi : n_lo;
            fill_r += add;
            fillp[mt][r >> 2] += (unsigned int)add << (8 * (r & 3));
            hw = fill_r > hw ? fill_r : hw;
            if constexpr (DBG == 4) { n_cand += __popcll(bal); t_evt += __builtin_amdgcn_s_memtime() - te0; ++n_evt; }
          }
        }
      }
    }
    if (have_ex) { exw[2 * t] = 0u; exw[2 * t + 1] = 0u; st_wave_fence(); }
  }

  if constexpr (DBG == 4) {
    if (lane == 0 && dbgbuf) {
      unsigned long long* d = dbgbuf + ((long)blockIdx.x * S3_WAVES + wave) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_evt; d[3] = n_cand; d[4] = n_evt; d[5] = n_ins; d[6] = t_cmp;
    }
  }
  // final compaction + output of the wave's 64 rows: the lane that holds the entry of rank j writes output position j
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      wfill[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = (int)((fillp[mt][r >> 2] >> (8 * (r & 3))) & 0xFFu);
  }
  st_wave_fence();
  const int myfill = wfill[lane];
  for (int row = 0; row < 64; ++row) {
    const long ur = row0 + wave * 64 + row;
    if (ur >= Bu) break;
    const int n = __builtin_amdgcn_readlane(myfill, row);
    unsigned long long e[2];
    bool kp[2];
    s3_select(wgb + (long)row * S3_CAP, n, k, lane, e, kp);
    // rank the (at most k) survivors among themselves: broadcast each of them with v_readlane
    int rk[2] = {0, 0};
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const int h32 = (int)(e[part] >> 32), l32 = (int)e[part];
      for (unsigned long long m = __ballot(kp[part]); m; m &= m - 1ull) {
        const int j = __ffsll((long long)m) - 1;
        const unsigned long long kj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(h32, j) << 32) |
                                      (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(l32, j);
        rk[0] += kj > e[0];
        rk[1] += kj > e[1];
      }
    }
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      if (kp[part]) {
        out_val[ur * k + rk[part]] = st_key2f((unsigned int)(e[part] >> 32));
        out_idx[ur * k + rk[part]] = (int)(0xFFFFFFFFu - (unsigned int)(e[part] & 0xFFFFFFFFull));
      }
    }
    if (lane >= n && lane < k) {                             // fewer than k candidates: empty slots behind them
      out_val[ur * k + lane] = -INFINITY;
      out_idx[ur * k + lane] = -1;
    }
  }
}

// =====================================================================================================================
// Transposed kernel (D <= 128, round 2): users on the LANES, items in the accumulator REGISTERS.
//
// The wide kernel above computes S = U x I^T tiles: a lane holds one item column of 16 user rows, so a row's candidates are
// spread over 32 lanes and every append needs a wave ballot, prefix counts and per-row fill bookkeeping (178 cycles per
// candidate block, 16 + 16 threshold / fill registers per 32-user tile, scalar branch ladders over 64 ballots per tile).
// Swapping the MFMA operands (A = item fragment, B = user fragment: the LDS image and the register fragments are the same
// bytes, the operands of v_mfma_f32_32x32x16_f16 trade places) yields S^T tiles: lane (u, h) holds, for ONE user u of the
// 32-user tile, the scores of the 16 items (r & 3) + 8 (r >> 2) + 4 h of every 32-item tile. Top-k state becomes lane-local:
//   * one threshold and one fill count per lane and user tile (4 VGPRs instead of 40);
//   * a candidate is appended by its own lane to its own buffer half (global workspace, S4_CAPH entries per (user, h)) with a
//     fire-and-forget store at base + fill: no ballot, no prefix count, no cross-lane traffic;
//   * only compaction is cooperative: the wave loads both halves of a user's buffer (<= 128 entries, two per lane), selects
//     the k best by bitwise binary search over ballot counts (s4_select), stores them back into half 0 and hands the new
//     threshold to the user's two lanes.
// Prefilter pass. A streaming top-k meets most of its candidates early (k ln(I / k) ~ 160 per user with perfect thresholds,
// 344 measured with thresholds refreshed only at compaction; two thirds of them in the first 10 % of the catalogue). The
// kernel therefore first streams a PREFIX of the catalogue (n_pre tiles, ~8 %) keeping only a running maximum per
// accumulator register (64 item classes per user: one v_max per score, no candidates). The k-th largest of a user's 64 class
// maxima is the score of an actual item with at least k - 1 distinct items above it: a safe lower bound of the user's final
// k-th best (its expected rank among the prefix items is ~24 for k = 20). The main pass then starts from tile 0 with that
// bound as threshold (inclusive: the bounding items themselves must come back), so the prefix contributes ~24 candidates per
// user instead of ~170. Exclusions and the catalogue end are honoured in both passes; results are exact and ordered by
// (score desc, item index asc) like every other path.
// Measured (100k x 50k x 128, top-20): 2.61 ms against 3.54 ms of the wide kernel (no exclusions; 2.68 / 3.69 ms with 50 per
// user); per wave 105 candidates per user instead of 344, 2.3 compactions per user instead of 5.4. In-kernel clock 2.2 GHz; the
// MFMA-only ablation of the same loop runs at 1.83 GHz and 74 % of the matrix pipe (DVFS: 1.29 ms), compares included 1.70 ms.
// Tried on top and dropped: software-pipelining the ladder of tile t with the MFMAs of tile t + 1 inside a wave (two
// accumulator sets, MFMA pairs issued in front of every ladder group, B fragments three groups ahead): 3.06 ms — an in-order
// wave stalls at every MFMA while the SIMD's other wave holds the matrix pipe, so the ladder is delayed instead of hidden, and
// the second accumulator set costs 17 spilled VGPRs; an 8-slot LDS ring instead of 6: 1 %.
// =====================================================================================================================
#define S4_WAVES 7
#define S4_ROWS (S4_WAVES * 64)
#define S4_THREADS ((S4_WAVES + 1) * 64)


template <int KS, int NS, int NJ, int DBG, bool PRE>   // KS = D / 16; NS = LDS ring slots; NJ = 32-item accumulator tiles per LDS tile; DBG: ablations; PRE: prefix pass compiled in
__global__ __launch_bounds__(S4_THREADS, 2) void score_topk_f16_t_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const long* __restrict__ u_idx,
    const long* __restrict__ excl_indptr, const int* __restrict__ excl_indices, int item_offset, int k, int n_pre,
    float* __restrict__ out_val, int* __restrict__ out_idx, unsigned long long* __restrict__ gbuf,
    unsigned long long* __restrict__ dbgbuf) {
  constexpr int D = KS * 16;
  constexpr int ST_TILE = 32 * NJ;
  constexpr int LIMIT = S4_CAPH - 16 * NJ;                 // a tile adds at most 16 NJ entries to a (user, half) buffer
  constexpr int ROWB = D * 2;
  constexpr int TILEB = ST_TILE * ROWB;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_T = (ST_TILE * CPR) / 64;
  constexpr int LFL0 = NS > 2 ? NS - 2 : 1;
  constexpr int LFL = LFL0 * PER_T <= 63 ? LFL0 : 63 / PER_T;
  static_assert(LFL >= 1 && LFL * PER_T <= 63, "vmcnt field");
  static_assert(LIMIT >= 32, "k <= 32 entries must fit below the compaction limit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* exw = reinterpret_cast<unsigned int*>(smem + NS * TILEB);                            // [S4_WAVES*64][2]
  int* enx = reinterpret_cast<int*>(exw + S4_WAVES * 64 * 2);                                        // [S4_WAVES*64]
  lds_int* full_lds = (lds_int*)(enx + S4_WAVES * 64);
  lds_int* free_lds = full_lds + NS;
  lds_int* enx_lds = (lds_int*)enx;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long row0 = (long)blockIdx.x * S4_ROWS;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
  const int n_virt = n_pre + n_tiles;                      // tile sequence: prefix tiles 0 .. n_pre - 1, then all tiles

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();                                         // the only workgroup barrier of the kernel

  if (wave == S4_WAVES) {
    // ---------------------------------------------- loader wave ------------------------------------------------------
    for (int v = 0; v < n_virt; ++v) {
      const int slot = v % NS;
      if (v >= NS) {
        const int need = S4_WAVES * (v / NS);
        while (st_peek(free_lds + slot) < need) __builtin_amdgcn_s_sleep(1);
      }
      const int j0 = (v < n_pre ? v : v - n_pre) * ST_TILE;
      unsigned char* dst = smem + slot * TILEB;
#pragma unroll
      for (int q = 0; q < PER_T; ++q) {
        const int P = q * 64 + lane;
        const int i = P / CPR, cp = P % CPR;
        int gi = j0 + i;
        gi = gi < I ? gi : I - 1;
        const _Float16* src = It + (long)gi * D + ((cp ^ (i & SWZ)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      if (v >= LFL) {
        st_wait_vmcnt<LFL * PER_T>();
        st_wave_fence();
        *(volatile lds_int*)(full_lds + (v - LFL) % NS) = v - LFL + 1;
      }
    }
    st_wait_vmcnt<0>();
    st_wave_fence();
    for (int v = (n_virt > LFL ? n_virt - LFL : 0); v < n_virt; ++v)
      *(volatile lds_int*)(full_lds + v % NS) = v + 1;
    return;
  }

  // ------------------------------------------------ consumer waves ------------------------------------------------------
  // B-operand fragments of the wave's two 32-user tiles: user 64 * wave + 32 * mt + l31, k = 16 s + 8 half + j
  f16x8 ufrag[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long r = row0 + wave * 64 + mt * 32 + l31;
    const long ur = r < Bu ? r : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) ufrag[mt][s] = src[2 * s + half];
  }
  unsigned long long* wgb = gbuf + (row0 + (long)wave * 64) * (2 * S4_CAPH);      // wave-uniform: buffers of the wave's 64 users
  exw[2 * t] = 0u;
  exw[2 * t + 1] = 0u;
  // exclusion cursor: lane L walks the sorted CSR row of user 64 * wave + L (user tile L >> 5, user column L & 31) in step with
  // the item tiles; e0 = next excluded item, the one after it sits in LDS (see the first kernel). lo0: restart point of pass 2.
  const long my_row = row0 + wave * 64 + lane;
  long eidx = 0, eend = 0, lo0 = 0;
  int e0 = 0x7FFFFFFF;
  {
    int e1 = 0x7FFFFFFF;
    if (my_row < Bu && excl_indptr) {
      const long u = u_idx ? u_idx[my_row] : my_row;
      long lo = excl_indptr[u];
      eend = excl_indptr[u + 1];
      long hi = eend;
      while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (excl_indices[mid] < item_offset) lo = mid + 1; else hi = mid;
      }
      lo0 = lo;
      if (lo < eend) e0 = excl_indices[lo];
      if (lo + 1 < eend) e1 = excl_indices[lo + 1];
      eidx = lo + 1;
    }
    enx[t] = e1;
  }
  st_wave_fence();
  bool e_pending = false;
  // lane (u, h): threshold and buffer fill of user 32 mt + u, half h (thresholds of the two halves of a user are equal)
  float thr[2] = {-INFINITY, -INFINITY};
  int cnt[2] = {0, 0};
  const unsigned int lane_base = (unsigned int)((l31 * 2 + half) * S4_CAPH * 8);     // + mt * 32 * 2 * S4_CAPH * 8

  unsigned long long t_wait = 0, t_evt = 0, n_evt = 0, n_ins = 0, n_cand = 0, t_cmp = 0, t_issue = 0, t_ladder = 0;
  const unsigned long long t_begin = DBG != 0 ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_begin = DBG != 0 ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // one item tile: wait, MFMAs (S^T = I x U^T), slot release, exclusion bits of the tile -> acc, have_ex
#define S4_TILE_BODY(V)                                                                                                  \
    const int slot = (V) % NS;                                                                                           \
    const unsigned long long tw0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;                                       \
    while (st_peek(full_lds + slot) != (V) + 1) __builtin_amdgcn_s_sleep(1);                                             \
    st_wave_fence();                                                                                                     \
    if constexpr (DBG == 4) t_wait += __builtin_amdgcn_s_memtime() - tw0;                                                \
    const unsigned char* cur = smem + slot * TILEB;                                                                      \
    f32x16 acc[2][2];                                                                                                    \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                                   \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) {                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mt][nj][r] = 0.f;                                             \
      }                                                                                                                  \
    }                                                                                                                    \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                     \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) {                                                                \
        const int i = nj * 32 + l31;                                                                                     \
        const int c = 2 * s + half;                                                                                      \
        const f16x8 b = *reinterpret_cast<const f16x8*>(cur + i * ROWB + ((c ^ (i & SWZ)) << 4));                        \
        acc[0][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, ufrag[0][s], acc[0][nj], 0, 0, 0);                        \
        acc[1][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, ufrag[1][s], acc[1][nj], 0, 0, 0);                        \
      }                                                                                                                  \
    }                                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(acc[0][0]), "v"(acc[0][NJ - 1]), "v"(acc[1][0]), "v"(acc[1][NJ - 1]) : "memory"); \
    if (lane == 0) atomicAdd((int*)(free_lds + slot), 1);                                                                \
    /* exclusions of this tile: item column `col` of the tile, user (L >> 5, L & 31): one bit for the lane that holds that  \
       accumulator — lane (L & 31) + 32 ((col >> 2) & 1), word L >> 5, bit (col >> 5) * 16 + (col & 3) + 4 ((col & 31) >> 3) */ \
    const int gbase = item_offset + j0;                                                                                  \
    bool wrote_ex = false;                                                                                               \
    for (int round = 0;; ++round) {                                                                                      \
      const bool take = e0 < gbase + ST_TILE;                                                                            \
      if (!__ballot(take)) break;                                                                                        \
      if (e_pending) st_wait_vmcnt<0>();                                                                                 \
      e_pending = false;                                                                                                 \
      wrote_ex = true;                                                                                                   \
      if (take) {                                                                                                        \
        const int col = e0 - gbase;                                                                                      \
        const int tgt = wave * 64 + l31 + 32 * ((col >> 2) & 1);                                                         \
        atomicOr(&exw[2 * tgt + half], 1u << ((col >> 5) * 16 + (col & 3) + 4 * ((col & 31) >> 3)));                     \
        st_wave_fence();                                                                                                 \
        e0 = enx_lds[t];                                                                                                 \
        ++eidx;                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        if (eidx < eend) {                                                                                               \
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(excl_indices + eidx),         \
                                           (__attribute__((address_space(3))) void*)(enx + wave * 64), 4, 0, 0);         \
        } else {                                                                                                         \
          enx_lds[t] = 0x7FFFFFFF;                                                                                       \
          st_wave_fence();                                                                                               \
        }                                                                                                                \
      }                                                                                                                  \
      e_pending = true;                                                                                                  \
    }                                                                                                                    \
    const bool have_ex = __ballot(wrote_ex) != 0ull;

  // ---- pass 1: prefix tiles, running maximum per accumulator register (item class) ----
  if (PRE && n_pre > 0) {
    f32x16 cm[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[mt][nj][r] = -INFINITY;
      }
    }
    for (int v = 0; v < n_pre; ++v) {
      const int j0 = v * ST_TILE;
      S4_TILE_BODY(v)
      if (have_ex) {                                       // excluded scores must not raise a class maximum
        st_wave_fence();
        const unsigned int ex0 = exw[2 * t], ex1 = exw[2 * t + 1];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[0][nj][r] = ((ex0 >> (nj * 16 + r)) & 1u) ? -INFINITY : acc[0][nj][r];
            acc[1][nj][r] = ((ex1 >> (nj * 16 + r)) & 1u) ? -INFINITY : acc[1][nj][r];
          }
        }
        exw[2 * t] = 0u;
        exw[2 * t + 1] = 0u;
        st_wave_fence();
      }
      if (j0 + ST_TILE > I) {                              // catalogue end inside the tile: padded columns do not count
        const int lim = I - j0 - 4 * half;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool in = nj * 32 + (r & 3) + 8 * (r >> 2) < lim;
            acc[0][nj][r] = in ? acc[0][nj][r] : -INFINITY;
            acc[1][nj][r] = in ? acc[1][nj][r] : -INFINITY;
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cm[mt][nj][r] = fmaxf(cm[mt][nj][r], acc[mt][nj][r]);
        }
      }
    }
    // k-th largest of each user's 32 NJ class maxima (16 NJ in each of its two lanes): bitwise binary search on the ordered
    // keys, every lane pair for its own user. The threshold admits scores EQUAL to the bound (its items are not in any buffer).
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      unsigned int T = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = T | (1u << bit);
        int c = 0;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c += st_f2key(cm[mt][nj][r]) >= trial;
        }
        c += __shfl_xor(c, 32, 64);
        T = c >= k ? trial : T;
      }
      // T = 0x007FFFFF is the key of -inf (fewer than k finite classes): no bound
      thr[mt] = T > 0x007FFFFFu ? st_key2f(T - 1u) : -INFINITY;
    }
    // restart the exclusion cursor for the main pass
    if (excl_indptr) st_wait_vmcnt<0>();
    e_pending = false;
    e0 = 0x7FFFFFFF;
    int e1 = 0x7FFFFFFF;
    if (lo0 < eend) e0 = excl_indices[lo0];
    if (lo0 + 1 < eend) e1 = excl_indices[lo0 + 1];
    eidx = lo0 + 1;
    enx_lds[t] = e1;
    st_wave_fence();
  }

  // ---- pass 2: all tiles, lane-local threshold filter and appends ----
  for (int tl = 0; tl < n_tiles; ++tl) {
    if (__ballot(cnt[0] > LIMIT || cnt[1] > LIMIT)) {
      // ---- maintenance (cold): compact the users with a half above LIMIT so that this tile's appends cannot overflow
      const unsigned long long tm0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        unsigned long long need = __ballot(cnt[mt] > LIMIT);
        need = (need | (need >> 32)) & 0xFFFFFFFFull;
        while (need) {
          const int u = __ffsll((long long)need) - 1;
          need &= need - 1ull;
          const int n0 = __builtin_amdgcn_readlane(cnt[mt], u), n1 = __builtin_amdgcn_readlane(cnt[mt], u + 32);
          unsigned long long* b0 = wgb + (long)(mt * 32 + u) * (2 * S4_CAPH);
          unsigned long long e[2];
          bool kp[2];
          const float nt = s4_select(b0, b0 + S4_CAPH, n0, n1, k, lane, e, kp);
          if (n0 + n1 >= k && l31 == u) {
            thr[mt] = nt;
            cnt[mt] = half ? 0 : k;
          }
          if constexpr (DBG == 4) ++n_ins;
        }
      }
      if constexpr (DBG == 4) t_cmp += __builtin_amdgcn_s_memtime() - tm0;
    }
    const int j0 = tl * ST_TILE;
    const int vseq = n_pre + tl;
    const unsigned long long ti0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    S4_TILE_BODY(vseq)
    const unsigned long long ti1 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    if constexpr (DBG == 4) t_issue += ti1 - ti0 - (__builtin_amdgcn_s_memtime() - ti1);
    if constexpr (DBG == 1) {
      asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][NJ - 1]), "v"(acc[1][0]), "v"(acc[1][NJ - 1]));
      continue;
    }
    unsigned int ex[2] = {0u, 0u};
    if (have_ex) { st_wave_fence(); ex[0] = exw[2 * t]; ex[1] = exw[2 * t + 1]; }
    if (j0 + ST_TILE > I) {                                // catalogue end inside the (last) tile: padded columns never qualify
      const int lim = I - j0 - 4 * half;                   // item (r & 3) + 8 (r >> 2) + 32 nj of this lane exists iff < lim
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool in = nj * 32 + (r & 3) + 8 * (r >> 2) < lim;
          acc[0][nj][r] = in ? acc[0][nj][r] : -INFINITY;
          acc[1][nj][r] = in ? acc[1][nj][r] : -INFINITY;
        }
      }
    }
    const unsigned int item_lane = 0xFFFFFFFFu - (unsigned int)(item_offset + j0 + 4 * half);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      char* bufp = reinterpret_cast<char*>(wgb) + mt * (32 * 2 * S4_CAPH * 8);
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // common path per FOUR accumulator registers: four v_cmp into SGPR pairs issued back to back, three s_or, one scalar
          // branch (a v_cmp -> branch pair per register serialises on the compare's latency: 27 cycles per register, 0.6 ms
          // per pass, measured with the compare-only ablation)
          unsigned long long bq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[q] = __ballot(acc[mt][nj][4 * g + q] > thr[mt]);
          if constexpr (DBG == 2) { if (bq[0] | bq[1] | bq[2] | bq[3]) asm volatile("s_nop 0"); continue; }
          if (bq[0] | bq[1] | bq[2] | bq[3]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = 4 * g + q;
              if (bq[q]) {
                const unsigned long long te0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
                // everything below hangs off values pinned inside the branch (hipcc otherwise if-converts the block and
                // evaluates exclusion / key arithmetic for every register of every tile)
                float v = acc[mt][nj][r];
                unsigned int exv = ex[mt];
                asm volatile("" : "+v"(v), "+v"(exv));
                const int C = nj * 32 + (r & 3) + 8 * (r >> 2);          // compile-time after unrolling
                const bool cand = (v > thr[mt]) && !((exv >> (nj * 16 + r)) & 1u);
                if (cand) {
                  const unsigned long long key = ((unsigned long long)st_f2key(v) << 32) | (unsigned long long)(item_lane - (unsigned int)C);
                  *reinterpret_cast<unsigned long long*>(bufp + lane_base + ((unsigned int)cnt[mt] << 3)) = key;
                  ++cnt[mt];
                }
                if constexpr (DBG == 4) { n_cand += __popcll(__ballot(cand)); t_evt += __builtin_amdgcn_s_memtime() - te0; ++n_evt; }
              }
            }
          }
        }
      }
    }
    if (have_ex) { exw[2 * t] = 0u; exw[2 * t + 1] = 0u; st_wave_fence(); }
    if constexpr (DBG == 4) t_ladder += __builtin_amdgcn_s_memtime() - ti1;
  }
#undef S4_TILE_BODY

  if constexpr (DBG != 0) {
    if (lane == 0 && dbgbuf) {
      unsigned long long* d = dbgbuf + ((long)blockIdx.x * S4_WAVES + wave) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_evt; d[3] = n_cand; d[4] = n_evt | (t_issue << 20); d[5] = n_ins | (t_ladder << 20); d[6] = t_cmp;
      d[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
  }
  // final selection + output of the wave's 64 users: the lane that holds the entry of rank j writes output position j
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    for (int u = 0; u < 32; ++u) {
      const long ur = row0 + wave * 64 + mt * 32 + u;
      if (ur >= Bu) break;
      const int n0 = __builtin_amdgcn_readlane(cnt[mt], u), n1 = __builtin_amdgcn_readlane(cnt[mt], u + 32);
      unsigned long long* b0 = wgb + (long)(mt * 32 + u) * (2 * S4_CAPH);
      unsigned long long e[2];
      bool kp[2];
      s4_select(b0, b0 + S4_CAPH, n0, n1, k, lane, e, kp);
      int rk[2] = {0, 0};
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        const int h32 = (int)(e[part] >> 32), l32 = (int)e[part];
        for (unsigned long long m = __ballot(kp[part]); m; m &= m - 1ull) {
          const int j = __ffsll((long long)m) - 1;
          const unsigned long long kj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(h32, j) << 32) |
                                        (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(l32, j);
          rk[0] += kj > e[0];
          rk[1] += kj > e[1];
        }
      }
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        if (kp[part]) {
          out_val[ur * k + rk[part]] = st_key2f((unsigned int)(e[part] >> 32));
          out_idx[ur * k + rk[part]] = (int)(0xFFFFFFFFu - (unsigned int)(e[part] & 0xFFFFFFFFull));
        }
      }
      const int n = n0 + n1;
      if (lane >= n && lane < k) {                           // fewer than k candidates: empty slots behind them
        out_val[ur * k + lane] = -INFINITY;
        out_idx[ur * k + lane] = -1;
      }
    }
  }
}

template <int KS, int NS, int NJ, bool PRE>
static int s4_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                     int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, hipStream_t s) {
  const long rows_padded = sbr_cdiv(Bu, S4_ROWS) * (long)S4_ROWS;
  const long need = rows_padded * 2 * S4_CAPH * 8;
  const int dbg = getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0;      // 1 | 2: timing-only ablations, 4: cycle stamps
  SBR_REQUIRE(workspace && workspace_bytes >= need + (dbg != 0 ? sbr_cdiv(Bu, S4_ROWS) * S4_WAVES * 64L : 0L),
              "sbr_score_topk_f16: workspace of %ld bytes needed (sbr_score_topk_f16_workspace), %ld given", need, workspace_bytes);
  void* dbg_buf = (char*)workspace + need;
  const size_t lds = (size_t)NS * (32 * NJ) * KS * 32 + S4_WAVES * 64 * 12 + 2 * NS * 4 + 16;
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  // prefix pass: ~1/12 of the catalogue (whole tiles), skipped for catalogues too short to repay it; SBR_ST_PRE overrides (tiles)
  const int n_tiles = sbr_cdiv(I, 32 * NJ);
  int n_pre = n_tiles >= 96 ? n_tiles / 12 : 0;
  if (getenv("SBR_ST_PRE")) n_pre = atoi(getenv("SBR_ST_PRE"));
  if (n_pre > n_tiles) n_pre = n_tiles;
  if (n_pre < 0 || !PRE) n_pre = 0;
  auto kern = dbg == 1 ? score_topk_f16_t_kernel<KS, NS, NJ, 1, PRE> : (dbg == 2 ? score_topk_f16_t_kernel<KS, NS, NJ, 2, PRE> :
              (dbg == 4 ? score_topk_f16_t_kernel<KS, NS, NJ, 4, PRE> : score_topk_f16_t_kernel<KS, NS, NJ, 0, PRE>));
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  kern<<<sbr_cdiv(Bu, S4_ROWS), S4_THREADS, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, u_idx, eptr, eidx, item_offset, k,
                                                      n_pre, out_val, out_idx, (unsigned long long*)workspace, (unsigned long long*)dbg_buf);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  return SBR_OK;
}

template <int KS, int NS, int NJ>
static int s3_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                     int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, hipStream_t s) {
  const long rows_padded = sbr_cdiv(Bu, S3_ROWS) * (long)S3_ROWS;
  const long need = rows_padded * S3_CAP * 8;
  SBR_REQUIRE(workspace && workspace_bytes >= need + (getenv("SBR_ST_DEBUG") && atoi(getenv("SBR_ST_DEBUG")) == 4 ? sbr_cdiv(Bu, S3_ROWS) * S3_WAVES * 64L : 0L),
              "sbr_score_topk_f16: workspace of %ld bytes needed (sbr_score_topk_f16_workspace), %ld given", need, workspace_bytes);
  void* dbg_buf = (char*)workspace + need;
  const size_t lds = (size_t)NS * (32 * NJ) * KS * 32 + S3_WAVES * 64 * 20 + 2 * NS * 4 + 16;
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  const int dbg = getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0;      // 1 | 2: timing-only ablations (see above)
  auto kern = dbg == 1 ? score_topk_f16_wide_kernel<KS, NS, NJ, 1> : (dbg == 2 ? score_topk_f16_wide_kernel<KS, NS, NJ, 2> :
              (dbg == 4 ? score_topk_f16_wide_kernel<KS, NS, NJ, 4> : score_topk_f16_wide_kernel<KS, NS, NJ, 0>));
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  kern<<<sbr_cdiv(Bu, S3_ROWS), S3_THREADS, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, u_idx, eptr, eidx,
                                                      item_offset, k, out_val, out_idx, (unsigned long long*)workspace, (unsigned long long*)dbg_buf);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  return SBR_OK;
}
static int st_cap(int k) { int c = 2 * k; if (c < k + 16) c = k + 16; if (c > 64) c = 64; return c; }

static bool st_use_wide(int D) {
  const bool v1 = getenv("SBR_SCORER_V1") && atoi(getenv("SBR_SCORER_V1")) != 0;
  return !v1 && (D == 128 || D == 64);
}

// bytes of the candidate-buffer workspace (wide kernel: S3_CAP entries per user, users padded to whole workgroups; + the
// cycle stamps of SBR_ST_DEBUG=4). D = 0: the largest over all D.
extern "C" long sbr_score_topk_f16_workspace(long Bu, int I, int k, long excl_nnz) {
  (void)I; (void)k;
  const long wgs = sbr_cdiv(Bu, S3_ROWS);
  const long older = wgs * S3_ROWS * (long)S3_CAP * 8 + wgs * S3_WAVES * 64L;
  const long narrow = s5_workspace_bytes(Bu, excl_nnz);
  return older > narrow ? older : narrow;
}

template <int KS, int NS, int ST_TILE>
static int st_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                     int item_offset, int k, float* out_val, int* out_idx, void* dbg_buf, hipStream_t s) {
  const int cap = st_cap(k);
  const size_t lds = (size_t)NS * ST_TILE * KS * 32 + (size_t)ST_ROWS * cap * 8 + ST_WAVES * 64 * 8 + 2 * NS * 4 + 16;
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  // SBR_ST_DEBUG=1|2 selects timing-only ablation builds (1: MFMA main loop only, 2: + threshold compares; results are
  // meaningless), 4 adds per-wave cycle stamps written to `workspace`. Unset / 0 = the real kernel.
  const int dbg = getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0;
  auto kern = dbg == 1 ? score_topk_f16_kernel<KS, NS, ST_TILE, 1> : (dbg == 2 ? score_topk_f16_kernel<KS, NS, ST_TILE, 2> :
              (dbg == 4 ? score_topk_f16_kernel<KS, NS, ST_TILE, 4> : score_topk_f16_kernel<KS, NS, ST_TILE, 0>));
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  kern<<<sbr_cdiv(Bu, ST_ROWS), ST_THREADS, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, u_idx, eptr, eidx,
                                                        item_offset, k, cap, out_val, out_idx, (unsigned long long*)dbg_buf);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  return SBR_OK;
}

extern "C" int sbr_score_topk_f16(const void* U_f16, const void* I_f16, int D, long Bu, int I, const long* u_idx,
                                  const long* excl_indptr, const int* excl_indices, long excl_nnz, int item_offset, int k, float* out_val,
                                  int* out_idx, void* workspace, long workspace_bytes, void* stream) {
  SBR_REQUIRE(k >= 1 && k <= 32, "sbr_score_topk_f16: k=%d outside [1, 32] (use sbr_gemm_f32 + sbr_topk_rows)", k);
  SBR_REQUIRE(I >= 1, "sbr_score_topk_f16: empty catalogue");
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(U_f16 && I_f16 && out_val && out_idx, "sbr_score_topk_f16: null operand");
  SBR_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr), "sbr_score_topk_f16: exclusion CSR must be given whole or not at all");
  hipStream_t s = (hipStream_t)stream;
  // default: the narrow-wave kernel (score_topk_f16_n.hip); SBR_SCORER_V3=1 keeps the transposed 64-users-per-wave kernel and
  // the older ones behind it for A/B timing
  if ((D == 64 || D == 128 || D == 256) && !(getenv("SBR_SCORER_V3") && atoi(getenv("SBR_SCORER_V3")) != 0))
    return s5_dispatch(U_f16, I_f16, D, Bu, I, u_idx, excl_indptr, excl_indices, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
  // D <= 128: the 64-users-per-wave kernel (SBR_SCORER_V1=1 keeps the first kernel for A/B timing). D = 256 stays on the first
  // kernel: two A fragment sets need 128 VGPRs; the wide kernel with 32-item tiles (<16, 6, 1>) spills A fragments to scratch
  // and reloads them inside the MFMA loop — measured 3.90 / 4.28 ms against 3.72 / 3.79 ms on 100k x 25k x 256.
  if (st_use_wide(D) && !(getenv("SBR_SCORER_V2") && atoi(getenv("SBR_SCORER_V2")) != 0)) {
    if (D == 128) return s4_launch<8, 6, 2, true>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
    return s4_launch<4, 8, 2, true>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
  }
  // D = 256: the same kernel without the prefix pass (two 16-step user fragment sets + the accumulators leave no room for the
  // 64 class-maximum registers): 2.50 ms against 3.2 ms of the first kernel on the c5 shard shape (100k x 25k x 256)
  if (D == 256 && !(getenv("SBR_SCORER_V1") && atoi(getenv("SBR_SCORER_V1")) != 0))
    return s4_launch<16, 4, 2, false>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
  if (st_use_wide(D)) {
    if (D == 128) return s3_launch<8, 6, 2>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
    return s3_launch<4, 8, 2>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
  }
  switch (D) {
    case 64: return st_launch<4, 6, 64>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, s);
    case 128: return st_launch<8, 5, 64>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, s);
    case 256: return st_launch<16, 2, 64>(U_f16, I_f16, Bu, I, u_idx, excl_indptr, excl_indices, item_offset, k, out_val, out_idx, workspace, s);
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
