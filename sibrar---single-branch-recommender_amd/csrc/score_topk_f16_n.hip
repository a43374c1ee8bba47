// Fused full-catalogue scorer, narrow-wave kernel (round 2, second half): 32 users per wave, up to 15 consumer waves + 1 loader
// wave per workgroup = four waves per SIMD. Replaces the transposed 64-users-per-wave kernel of score_topk_f16.hip
// (eval/eval.py:205-222: scores = U x I^T, out[excluded] = -inf, top-k; the score matrix is never written).
//
// Why. The transposed kernel ran two waves per SIMD (240 VGPRs each). A wave is in-order: while it walks its threshold ladder and
// its candidate blocks the matrix pipe only has the SIMD's other wave to draw on, and when both are in their ladders it idles —
// cycle stamps put 55 % of a wave's life in the ladder and the PMC matrix-pipe utilisation at 19 %. Here a wave keeps ONE
// 32-user B fragment set (D / 16 x 4 VGPRs) and 16 NJ accumulators, fits in 128 VGPRs, and four of them share a SIMD: the
// ladder, the candidate blocks and the compactions of three waves hide under the MFMAs of the fourth. The LDS bytes read per
// flop double (a fragment read feeds one MFMA instead of two): 15 x 16 KB per 64-item tile = 940 LDS cycles against 1,920 cycles
// of MFMA per SIMD, still under half the array's rate. The number of consumer waves is a launch parameter: users are dealt in
// 32-user units, so 100k users become 241 workgroups of 13 waves (94 % of the CUs busy for the whole kernel) instead of 224 of
// 7 x 64 (87.5 %).
//
// Kept from the transposed kernel: A = item fragment (LDS ring filled by LDS-DMA, XOR swizzle on the source address, FULL / FREE
// counters per slot, no workgroup barrier in the loop), B = user fragment, so lane (u, h) holds for ONE user the scores of 16
// items of every 32-item tile; lane-local threshold and fill, fire-and-forget appends, cooperative compaction by selection, the
// prefix pass with class maxima, exclusions delivered as bits to the owning lane. New besides the geometry:
//   * the first MFMA of a chain takes the inline constant 0 as C (no accumulator clears: 32 v_mov per tile saved);
//   * a candidate is stored RAW (score bits, ~item) through a buffer descriptor of the wave's 32 KB buffer block at a per-lane
//     byte cursor: the append is one buffer_store_dwordx2 + one v_add, no 64-bit address arithmetic and no key conversion
//     (keys are built at compaction);
//   * thresholds are compared as floats on raw accumulators, ordering / tie rules unchanged (score desc, item index asc).
#include "score_topk_common.h"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef S5_NL
#define S5_NL 2                          // loader waves per workgroup
#endif
#define S5_MAXW (16 - S5_NL)             // consumer waves per workgroup (+ the loader waves = 1024 threads)
#ifndef S5_PF1
#define S5_PF1 3                         // prefetch distance (K steps of 16) with one accumulator tile per LDS tile (D = 256)
#endif
#ifndef S5_PF2
#define S5_PF2 1                         // ... with two (D = 64, 128)
#endif
#ifndef S5_NS
#define S5_NS 6                          // LDS ring slots of 16 KB (D = 128: 64-item tiles, D = 256: 32-item tiles)
#endif
#define S5_CAPH 64                       // candidate entries per (user, lane half)

// All 64 lanes: the k best of the n0 + n1 raw entries of a user's two buffer halves are stored back (raw, unsorted), k - k / 2 to
// b0 and k / 2 to b1 (both halves keep room: a compaction is due when ONE half passes the limit);
// returns the k-th best score (-inf, nothing moved, while fewer than k entries exist). e: the lane's two entries as composite
// keys (ordered score key << 32 | ~item), keep: whether they survived.
__device__ __forceinline__ float s5_select(unsigned long long* b0, unsigned long long* b1, int n0_any, int n1_any, int k, int lane,
                                           unsigned long long e[2], bool keep[2]) {
  const int n0 = __builtin_amdgcn_readfirstlane(n0_any), n1 = __builtin_amdgcn_readfirstlane(n1_any);
  // written and read by this wave only: same-CU vector memory path, in order
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const unsigned long long r0 = lane < n0 ? b0[lane] : 0ull, r1 = lane < n1 ? b1[lane] : 0ull;
  e[0] = lane < n0 ? (((unsigned long long)st_f2key(__uint_as_float((unsigned int)(r0 >> 32))) << 32) | (r0 & 0xFFFFFFFFull)) : 0ull;
  e[1] = lane < n1 ? (((unsigned long long)st_f2key(__uint_as_float((unsigned int)(r1 >> 32))) << 32) | (r1 & 0xFFFFFFFFull)) : 0ull;
  keep[0] = lane < n0;
  keep[1] = lane < n1;
  if (n0 + n1 < k) return -INFINITY;
  const unsigned int h0 = (unsigned int)(e[0] >> 32), h1 = (unsigned int)(e[1] >> 32);
  unsigned int T = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    const int cnt = __popcll(__ballot(h0 >= trial)) + __popcll(__ballot(h1 >= trial));
    T = cnt >= k ? trial : T;
  }
  unsigned long long C = (unsigned long long)T << 32;
  const int c_ge = __popcll(__ballot(h0 >= T)) + __popcll(__ballot(h1 >= T));
  if (c_ge != k) {                                           // several entries share the k-th key: smallest item indices stay
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
  const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
  const int p0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m0, 0u));
  const int p1 = __popcll(m0) + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m1, 0u));
  const int kh = k - (k >> 1);                               // survivors 0 .. kh - 1 stay in half 0, the rest go to half 1
  if (keep[0]) (p0 < kh ? b0 + p0 : b1 + (p0 - kh))[0] = r0;
  if (keep[1]) (p1 < kh ? b0 + p1 : b1 + (p1 - kh))[0] = r1;
  return st_key2f(T);
}

// append of one raw candidate entry at byte offset `pos` of the wave's buffer block (`block`: wave-uniform, so the descriptor is
// four SGPRs the compiler builds once per kernel): buffer_store_dwordx2 v[ent], v[pos], s[rsrc], 0 offen
__device__ __forceinline__ void s5_append(unsigned long long* block, int pos, u32x2 ent) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(block, 0, 32 * 2 * S5_CAPH * 8, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(ent, rs, pos, 0, 0);
}

// all LDS reads of the tile have returned (the accumulators are named so that the wait stays behind the MFMAs that consume the
// fragments); device-only helpers: the host pass of hipcc rejects 64-byte "v" operands and then silently drops the kernel's stub
__device__ __forceinline__ void s5_lds_done(const f32x16& a, const f32x16& b) { asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(b) : "memory"); }
__device__ __forceinline__ void s5_pin(const f32x16& a, const f32x16& b) { asm volatile("" ::"v"(a), "v"(b)); }
// no-return LDS atomics as bare instructions: hipcc puts s_waitcnt vmcnt(0) in front of every LDS atomic of a wave that also
// issues LDS-DMA (it cannot tell the DMA destination from the atomic's word), i.e. a wait for all candidate stores in flight,
// once per tile. LDS operations of a wave execute in order; callers place the waits they need themselves.
__device__ __forceinline__ void s5_lds_add(lds_int* p, int v) { asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned int)(size_t)p), "v"(v) : "memory"); }
__device__ __forceinline__ void s5_lds_or(lds_int* p, unsigned int v) { asm volatile("ds_or_b32 %0, %1" ::"v"((unsigned int)(size_t)p), "v"(v) : "memory"); }
__device__ __forceinline__ void s5_pin8(const f16x8& a) { asm volatile("" ::"v"(a)); }

template <int KS, int NS, int NJ, int DBG, bool PRE>   // KS = D / 16; NS = LDS ring slots; NJ = 32-item accumulator tiles per LDS tile; DBG: ablations; PRE: prefix pass compiled in
__global__ __launch_bounds__(1024) void score_topk_f16_n_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const long* __restrict__ u_idx,
    const long* __restrict__ excl_indptr, const int* __restrict__ excl_indices, int item_offset, int k, int n_pre, int W,
    float* __restrict__ out_val, int* __restrict__ out_idx, unsigned long long* __restrict__ gbuf,
    unsigned long long* __restrict__ dbgbuf) {
  constexpr int D = KS * 16;
  constexpr int ST_TILE = 32 * NJ;
  constexpr int PF = NJ == 1 ? S5_PF1 : S5_PF2;            // fragment prefetch distance in K steps
  constexpr int PF_PRE = KS >= 16 ? 1 : PF;                // ... in the prefix pass (D = 256: the class maxima need the registers)
  constexpr int LIMIT = S5_CAPH - 16 * NJ;                 // a tile adds at most 16 NJ entries to a (user, half) buffer
  constexpr int ROWB = D * 2;
  constexpr int TILEB = ST_TILE * ROWB;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_T = (ST_TILE * CPR) / 64;
  constexpr int LFL0 = (NS - 2) / S5_NL >= 1 ? (NS - 2) / S5_NL : 1;     // tiles in flight per loader wave
  constexpr int LFL = LFL0 * PER_T <= 63 ? LFL0 : 63 / PER_T;
  static_assert(LFL >= 1 && LFL * PER_T <= 63, "vmcnt field");
  static_assert(LIMIT >= 32, "k <= 32 entries must fit below the compaction limit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* exw = reinterpret_cast<unsigned int*>(smem + NS * TILEB);                            // [S5_MAXW * 64]
  int* enx = reinterpret_cast<int*>(exw + S5_MAXW * 64);                                             // [S5_MAXW * 64]
  lds_int* full_lds = (lds_int*)(enx + S5_MAXW * 64);
  lds_int* free_lds = full_lds + NS;
  lds_int* enx_lds = (lds_int*)enx;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long row0 = (long)blockIdx.x * (W * 32);
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
  const int n_virt = n_pre + n_tiles;                      // tile sequence: prefix tiles 0 .. n_pre - 1, then all tiles

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();                                         // the only workgroup barrier of the kernel

  if (wave >= W) {
    if constexpr (DBG == 5) return;                        // lab: consumers run over whatever the ring holds, no loads, no hand-off
    // ---------------------------------------------- loader waves -----------------------------------------------------
    // S5_NL waves take the tiles in turn (tile v belongs to loader v % S5_NL): one wave's LDS-DMA stream tops out near one
    // 16 KB tile per 0.65 us, which is what fourteen consumer waves eat
    const int lw = wave - W;
    int n_mine = 0, v_last = -1;
    for (int v = lw; v < n_virt; v += S5_NL) {
      const int slot = v % NS;
      if (v >= NS) {
        const int need = W * (v / NS);
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
      v_last = v;
      if (++n_mine > LFL) {
        st_wait_vmcnt<LFL * PER_T>();
        st_wave_fence();
        const int vp = v - LFL * S5_NL;
        *(volatile lds_int*)(full_lds + vp % NS) = vp + 1;
      }
    }
    st_wait_vmcnt<0>();
    st_wave_fence();
    if (v_last >= 0) {
      int vp = v_last - (LFL - 1) * S5_NL;
      if (vp < lw) vp = lw;
      for (; vp <= v_last; vp += S5_NL) *(volatile lds_int*)(full_lds + vp % NS) = vp + 1;
    }
    return;
  }

  // ------------------------------------------------ consumer waves ------------------------------------------------------
  // B-operand fragments of the wave's 32-user tile: user 32 * wave + l31, k = 16 s + 8 half + j
  f16x8 ufrag[KS];
  {
    const long r = row0 + wave * 32 + l31;
    const long ur = r < Bu ? r : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
  }
  unsigned long long* wgb = gbuf + (row0 + (long)wave * 32) * (2 * S5_CAPH);      // wave-uniform: buffers of the wave's 32 users
  exw[t] = 0u;
  // exclusion cursor: lane L < 32 walks the sorted CSR row of user 32 * wave + L in step with the item tiles; e0 = next excluded
  // item, the one after it sits in LDS. lo0: restart point of the main pass.
  const long my_row = row0 + wave * 32 + l31;
  long eidx = 0, eend = 0, lo0 = 0;
  int e0 = 0x7FFFFFFF;
  {
    int e1 = 0x7FFFFFFF;
    if (half == 0 && my_row < Bu && excl_indptr) {
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
  int peek = 0;
  // lane (u, h): threshold of user u and byte cursor into its buffer half h (thresholds of the two halves of a user are equal)
  float thr = -INFINITY;
  const int lane_base = (l31 * 2 + half) * S5_CAPH * 8;
  int pos = lane_base;

  unsigned long long t_wait = 0, t_evt = 0, n_evt = 0, n_ins = 0, n_cand = 0, t_cmp = 0, t_issue = 0, t_ladder = 0;
  const unsigned long long t_begin = DBG != 0 ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_begin = DBG != 0 ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // one item tile: wait, MFMAs (S^T = I x U^T), slot release, exclusion bits of the tile -> acc, have_ex
#define S5_TILE_BODY(V, PFV)                                                                                                 \
    const int slot = (V) % NS;                                                                                           \
    const unsigned long long tw0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;                                       \
    /* `peek` = FULL word of this slot as read while the previous tile was in its MFMAs (stale at worst: slow poll) */     \
    if (DBG != 5 && __builtin_amdgcn_readfirstlane(peek) != (V) + 1) {                                                   \
      while (st_peek(full_lds + slot) != (V) + 1) __builtin_amdgcn_s_sleep(1);                                           \
    }                                                                                                                    \
    st_wave_fence();                                                                                                     \
    if constexpr (DBG == 4) t_wait += __builtin_amdgcn_s_memtime() - tw0;                                                \
    f32x16 acc[NJ];                                                                                                      \
    /* fragment reads run PF steps ahead of the MFMAs that consume them (register ring of PF + 1 steps); the scheduling    \
       barriers keep hipcc from sinking the reads back to their use (it otherwise issues read, wait, MFMA in turn and a     \
       wave shows the LDS latency sixteen times per tile) */                                                              \
    f16x8 bf[(PFV) + 1][NJ];                                                                                                \
    /* fragment of K step s, tile nj: row nj * 32 + l31, 16-byte chunk (2 s + half) ^ (l31 & SWZ) = byte offset             \
       (s << 5) ^ lxh; lxh is pinned per tile so that the KS offsets are not kept in registers (one v_xad_u32 per read) */   \
    const unsigned char* rowp = smem + slot * TILEB + l31 * ROWB;                                                        \
    unsigned int lxh = (unsigned int)(((l31 & SWZ) << 4) ^ (half << 4));                                                 \
    asm volatile("" : "+v"(lxh));                                                                                        \
    _Pragma("unroll") for (int s = 0; s < (PFV) && s < KS; ++s) {                                                           \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                  \
        bf[s][nj] = DBG == 6 ? ufrag[(s + nj) % KS] : *reinterpret_cast<const f16x8*>(rowp + nj * 32 * ROWB + (((unsigned int)s << 5) ^ lxh)); \
    }                                                                                                                    \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                     \
      if (s + (PFV) < KS) {                                                                                                 \
        _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                \
          bf[(s + (PFV)) % ((PFV) + 1)][nj] = DBG == 6 ? ufrag[(s + nj + 1) % KS] : *reinterpret_cast<const f16x8*>(rowp + nj * 32 * ROWB + (((unsigned int)(s + (PFV)) << 5) ^ lxh)); \
      }                                                                                                                  \
      if (s == KS / 2) peek = *(volatile lds_int*)(full_lds + ((V) + 1) % NS);                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                  \
        if constexpr (DBG == 7) { s5_pin8(bf[s % ((PFV) + 1)][nj]); acc[nj] = zero16; }                                       \
        else acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s % ((PFV) + 1)][nj], ufrag[s], s == 0 ? zero16 : acc[nj], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
    s5_lds_done(acc[0], acc[NJ - 1]);                                                                                    \
    if (DBG != 5 && lane == 0) s5_lds_add(free_lds + slot, 1);                                                                \
    /* exclusions of this tile: item column `col` of the tile, user L: one bit for the lane that holds that accumulator —  \
       lane L + 32 ((col >> 2) & 1), bit (col >> 5) * 16 + (col & 3) + 4 ((col & 31) >> 3) */                             \
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
        s5_lds_or((lds_int*)(exw + tgt), 1u << ((col >> 5) * 16 + (col & 3) + 4 * ((col & 31) >> 3)));                     \
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
    f32x16 cm;                                             // item class = (lane half, accumulator register): 32 per user
#pragma unroll
    for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;
    for (int v = 0; v < n_pre; ++v) {
      const int j0 = v * ST_TILE;
      S5_TILE_BODY(v, PF_PRE)
      if (have_ex) {                                       // excluded scores must not raise a class maximum
        st_wave_fence();
        const unsigned int ex0 = exw[t];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nj][r] = ((ex0 >> (nj * 16 + r)) & 1u) ? -INFINITY : acc[nj][r];
        }
        exw[t] = 0u;
        st_wave_fence();
      }
      if (j0 + ST_TILE > I) {                              // catalogue end inside the tile: padded columns do not count
        const int lim = I - j0 - 4 * half;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool in = nj * 32 + (r & 3) + 8 * (r >> 2) < lim;
            acc[nj][r] = in ? acc[nj][r] : -INFINITY;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (NJ == 2) cm[r] = __builtin_fmaxf(cm[r], __builtin_fmaxf(acc[0][r], acc[1][r]));      // one v_max3
        else cm[r] = fmaxf(cm[r], acc[0][r]);
      }
    }
    // k-th largest of the user's 32 class maxima (16 in each of its two lanes): bitwise binary search on the ordered keys,
    // every lane pair for its own user. The threshold admits scores EQUAL to the bound (its items are not in any buffer).
    {
      unsigned int T = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = T | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c += st_f2key(cm[r]) >= trial;
        c += __shfl_xor(c, 32, 64);
        T = c >= k ? trial : T;
      }
      // T = 0x007FFFFF is the key of -inf (fewer than k finite classes): no bound
      thr = T > 0x007FFFFFu ? st_key2f(T - 1u) : -INFINITY;
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
  const int pos_limit = lane_base + LIMIT * 8;
  for (int tl = 0; tl < n_tiles; ++tl) {
    if (__ballot(pos > pos_limit)) {
      // ---- maintenance (cold): compact the users with a half above LIMIT so that this tile's appends cannot overflow
      const unsigned long long tm0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
      unsigned long long need = __ballot(pos > pos_limit);
      need = (need | (need >> 32)) & 0xFFFFFFFFull;
      const int cnt = (pos - lane_base) >> 3;
      while (need) {
        const int u = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        const int n0 = __builtin_amdgcn_readlane(cnt, u), n1 = __builtin_amdgcn_readlane(cnt, u + 32);
        unsigned long long* b0 = wgb + (long)u * (2 * S5_CAPH);
        unsigned long long e[2];
        bool kp[2];
        const float nt = s5_select(b0, b0 + S5_CAPH, n0, n1, k, lane, e, kp);
        if (n0 + n1 >= k && l31 == u) {
          thr = nt;
          pos = lane_base + (half ? (k >> 1) : k - (k >> 1)) * 8;
        }
        if constexpr (DBG == 4) ++n_ins;
      }
      if constexpr (DBG == 4) t_cmp += __builtin_amdgcn_s_memtime() - tm0;
    }
    const int j0 = tl * ST_TILE;
    const int vseq = n_pre + tl;
    const unsigned long long ti0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    S5_TILE_BODY(vseq, PF)
    const unsigned long long ti1 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    if constexpr (DBG == 4) t_issue += ti1 - ti0 - (__builtin_amdgcn_s_memtime() - ti1);
    if constexpr (DBG == 1 || DBG >= 5) {
      s5_pin(acc[0], acc[NJ - 1]);
      continue;
    }
    unsigned int ex = 0u;
    if (have_ex) { st_wave_fence(); ex = exw[t]; }
    if (j0 + ST_TILE > I) {                                // catalogue end inside the (last) tile: padded columns never qualify
      const int lim = I - j0 - 4 * half;                   // item (r & 3) + 8 (r >> 2) + 32 nj of this lane exists iff < lim
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool in = nj * 32 + (r & 3) + 8 * (r >> 2) < lim;
          acc[nj][r] = in ? acc[nj][r] : -INFINITY;
        }
      }
    }
    const unsigned int item_lane = 0xFFFFFFFFu - (unsigned int)(item_offset + j0 + 4 * half);
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // common path per FOUR accumulator registers: four v_cmp into SGPR pairs issued back to back, three s_or, one scalar
        // branch (a v_cmp -> branch pair per register serialises on the compare's latency)
        unsigned long long bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = __ballot(acc[nj][4 * g + q] > thr);
        if constexpr (DBG == 2) { if (bq[0] | bq[1] | bq[2] | bq[3]) asm volatile("s_nop 0"); continue; }
        if (bq[0] | bq[1] | bq[2] | bq[3]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 4 * g + q;
            if (bq[q]) {
              const unsigned long long te0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
              // everything below hangs off values pinned inside the branch (hipcc otherwise if-converts the block and
              // evaluates exclusion arithmetic for every register of every tile)
              float v = acc[nj][r];
              unsigned int exv = ex;
              asm volatile("" : "+v"(v), "+v"(exv));
              const int C = nj * 32 + (r & 3) + 8 * (r >> 2);          // compile-time after unrolling
              const bool cand = (v > thr) && !((exv >> (nj * 16 + r)) & 1u);
              if (cand) {
                const u32x2 ent = {item_lane - (unsigned int)C, __float_as_uint(v)};
                s5_append(wgb, pos, ent);
                pos += 8;
              }
              if constexpr (DBG == 4) { n_cand += __popcll(__ballot(cand)); t_evt += __builtin_amdgcn_s_memtime() - te0; ++n_evt; }
            }
          }
        }
      }
    }
    if (have_ex) { exw[t] = 0u; st_wave_fence(); }
    if constexpr (DBG == 4) t_ladder += __builtin_amdgcn_s_memtime() - ti1;
  }
#undef S5_TILE_BODY

  if constexpr (DBG != 0) {
    if (lane == 0 && dbgbuf) {
      unsigned long long* d = dbgbuf + ((long)blockIdx.x * S5_MAXW + wave) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_evt; d[3] = n_cand; d[4] = n_evt | (t_issue << 20); d[5] = n_ins | (t_ladder << 20); d[6] = t_cmp;
      d[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
  }
  // final selection + output of the wave's 32 users: the lane that holds the entry of rank j writes output position j
  const int cnt_fin = (pos - lane_base) >> 3;
  for (int u = 0; u < 32; ++u) {
    const long ur = row0 + wave * 32 + u;
    if (ur >= Bu) break;
    const int n0 = __builtin_amdgcn_readlane(cnt_fin, u), n1 = __builtin_amdgcn_readlane(cnt_fin, u + 32);
    unsigned long long* b0 = wgb + (long)u * (2 * S5_CAPH);
    unsigned long long e[2];
    bool kp[2];
    s5_select(b0, b0 + S5_CAPH, n0, n1, k, lane, e, kp);
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
    if (lane >= n && lane < k) {                             // fewer than k candidates: empty slots behind them
      out_val[ur * k + lane] = -INFINITY;
      out_idx[ur * k + lane] = -1;
    }
  }
}

// consumer waves per workgroup: users are dealt in 32-user units over the CUs; the smallest W that keeps the number of rounds
// (workgroups per CU, one resident at a time) at its minimum
static int s5_pick_waves(long Bu) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  if (getenv("SBR_ST_WAVES")) {
    int w = atoi(getenv("SBR_ST_WAVES"));
    return w < 1 ? 1 : (w > S5_MAXW ? S5_MAXW : w);
  }
  const long units = sbr_cdiv(Bu, 32);
  const long rounds = sbr_cdiv(units, (long)n_cu * S5_MAXW);
  long w = sbr_cdiv(units, rounds * n_cu);
  return (int)(w < 1 ? 1 : (w > S5_MAXW ? S5_MAXW : w));
}

long s5_workspace_bytes(long Bu) {
  // users padded to whole workgroups of any wave count (< 32 * S5_MAXW extra) + the cycle stamps of SBR_ST_DEBUG
  const long padded = Bu + 32L * S5_MAXW;
  return padded * 2 * S5_CAPH * 8 + (sbr_cdiv(Bu, 32) + S5_MAXW) * S5_MAXW * 64L;
}

template <int KS, int NS, int NJ, bool PRE>
static int s5_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                     int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, hipStream_t s) {
  const int W = s5_pick_waves(Bu);
  const long n_wg = sbr_cdiv(Bu, 32L * W);
  const long need = n_wg * 32L * W * 2 * S5_CAPH * 8;
  const int dbg = getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0;      // 1 | 2: timing-only ablations, 4: cycle stamps
  SBR_REQUIRE(workspace && workspace_bytes >= need + (dbg != 0 ? n_wg * S5_MAXW * 64L : 0L),
              "sbr_score_topk_f16: workspace of %ld bytes needed (sbr_score_topk_f16_workspace), %ld given", need, workspace_bytes);
  void* dbg_buf = (char*)workspace + need;
  const size_t lds = (size_t)NS * (32 * NJ) * KS * 32 + S5_MAXW * 64 * 8 + 2 * NS * 4 + 16;
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  // prefix pass: ~1/12 of the catalogue (whole tiles), skipped for catalogues too short to repay it; SBR_ST_PRE overrides (tiles)
  const int n_tiles = sbr_cdiv(I, 32 * NJ);
  int n_pre = n_tiles >= 96 ? n_tiles / 12 : 0;
  if (getenv("SBR_ST_PRE")) n_pre = atoi(getenv("SBR_ST_PRE"));
  if (n_pre > n_tiles) n_pre = n_tiles;
  if (n_pre < 0 || !PRE || k > 24) n_pre = 0;             // the bound is the k-th of 32 class maxima: needs k below that
  // 1 | 2 | 5 | 6 | 7: timing-only ablations (1: MFMA loop only; 2: + threshold compares; of the MFMA loop 5: without loads and
  // hand-off, 6: without fragment reads, 7: without MFMAs), 4: cycle stamps
  auto kern = dbg == 1 ? score_topk_f16_n_kernel<KS, NS, NJ, 1, PRE> : (dbg == 2 ? score_topk_f16_n_kernel<KS, NS, NJ, 2, PRE> :
              (dbg == 4 ? score_topk_f16_n_kernel<KS, NS, NJ, 4, PRE> : (dbg == 5 ? score_topk_f16_n_kernel<KS, NS, NJ, 5, PRE> :
              (dbg == 6 ? score_topk_f16_n_kernel<KS, NS, NJ, 6, PRE> : (dbg == 7 ? score_topk_f16_n_kernel<KS, NS, NJ, 7, PRE> :
               score_topk_f16_n_kernel<KS, NS, NJ, 0, PRE>)))));
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  kern<<<(unsigned int)n_wg, (W + S5_NL) * 64, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, u_idx, eptr, eidx, item_offset, k,
                                                   n_pre, W, out_val, out_idx, (unsigned long long*)workspace, (unsigned long long*)dbg_buf);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  return SBR_OK;
}

// D in {64, 128, 256}; called by sbr_score_topk_f16 (score_topk_f16.hip)
int s5_dispatch(const void* U, const void* It, int D, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx,
                int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, hipStream_t s) {
  switch (D) {
    case 64: return s5_launch<4, 8, 2, true>(U, It, Bu, I, u_idx, eptr, eidx, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
    case 128: return s5_launch<8, S5_NS, 2, true>(U, It, Bu, I, u_idx, eptr, eidx, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
    case 256: return s5_launch<16, S5_NS, 1, true>(U, It, Bu, I, u_idx, eptr, eidx, item_offset, k, out_val, out_idx, workspace, workspace_bytes, s);
    default:
      sbr_set_error("sbr_score_topk_f16: D=%d not supported by the narrow-wave kernel", D);
      return SBR_ERR_ARG;
  }
}
