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
//
// Round 3: thresholds without compactions. A user's candidate buffers are large (S5_CAPH entries per lane half, in the workspace) and
// are not compacted during the stream; instead every lane keeps the running MAXIMUM of what it appended per accumulator register
// ("class": 16 per lane, 32 per user, updated by one v_max inside the branch-free append, under its EXEC mask — so excluded scores
// and scores below the threshold never enter). The k-th largest of a user's 32 class maxima is the score of a real, already
// buffered item with k - 1 buffered items of other classes at or above it: every later item (larger index) needs a strictly larger
// score to reach the top k. All 32 users of a wave refresh their threshold from it at once, lane-parallel (sort 16 registers by a
// Batcher network, exchange with the partner lane, bitonic half-merge: ~270 vector instructions, no ballots, no memory), every few
// tiles at first and every 32 tiles later. The cooperative per-user selection (s5_select: 32 ballot rounds for ONE user) remains
// only as the overflow path of a (user, half) buffer, and the final selection + ranking runs in a kernel of its own
// (score_topk_finalize_kernel: one wave per user, all CUs busy) instead of serially per user at the tail of the scorer's waves.
#include "score_topk_shared.h"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef S5_CAPH
#define S5_CAPH 256                      // candidate entries per (user, lane half); multiple of 64
#endif
#define S5_EH (S5_CAPH / 64)             // entries of one buffer half per lane when a wave holds a whole user's buffers
#ifndef S5_PRE_TILES
#define S5_PRE_TILES 16                  // tiles of the prefix pass (class maxima only, no appends) of catalogues of >= 96 tiles
#endif
#ifndef S5_CML_KS
#define S5_CML_KS 16                     // class maxima of the main pass in LDS for D >= 16 * S5_CML_KS (else in registers)
#endif
#ifndef S5_EXSPLIT
#define S5_EXSPLIT 0                     // 1: tiles without exclusion events append without testing exclusion bits (two code copies;
                                         // measured slower: the copies cost eight register moves of the class maxima per tile)
#endif
#ifndef S5_RF
#define S5_RF 32                         // tiles between two threshold refreshes in the steady state
#endif
#ifndef S5_SHAPE16
#define S5_SHAPE16 0                     // lab (timing only, wrong results): every 32x32x16 MFMA as two 16x16x32 MFMAs — does the other shape
                                         // hold a higher clock under the power limit in THIS kernel? With DISTINCT operands for the two:
                                         // -5 % (1.370 against 1.437 ms); with the same operands twice -24 % (the clock follows the data:
                                         // repeated operands toggle less). A complete 16x16x32 kernel (a user over four lanes, two users per
                                         // lane, class maxima in LDS) was written, passed the scorer tests and measured 0 .. +6 %: dropped.
#endif
#ifndef S5_LADDER
#define S5_LADDER 0                      // lab: 1 = maximum + compare of a register pair in one asm block (no s_nop pads), 2 = and no OR
                                         // over the pairs. Fewer instructions, not faster (in-process A/B: +1 %, +2 %): the consumer waves are
                                         // not issue-bound in the ladder
#endif

// All 64 lanes: the scorer's OVERFLOW path (a (user, half) buffer ran full: ties at the threshold, or a threshold that cannot rise).
// The k best of the n0 + n1 raw entries of a user's two buffer halves are found by a bitwise binary search for the k-th largest
// composite key (score key << 32 | ~item: ties at the k-th score keep the smallest item indices) over ballot counts; written for few
// registers instead of speed — the entries are re-read from the buffers in every round of the search instead of being held in
// 2 x S5_EH register pairs per lane, which would cost the hot loop its fourth wave per SIMD. Survivors go back split over both
// halves (k - k / 2 and k / 2: both keep room). Returns the k-th best score; -inf (nothing moved) below k entries.
__device__ __forceinline__ float s5_overflow_select(unsigned long long* b0, unsigned long long* b1, int n0_any, int n1_any, int k, int lane) {
  const int n0 = __builtin_amdgcn_readfirstlane(n0_any), n1 = __builtin_amdgcn_readfirstlane(n1_any);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     // written and read by this wave only
  if (n0 + n1 < k) return -INFINITY;
  const int c0 = (n0 + 63) >> 6, c1 = (n1 + 63) >> 6;
  auto raw_at = [&](bool first, int j) -> unsigned long long {
    const int q = lane + 64 * j;
    return q < (first ? n0 : n1) ? (first ? b0 : b1)[q] : 0ull;
  };
  auto key_of = [&](unsigned long long raw) -> unsigned long long {     // 0 for an empty slot (raw entries are never 0: ~item != 0)
    return raw ? (((unsigned long long)st_f2key(__uint_as_float((unsigned int)(raw >> 32))) << 32) | (raw & 0xFFFFFFFFull)) : 0ull;
  };
  auto count_ge = [&](unsigned long long C) {
    int cnt = 0;
    for (int j = 0; j < c0; ++j) cnt += __popcll(__ballot(key_of(raw_at(true, j)) >= C));
    for (int j = 0; j < c1; ++j) cnt += __popcll(__ballot(key_of(raw_at(false, j)) >= C));
    return cnt;
  };
  unsigned int T = 0u;
  int c_ge = n0 + n1;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    const int cnt = count_ge((unsigned long long)trial << 32);
    if (cnt >= k) { T = trial; c_ge = cnt; if (cnt == k) break; }
  }
  unsigned long long C = (unsigned long long)T << 32;
  if (c_ge != k) {
    unsigned int Lw = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = Lw | (1u << bit);
      Lw = count_ge(((unsigned long long)T << 32) | trial) >= k ? trial : Lw;
    }
    C |= (unsigned long long)Lw;
  }
  // survivors land in entries [0, 32) of the two halves: chunk 0 of both is taken into registers first, the other chunks are
  // streamed (read, keep, store), chunk 0's survivors go last
  const unsigned long long r00 = raw_at(true, 0), r10 = raw_at(false, 0);
  const int kh = k - (k >> 1);
  int before = 0;
  auto place = [&](unsigned long long raw) {
    const bool keep = key_of(raw) >= C;
    const unsigned long long m = __ballot(keep);
    const int p = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    if (keep) (p < kh ? b0 + p : b1 + (p - kh))[0] = raw;
    before += __popcll(m);
  };
  for (int j = 1; j < c0; ++j) place(raw_at(true, j));
  for (int j = 1; j < c1; ++j) place(raw_at(false, j));
  place(r00);
  place(r10);
  const float t = st_key2f(T);
  return t == t ? t : -INFINITY;
}

// append of one raw candidate entry at byte offset `pos` of the wave's buffer block (`block`: wave-uniform, so the descriptor is
// four SGPRs the compiler builds once per kernel): buffer_store_dwordx2 v[ent], v[pos], s[rsrc], 0 offen
__device__ __forceinline__ void s5_append(unsigned long long* block, int pos, u32x2 ent) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(block, 0, 32 * 2 * S5_CAPH * 8, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(ent, rs, pos, 0, 0);
}

// Branch-free append of one accumulator value: lanes with a > thr whose exclusion bit is clear store the raw entry
// (~item = il - C, score bits) at their byte cursor and advance it. The vector ALU writes EXEC itself (v_cmpx), so the
// sequence has no vector -> scalar hand-over and no branch; all lanes are active on entry and on exit. rs: buffer descriptor of
// the wave's candidate block (s5_block_rsrc).
#ifndef S5_NOSTORE
#define S5_NOSTORE 0                      // lab (timing only, wrong results): the append stores nothing
#endif
#if S5_NOSTORE
#define S5_STORE_ASM
#else
#define S5_STORE_ASM "buffer_store_dword %[tmp], %[pos], %[rs], 0 offen\n\t" "buffer_store_dword %[a], %[pos], %[rs], 0 offen offset:4\n\t"
#endif
// CML (D = 256: the user fragments alone take 64 registers): the class maxima live in LDS ([register][lane] floats per wave, the lane's
// slot of class r at cm_addr + 256 r) and the append updates them with a no-return ds_max_f32 under the same EXEC mask.
template <unsigned int BIT, int C, bool EX, bool CML, int R>
__device__ __forceinline__ void s5_try_append(float a, float thr, unsigned int ex, int& pos, unsigned int il, i32x4 rs, float& cmax,
                                              unsigned int cm_addr) {
  unsigned int tmp;
  if constexpr (CML) {
    asm volatile(
        "v_cmpx_gt_f32_e32 %[a], %[thr]\n\t"
        "v_and_b32_e32 %[tmp], %[bit], %[ex]\n\t"
        "v_cmpx_eq_u32_e32 0, %[tmp]\n\t"
        "v_subrev_u32_e32 %[tmp], %[c], %[il]\n\t"
        S5_STORE_ASM
        "v_add_u32_e32 %[pos], 8, %[pos]\n\t"
        "ds_max_f32 %[cma], %[a] offset:%[off]\n\t"
        "s_mov_b64 exec, -1"
        : [pos] "+v"(pos), [tmp] "=&v"(tmp)
        : [a] "v"(a), [thr] "v"(thr), [ex] "v"(ex), [il] "v"(il), [rs] "s"(rs), [bit] "n"(BIT), [c] "n"(C), [cma] "v"(cm_addr), [off] "n"(R * 256)
        : "vcc", "memory");
    return;
  }
  if constexpr (!EX) {                                       // a tile without exclusion events: no exclusion bit to test
    asm volatile(
        "v_cmpx_gt_f32_e32 %[a], %[thr]\n\t"
        "v_subrev_u32_e32 %[tmp], %[c], %[il]\n\t"
        S5_STORE_ASM
        "v_add_u32_e32 %[pos], 8, %[pos]\n\t"
        "v_max_f32_e32 %[cm], %[cm], %[a]\n\t"
        "s_mov_b64 exec, -1"
        : [pos] "+v"(pos), [tmp] "=&v"(tmp), [cm] "+v"(cmax)
        : [a] "v"(a), [thr] "v"(thr), [il] "v"(il), [rs] "s"(rs), [c] "n"(C)
        : "vcc", "memory");
    return;
  }
  asm volatile(
      "v_cmpx_gt_f32_e32 %[a], %[thr]\n\t"
      "v_and_b32_e32 %[tmp], %[bit], %[ex]\n\t"
      "v_cmpx_eq_u32_e32 0, %[tmp]\n\t"
      "v_subrev_u32_e32 %[tmp], %[c], %[il]\n\t"
      S5_STORE_ASM
      "v_add_u32_e32 %[pos], 8, %[pos]\n\t"
      "v_max_f32_e32 %[cm], %[cm], %[a]\n\t"               /* class maximum of what was appended (same EXEC mask) */
      "s_mov_b64 exec, -1"
      : [pos] "+v"(pos), [tmp] "=&v"(tmp), [cm] "+v"(cmax)
      : [a] "v"(a), [thr] "v"(thr), [ex] "v"(ex), [il] "v"(il), [rs] "s"(rs), [bit] "n"(BIT), [c] "n"(C)
      : "vcc", "memory");
}
// raw buffer descriptor of a wave's candidate block: base, stride 0, 32 users x 2 halves x S5_CAPH entries of 8 bytes, gfx950 format word
__device__ __forceinline__ i32x4 s5_block_rsrc(const void* block) {
  const unsigned long long b = (unsigned long long)block;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)(b >> 32) & 0xFFFF);
  r[2] = 32 * 2 * S5_CAPH * 8;
  r[3] = 0x00020000;
  return r;
}

template <int KS, int NS, int NJ, int DBG, bool PRE>   // KS = D / 16; NS = LDS ring slots; NJ = 32-item accumulator tiles per LDS tile; DBG: ablations; PRE: prefix pass compiled in
__global__ __launch_bounds__(1024) void score_topk_f16_n_kernel(
    const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, const unsigned int* __restrict__ events,
    const int* __restrict__ group_base, int item_offset, int k, int n_pre, int W, int n_part, int P,
    int* __restrict__ cnt_out, unsigned long long* __restrict__ gbuf, unsigned long long* __restrict__ dbgbuf) {
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
  constexpr bool CML = KS >= S5_CML_KS;                    // class maxima of the main pass in LDS instead of registers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_int* full_lds = (lds_int*)(smem + NS * TILEB);
  lds_int* free_lds = full_lds + NS;
  // CML: [consumer wave][16 classes][64 lanes] floats behind the ring and its counters
  const unsigned int cm_addr = (unsigned int)(size_t)(smem + NS * TILEB + 2 * NS * 4 + 16) + (unsigned int)((threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 4);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // Work units = 32 users x the whole catalogue. A workgroup has W FULL consumer waves (unit blockIdx.x * W + wave); when the units
  // do not divide evenly over the CUs, the remainder units are each cut into P PARTS by item tile (tile t belongs to part t % P) and
  // workgroup b < n_part gets one more consumer wave for part b % P of remainder unit b / P — with its own candidate buffers, merged by
  // the final selection. (One unit more per workgroup instead would put a fourth consumer wave on ONE SIMD of every CU: that SIMD's
  // instruction stream sets the pace of the whole workgroup through the tile ring — cycle stamps: the other waves waited a quarter
  // of their time.)
  const int Wb = W + ((int)blockIdx.x < n_part ? 1 : 0);  // consumer waves of THIS workgroup
  const bool partial = wave == W && (int)blockIdx.x < n_part;                       // wave-uniform
  const int part = partial ? (int)blockIdx.x % P : 0, n_parts = partial ? P : 1;
  const long n_full_units = (long)gridDim.x * W;
  const long unit = partial ? n_full_units + (int)blockIdx.x / P : (long)blockIdx.x * W + wave;      // 32-user group of this wave
  // rows of the candidate buffers / fill counts: a full wave uses its users' rows, a partial wave rows behind all units
  const long n_units = (Bu + 31) >> 5;
  const long brow0 = partial ? n_units * 32 + (long)blockIdx.x * 32 : unit * 32;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
  const int n_virt = n_pre + n_tiles;                      // tile sequence: prefix tiles 0 .. n_pre - 1, then all tiles

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();                                         // the only workgroup barrier of the kernel

  const int cslots = W + (n_part > 0 ? 1 : 0);             // wave slots in front of the loader waves
  if (wave == W && n_part > 0 && !partial) return;         // the slot of the partial wave in a workgroup that has none
  if (wave >= cslots) {
    if constexpr (DBG == 5) return;                        // lab: consumers run over whatever the ring holds, no loads, no hand-off
    // ---------------------------------------------- loader waves -----------------------------------------------------
    // S5_NL waves take the tiles in turn (tile v belongs to loader v % S5_NL): one wave's LDS-DMA stream tops out near one
    // 16 KB tile per 0.65 us, which is what fourteen consumer waves eat
    const int lw = wave - cslots;
    int n_mine = 0, v_last = -1;
    for (int v = lw; v < n_virt; v += S5_NL) {
      const int slot = v % NS;
      if (v >= NS) {
        const int need = Wb * (v / NS);
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
    const long r = unit * 32 + l31;
    const long ur = r < Bu ? r : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
    // Name the fragments once before the tile loops: hipcc then waits for these loads HERE. Left pending, its wait lands in front of
    // the first MFMA inside the loop as s_waitcnt vmcnt(0) — which, on every later tile, waits for the candidate stores of the
    // previous tile (appends are inline assembly the compiler's counter model does not see).
#pragma unroll
    for (int s = 0; s < KS; ++s) s5_pin8(ufrag[s]);
  }
  unsigned long long* wgb = gbuf + brow0 * (2 * S5_CAPH);      // wave-uniform: buffers of the wave's 32 users
  const i32x4 wrs = s5_block_rsrc(wgb);
  // Exclusions arrive as a wave-uniform EVENT stream (s5_build_events): for this wave's 32 users, one 32-bit word per excluded
  // (user, item) of the scored item range, ordered by item tile: tile << 11 | lane that holds the accumulator << 5 | its bit.
  // The wave reads it with scalar loads, a quad at a time and one quad ahead (w: current, shifted down as events are consumed;
  // n: next), and applies an event with one v_cmp / v_cndmask / v_or. Scalar loads do not share a counter with the candidate
  // stores (a per-lane walk of the CSR rows has to wait on vmcnt, i.e. for every store in flight), and a user with thousands of
  // exclusions costs its events, not a serialised round per entry for the whole wave.
  // (a wave past the last user group — padding of the last workgroup — has no group_base entry: it runs without events)
  const bool has_excl = events != nullptr && unit < n_units;      // wave-uniform
  // (read through the CONSTANT address space: hipcc turns a wave-uniform load from global memory into s_load only when it can
  // prove that nothing in the kernel writes there; it could not, used global_load_dwordx4 + VGPRs for the window, and the wait for
  // that load — vmcnt(0), i.e. for every candidate store in flight — sat inside the event loop: +0.28 ms per pass)
  typedef const __attribute__((address_space(4))) unsigned int* ev_ptr;
  typedef unsigned int ev_quad __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) ev_quad* ev_quad_ptr;
  ev_ptr evp = nullptr;
  unsigned int w0 = S5_EV_NONE, w1 = S5_EV_NONE, w2 = S5_EV_NONE, w3 = S5_EV_NONE, n0 = S5_EV_NONE, n1 = S5_EV_NONE, n2 = S5_EV_NONE, n3 = S5_EV_NONE;
  int ev_rem = 4, ev_q = 8;
#define S5_EV_RESTART()                                                                                                  \
  if (has_excl) {                                                                                                        \
    const ev_quad qa = *(ev_quad_ptr)(evp), qb = *(ev_quad_ptr)(evp + 4);                                                \
    w0 = qa.x; w1 = qa.y; w2 = qa.z; w3 = qa.w; n0 = qb.x; n1 = qb.y; n2 = qb.z; n3 = qb.w;                             \
    ev_rem = 4; ev_q = 8;                                                                                                \
  }
  if (has_excl) evp = (ev_ptr)events + ((const __attribute__((address_space(4))) int*)group_base)[unit];
  S5_EV_RESTART()
  int peek = 0;
  int slot_next = 0;                                       // ring slot of the next tile of the sequence
  // lane (u, h): threshold of user u and byte cursor into its buffer half h (thresholds of the two halves of a user are equal)
  float thr = -INFINITY;
  const int lane_base = (l31 * 2 + half) * S5_CAPH * 8;
  int pos = lane_base;

  unsigned long long t_mid = 0;
  unsigned long long t_wait = 0, t_evt = 0, n_evt = 0, n_ins = 0, n_cand = 0, t_cmp = 0, t_issue = 0, t_ladder = 0;
  const unsigned long long t_begin = DBG != 0 ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_begin = DBG != 0 ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef float f32x4q __attribute__((ext_vector_type(4)));
  const f32x4q zero4 = {0.f, 0.f, 0.f, 0.f};

  // the event window moves on by one event (a quad at a time is refilled by a scalar load, one quad ahead)
#define S5_EV_NEXT()                                                                                                     \
        w0 = w1; w1 = w2; w2 = w3;                                                                                       \
        if (--ev_rem == 0) {                                                                                             \
          w0 = n0; w1 = n1; w2 = n2; w3 = n3;                                                                            \
          if constexpr (S5_EVABL == 1) { n0 = n1 = n2 = n3 = S5_EV_NONE; } else {                                       \
          const ev_quad qn = *(ev_quad_ptr)(evp + ev_q);                                                                 \
          n0 = qn.x; n1 = qn.y; n2 = qn.z; n3 = qn.w; }                                                                  \
          ev_rem = 4; ev_q += 4;                                                                                         \
        }
  // a tile that is not this (partial) wave's: wait for it, release it, pass its exclusion events by — the wave stays in step with
  // the ring (a slot may only be released after its tile has been published: the loader counts releases per slot)
#define S5_TILE_SKIP(V, J0)                                                                                              \
  {                                                                                                                      \
    const int slot = slot_next;                                                                                          \
    slot_next = slot + 1 == NS ? 0 : slot + 1;                                                                           \
    if (DBG != 5) { while (st_peek(full_lds + slot) != (V) + 1) __builtin_amdgcn_s_sleep(1); }                           \
    st_wave_fence();                                                                                                     \
    if constexpr (DBG != 5) s5_lds_add_lane0(free_lds + slot, 1);                                                        \
    peek = 0;                                                                                                            \
    if (has_excl) {                                                                                                      \
      const unsigned int tkey = (unsigned int)((J0) / ST_TILE);                                                          \
      while ((w0 >> 11) == tkey) { S5_EV_NEXT() }                                                                        \
    }                                                                                                                    \
  }
  // one item tile: wait, MFMAs (S^T = I x U^T), slot release, exclusion bits of the tile -> acc, have_ex
#define S5_TILE_BODY(V, PFV)                                                                                                 \
    const int slot = slot_next;                            /* = (V) % NS, kept as a wrapping counter */                    \
    slot_next = slot + 1 == NS ? 0 : slot + 1;                                                                           \
    const unsigned long long tw0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;                                       \
    /* `peek` = FULL word of this slot as read while the previous tile was in its MFMAs (stale at worst: slow poll) */     \
    if (DBG != 5 && __builtin_amdgcn_readfirstlane(peek) != (V) + 1) {                                                   \
      while (st_peek(full_lds + slot) != (V) + 1) __builtin_amdgcn_s_sleep(1);                                           \
    }                                                                                                                    \
    st_wave_fence();                                                                                                     \
    if constexpr (DBG == 4) t_wait += __builtin_amdgcn_s_memtime() - tw0;                                                \
    f32x16 acc[NJ];                                                                                                      \
    f32x4q accq[NJ][4];                                                                                                  \
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
    if constexpr (S5_PRIO != 0) __builtin_amdgcn_s_setprio(S5_PRIO);   /* MFMA phase wins the SIMD's issue arbitration */    \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                     \
      if (s + (PFV) < KS) {                                                                                                 \
        _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                \
          bf[(s + (PFV)) % ((PFV) + 1)][nj] = DBG == 6 ? ufrag[(s + nj + 1) % KS] : *reinterpret_cast<const f16x8*>(rowp + nj * 32 * ROWB + (((unsigned int)(s + (PFV)) << 5) ^ lxh)); \
      }                                                                                                                  \
      if (s == KS / 2) peek = *(volatile lds_int*)(full_lds + slot_next);                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                  \
        if constexpr (DBG == 7) { s5_pin8(bf[s % ((PFV) + 1)][nj]); acc[nj] = zero16; }                                       \
        else if constexpr (S5_SHAPE16 != 0) {                                                                                \
          /* lab (timing only, wrong results): the same FLOP as two 16x16x32 MFMAs on quarters of the accumulator */         \
          accq[nj][(2 * s) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s % ((PFV) + 1)][nj], ufrag[s], s < 2 ? zero4 : accq[nj][(2 * s) & 3], 0, 0, 0); \
          accq[nj][(2 * s + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s % ((PFV) + 1)][nj], ufrag[(s + 3) % KS], s < 2 ? zero4 : accq[nj][(2 * s + 1) & 3], 0, 0, 0); \
        }                                                                                                                  \
        else acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s % ((PFV) + 1)][nj], ufrag[s], s == 0 ? zero16 : acc[nj], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
    if constexpr (S5_SHAPE16 != 0) {                                                                                       \
      _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nj][r] = accq[nj][r >> 2][r & 3];                              \
    }                                                                                                                    \
    if constexpr (S5_PRIO != 0) __builtin_amdgcn_s_setprio(0);                                                           \
    s5_lds_done(acc[0], acc[NJ - 1]);                                                                                    \
    if constexpr (DBG == 3) t_mid = __builtin_amdgcn_s_memtime();                                                        \
    if constexpr (DBG != 5) s5_lds_add_lane0(free_lds + slot, 1);                                                        \
    /* exclusion events of this tile -> one bit per excluded score in the lane that holds it */                          \
    unsigned int ex = 0u;                                                                                                \
    bool have_ex = false;                                                                                                \
    if (has_excl) {                                                                                                      \
      const unsigned int tkey = (unsigned int)(j0 / ST_TILE);                                                            \
      while ((w0 >> 11) == tkey) {                                                                                       \
        ex |= lane == (int)((w0 >> 5) & 63u) ? 1u << (w0 & 31u) : 0u;                                                    \
        have_ex = true;                                                                                                  \
        S5_EV_NEXT()                                                                                                     \
      }                                                                                                                  \
    }


  // ---- pass 1: prefix tiles, running maximum per accumulator register (item class) ----
  float cm[16];                                            // item class = (lane half, accumulator register): 32 per user
#pragma unroll
  for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;
  if (partial) {
    for (int v = 0; v < n_pre; ++v) S5_TILE_SKIP(v, v * ST_TILE)
    S5_EV_RESTART()
  } else if (PRE && n_pre > 0) {
    for (int v = 0; v < n_pre; ++v) {
      const int j0 = v * ST_TILE;
      S5_TILE_BODY(v, PF_PRE)
      if (have_ex) {                                       // excluded scores must not raise a class maximum
        const unsigned int ex0 = ex;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nj][r] = ((ex0 >> (nj * 16 + r)) & 1u) ? -INFINITY : acc[nj][r];
        }
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
    // k-th largest of the user's 32 class maxima (16 in each of its two lanes), every lane pair for its own user. The threshold
    // admits scores EQUAL to the bound (its items are not in any buffer: the main pass meets them again): one ulp below it.
    {
      const float tk = s5_kth_of_32(cm, k);
      const unsigned int key = st_f2key(tk);
      // key 0x007FFFFF is -inf (fewer than k finite classes): no bound. One step below +0.0 in KEY order is -0.0, which the float
      // compare of the appends treats as EQUAL to +0.0 (a > -0.0 is false for a = +0.0: a user whose scores are all exactly zero got an
      // empty list in round 3) — the value below both zeros is the negative denormal -1.4e-45
      unsigned int below = key - 1u;
      below = below == 0x7FFFFFFFu ? 0x7FFFFFFEu : below;
      thr = key > 0x007FFFFFu ? st_key2f(below) : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;       // the main pass keeps the class maxima of what it APPENDS
    S5_EV_RESTART()                                        // the main pass starts again from the first tile
  }
  if constexpr (CML) {
#pragma unroll
    for (int r = 0; r < 16; ++r) *(__attribute__((address_space(3))) float*)(size_t)(cm_addr + r * 256) = -INFINITY;
  }

  // ---- pass 2: all tiles, lane-local threshold filter and appends ----
  const int pos_limit = lane_base + LIMIT * 8;
  // threshold refresh from the class maxima: after tiles 0, 1, 2, 3, 5, 8, 12, ... (gaps growing by half) while the thresholds are
  // still crude, every S5_RF tiles in the steady state
  int next_rf = (n_pre > 0 && !partial) ? S5_RF - 1 : 0;
  int part_next = part;                                    // next tile of this part (a full wave: every tile)
  for (int tl = 0; tl < n_tiles; ++tl) {
    if (tl != part_next) {                                 // (partial waves only) another part's tile
      S5_TILE_SKIP(n_pre + tl, tl * ST_TILE)
      continue;
    }
    part_next += n_parts;
    if (__ballot(pos > pos_limit)) {
      // ---- overflow (cold): a (user, half) buffer is nearly full — select that user's k best so that this tile's appends fit
      const unsigned long long tm0 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
      unsigned long long need = __ballot(pos > pos_limit);
      need = (need | (need >> 32)) & 0xFFFFFFFFull;
      const int cnt = (pos - lane_base) >> 3;
      while (need) {
        const int u = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        const int n0 = __builtin_amdgcn_readlane(cnt, u), n1 = __builtin_amdgcn_readlane(cnt, u + 32);
        unsigned long long* b0 = wgb + (long)u * (2 * S5_CAPH);
        const float nt = s5_overflow_select(b0, b0 + S5_CAPH, n0, n1, k, lane);
        if (n0 + n1 >= k && l31 == u) {
          thr = nt > thr ? nt : thr;
          pos = lane_base + (half ? (k >> 1) : k - (k >> 1)) * 8;
        }
        if constexpr (DBG == 4) ++n_ins;
      }
      if constexpr (DBG == 4) t_cmp += __builtin_amdgcn_s_memtime() - tm0;
    }
    const int j0 = tl * ST_TILE;
    const int vseq = n_pre + tl;
    const unsigned long long ti0 = (DBG == 4 || DBG == 3) ? __builtin_amdgcn_s_memtime() : 0ull;
    S5_TILE_BODY(vseq, PF)
    if constexpr (DBG == 3) t_issue += t_mid - ti0;
    const unsigned long long ti1 = DBG == 4 ? __builtin_amdgcn_s_memtime() : 0ull;
    if constexpr (DBG == 4) t_issue += ti1 - ti0 - (__builtin_amdgcn_s_memtime() - ti1);
    if constexpr (DBG == 8 || DBG == 9) {
      // lab (timing only): the class-maxima-only main pass of a two-pass scorer — per tile one v_max3 per accumulator register pair
      // into the lane's 16 class maxima, stored and reset every ST_X tiles (a group = one register class over ST_X tiles = 64 items);
      // 9: with the exclusion bits applied first
      constexpr int ST_X = NJ == 2 ? 32 : 64;
      if constexpr (DBG == 9) {
        if (have_ex) {
          const unsigned int ex0 = ex;
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nj][r] = ((ex0 >> (nj * 16 + r)) & 1u) ? -INFINITY : acc[nj][r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (NJ == 2) cm[r] = __builtin_fmaxf(cm[r], __builtin_fmaxf(acc[0][r], acc[1][r]));
        else cm[r] = fmaxf(cm[r], acc[0][r]);
      }
      if ((tl & (ST_X - 1)) == ST_X - 1 || tl == n_tiles - 1) {
        const int n_st = (n_tiles + ST_X - 1) / ST_X;
        f32x4q* o = reinterpret_cast<f32x4q*>(gbuf) + (((brow0 >> 5) * n_st + tl / ST_X) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4q v = {cm[4 * q], cm[4 * q + 1], cm[4 * q + 2], cm[4 * q + 3]};
          o[q * 64] = v;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;
      }
      continue;
    }
    if constexpr (DBG == 1 || (DBG >= 5 && DBG <= 7)) {
      s5_pin(acc[0], acc[NJ - 1]);
      continue;
    }
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
    // Threshold ladder. A wave is in-order and a vector -> scalar hand-over (v_cmp -> s_cbranch, v_cmp -> s_and_saveexec) costs
    // ~25 cycles every time, so the ladder does ALL its vector work first — the maximum of every PAIR of accumulator registers
    // and one v_cmp per pair into its own SGPR pair, back to back — then dispatches on scalar registers only (one branch for
    // "nothing in this tile", s_cmp + branch per pair), and a pair that fired runs a branch-free append per register in which
    // the vector ALU writes EXEC itself (s5_try_append). Measured with one wave per SIMD, per tile: 1,550 cycles when every
    // group and every register paid hand-overs of its own, ~? after.
    unsigned long long gm[8 * NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
#if S5_LADDER >= 1
        // maximum and compare in ONE asm block: behind a separate v_max block hipcc pads every v_cmp with an s_nop (sixteen issue
        // slots per tile that no hazard asks for)
        float mx;
        asm("v_max_f32 %1, %2, %3\n\tv_cmp_gt_f32_e64 %0, %1, %4" : "=s"(gm[nj * 8 + g]), "=&v"(mx) : "v"(acc[nj][2 * g]), "v"(acc[nj][2 * g + 1]), "v"(thr));
#else
        gm[nj * 8 + g] = __ballot(s5_max2(acc[nj][2 * g], acc[nj][2 * g + 1]) > thr);
#endif
      }
    }
#if S5_LADDER >= 2
    const bool any_g = true;                                 // the sixteen scalar tests below cost what their OR would
#else
    unsigned long long any_g = 0ull;
#pragma unroll
    for (int i = 0; i < 8 * NJ; ++i) any_g |= gm[i];
#endif
    if constexpr (DBG == 2) { if (any_g) asm volatile("s_nop 0"); continue; }
    if (any_g) {
      // item of register r: nj * 32 + (r & 3) + 8 * (r >> 2) (+ 4 * half, in item_lane); exclusion bit nj * 16 + r. A pair that
      // did not fire is the common case: its test falls through (the append blocks are laid out of line: a taken branch costs the
      // wave an instruction-fetch bubble, sixteen of them per tile)
#define S5_PAIR(NJI, G, EXF)                                                                                             \
      if (__builtin_expect(gm[(NJI) * 8 + (G)] != 0ull, 0)) {                                                            \
        s5_try_append<(1u << ((NJI) * 16 + 2 * (G))), (NJI) * 32 + ((2 * (G)) & 3) + 8 * ((2 * (G)) >> 2), EXF, CML, 2 * (G)>(acc[NJI][2 * (G)], thr, ex, pos, item_lane, wrs, cm[2 * (G)], cm_addr);             \
        s5_try_append<(1u << ((NJI) * 16 + 2 * (G) + 1)), (NJI) * 32 + ((2 * (G) + 1) & 3) + 8 * ((2 * (G) + 1) >> 2), EXF, CML, 2 * (G) + 1>(acc[NJI][2 * (G) + 1], thr, ex, pos, item_lane, wrs, cm[2 * (G) + 1], cm_addr); \
        if constexpr (DBG == 4) ++n_evt;                                                                                 \
      }
#define S5_PAIRS(EXF)                                                                                                    \
      S5_PAIR(0, 0, EXF) S5_PAIR(0, 1, EXF) S5_PAIR(0, 2, EXF) S5_PAIR(0, 3, EXF) S5_PAIR(0, 4, EXF) S5_PAIR(0, 5, EXF) S5_PAIR(0, 6, EXF) S5_PAIR(0, 7, EXF) \
      if constexpr (NJ == 2) {                                                                                           \
        S5_PAIR(NJ - 1, 0, EXF) S5_PAIR(NJ - 1, 1, EXF) S5_PAIR(NJ - 1, 2, EXF) S5_PAIR(NJ - 1, 3, EXF) S5_PAIR(NJ - 1, 4, EXF) S5_PAIR(NJ - 1, 5, EXF) S5_PAIR(NJ - 1, 6, EXF) S5_PAIR(NJ - 1, 7, EXF) \
      }
#if S5_EXSPLIT
      if (have_ex) { S5_PAIRS(true) } else { S5_PAIRS(false) }
#else
      S5_PAIRS(true)
#endif
#undef S5_PAIRS
#undef S5_PAIR
    }
    if constexpr (DBG == 4) t_ladder += __builtin_amdgcn_s_memtime() - ti1;
    if constexpr (DBG == 3) t_ladder += __builtin_amdgcn_s_memtime() - t_mid;
    if (tl >= next_rf) {
      // every later item has a larger index than the k buffered items at or above the bound: it needs a strictly larger score
      if constexpr (CML) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ds_max updates have been performed
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[r] = *(volatile __attribute__((address_space(3))) float*)(size_t)(cm_addr + r * 256);
      }
      const float tk = s5_kth_of_32(cm, k);
      thr = tk > thr ? tk : thr;
      const int gap = (tl + 2) >> 1;
      next_rf = tl + (gap < S5_RF ? gap : S5_RF);
      if constexpr (DBG == 4) ++n_cand;
    }
  }
#undef S5_TILE_BODY
#undef S5_TILE_SKIP
#undef S5_EV_NEXT

  if constexpr (DBG != 0) {
    if (lane == 0 && dbgbuf) {
      unsigned long long* d = dbgbuf + ((long)blockIdx.x * S5_MAXW + wave) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_evt; d[3] = n_cand; d[4] = n_evt | (t_issue << 20); d[5] = n_ins | (t_ladder << 20); d[6] = t_cmp;
      d[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
  }
  // fill counts and final thresholds of the wave's 64 buffer halves: the final selection + ranking is score_topk_finalize_kernel's
  {
    int2 o;
    o.x = (pos - lane_base) >> 3;
    o.y = (int)__float_as_uint(thr);
    reinterpret_cast<int2*>(cnt_out)[(brow0 + l31) * 2 + half] = o;
  }
}

// ---- final selection: one wave per user, four users per workgroup. A user's candidates sit in ONE pair of buffer halves (its unit
// was scored by a full wave) or in P pairs (a remainder unit cut into P parts by item tile, see the scorer). Only candidates at or
// above the largest of the sources' final thresholds can be among the k best (k buffered items lie at or above each): they are
// filtered first (~25 of ~110), gathered into one entry per lane through LDS, ranked by counting (score desc, item index asc) and the
// lanes of rank < k write the output. More than 64 survivors (ties at the threshold, a threshold that never rose): a bitwise binary
// search over ballot counts (entries re-read per round: cold) finds the k-th largest composite key first and exactly k survive.
// Empty slots (-inf, -1) behind fewer than k candidates.
__global__ __launch_bounds__(256) void score_topk_finalize_kernel(long Bu, int k, long n_full_units, int P, const int* __restrict__ cnt,
                                                                  const unsigned long long* __restrict__ gbuf, float* __restrict__ out_val,
                                                                  int* __restrict__ out_idx) {
  constexpr int CAP = 256;                                   // survivors a wave can stage in LDS
  __shared__ unsigned long long stage[4][CAP];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long ur = (long)blockIdx.x * 4 + w;
  if (ur >= Bu) return;                                      // wave-uniform
  const long unit = ur >> 5, n_units = (Bu + 31) >> 5;
  const int n_src = unit < n_full_units ? 1 : P;
  const long row_first = unit < n_full_units ? ur : n_units * 32 + (unit - n_full_units) * P * 32 + (ur & 31);      // + 32 per part
  auto key_of = [&](unsigned long long raw) -> unsigned long long {     // 0 for an empty slot (raw entries are never 0: ~item != 0)
    return raw ? (((unsigned long long)st_f2key(__uint_as_float((unsigned int)(raw >> 32))) << 32) | (raw & 0xFFFFFFFFull)) : 0ull;
  };
  const int4 c_first = reinterpret_cast<const int4*>(cnt)[row_first];
  float thr = __uint_as_float((unsigned int)__builtin_amdgcn_readfirstlane(c_first.y));
  // every (source, half, 64-entry chunk) in a fixed order; f(raw entry of this lane or 0). The loads of a source are issued together.
  auto for_chunks = [&](auto&& f) {
    for (int sidx = 0; sidx < n_src; ++sidx) {
      const long row = row_first + 32L * sidx;
      const int4 c = sidx == 0 ? c_first : reinterpret_cast<const int4*>(cnt)[row];       // (n0, thr bits, n1, thr bits)
      const int n0 = __builtin_amdgcn_readfirstlane(c.x), n1 = __builtin_amdgcn_readfirstlane(c.z);
      const unsigned long long* b0 = gbuf + row * (2 * S5_CAPH);
      unsigned long long raw[2 * S5_EH];
#pragma unroll
      for (int j = 0; j < 2 * S5_EH; ++j) {
        const int hh = j / S5_EH, q = (j % S5_EH) * 64 + lane;
        raw[j] = q < (hh ? n1 : n0) ? b0[hh * S5_CAPH + q] : 0ull;
      }
#pragma unroll
      for (int j = 0; j < 2 * S5_EH; ++j) {
        if ((j % S5_EH) * 64 < (j / S5_EH ? n1 : n0)) f(raw[j]);                  // wave-uniform
      }
    }
  };
  for (int sidx = 1; sidx < n_src; ++sidx) {
    const float t = __uint_as_float((unsigned int)__builtin_amdgcn_readfirstlane(cnt[(row_first + 32L * sidx) * 4 + 1]));
    thr = t > thr ? t : thr;
  }
  // ---- gather the candidates at or above the threshold, one per lane
  unsigned long long cut = 0ull;                             // survivors: composite key >= cut (when the filter lets too many through)
  int n = 0;
  auto gather = [&](unsigned long long raw) {
    const bool keep = raw != 0ull && (cut ? key_of(raw) >= cut : __uint_as_float((unsigned int)(raw >> 32)) >= thr);
    const unsigned long long m = __ballot(keep);
    const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    if (keep && p < CAP) stage[w][p] = raw;
    n += __popcll(m);
  };
  for_chunks(gather);
  if (n > CAP) {                                             // cold: exactly k survive a cut at the k-th largest composite key
    auto count_ge = [&](unsigned long long C) {
      int cn = 0;
      for_chunks([&](unsigned long long raw) { cn += __popcll(__ballot(key_of(raw) >= C && raw != 0ull)); });
      return cn;
    };
    unsigned int T = 0u;
    int c_ge = 1 << 30;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = T | (1u << bit);
      const int cn = count_ge((unsigned long long)trial << 32);
      if (cn >= k) { T = trial; c_ge = cn; if (cn == k) break; }
    }
    cut = (unsigned long long)T << 32;
    if (c_ge != k) {
      unsigned int Lw = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = Lw | (1u << bit);
        Lw = count_ge(((unsigned long long)T << 32) | trial) >= k ? trial : Lw;
      }
      cut |= (unsigned long long)Lw;
    }
    st_wave_fence();
    n = 0;
    for_chunks(gather);
  }
  st_wave_fence();                                           // LDS operations of a wave execute in order
  unsigned long long e;
  if (n <= 64) {
    e = key_of(lane < n ? stage[w][lane] : 0ull);
  } else {
    // 65 .. CAP survivors (a user of a remainder unit: the parts' thresholds are those of a quarter of the catalogue each): the same
    // search for the k-th largest composite key, over registers
    unsigned long long e4[CAP / 64];
#pragma unroll
    for (int q = 0; q < CAP / 64; ++q) e4[q] = key_of(q * 64 + lane < n ? stage[w][q * 64 + lane] : 0ull);
    auto count_ge = [&](unsigned long long C) {
      int cn = 0;
#pragma unroll
      for (int q = 0; q < CAP / 64; ++q) cn += __popcll(__ballot(e4[q] >= C));
      return cn;
    };
    unsigned int T = 0u;
    int c_ge = 1 << 30;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = T | (1u << bit);
      const int cn = count_ge((unsigned long long)trial << 32);
      if (cn >= k) { T = trial; c_ge = cn; if (cn == k) break; }
    }
    unsigned long long kcut = (unsigned long long)T << 32;
    if (c_ge != k) {
      unsigned int Lw = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = Lw | (1u << bit);
        Lw = count_ge(((unsigned long long)T << 32) | trial) >= k ? trial : Lw;
      }
      kcut |= (unsigned long long)Lw;
    }
    st_wave_fence();
    n = 0;
#pragma unroll
    for (int q = 0; q < CAP / 64; ++q) {
      const bool keep = e4[q] >= kcut && e4[q] != 0ull;
      const unsigned long long m = __ballot(keep);
      const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (keep && p < 64) stage[w][p] = e4[q];
      n += __popcll(m);
    }
    st_wave_fence();
    e = lane < n ? stage[w][lane] : 0ull;
  }
  const int h32 = (int)(e >> 32), l32 = (int)e;
  int rk = 0;
  for (int j = 0; j < n; ++j) {
    const unsigned long long kj = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(h32, j) << 32) |
                                  (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(l32, j);
    rk += kj > e;
  }
  const int nk = n < k ? n : k;
  if (lane < n && rk < k) {
    out_val[ur * k + rk] = st_key2f((unsigned int)(e >> 32));
    out_idx[ur * k + rk] = (int)(0xFFFFFFFFu - (unsigned int)(e & 0xFFFFFFFFull));
  }
  if (lane >= nk && lane < k) {                              // fewer than k candidates: empty slots behind them
    out_val[ur * k + lane] = -INFINITY;
    out_idx[ur * k + lane] = -1;
  }
}

extern "C" long sbr_score_topk_f16_events_bytes(long Bu, long excl_nnz) { return s5_event_bytes(Bu, excl_nnz); }

static long s5_padded_users(long Bu) { return sbr_cdiv(Bu, 32) * 32 + 32L * S5_MAXW + 32L * s5_n_cu(); }      // whole units + the last workgroup's padding + one row group per partial wave
static long s5_workspace_bytes(long Bu) {
  // candidate buffers + fill counts + the cycle stamps of SBR_ST_DEBUG
  const long padded = s5_padded_users(Bu);
  return padded * 2 * S5_CAPH * 8 + s5_al16(padded * 2 * 8) + s5_al16((sbr_cdiv(Bu, 32) + S5_MAXW + s5_n_cu()) * S5_MAXW * 64L);
}

template <int KS, int NS, int NJ, bool PRE>
static int s5_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                     int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, void* ev_buf,
                     long ev_bytes, int build_events, hipStream_t s) {
  const S5Plan plan = s5_plan(Bu);
  const int W = plan.W;
  const long n_wg = plan.n_wg;
  const long padded = s5_padded_users(Bu);
  const long buf_bytes = padded * 2 * S5_CAPH * 8, cnt_bytes = s5_al16(padded * 2 * 8);
#ifdef SBR_LAB
  // lab builds only (tools/lab/build_scorer_variants.sh defines SBR_LAB; output under tools/lab/bin/): timing-only ablations of the
  // kernel (1 | 2 | 5 | 6 | 7 | 8 | 9: results are garbage by design) and cycle stamps (3 | 4). The product library compiles the
  // DBG = 0 kernel only and reads no environment variable on a launch path.
  const int dbg = getenv("SBR_ST_DEBUG") ? atoi(getenv("SBR_ST_DEBUG")) : 0;
#endif
  SBR_REQUIRE(n_wg * 32L * W + 32L * plan.n_part <= padded && sbr_cdiv(Bu, 32) * 32 + 32L * plan.n_part <= padded, "sbr_score_topk_f16: internal: padding");
  SBR_REQUIRE(workspace && workspace_bytes >= s5_workspace_bytes(Bu),
              "sbr_score_topk_f16: workspace of %ld bytes needed (sbr_score_topk_f16_workspace), %ld given", s5_workspace_bytes(Bu), workspace_bytes);
  int* cnt = (int*)((char*)workspace + buf_bytes);
  void* dbg_buf = (char*)workspace + buf_bytes + cnt_bytes;
  const bool with_excl = eptr != nullptr && excl_nnz > 0;
  S5Events evs = {nullptr, nullptr};
  if (with_excl) {
    const int rc = s5_build_events(ev_buf, ev_bytes, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, 32 * NJ, build_events != 0, &evs, s);
    if (rc) return rc;
  }
  const size_t lds = (size_t)NS * (32 * NJ) * KS * 32 + 2 * NS * 4 + 16 + (KS >= S5_CML_KS ? (size_t)S5_MAXW * 4096 : 0);      // + the class maxima of D = 256
  SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16: LDS budget exceeded (%zu bytes)", lds);
  // prefix pass (class maxima only, no appends) over the first S5_PRE_TILES tiles: its bound spares the main pass the appends of its
  // first tiles (every score passes a threshold of -inf), at the price of scoring those tiles twice (measured on c2: 0 tiles 1.76 ms,
  // 8: 1.59, 16: 1.54, 32: 1.55, 65: 1.58); SBR_ST_PRE overrides (tiles, lab)
  const int n_tiles = sbr_cdiv(I, 32 * NJ);
  int n_pre = n_tiles >= 96 ? S5_PRE_TILES : 0;
#ifdef SBR_LAB
  if (getenv("SBR_ST_PRE")) n_pre = atoi(getenv("SBR_ST_PRE"));
#endif
  if (n_pre > n_tiles) n_pre = n_tiles;
  if (n_pre < 0 || !PRE) n_pre = 0;
#ifdef SBR_LAB
  auto kern = dbg == 1 ? score_topk_f16_n_kernel<KS, NS, NJ, 1, PRE> : (dbg == 2 ? score_topk_f16_n_kernel<KS, NS, NJ, 2, PRE> :
              (dbg == 4 ? score_topk_f16_n_kernel<KS, NS, NJ, 4, PRE> : (dbg == 3 ? score_topk_f16_n_kernel<KS, NS, NJ, 3, PRE> : (dbg == 5 ? score_topk_f16_n_kernel<KS, NS, NJ, 5, PRE> :
              (dbg == 6 ? score_topk_f16_n_kernel<KS, NS, NJ, 6, PRE> : (dbg == 7 ? score_topk_f16_n_kernel<KS, NS, NJ, 7, PRE> :
              (dbg == 8 ? score_topk_f16_n_kernel<KS, NS, NJ, 8, PRE> : (dbg == 9 ? score_topk_f16_n_kernel<KS, NS, NJ, 9, PRE> :
               score_topk_f16_n_kernel<KS, NS, NJ, 0, PRE>))))))));
#else
  auto kern = score_topk_f16_n_kernel<KS, NS, NJ, 0, PRE>;
#endif
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit to %zu", lds);
    return SBR_ERR_HIP;
  }
  // (a full wave of the last workgroups may own a unit past the last user: it scores a copy of the last user and nobody reads its buffers)
  kern<<<(unsigned int)n_wg, (W + (plan.n_part > 0 ? 1 : 0) + S5_NL) * 64, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, evs.events, evs.group_base, item_offset, k,
                                                       n_pre, W, plan.n_part, plan.P, cnt, (unsigned long long*)workspace, (unsigned long long*)dbg_buf);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16");
  score_topk_finalize_kernel<<<(unsigned int)sbr_cdiv(Bu, 4), 256, 0, s>>>(Bu, k, plan.n_part > 0 ? (long)n_wg * W : (1L << 40), plan.P, cnt,
                                                                            (const unsigned long long*)workspace, out_val, out_idx);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16 (final selection)");
  return SBR_OK;
}

// D in {64, 128, 256}
static int s5_dispatch(const void* U, const void* It, int D, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                       int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, void* ev_buf, long ev_bytes,
                       int build_events, hipStream_t s) {
  switch (D) {
    case 64: return s5_launch<4, 8, 2, true>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    case 128: return s5_launch<8, S5_NS, 2, true>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    case 256: return s5_launch<16, S5_NS, 1, true>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    default:
      sbr_set_error("sbr_score_topk_f16: D=%d not supported by the narrow-wave kernel", D);
      return SBR_ERR_ARG;
  }
}

// the two-pass scorer (score_topk_f16_2p.hip)
bool s2_supported(int D, long Bu, int I, int k);
long s2_workspace_bytes(long Bu, int I);
int s2_dispatch(const void* U, const void* It, int D, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, void* ev_buf, long ev_bytes,
                int build_events, hipStream_t s);

// 0: automatic (two-pass where it applies: catalogues of >= 8,192 items), 1: always the one-pass kernel, 2: two-pass or an error.
// Both routes return the same lists bit for bit; the switch exists for tests and A/B timing.
static int g_route = 0;
extern "C" int sbr_score_topk_f16_route(int route) {
  const int prev = g_route;
  if (route >= 0 && route <= 2) g_route = route;
  return prev;
}

// bytes of the workspace of one call, whichever route it takes (one-pass: candidate buffers, fill counts; two-pass: group maxima, pair
// lists, candidate regions)
extern "C" long sbr_score_topk_f16_workspace(long Bu, int I, int k) {
  const long one = s5_workspace_bytes(Bu);
  const long two = s2_supported(128, Bu, I, k < 1 ? 1 : (k > 32 ? 32 : k)) ? s2_workspace_bytes(Bu, I) : 0;
  return one > two ? one : two;
}

// events / events_bytes: caller-owned buffer of sbr_score_topk_f16_events_bytes(Bu, excl_nnz) bytes (NULL without exclusions);
// build_events != 0: the event stream of (u_idx, exclusion CSR, item range, D) is built into it first (three small launches),
// 0: it holds the stream a previous call with the same (u_idx, CSR, item_offset, I, D) built.
extern "C" int sbr_score_topk_f16(const void* U_f16, const void* I_f16, int D, long Bu, int I, const long* u_idx,
                                  const long* excl_indptr, const int* excl_indices, long excl_nnz, int item_offset, int k, float* out_val,
                                  int* out_idx, void* workspace, long workspace_bytes, void* events, long events_bytes, int build_events,
                                  void* stream) {
  SBR_REQUIRE(k >= 1 && k <= 32, "sbr_score_topk_f16: k=%d outside [1, 32] (use sbr_gemm_f32 + sbr_topk_rows)", k);
  SBR_REQUIRE(I >= 1, "sbr_score_topk_f16: empty catalogue");
  if (Bu == 0) return SBR_OK;
  SBR_REQUIRE(U_f16 && I_f16 && out_val && out_idx, "sbr_score_topk_f16: null operand");
  SBR_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr), "sbr_score_topk_f16: exclusion CSR must be given whole or not at all");
  SBR_REQUIRE(D == 64 || D == 128 || D == 256, "sbr_score_topk_f16: D=%d not supported (64, 128, 256)", D);
  const bool two = s2_supported(D, Bu, I, k);
  SBR_REQUIRE(g_route != 2 || two, "sbr_score_topk_f16: the two-pass route was requested for a shape it does not take (I=%d)", I);
  if (two && g_route == 2)      // (checkpoint: the automatic route stays on the one-pass kernel until the two-pass scorer is the faster one)
    return s2_dispatch(U_f16, I_f16, D, Bu, I, u_idx, excl_indptr, excl_indices, excl_nnz, item_offset, k, out_val, out_idx, workspace,
                       workspace_bytes, events, events_bytes, build_events, (hipStream_t)stream);
  return s5_dispatch(U_f16, I_f16, D, Bu, I, u_idx, excl_indptr, excl_indices, excl_nnz, item_offset, k, out_val, out_idx, workspace,
                     workspace_bytes, events, events_bytes, build_events, (hipStream_t)stream);
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
