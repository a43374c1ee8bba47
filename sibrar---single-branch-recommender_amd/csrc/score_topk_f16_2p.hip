// Fused full-catalogue scorer, TWO-PASS form (round 4) — eval/eval.py:205-222: scores = U x I^T, out[excluded] = -inf, top-k; the score
// matrix is never written. Same results as the one-pass kernel (score_topk_f16_n.hip), bit for bit.
//
// Why. The one-pass kernel runs its MFMA stream at the rate the chip sustains (1.08 ms for 100k x 50k x 128) but pays another 0.35 -
// 0.45 ms for the top-k machinery — threshold ladder, appends, refreshes — issued by the same in-order waves that issue the MFMAs.
// Here the first pass does nothing but the MFMA stream and ONE running maximum per class of accumulator registers; everything that
// depends on a threshold happens afterwards, on 2 - 4 % of the scores:
//
//   pass 1  score_max_f16_kernel    the one-pass kernel's geometry (32 users per wave, item tiles through an LDS ring filled by LDS-DMA,
//                                   exclusions as a wave-uniform event stream). Per 512-item SUPERTILE and lane: the maximum of each of
//                                   4 register classes (class c = accumulator registers 4c .. 4c + 3 of every 32-item MFMA tile), i.e.
//                                   per user 8 GROUPS of 64 items per supertile: group (st, h, c) = items st * 512 + 8c + 4h + 32m + j
//                                   (m < 16, j < 4). Excluded scores do not count (an event names the lane and the register; the
//                                   register quads an event names are masked on their way into the maximum). The maxima are rounded DOWN to bf16 and
//                                   stored (M: 8 bytes per lane and supertile); each also feeds one of 16 threshold classes per lane
//                                   (32 per user), and at the end of the pass L = the k-th largest of the user's 32 class maxima: k
//                                   groups of different classes hold an item with score >= L, so the k-th best score is >= L.
//   select  score_select_kernel     one wave per 32-user unit, one sweep over M: every group with M >= L is a (user, group) PAIR (~31 per
//                                   user). A ballot over the unit's lanes IS the pair set of a group: one 32-bit word per (group, unit),
//                                   bit = user — a bitmap of G x units words (10 MB for 100k users x 50k items), every word written once.
//   pass 2  score_rescore_kernel    one workgroup per (superblock of 8,192 / 4,096 users, group): the group's 64 item rows in LDS, the
//                                   superblock's row of the bitmap expanded into the user list, the users' rows gathered 32 at a time as
//                                   the B operand, the SAME MFMA chain as pass 1 (same instruction, operand roles and k order: every
//                                   score comes out bit-identical); scores >= L are appended to the user's candidate list (one atomic
//                                   add per lane reserves its entries). Work items are ordered superblock-major so that the user rows a
//                                   superblock touches (2 MB) stay in the XCDs' L2 while its ~31 pairs per user are served.
//   final   score_finalize2_kernel  one wave per user: its candidate list (~45 entries, one coalesced load), exclusion filter (the
//                                   user's sorted CSR row in LDS: pass 2 does not see exclusions), exact ranking (score desc, item asc).
//
// Exactness. Every non-excluded item with score >= L lies in a group whose (masked) maximum is >= L, hence in a selected pair, hence
// among the candidates; at least k such items exist; so the k best of the candidates are the k best of the catalogue. Users whose
// candidate list overflows (massive ties: every group ties at the bound) or who have fewer than k scoreable items are HARD: their wave
// of the final kernel streams the whole catalogue itself with the same MFMA chain (slow, exact).
#include "score_topk_shared.h"

#define S2_SUPER 512                     // items per supertile
#define S2_CAND_CAP 128                  // candidate entries per user (typical: ~45)
#define S2_MIN_ITEMS 8192                // below: the one-pass kernel (all 32 threshold classes need groups)
#define S2_STAGE 256                     // candidates a wave of the final kernel can stage (the hard path compacts beyond)
#define S2_CHUNK 512                     // pairs of a work item staged in LDS at a time
#ifndef S2_ABL
#define S2_ABL 0                         // lab builds only (timing, wrong results): 1 = no atomics in pass 2's flush, 2 = no flush, 3 = no user-row gather, 4 = no epilogue, 5 = no epilogue and no MFMA, 6 = set-up only
#endif
#define S2_ROWCAP 256                    // exclusion-row entries a wave keeps in LDS (longer rows: binary search in memory)

typedef float f32x4q __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------------------
// pass 1
// ---------------------------------------------------------------------------------------------------------------------------------
// x rounded DOWN to a bf16 value (as fp32 bits with the low half clear): negative numbers grow in magnitude; -inf stays -inf
__device__ __forceinline__ unsigned int s2_floor_bf16(float x) {
  const unsigned int u = __float_as_uint(x);
  return (u + ((unsigned int)((int)u >> 31) & 0xFFFFu)) & 0xFFFF0000u;
}

// v_max3 as an instruction (fmaxf makes hipcc canonicalise every operand first: a v_max x, x each — seven instructions per register
// quad instead of two)
__device__ __forceinline__ float s2_max3(float a, float b, float c) {
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  return m;
}

template <int KS, int NS, int NJ>
__global__ __launch_bounds__(1024) void score_max_f16_kernel(const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I,
                                                             const unsigned int* __restrict__ events, const int* __restrict__ group_base, int k, int W,
                                                             int n_part, int P, uint4* __restrict__ M, float* __restrict__ Lbuf) {
  constexpr int D = KS * 16;
  constexpr int ST_TILE = 32 * NJ;
  constexpr int X = S2_SUPER / ST_TILE;                    // tiles per supertile
  constexpr int PF = NJ == 1 ? S5_PF1 : S5_PF2;
  constexpr int ROWB = D * 2;
  constexpr int TILEB = ST_TILE * ROWB;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_T = (ST_TILE * CPR) / 64;
  constexpr int LFL0 = (NS - 2) / S5_NL >= 1 ? (NS - 2) / S5_NL : 1;
  constexpr int LFL = LFL0 * PER_T <= 63 ? LFL0 : 63 / PER_T;
  static_assert(LFL >= 1 && LFL * PER_T <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_int* full_lds = (lds_int*)(smem + NS * TILEB);
  lds_int* free_lds = full_lds + NS;
  // threshold classes: [consumer wave][16 slots][64 lanes] floats behind the ring and its counters
  const unsigned int tc_addr = (unsigned int)(size_t)(smem + NS * TILEB + 2 * NS * 4 + 16) + (unsigned int)((threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 4);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // units as in the one-pass kernel: W full consumer waves per workgroup; remainder units are cut into P parts — here by supertile
  // PAIR (supertiles 2q, 2q + 1 belong to part q % P), so that every 16-byte row of M is produced whole by one wave
  const int Wb = W + ((int)blockIdx.x < n_part ? 1 : 0);
  const bool partial = wave == W && (int)blockIdx.x < n_part;
  const int part = partial ? (int)blockIdx.x % P : 0, n_parts = partial ? P : 1;
  const long n_full_units = (long)gridDim.x * W;
  const long unit = partial ? n_full_units + (int)blockIdx.x / P : (long)blockIdx.x * W + wave;
  const long n_units = (Bu + 31) >> 5;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
  const int n_st = (n_tiles + X - 1) / X;
  const int n_st2 = (n_st + 1) >> 1;                       // 16-byte rows of M per lane: two supertiles each

  if (t < NS) { full_lds[t] = 0; free_lds[t] = 0; }
  __syncthreads();

  const int cslots = W + (n_part > 0 ? 1 : 0);
  if (wave == W && n_part > 0 && !partial) return;
  if (wave >= cslots) {
    // ---------------------------------------------- loader waves (as in the one-pass kernel) ---------------------------------------
    const int lw = wave - cslots;
    int n_mine = 0, v_last = -1;
    for (int v = lw; v < n_tiles; v += S5_NL) {
      const int slot = v % NS;
      if (v >= NS) {
        const int need = Wb * (v / NS);
        while (st_peek(free_lds + slot) < need) __builtin_amdgcn_s_sleep(1);
      }
      const int j0 = v * ST_TILE;
      unsigned char* dst = smem + slot * TILEB;
#pragma unroll
      for (int q = 0; q < PER_T; ++q) {
        const int Pq = q * 64 + lane;
        const int i = Pq / CPR, cp = Pq % CPR;
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

  // ------------------------------------------------ consumer waves ------------------------------------------------------------------
  f16x8 ufrag[KS];
  {
    const long r = unit * 32 + l31;
    const long ur = r < Bu ? r : Bu - 1;
    const f16x8* src = reinterpret_cast<const f16x8*>(U + ur * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
#pragma unroll
    for (int s = 0; s < KS; ++s) s5_pin8(ufrag[s]);
  }
  // exclusion events (same stream as the one-pass kernel reads): tile << 11 | lane << 5 | accumulator bit (nj * 16 + register)
  const bool has_excl = events != nullptr && unit < n_units;
  typedef const __attribute__((address_space(4))) unsigned int* ev_ptr;
  typedef unsigned int ev_quad __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) ev_quad* ev_quad_ptr;
  ev_ptr evp = nullptr;
  unsigned int w0 = S5_EV_NONE, w1 = S5_EV_NONE, w2 = S5_EV_NONE, w3 = S5_EV_NONE, n0 = S5_EV_NONE, n1 = S5_EV_NONE, n2 = S5_EV_NONE, n3 = S5_EV_NONE;
  int ev_rem = 4, ev_q = 8;
  if (has_excl) {
    evp = (ev_ptr)events + ((const __attribute__((address_space(4))) int*)group_base)[unit];
    const ev_quad qa = *(ev_quad_ptr)(evp), qb = *(ev_quad_ptr)(evp + 4);
    w0 = qa.x; w1 = qa.y; w2 = qa.z; w3 = qa.w; n0 = qb.x; n1 = qb.y; n2 = qb.z; n3 = qb.w;
  }
#define S2_EV_NEXT()                                                                                                     \
        w0 = w1; w1 = w2; w2 = w3;                                                                                       \
        if (--ev_rem == 0) {                                                                                             \
          w0 = n0; w1 = n1; w2 = n2; w3 = n3;                                                                            \
          const ev_quad qn = *(ev_quad_ptr)(evp + ev_q);                                                                 \
          n0 = qn.x; n1 = qn.y; n2 = qn.z; n3 = qn.w;                                                                    \
          ev_rem = 4; ev_q += 4;                                                                                         \
        }
  int peek = 0;
  int slot_next = 0;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  float cm[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) cm[c] = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) *(__attribute__((address_space(3))) float*)(size_t)(tc_addr + r * 256) = -INFINITY;
  uint4* mrow = M + (unit * n_st2) * 64 + lane;            // this lane's slot of supertile pair 0 (a unit's rows are contiguous)
  unsigned int pk0 = 0xFF80FF80u, pk1 = 0xFF80FF80u;       // the even supertile's packed maxima, kept until the odd one is done (-inf)

  int st = 0, tin = 0;                                      // supertile of the tile, tile inside the supertile
  int st_part = 0;                                          // st % n_parts
  for (int tl = 0; tl < n_tiles; ++tl) {
    const int slot = slot_next;
    slot_next = slot + 1 == NS ? 0 : slot + 1;
    const bool mine = st_part == part;                      // wave-uniform (full waves: always)
    if (!mine) {
      // another part's tile: wait for it, release it, pass its events by (the ring's bookkeeping counts every consumer wave)
      while (st_peek(full_lds + slot) != tl + 1) __builtin_amdgcn_s_sleep(1);
      st_wave_fence();
      s5_lds_add_lane0(free_lds + slot, 1);
      peek = 0;
      if (has_excl) {
        const unsigned int tkey = (unsigned int)tl;
        while ((w0 >> 11) == tkey) { S2_EV_NEXT() }
      }
    } else {
      if (__builtin_amdgcn_readfirstlane(peek) != tl + 1) {
        while (st_peek(full_lds + slot) != tl + 1) __builtin_amdgcn_s_sleep(1);
      }
      st_wave_fence();
      f32x16 acc[NJ];
      f16x8 bf[PF + 1][NJ];
      const unsigned char* rowp = smem + slot * TILEB + l31 * ROWB;
      unsigned int lxh = (unsigned int)(((l31 & SWZ) << 4) ^ (half << 4));
      asm volatile("" : "+v"(lxh));
#pragma unroll
      for (int s = 0; s < PF && s < KS; ++s) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) bf[s][nj] = *reinterpret_cast<const f16x8*>(rowp + nj * 32 * ROWB + (((unsigned int)s << 5) ^ lxh));
      }
      if constexpr (S5_PRIO != 0) __builtin_amdgcn_s_setprio(S5_PRIO);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + PF < KS) {
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            bf[(s + PF) % (PF + 1)][nj] = *reinterpret_cast<const f16x8*>(rowp + nj * 32 * ROWB + (((unsigned int)(s + PF) << 5) ^ lxh));
        }
        if (s == KS / 2) peek = *(volatile lds_int*)(full_lds + slot_next);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          acc[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s % (PF + 1)][nj], ufrag[s], s == 0 ? zero16 : acc[nj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (S5_PRIO != 0) __builtin_amdgcn_s_setprio(0);
      s5_lds_done(acc[0], acc[NJ - 1]);
      s5_lds_add_lane0(free_lds + slot, 1);
      // exclusion events of this tile: one bit per excluded score in the lane that holds it (ex), and — scalar — which accumulator
      // registers are named by any event of the tile (sbits). The accumulators themselves are never written: a register-indirect write
      // (s_set_gpr_idx) made hipcc copy a whole 16-register tuple per tile and per event, a scalar switch over single-register
      // v_cndmasks merged every case through copies.
      unsigned int ex = 0u, sbits = 0u;
      if (has_excl) {
        const unsigned int tkey = (unsigned int)tl;
        while ((w0 >> 11) == tkey) {
          ex |= lane == (int)((w0 >> 5) & 63u) ? 1u << (w0 & 31u) : 0u;
          sbits |= 1u << (w0 & 31u);
          S2_EV_NEXT()
        }
      }
      const int j0 = tl * ST_TILE;
      if (j0 + ST_TILE > I) {                               // catalogue end inside the tile: padded columns do not count
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
      // class maxima: four registers of a tile at a time; a quad that an event names (rare, wave-uniform) masks its excluded scores first
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float m = cm[c];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          const unsigned int qmask = 0xFu << (nj * 16 + 4 * c);
          if (__builtin_expect((sbits & qmask) != 0u, 0)) {
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ((ex >> (nj * 16 + 4 * c + j)) & 1u) ? -INFINITY : acc[nj][4 * c + j];
            m = s2_max3(m, a[0], a[1]);
            m = s2_max3(m, a[2], a[3]);
          } else {
            m = s2_max3(m, acc[nj][4 * c], acc[nj][4 * c + 1]);
            m = s2_max3(m, acc[nj][4 * c + 2], acc[nj][4 * c + 3]);
          }
        }
        cm[c] = m;
      }
    }
    if (tin == X - 1 || tl == n_tiles - 1) {
      if (mine) {
        // the supertile's four group maxima of this lane: rounded down to bf16, stored, and fed to the threshold classes
        const unsigned int f0 = s2_floor_bf16(cm[0]), f1 = s2_floor_bf16(cm[1]), f2 = s2_floor_bf16(cm[2]), f3 = s2_floor_bf16(cm[3]);
        const unsigned int q0 = (f0 >> 16) | f1, q1 = (f2 >> 16) | f3;
        if ((st & 1) == 0 && st + 1 < n_st) {
          pk0 = q0; pk1 = q1;
        } else {
          uint4 o;
          if (st & 1) { o.x = pk0; o.y = pk1; o.z = q0; o.w = q1; }
          else { o.x = q0; o.y = q1; o.z = 0xFF80FF80u; o.w = 0xFF80FF80u; }      // the last supertile has no partner
          mrow[(long)(st >> 1) * 64] = o;
        }
        const unsigned int a = tc_addr + (unsigned int)((st & 3) * 1024);
        asm volatile("ds_max_f32 %0, %1\n\tds_max_f32 %0, %2 offset:256\n\tds_max_f32 %0, %3 offset:512\n\tds_max_f32 %0, %4 offset:768"
                     ::"v"(a), "v"(__uint_as_float(f0)), "v"(__uint_as_float(f1)), "v"(__uint_as_float(f2)), "v"(__uint_as_float(f3)) : "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) cm[c] = -INFINITY;
      }
      tin = 0;
      if (st & 1) st_part = st_part + 1 == n_parts ? 0 : st_part + 1;
      ++st;
    } else {
      ++tin;
    }
  }
#undef S2_EV_NEXT
  // L = the k-th largest of the user's 32 threshold-class maxima. A part wave has seen only its own supertiles: the selection kernel
  // recomputes the bound of remainder units from M.
  if (!partial) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float tc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tc[r] = *(volatile __attribute__((address_space(3))) float*)(size_t)(tc_addr + r * 256);
    const float L = s5_kth_of_32(tc, k);
    const long user = unit * 32 + l31;
    if (half == 0 && user < Bu) Lbuf[user] = L;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// selection: the (user, group) pairs with M >= L as a bitmap — one 32-bit word per (group, unit), bit = user of the unit
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s2_unpack8(uint4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
}

__global__ __launch_bounds__(256) void score_select_kernel(const uint4* __restrict__ M, const float* __restrict__ Lbuf, long n_units, int G,
                                                           long n_full_units, int n_st, long Bu, int k, const long* __restrict__ u_idx,
                                                           const long* __restrict__ indptr, unsigned int* __restrict__ bitmap,
                                                           float* __restrict__ Lout, int* __restrict__ cnt, int* __restrict__ hard,
                                                           long* __restrict__ row_lo, int* __restrict__ row_len) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long unit = (long)blockIdx.x * 4 + wave;
  if (unit >= n_units) return;                               // wave-uniform
  const int n_st2 = (n_st + 1) >> 1;
  const long user = unit * 32 + l31;
  const bool uvalid = user < Bu;
  const uint4* mrow = M + (unit * n_st2) * 64 + lane;
  float L = INFINITY;
  if (unit >= n_full_units) {
    // remainder unit (scored by part waves, none of which saw the whole catalogue): the bound from M itself, same threshold classes
    float tc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tc[r] = -INFINITY;
    for (int s2 = 0; s2 < n_st2; s2 += 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (s2 + q < n_st2) {
          float f[8];
          s2_unpack8(mrow[(long)(s2 + q) * 64], f);           // supertiles 2 (s2 + q), + 1: threshold slots ((st & 3) * 4 + c)
#pragma unroll
          for (int e = 0; e < 8; ++e) tc[q * 8 + e] = fmaxf(tc[q * 8 + e], f[e]);
        }
      }
    }
    L = s5_kth_of_32(tc, k);
  } else if (uvalid) {
    L = Lbuf[user];
  }
  if (!uvalid) L = INFINITY;
  if (uvalid && half == 0) {
    Lout[user] = L; cnt[user] = 0; hard[user] = 0;
    // the user's exclusion row for the final kernel (one dependent load chain less there)
    long eb = 0, ee = 0;
    if (indptr != nullptr) {
      const long row = u_idx ? u_idx[user] : user;
      eb = indptr[row];
      ee = indptr[row + 1];
    }
    row_lo[user] = eb;
    row_len[user] = (int)(ee - eb);
  }
  // one sweep: eight 16-byte rows (sixteen supertiles) in flight per lane; a ballot per class is the pair word of two groups
  for (int s8 = 0; s8 < n_st2; s8 += 8) {
    uint4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = s8 + q < n_st2 ? mrow[(long)(s8 + q) * 64] : make_uint4(0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float f[8];
      s2_unpack8(v[q], f);
      unsigned int my = 0u;                                  // lane e < 16 of the wave keeps the word of group slot e of this row
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned long long m = __ballot(f[e] >= L && f[e] > -INFINITY);
        // supertile 2 (s8 + q) + (e >> 2), class e & 3: half 0 = lanes 0 .. 31, half 1 = lanes 32 .. 63
        my = lane == e * 2 ? (unsigned int)m : my;
        my = lane == e * 2 + 1 ? (unsigned int)(m >> 32) : my;
      }
      const int st = 2 * (s8 + q) + (lane >> 3);             // lane = (st & 1) * 8 + c * 2 + h
      if (lane < 16 && s8 + q < n_st2 && st < n_st) {
        const int g = st * 8 + (lane & 1) * 4 + ((lane >> 1) & 3);      // the sixteen lanes cover sixteen consecutive g: one 64-byte store
        bitmap[unit * G + g] = my;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pass 2: re-score the selected (user, group) pairs, group by group
// ---------------------------------------------------------------------------------------------------------------------------------
// Branch-free append of one accumulator value to the lane's PRIVATE slots in LDS (four entries of (~item, score bits) per lane and
// MFMA block): lanes with a >= L write slot min(n, 3) and count. A fifth candidate of a 32-item half group overwrites the fourth — the
// lane then reports its user as hard (massive ties only).
#define S2_WCAP 256                      // candidate entries a wave collects (in LDS) before it flushes them to the users' lists
template <int OFF>
__device__ __forceinline__ void s2_try_append(float a, float L, int& n, unsigned int priv, unsigned int il) {
  unsigned int tmp, t2;
  asm volatile(
      "v_cmpx_le_f32_e32 %[L], %[a]\n\t"
      "v_subrev_u32_e32 %[tmp], %[off], %[il]\n\t"
      "v_min_u32_e32 %[t2], 3, %[n]\n\t"
      "v_lshl_add_u32 %[t2], %[t2], 3, %[priv]\n\t"
      "ds_write2_b32 %[t2], %[tmp], %[a] offset1:1\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "s_mov_b64 exec, -1"
      : [n] "+v"(n), [tmp] "=&v"(tmp), [t2] "=&v"(t2)
      : [a] "v"(a), [L] "v"(L), [il] "v"(il), [priv] "v"(priv), [off] "n"(OFF)
      : "vcc", "memory");
}

template <int KS>
__global__ __launch_bounds__(256) void score_rescore_kernel(const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, int item_offset,
                                                            const float* __restrict__ Lbuf, const unsigned int* __restrict__ bitmap, long n_units,
                                                            int ups, int G, int* __restrict__ cnt, unsigned long long* __restrict__ cand,
                                                            int* __restrict__ hard) {
  constexpr int D = KS * 16;
  constexpr int ROWB = D * 2;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_W = CPR / 4;                             // LDS-DMA instructions per wave for the 64-row item tile
  static_assert(PER_W >= 1, "D >= 32");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [64 rows][ROWB] item tile | wave sums | staged users | private slots [4 waves][64 lanes][4] | the waves' collected lists
  int* wsum = (int*)(smem + 64 * ROWB);                      // [8]
  int* ch_user = wsum + 8;                                   // [S2_CHUNK]
  unsigned long long* priv_all = (unsigned long long*)(ch_user + S2_CHUNK);      // [4][64][4]
  unsigned long long* l_raw_all = priv_all + 4 * 64 * 4;     // [4][S2_WCAP]
  int* l_user_all = (int*)(l_raw_all + 4 * S2_WCAP);         // [4][S2_WCAP]
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int sb = (int)(blockIdx.x / (unsigned int)G), g = (int)(blockIdx.x % (unsigned int)G);
  const int st = g >> 3, gh = (g >> 2) & 1, gc = g & 3;
  // the group's 64 item rows -> LDS first (rows of the one-pass tile layout: 16-byte chunk cp of row i at chunk cp ^ (i & SWZ)): the
  // transfer runs under the set-up below
  const int item_base = st * S2_SUPER + 8 * gc + 4 * gh;     // item of element e: item_base + 32 (e >> 2) + (e & 3)
#pragma unroll
  for (int q = 0; q < PER_W; ++q) {
    const int Pq = (wave * PER_W + q) * 64 + lane;
    const int i = Pq / CPR, cp = Pq % CPR;
    int gi = item_base + 32 * (i >> 2) + (i & 3);
    gi = gi < I ? gi : I - 1;
    const _Float16* src = It + (long)gi * D + ((cp ^ (i & SWZ)) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + (wave * PER_W + q) * 1024), 16, 0, 0);
  }
  // the superblock's column of the bitmap: thread t holds the word of unit u0 + t (bit = user of the unit)
  const long u0 = (long)sb * ups;
  const unsigned int word = (t < ups && u0 + t < n_units) ? bitmap[(u0 + t) * G + g] : 0u;
  const int nw = __popc(word);
  int incl = nw;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
  if (lane == 63) wsum[wave] = incl;
  st_wait_vmcnt<0>();                                        // the item tile has landed (and the word)
  __syncthreads();
  const int w0s = wsum[0], w1s = wsum[1], w2s = wsum[2], w3s = wsum[3];
  const int total = w0s + w1s + w2s + w3s;
  if (total == 0) return;                                    // workgroup-uniform
  const int first = incl - nw + (wave > 0 ? w0s : 0) + (wave > 1 ? w1s : 0) + (wave > 2 ? w2s : 0);      // position of this thread's first pair
  const unsigned char* rowp = smem + l31 * ROWB;
  const unsigned int lxh = (unsigned int)(((l31 & SWZ) << 4) ^ (half << 4));
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool tail_st = (st + 1) * S2_SUPER > I;              // the catalogue ends inside this supertile
  const int lane_item = item_base + 32 * half;               // element e = rb * 32 + (r & 3) + 8 (r >> 2) + 4 half: item lane_item + rb * 256 + (r >> 2) * 64 + (r & 3)
  const unsigned int il = 0xFFFFFFFFu - (unsigned int)(item_offset + lane_item);
  unsigned long long* priv = priv_all + (wave * 64 + lane) * 4;
  const unsigned int priv_addr = (unsigned int)(size_t)priv;
  unsigned long long* l_raw = l_raw_all + wave * S2_WCAP;
  int* l_user = l_user_all + wave * S2_WCAP;
  int wn = 0;                                                // entries in this wave's list (wave-uniform)
  // every collected candidate takes a slot of its user's list: one returning atomic each, the wave's in flight together
  auto flush = [&]() {
    st_wave_fence();
    for (int e = lane; e < wn; e += 64) {
      if (S2_ABL == 2) break;
      const int user = l_user[e];
      const int at = S2_ABL == 1 ? (e & 63) : atomicAdd(cnt + user, 1);
      if (at < S2_CAND_CAP) cand[(long)user * S2_CAND_CAP + at] = l_raw[e];
    }
    st_wave_fence();
    wn = 0;
  };
  for (int c0 = 0; c0 < total; c0 += S2_CHUNK) {
    const int cn = total - c0 < S2_CHUNK ? total - c0 : S2_CHUNK;
    if (c0 > 0) __syncthreads();                             // the previous chunk's users have been consumed
    // stage the chunk's users (expanding this thread's word)
    {
      unsigned int wbits = word;
      int q = first;
      while (wbits) {
        const int b = __ffs((int)wbits) - 1;
        wbits &= wbits - 1u;
        if (q >= c0 && q < c0 + S2_CHUNK) ch_user[q - c0] = (int)((u0 + t) * 32 + b);
        ++q;
      }
    }
    __syncthreads();
    if (S2_ABL == 6) continue;
    // the wave's MFMA blocks: 32 pairs each; the rows and the bound of the NEXT block are requested before the epilogue of the current one
    f16x8 ufrag[KS];
    int mb = wave;
    int user = 0;
    float L = INFINITY;
    if (mb * 32 < cn) {
      const int q = mb * 32 + l31;
      user = S2_ABL == 3 ? l31 : ch_user[q < cn ? q : 0];
      L = q < cn ? Lbuf[user] : INFINITY;
      const f16x8* src = reinterpret_cast<const f16x8*>(U + (long)user * D);
      if (S2_ABL == 7) {
        // lab (timing only): the same rows with whole-row wave instructions — instruction s reads rows 4 s .. 4 s + 3 of the block, 16 lanes per row
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int qq = mb * 32 + (4 * s + (lane >> 4)) % 32;
          ufrag[s] = reinterpret_cast<const f16x8*>(U + (long)ch_user[qq < cn ? qq : 0] * D)[lane & 15];
        }
      } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
      }
    }
    for (; mb * 32 < cn; mb += 4) {
      const int cur_user = user;
      const float cur_L = L;
      f32x16 acc[2];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(rowp + rb * 32 * ROWB + (((unsigned int)s << 5) ^ lxh));
          if (S2_ABL == 5 || S2_ABL == 7) { asm volatile("" ::"v"(af), "v"(ufrag[s])); acc[rb] = zero16; }      // lab: no MFMA
          else acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, ufrag[s], s == 0 ? zero16 : acc[rb], 0, 0, 0);
        }
      }
      if ((mb + 4) * 32 < cn) {
        const int qn = (mb + 4) * 32 + l31;
        user = S2_ABL == 3 ? l31 : ch_user[qn < cn ? qn : 0];
        L = qn < cn ? Lbuf[user] : INFINITY;
        const f16x8* src = reinterpret_cast<const f16x8*>(U + (long)user * D);
        if (S2_ABL == 7) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int qq = (mb + 4) * 32 + (4 * s + (lane >> 4)) % 32;
            ufrag[s] = reinterpret_cast<const f16x8*>(U + (long)ch_user[qq < cn ? qq : 0] * D)[lane & 15];
          }
        } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
        }
      }
      if (tail_st) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][r] = lane_item + rb * 256 + (r >> 2) * 64 + (r & 3) < I ? acc[rb][r] : -INFINITY;
        }
      }
      int n = 0;
      if (S2_ABL == 4 || S2_ABL == 5 || S2_ABL == 7) { asm volatile("" ::"v"(acc[0]), "v"(acc[1])); continue; }      // lab: no epilogue
#define S2_AP(RB, R) s2_try_append<(RB) * 256 + ((R) >> 2) * 64 + ((R) & 3)>(acc[RB][R], cur_L, n, priv_addr, il);
#define S2_AP16(RB) S2_AP(RB, 0) S2_AP(RB, 1) S2_AP(RB, 2) S2_AP(RB, 3) S2_AP(RB, 4) S2_AP(RB, 5) S2_AP(RB, 6) S2_AP(RB, 7) \
                    S2_AP(RB, 8) S2_AP(RB, 9) S2_AP(RB, 10) S2_AP(RB, 11) S2_AP(RB, 12) S2_AP(RB, 13) S2_AP(RB, 14) S2_AP(RB, 15)
      S2_AP16(0)
      S2_AP16(1)
#undef S2_AP16
#undef S2_AP
      // the block's candidates move from the private slots to the wave's list
      if (n > 4) { hard[cur_user] = 1; n = 4; }              // a fifth candidate in one half group: the exact path takes the user
      int incl_n = n;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl_n, d, 64); if (lane >= d) incl_n += v; }
      const int wave_n = __builtin_amdgcn_readlane(incl_n, 63);
      if (wave_n > 0) {
        if (wn + wave_n > S2_WCAP) flush();
        int o = wn + incl_n - n;
        st_wave_fence();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (e < n) { l_raw[o] = priv[e]; l_user[o] = cur_user; ++o; }
        }
        wn += wave_n;
      }
    }
  }
  flush();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// final selection
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long s2_key_of(unsigned long long raw) {      // 0 for an empty slot (raw entries are never 0: ~item != 0)
  return raw ? (((unsigned long long)st_f2key(__uint_as_float((unsigned int)(raw >> 32))) << 32) | (raw & 0xFFFFFFFFull)) : 0ull;
}

// is `item` (global index) in the sorted CSR row [b, e)?
__device__ __forceinline__ bool s2_excluded(const int* __restrict__ indices, long b, long e, int item) {
  while (b < e) {
    const long mid = (b + e) >> 1;
    const int v = indices[mid];
    if (v == item) return true;
    if (v < item) b = mid + 1; else e = mid;
  }
  return false;
}

// the k best of the n <= S2_STAGE raw entries staged in `stage` (one wave), written to the user's output row in order (score desc, item
// asc); empty slots (-inf, -1) behind fewer than k
__device__ __forceinline__ void s2_rank_write(unsigned long long* stage, int n, int k, int lane, float* __restrict__ out_val, int* __restrict__ out_idx) {
  unsigned long long e;
  if (n <= 64) {
    e = s2_key_of(lane < n ? stage[lane] : 0ull);
  } else {
    unsigned long long e4[S2_STAGE / 64];
#pragma unroll
    for (int q = 0; q < S2_STAGE / 64; ++q) e4[q] = s2_key_of(q * 64 + lane < n ? stage[q * 64 + lane] : 0ull);
    auto count_ge = [&](unsigned long long C) {
      int cn = 0;
#pragma unroll
      for (int q = 0; q < S2_STAGE / 64; ++q) cn += __popcll(__ballot(e4[q] >= C));
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
    for (int q = 0; q < S2_STAGE / 64; ++q) {
      const bool keep = e4[q] >= kcut && e4[q] != 0ull;
      const unsigned long long m = __ballot(keep);
      const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (keep && p < 64) stage[p] = e4[q];
      n += __popcll(m);
    }
    st_wave_fence();
    e = lane < n ? stage[lane] : 0ull;                       // (keys now, not raw entries)
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
    out_val[rk] = st_key2f((unsigned int)(e >> 32));
    out_idx[rk] = (int)(0xFFFFFFFFu - (unsigned int)(e & 0xFFFFFFFFull));
  }
  if (lane >= nk && lane < k) {
    out_val[lane] = -INFINITY;
    out_idx[lane] = -1;
  }
}

// HARD user: the wave streams the whole catalogue itself — item rows straight from memory as the A operand, the user's row in every
// column of the B operand, the same MFMA chain as both passes — keeps the candidates at or above its running k-th best in LDS and
// compacts them when the stage runs full. Exact for any input (massive ties, fewer than k scoreable items); slow (one wave, the whole
// catalogue).
template <int KS>
__device__ void s2_hard_user(const _Float16* __restrict__ U, const _Float16* __restrict__ It, long user, int I, int item_offset,
                             const int* __restrict__ indices, long eb, long ee, int k, unsigned long long* stage, int lane,
                             float* __restrict__ out_val, int* __restrict__ out_idx) {
  constexpr int D = KS * 16;
  const int l31 = lane & 31, half = lane >> 5;
  f16x8 ufrag[KS];
  {
    const f16x8* src = reinterpret_cast<const f16x8*>(U + user * D);
#pragma unroll
    for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned long long cut = 0ull;                             // composite key of the k-th best so far (0: fewer than k staged)
  int n = 0;
  for (int j0 = 0; j0 < I; j0 += 32) {
    int gi = j0 + l31;
    gi = gi < I ? gi : I - 1;
    const f16x8* arow = reinterpret_cast<const f16x8*>(It + (long)gi * D);
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(arow[2 * s + half], ufrag[s], s == 0 ? zero16 : acc, 0, 0, 0);
    // every column holds the same user: lane (l31 < 16, half) takes register l31 -> 32 distinct scores per block
    float sc = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) sc = (l31 & 15) == r ? acc[r] : sc;
    const int r = l31 & 15;
    const int item = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const unsigned long long raw = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)(item_offset + item));
    bool keep = l31 < 16 && item < I && sc == sc && s2_key_of(raw) > cut;
    if (keep) keep = !s2_excluded(indices, eb, ee, item_offset + item);
    const unsigned long long m = __ballot(keep);
    if (m) {
      const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (keep) stage[p] = raw;
      n += __popcll(m);
      st_wave_fence();
      if (n > S2_STAGE - 32) {
        // compaction: the k best stay, `cut` becomes the k-th best key (later entries must beat it: a later item has a larger index)
        unsigned long long e4[S2_STAGE / 64];
#pragma unroll
        for (int q = 0; q < S2_STAGE / 64; ++q) e4[q] = s2_key_of(q * 64 + lane < n ? stage[q * 64 + lane] : 0ull);
        auto count_ge = [&](unsigned long long C) {
          int cn = 0;
#pragma unroll
          for (int q = 0; q < S2_STAGE / 64; ++q) cn += __popcll(__ballot(e4[q] >= C));
          return cn;
        };
        unsigned long long C = 0ull;
        for (int bit = 63; bit >= 0; --bit) {
          const unsigned long long trial = C | (1ull << bit);
          if (count_ge(trial) >= k) C = trial;
        }
        st_wave_fence();
        int n2 = 0;
#pragma unroll
        for (int q = 0; q < S2_STAGE / 64; ++q) {
          const unsigned long long raw_q = q * 64 + lane < n ? stage[q * 64 + lane] : 0ull;
          const bool kp = e4[q] >= C && e4[q] != 0ull;
          const unsigned long long mm = __ballot(kp);
          const int pp = n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mm, 0u));
          st_wave_fence();
          if (kp) stage[pp] = raw_q;                          // pp <= the entry's old position: no entry is overwritten before it is read
          n2 += __popcll(mm);
        }
        st_wave_fence();
        n = n2;
        cut = C;
      }
    }
  }
  st_wave_fence();
  s2_rank_write(stage, n, k, lane, out_val, out_idx);
}

template <int KS>
__global__ __launch_bounds__(256) void score_finalize2_kernel(const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, int item_offset,
                                                              int k, const long* __restrict__ row_lo, const int* __restrict__ row_len,
                                                              const int* __restrict__ indices, const int* __restrict__ cnt, const int* __restrict__ hard,
                                                              const unsigned long long* __restrict__ cand, float* __restrict__ out_val,
                                                              int* __restrict__ out_idx) {
  __shared__ unsigned long long stage_all[4][S2_STAGE];
  __shared__ int row_all[4][S2_ROWCAP];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long user = (long)blockIdx.x * 4 + w;
  if (user >= Bu) return;                                    // wave-uniform
  unsigned long long* stage = stage_all[w];
  int* rowbuf = row_all[w];
  const int nc = __builtin_amdgcn_readfirstlane(cnt[user]);
  bool is_hard = __builtin_amdgcn_readfirstlane(hard[user]) != 0 || nc > S2_CAND_CAP;
  const long eb = row_lo[user];                              // the user's exclusion row (bounds looked up by the selection kernel)
  const int rowlen = __builtin_amdgcn_readfirstlane(row_len[user]);
  const long ee = eb + rowlen;
  const bool row_lds = rowlen <= S2_ROWCAP;
  float* ov = out_val + user * k;
  int* oi = out_idx + user * k;
  int n = 0;
  if (!is_hard) {
    // the candidate list (at most two entries per lane) and the exclusion row are requested together
    const unsigned long long* list = cand + user * S2_CAND_CAP;
    unsigned long long raw[S2_CAND_CAP / 64];
#pragma unroll
    for (int i = 0; i < S2_CAND_CAP / 64; ++i) raw[i] = i * 64 + lane < nc ? list[i * 64 + lane] : 0ull;
    if (row_lds) {
      for (int i = lane; i < rowlen; i += 64) rowbuf[i] = indices[eb + i];
    }
    st_wave_fence();
#pragma unroll
    for (int i = 0; i < S2_CAND_CAP / 64; ++i) {
      bool keep = raw[i] != 0ull;
      if (keep && rowlen > 0) {
        // exclusion filter (pass 2 does not see exclusions): binary search in the user's sorted CSR row
        const int item = (int)(0xFFFFFFFFu - (unsigned int)(raw[i] & 0xFFFFFFFFull));
        if (row_lds) {
          int lo = 0, hi = rowlen;
          while (lo < hi) { const int mid = (lo + hi) >> 1; const int v = rowbuf[mid]; if (v < item) lo = mid + 1; else hi = mid; }
          keep = !(lo < rowlen && rowbuf[lo] == item);
        } else {
          keep = !s2_excluded(indices, eb, ee, item);
        }
      }
      const unsigned long long m = __ballot(keep);
      const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (keep) stage[p] = raw[i];
      n += __popcll(m);
    }
    st_wave_fence();
    if (n < k) is_hard = true;                               // fewer than k: the catalogue holds fewer than k scoreable items (or a bug) — the exact path decides
  }
  if (is_hard) {
    s2_hard_user<KS>(U, It, user, I, item_offset, indices, eb, ee, k, stage, lane, ov, oi);
    return;
  }
  s2_rank_write(stage, n, k, lane, ov, oi);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------------------
static long s2_al(long b) { return (b + 255) & ~255L; }

struct S2Layout {
  long n_units, m_units, nu_pad, n_st, n_st2, G;
  long off_M, off_L, off_L2, off_cnt, off_hard, off_rowlo, off_rowlen, off_bitmap, off_cand, total;
};

static S2Layout s2_layout(long Bu, int I, const S5Plan& plan) {
  S2Layout l;
  l.n_units = sbr_cdiv(Bu, 32);
  const long rem_units = plan.n_part > 0 ? plan.n_part / plan.P : 0;
  l.m_units = (long)plan.n_wg * plan.W + rem_units;          // rows of M the pass-1 grid can write (>= n_units)
  if (l.m_units < l.n_units) l.m_units = l.n_units;
  l.nu_pad = (l.n_units + 63) & ~63L;                        // row length of the bitmap in words
  l.n_st = sbr_cdiv(I, S2_SUPER);
  l.n_st2 = (l.n_st + 1) / 2;
  l.G = l.n_st * 8;
  const long users = l.n_units * 32;
  long o = 0;
  l.off_M = o; o += s2_al(l.m_units * l.n_st2 * 64 * 16);
  l.off_L = o; o += s2_al(users * 4);
  l.off_L2 = o; o += s2_al(users * 4);
  l.off_cnt = o; o += s2_al(users * 4);
  l.off_hard = o; o += s2_al(users * 4);
  l.off_rowlo = o; o += s2_al(users * 8);
  l.off_rowlen = o; o += s2_al(users * 4);
  l.off_bitmap = o; o += s2_al(l.G * l.n_units * 4);
  l.off_cand = o; o += s2_al(users * S2_CAND_CAP * 8);
  l.total = o + 256;
  return l;
}

bool s2_supported(int D, long Bu, int I, int k) {
  if (!(D == 64 || D == 128 || D == 256) || k < 1 || k > 32 || Bu < 1) return false;
  if (I < S2_MIN_ITEMS || Bu > 4000000L) return false;       // (candidate lists behind one buffer descriptor: < 4 GB)
  return sbr_cdiv(Bu, D >= 256 ? 4096 : 8192) * (long)sbr_cdiv(I, S2_SUPER) * 8 < (1L << 31);
}

long s2_workspace_bytes(long Bu, int I) { return s2_layout(Bu, I, s5_plan(Bu)).total; }

template <int KS, int NS, int NJ>
static int s2_launch(const void* U, const void* It, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                     int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, void* ev_buf,
                     long ev_bytes, int build_events, hipStream_t s) {
  constexpr int D = KS * 16;
  const S5Plan plan = s5_plan(Bu);
  const S2Layout l = s2_layout(Bu, I, plan);
  SBR_REQUIRE(workspace && workspace_bytes >= l.total, "sbr_score_topk_f16: workspace of %ld bytes needed (sbr_score_topk_f16_workspace), %ld given",
              l.total, workspace_bytes);
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint4* M = (uint4*)(ws + l.off_M);
  float* Lb = (float*)(ws + l.off_L);
  float* Lb2 = (float*)(ws + l.off_L2);
  int* cnt = (int*)(ws + l.off_cnt);
  int* hard = (int*)(ws + l.off_hard);
  long* row_lo = (long*)(ws + l.off_rowlo);
  int* row_len = (int*)(ws + l.off_rowlen);
  unsigned int* bitmap = (unsigned int*)(ws + l.off_bitmap);
  unsigned long long* cand = (unsigned long long*)(ws + l.off_cand);
  const bool with_excl = eptr != nullptr && excl_nnz > 0;
  S5Events evs = {nullptr, nullptr};
  if (with_excl) {
    const int rc = s5_build_events(ev_buf, ev_bytes, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, 32 * NJ, build_events != 0, &evs, s);
    if (rc) return rc;
  }
  // ---- pass 1
  {
    const size_t lds = (size_t)NS * (32 * NJ) * KS * 32 + 2 * NS * 4 + 16 + (size_t)S5_MAXW * 4096;
    SBR_REQUIRE(lds <= 160 * 1024, "sbr_score_topk_f16 (two-pass): LDS budget exceeded (%zu bytes)", lds);
    auto kern = score_max_f16_kernel<KS, NS, NJ>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      sbr_set_error("sbr_score_topk_f16 (two-pass): cannot raise the dynamic LDS limit to %zu", lds);
      return SBR_ERR_HIP;
    }
    kern<<<(unsigned int)plan.n_wg, (plan.W + (plan.n_part > 0 ? 1 : 0) + S5_NL) * 64, lds, s>>>(
        (const _Float16*)U, (const _Float16*)It, Bu, I, evs.events, evs.group_base, k, plan.W, plan.n_part, plan.P, M, Lb);
    SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, pass 1)");
  }
  // ---- selection
  {
    const long n_full_units = plan.n_part > 0 ? (long)plan.n_wg * plan.W : (1L << 40);
    score_select_kernel<<<(unsigned int)sbr_cdiv(l.n_units, 4), 256, 0, s>>>(M, Lb, l.n_units, (int)l.G, n_full_units, (int)l.n_st, Bu, k, u_idx,
                                                                               with_excl ? eptr : nullptr, bitmap, Lb2, cnt, hard, row_lo, row_len);
    SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, selection)");
  }
  // ---- pass 2
  {
#ifdef S2_UPS
    const int ups = S2_UPS;                                  // lab
#else
    const int ups = D >= 256 ? 128 : 256;                    // units per superblock: 4,096 / 8,192 users = 2 MB of rows
#endif
    const long n_sb = sbr_cdiv(l.n_units, ups);
    const size_t lds = (size_t)64 * D * 2 + 32 + (size_t)S2_CHUNK * 4 + 4 * 64 * 4 * 8 + (size_t)4 * S2_WCAP * 12;
    score_rescore_kernel<KS><<<(unsigned int)(n_sb * l.G), 256, lds, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, item_offset, Lb2, bitmap,
                                                                            l.n_units, ups, (int)l.G, cnt, cand, hard);
    SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, pass 2)");
  }
  // ---- final selection
  score_finalize2_kernel<KS><<<(unsigned int)sbr_cdiv(Bu, 4), 256, 0, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, item_offset, k,
                                                                             row_lo, row_len, eidx, cnt, hard, cand, out_val, out_idx);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, final selection)");
  return SBR_OK;
}

int s2_dispatch(const void* U, const void* It, int D, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, void* ev_buf, long ev_bytes,
                int build_events, hipStream_t s) {
  switch (D) {
    case 64: return s2_launch<4, 8, 2>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    case 128: return s2_launch<8, S5_NS, 2>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    case 256: return s2_launch<16, S5_NS, 1>(U, It, Bu, I, u_idx, eptr, eidx, excl_nnz, item_offset, k, out_val, out_idx, workspace, workspace_bytes, ev_buf, ev_bytes, build_events, s);
    default:
      sbr_set_error("sbr_score_topk_f16 (two-pass): D=%d not supported", D);
      return SBR_ERR_ARG;
  }
}
