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
//   select  score_select_kernel     per block of 512 users: every group with M >= L is a (user, group) PAIR (~31 per user); the pairs
//                                   of the block are counting-sorted by group in LDS (two sweeps over M: count, scan, scatter).
//   pass 2  score_rescore_kernel    one workgroup per (superblock of 8,192 / 4,096 users, group): the group's 64 item rows in LDS, the
//                                   pairs' user rows gathered 32 at a time as the B operand, the SAME MFMA chain as pass 1 (same
//                                   instruction, operand roles and k order: every score comes out bit-identical), scores >= L are
//                                   appended to the pair's 64-byte candidate region (3 entries per lane half + a count; more go to a
//                                   per-user overflow list). Work items are ordered superblock-major so that the user rows a
//                                   superblock touches (2 MB) stay in the XCDs' L2 while its ~31 pairs per user are served.
//   final   score_finalize2_kernel  one wave per user: candidates of its pairs, exclusion filter (binary search in the user's CSR row:
//                                   pass 2 does not see exclusions), exact ranking by (score desc, item asc).
//
// Exactness. Every non-excluded item with score >= L lies in a group whose (masked) maximum is >= L, hence in a selected pair, hence
// among the candidates; at least k such items exist; so the k best of the candidates are the k best of the catalogue. Users for whom
// the bookkeeping does not fit (more than S2_JMAX groups at or above L — massive ties —, overflowing overflow lists, fewer than k
// scoreable items) are HARD: their wave of the final kernel streams the whole catalogue itself with the same MFMA chain (slow, exact).
#include "score_topk_shared.h"

#define S2_SUPER 512                     // items per supertile
#define S2_JMAX 96                       // pairs per user
#define S2_BU 16                         // units (32 users) per selection block
#define S2_BLOCK_USERS (32 * S2_BU)
#define S2_PAIRCAP (S2_BLOCK_USERS * S2_JMAX)      // pair slots of a selection block
#define S2_OVF_CAP 64                    // overflow candidates per user
#define S2_MIN_ITEMS 8192                // below: the one-pass kernel (all 32 threshold classes need groups)
#define S2_MAX_GROUPS 16000              // LDS histogram of the selection kernel
#define S2_STAGE 256                     // candidates a wave of the final kernel can stage

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
                                                             int n_part, int P, uint2* __restrict__ M, float* __restrict__ Lbuf) {
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
  // units as in the one-pass kernel: W full consumer waves per workgroup; remainder units are cut into P parts — here by SUPERTILE
  // (supertile st belongs to part st % P), so that every group maximum is produced whole by one wave
  const int Wb = W + ((int)blockIdx.x < n_part ? 1 : 0);
  const bool partial = wave == W && (int)blockIdx.x < n_part;
  const int part = partial ? (int)blockIdx.x % P : 0, n_parts = partial ? P : 1;
  const long n_full_units = (long)gridDim.x * W;
  const long unit = partial ? n_full_units + (int)blockIdx.x / P : (long)blockIdx.x * W + wave;
  const long n_units = (Bu + 31) >> 5;
  const int n_tiles = (I + ST_TILE - 1) / ST_TILE;
  const int n_st = (n_tiles + X - 1) / X;

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
  uint2* mrow = M + (unit * n_st) * 64 + lane;             // this lane's slot of supertile 0 (a unit's rows are contiguous)

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
        uint2 o;
        o.x = (f0 >> 16) | f1;
        o.y = (f2 >> 16) | f3;
        mrow[(long)st * 64] = o;
        const unsigned int a = tc_addr + (unsigned int)((st & 3) * 1024);
        asm volatile("ds_max_f32 %0, %1\n\tds_max_f32 %0, %2 offset:256\n\tds_max_f32 %0, %3 offset:512\n\tds_max_f32 %0, %4 offset:768"
                     ::"v"(a), "v"(__uint_as_float(f0)), "v"(__uint_as_float(f1)), "v"(__uint_as_float(f2)), "v"(__uint_as_float(f3)) : "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) cm[c] = -INFINITY;
      }
      tin = 0;
      ++st;
      st_part = st_part + 1 == n_parts ? 0 : st_part + 1;
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
// selection: (user, group) pairs with M >= L, counting-sorted by group per block of S2_BLOCK_USERS users
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s2_unpack(uint2 v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
}

__global__ __launch_bounds__(1024) void score_select_kernel(const uint2* __restrict__ M, const float* __restrict__ Lbuf, long n_units, long n_full_units,
                                                            int n_st, long Bu, int k, int* __restrict__ nsel, int* __restrict__ hard,
                                                            int* __restrict__ ovf_cnt, int* __restrict__ pair_of, int* __restrict__ offs,
                                                            int* __restrict__ pairs, float* __restrict__ Lout) {
  extern __shared__ int hist[];                              // [G + 1]
  __shared__ int part[1024];
  const int G = n_st * 8;
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long unit = (long)blockIdx.x * S2_BU + wave;
  const bool valid = unit < n_units;                         // wave-uniform
  const long user = unit * 32 + l31;
  const bool uvalid = valid && user < Bu;
  const uint2* mrow = M + (unit * n_st) * 64 + lane;
  float L = INFINITY;
  if (valid && unit >= n_full_units) {
    // remainder unit (scored by part waves): the bound from M itself, same threshold classes as pass 1
    float tc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tc[r] = -INFINITY;
    for (int s4 = 0; s4 < n_st; s4 += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s4 + q < n_st) {
          float f[4];
          s2_unpack(mrow[(long)(s4 + q) * 64], f);
#pragma unroll
          for (int c = 0; c < 4; ++c) tc[q * 4 + c] = fmaxf(tc[q * 4 + c], f[c]);
        }
      }
    }
    L = s5_kth_of_32(tc, k);
  } else if (uvalid) {
    L = Lbuf[user];
  }
  if (!uvalid) L = INFINITY;
  for (int i = t; i <= G; i += 1024) hist[i] = 0;
  __syncthreads();
  // ---- sweep A: count (eight supertiles' loads in flight per lane)
  int n = 0;
  if (valid) {
    for (int s8 = 0; s8 < n_st; s8 += 8) {
      uint2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = s8 + q < n_st ? mrow[(long)(s8 + q) * 64] : make_uint2(0xFF80FF80u, 0xFF80FF80u);      // -inf
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float f[4];
        s2_unpack(v[q], f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (f[c] >= L && f[c] > -INFINITY) { atomicAdd(hist + (s8 + q) * 8 + half * 4 + c, 1); ++n; }
        }
      }
    }
  }
  const int n_other = __shfl_xor(n, 32, 64);
  const int n_user = n + n_other;
  const bool is_hard = uvalid && n_user > S2_JMAX;
  if (__ballot(is_hard)) {                                   // cold: a hard user's pairs are not emitted
    if (is_hard) {
      for (int st = 0; st < n_st; ++st) {
        float f[4];
        s2_unpack(mrow[(long)st * 64], f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (f[c] >= L && f[c] > -INFINITY) atomicSub(hist + st * 8 + half * 4 + c, 1);
        }
      }
    }
  }
  __syncthreads();
  // ---- exclusive scan of the group counts -> offsets of the block's sorted pair list
  {
    const int per = (G + 1023) / 1024;
    const int i0 = t * per, i1 = i0 + per < G ? i0 + per : G;
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += hist[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int v = t >= d ? part[t - d] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    int run = part[t] - sum;
    int* o = offs + (long)blockIdx.x * (G + 1);
    for (int i = i0; i < i1; ++i) { const int c = hist[i]; hist[i] = run; o[i] = run; run += c; }
    if (t == 1023) { hist[G] = part[1023]; o[G] = part[1023]; }
  }
  __syncthreads();
  // ---- sweep B: scatter (hist[g] is now the running cursor of group g)
  if (valid && !is_hard) {
    int j = half ? n_other : 0;                              // the half-0 lane's pairs come first in the user's list
    const long pbase = (long)blockIdx.x * S2_PAIRCAP;
    for (int s8 = 0; s8 < n_st; s8 += 8) {
      uint2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = s8 + q < n_st ? mrow[(long)(s8 + q) * 64] : make_uint2(0xFF80FF80u, 0xFF80FF80u);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float f[4];
        s2_unpack(v[q], f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (f[c] >= L && f[c] > -INFINITY) {
            const int pos = atomicAdd(hist + (s8 + q) * 8 + half * 4 + c, 1);
            pairs[pbase + pos] = (int)user;
            pair_of[user * S2_JMAX + j] = (int)(pbase + pos);
            ++j;
          }
        }
      }
    }
  }
  if (uvalid && half == 0) {
    nsel[user] = is_hard ? 0 : n_user;
    hard[user] = is_hard ? 1 : 0;
    ovf_cnt[user] = 0;
    Lout[user] = L;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pass 2: re-score the selected (user, group) pairs, group by group
// ---------------------------------------------------------------------------------------------------------------------------------
// Branch-free append of one accumulator value: lanes with a >= L store the raw entry (~item = il - OFF, score bits) at slot min(n, 3) of
// their half region and count it. Slot 3 is the header slot: the 4th and later candidates of a lane land there and are overwritten by
// the count afterwards (stores of one lane to one address complete in order); a count above 3 sends the lane through the overflow path.
template <int OFF>
__device__ __forceinline__ void s2_try_append(float a, float L, int& n, int pos, unsigned int il, i32x4 rs) {
  unsigned int tmp, t2;
  asm volatile(
      "v_cmpx_le_f32_e32 %[L], %[a]\n\t"
      "v_subrev_u32_e32 %[tmp], %[off], %[il]\n\t"
      "v_min_u32_e32 %[t2], 3, %[n]\n\t"
      "v_lshl_add_u32 %[t2], %[t2], 3, %[pos]\n\t"
      "buffer_store_dword %[tmp], %[t2], %[rs], 0 offen\n\t"
      "buffer_store_dword %[a], %[t2], %[rs], 0 offen offset:4\n\t"
      "v_add_u32_e32 %[n], 1, %[n]\n\t"
      "s_mov_b64 exec, -1"
      : [n] "+v"(n), [tmp] "=&v"(tmp), [t2] "=&v"(t2)
      : [a] "v"(a), [L] "v"(L), [il] "v"(il), [pos] "v"(pos), [rs] "s"(rs), [off] "n"(OFF)
      : "vcc", "memory");
}

#define S2_CHUNK 1024                    // pairs of a work item staged in LDS at a time

template <int KS>
__global__ __launch_bounds__(256) void score_rescore_kernel(const _Float16* __restrict__ U, const _Float16* __restrict__ It, int I, int item_offset,
                                                            const float* __restrict__ Lbuf, const int* __restrict__ offs, const int* __restrict__ pairs,
                                                            int n_blocks, int bps, int G, unsigned long long* __restrict__ cand,
                                                            int* __restrict__ ovf_cnt, unsigned long long* __restrict__ ovf, int* __restrict__ hard) {
  constexpr int D = KS * 16;
  constexpr int ROWB = D * 2;
  constexpr int CPR = D / 8;
  constexpr int SWZ = (CPR >= 16) ? 15 : (CPR - 1);
  constexpr int PER_W = CPR / 4;                             // LDS-DMA instructions per wave for the 64-row item tile
  static_assert(PER_W >= 1, "D >= 32");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [64 rows][ROWB], segment tables, the staged pair chunk
  int* segstart = (int*)(smem + 64 * ROWB);                  // [17]
  int* segsrc = segstart + 32;                               // [16] first pair slot of the segment, relative to the superblock's first slot
  int* ch_user = segsrc + 32;                                // [S2_CHUNK]
  float* ch_L = (float*)(ch_user + S2_CHUNK);                // [S2_CHUNK]
  int* ch_p = (int*)(ch_L + S2_CHUNK);                       // [S2_CHUNK] pair slot relative to the superblock's first slot
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int sb = (int)(blockIdx.x / (unsigned int)G), g = (int)(blockIdx.x % (unsigned int)G);
  const int st = g >> 3, gh = (g >> 2) & 1, gc = g & 3;
  const int b0 = sb * bps;
  const int nb = n_blocks - b0 < bps ? n_blocks - b0 : bps;
  if (t < 64) {
    // segment lengths of the superblock's blocks for this group, prefix sum over the first 16 lanes
    int s0 = 0, len = 0;
    if (t < nb) {
      const int* o = offs + (long)(b0 + t) * (G + 1) + g;
      s0 = o[0];
      len = o[1] - s0;
    }
    int incl = len;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(incl, d, 64); if ((t & 63) >= d) incl += v; }
    if (t < 16) { segstart[t] = incl - len; segsrc[t] = t * S2_PAIRCAP + s0; }
    if (t == 15) segstart[16] = incl;
  }
  __syncthreads();
  const int total = segstart[16];
  if (total == 0) return;                                    // workgroup-uniform
  // the group's 64 item rows -> LDS (rows of the one-pass tile layout: 16-byte chunk cp of row i at chunk cp ^ (i & SWZ))
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
  // candidate regions of this superblock through one buffer descriptor (offsets stay far below 4 GB)
  const long cand_lo = (long)b0 * S2_PAIRCAP;                // first pair slot of the superblock
  i32x4 rs;
  {
    const unsigned long long cb = (unsigned long long)(cand + cand_lo * 8);
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)cb);
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)(cb >> 32) & 0xFFFF);
    rs[2] = (int)((long)nb * S2_PAIRCAP * 64);
    rs[3] = 0x00020000;
  }
  const int* pairs_sb = pairs + cand_lo;
  const unsigned char* rowp = smem + l31 * ROWB;
  const unsigned int lxh = (unsigned int)(((l31 & SWZ) << 4) ^ (half << 4));
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool tail_st = (st + 1) * S2_SUPER > I;              // the catalogue ends inside this supertile
  const int lane_item = item_base + 32 * half;               // element e = rb * 32 + (r & 3) + 8 (r >> 2) + 4 half: item lane_item + rb * 256 + (r >> 2) * 64 + (r & 3)
  const unsigned int il = 0xFFFFFFFFu - (unsigned int)(item_offset + lane_item);
  for (int c0 = 0; c0 < total; c0 += S2_CHUNK) {
    const int cn = total - c0 < S2_CHUNK ? total - c0 : S2_CHUNK;
    // stage the chunk's pairs: slot, user, bound (two dependent loads per pair, all of the chunk's in flight together)
    for (int q = t; q < cn; q += 256) {
      const int qq = c0 + q;
      int b = 0;
#pragma unroll
      for (int s_ = 1; s_ < 16; ++s_) b += qq >= segstart[s_] ? 1 : 0;
      const int p = segsrc[b] + (qq - segstart[b]);
      const int user = pairs_sb[p];
      ch_p[q] = p;
      ch_user[q] = user;
      ch_L[q] = Lbuf[user];
    }
    st_wait_vmcnt<0>();                                      // (also the item tile's LDS-DMA of the first chunk)
    __syncthreads();
    for (int mb = wave; mb * 32 < cn; mb += 4) {
      const int q = mb * 32 + l31;
      const bool active = q < cn;
      const int user = active ? ch_user[q] : 0;
      const float L = active ? ch_L[q] : INFINITY;
      const int p = active ? ch_p[q] : 0;
      f16x8 ufrag[KS];
      {
        const f16x8* src = reinterpret_cast<const f16x8*>(U + (long)user * D);
#pragma unroll
        for (int s = 0; s < KS; ++s) ufrag[s] = src[2 * s + half];
      }
      f32x16 acc[2];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(rowp + rb * 32 * ROWB + (((unsigned int)s << 5) ^ lxh));
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, ufrag[s], s == 0 ? zero16 : acc[rb], 0, 0, 0);
        }
      }
      if (tail_st) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][r] = lane_item + rb * 256 + (r >> 2) * 64 + (r & 3) < I ? acc[rb][r] : -INFINITY;
        }
      }
      const int pos = p * 64 + half * 32;
      int n = 0;
#define S2_AP(RB, R) s2_try_append<(RB) * 256 + ((R) >> 2) * 64 + ((R) & 3)>(acc[RB][R], L, n, pos, il, rs);
#define S2_AP16(RB) S2_AP(RB, 0) S2_AP(RB, 1) S2_AP(RB, 2) S2_AP(RB, 3) S2_AP(RB, 4) S2_AP(RB, 5) S2_AP(RB, 6) S2_AP(RB, 7) \
                    S2_AP(RB, 8) S2_AP(RB, 9) S2_AP(RB, 10) S2_AP(RB, 11) S2_AP(RB, 12) S2_AP(RB, 13) S2_AP(RB, 14) S2_AP(RB, 15)
      S2_AP16(0)
      S2_AP16(1)
#undef S2_AP16
#undef S2_AP
      if (active) cand[(cand_lo + p) * 8 + half * 4 + 3] = (unsigned long long)(unsigned int)n;      // header: the count
      if (__ballot(n > 3)) {
        // cold: a lane half with more than three candidates writes ALL of them to the user's overflow list (the final kernel ignores the
        // region of a half whose count is above 3)
        if (n > 3) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (acc[rb][r] >= L) {
                const int o = atomicAdd(ovf_cnt + user, 1);
                if (o < S2_OVF_CAP) {
                  ovf[(long)user * S2_OVF_CAP + o] = ((unsigned long long)__float_as_uint(acc[rb][r]) << 32) |
                                                     (unsigned long long)(il - (unsigned int)(rb * 256 + (r >> 2) * 64 + (r & 3)));
                } else {
                  hard[user] = 1;
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();                                         // the chunk's staging arrays are free again
  }
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

#define S2_ROWCAP 256                     // exclusion-row entries a wave keeps in LDS (longer rows: binary search in memory)

template <int KS>
__global__ __launch_bounds__(256) void score_finalize2_kernel(const _Float16* __restrict__ U, const _Float16* __restrict__ It, long Bu, int I, int item_offset,
                                                              int k, const long* __restrict__ u_idx, const long* __restrict__ indptr,
                                                              const int* __restrict__ indices, const int* __restrict__ nsel, const int* __restrict__ hard,
                                                              const int* __restrict__ ovf_cnt, const int* __restrict__ pair_of,
                                                              const unsigned long long* __restrict__ cand, const unsigned long long* __restrict__ ovf,
                                                              float* __restrict__ out_val, int* __restrict__ out_idx) {
  __shared__ unsigned long long stage_all[4][S2_STAGE];
  __shared__ int row_all[4][S2_ROWCAP];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long user = (long)blockIdx.x * 4 + w;
  if (user >= Bu) return;                                    // wave-uniform
  unsigned long long* stage = stage_all[w];
  int* rowbuf = row_all[w];
  // everything the wave needs is requested up front; the only dependent chain is pair_of -> candidate regions
  const int np = __builtin_amdgcn_readfirstlane(nsel[user]);
  const int no = __builtin_amdgcn_readfirstlane(ovf_cnt[user]);
  bool is_hard = __builtin_amdgcn_readfirstlane(hard[user]) != 0;
  long eb = 0, ee = 0;
  if (indptr != nullptr) {
    const long row = u_idx ? u_idx[user] : user;
    eb = indptr[row];
    ee = indptr[row + 1];
  }
  const int rowlen = (int)(ee - eb);
  const bool row_lds = rowlen <= S2_ROWCAP;
  float* ov = out_val + user * k;
  int* oi = out_idx + user * k;
  int n = 0;
  if (!is_hard) {
    const int p0 = lane < np ? pair_of[user * S2_JMAX + lane] : 0;
    const int p1 = lane + 64 < np ? pair_of[user * S2_JMAX + 64 + lane] : 0;
    if (row_lds) {
      for (int i = lane; i < rowlen; i += 64) rowbuf[i] = indices[eb + i];
    }
    auto take = [&](unsigned long long raw, bool ok) {
      const bool keep = ok && raw != 0ull;
      const unsigned long long m = __ballot(keep);
      const int p = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (keep && p < S2_STAGE) stage[p] = raw;
      n += __popcll(m);
    };
    const int nslots = np * 8;                               // 64-byte region per pair: slots 0-2 / 4-6 entries, 3 / 7 the two halves' counts
    for (int base = 0; base < nslots; base += 256) {
      unsigned long long raw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sl = base + i * 64 + lane;
        const int pair = sl >> 3;
        const int pa = __shfl(p0, pair & 63, 64), pb = __shfl(p1, pair & 63, 64);
        raw[i] = sl < nslots ? cand[(long)(pair < 64 ? pa : pb) * 8 + (sl & 7)] : 0ull;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sl = base + i * 64 + lane;
        const int hdr = __shfl((int)(unsigned int)raw[i], (lane & ~3) | 3, 64);        // the count of this slot's half: slot 3 / 7 of the pair
        take(raw[i], sl < nslots && (lane & 3) < 3 && (lane & 3) < hdr && hdr <= 3);
      }
    }
    if (no > 0) {
      const int m = no < S2_OVF_CAP ? no : S2_OVF_CAP;
      for (int i0 = 0; i0 < m; i0 += 64) take(i0 + lane < m ? ovf[user * S2_OVF_CAP + i0 + lane] : 0ull, i0 + lane < m);
    }
    if (n > S2_STAGE) is_hard = true;
    st_wave_fence();
    if (!is_hard && rowlen > 0) {
      // exclusion filter (pass 2 does not see exclusions): every staged candidate against the user's sorted CSR row, in place
      int n2 = 0;
      for (int c0 = 0; c0 < n; c0 += 64) {
        const unsigned long long raw = c0 + lane < n ? stage[c0 + lane] : 0ull;
        const int item = (int)(0xFFFFFFFFu - (unsigned int)(raw & 0xFFFFFFFFull));
        bool keep = raw != 0ull;
        if (keep) {
          if (row_lds) {
            int lo = 0, hi = rowlen;
            while (lo < hi) { const int mid = (lo + hi) >> 1; const int v = rowbuf[mid]; if (v < item) lo = mid + 1; else hi = mid; }
            keep = !(lo < rowlen && rowbuf[lo] == item);
          } else {
            keep = !s2_excluded(indices, eb, ee, item);
          }
        }
        const unsigned long long m = __ballot(keep);
        const int p = n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
        st_wave_fence();
        if (keep) stage[p] = raw;                            // p <= c0 + lane: nothing unread is overwritten
        n2 += __popcll(m);
      }
      n = n2;
      st_wave_fence();
    }
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
  long n_units, m_units, n_blocks, n_st, G;
  long off_M, off_L, off_L2, off_nsel, off_hard, off_ovfc, off_pairof, off_offs, off_pairs, off_cand, off_ovf, total;
};

static S2Layout s2_layout(long Bu, int I, const S5Plan& plan) {
  S2Layout l;
  l.n_units = sbr_cdiv(Bu, 32);
  const long rem_units = plan.n_part > 0 ? plan.n_part / plan.P : 0;
  l.m_units = (long)plan.n_wg * plan.W + rem_units;          // rows of M the pass-1 grid can write (>= n_units)
  if (l.m_units < l.n_units) l.m_units = l.n_units;
  l.n_blocks = sbr_cdiv(l.n_units, S2_BU);
  l.n_st = sbr_cdiv(I, S2_SUPER);
  l.G = l.n_st * 8;
  const long users = l.n_blocks * S2_BLOCK_USERS;
  long o = 0;
  l.off_M = o; o += s2_al(l.m_units * l.n_st * 64 * 8);
  l.off_L = o; o += s2_al(users * 4);
  l.off_L2 = o; o += s2_al(users * 4);
  l.off_nsel = o; o += s2_al(users * 4);
  l.off_hard = o; o += s2_al(users * 4);
  l.off_ovfc = o; o += s2_al(users * 4);
  l.off_pairof = o; o += s2_al(users * S2_JMAX * 4);
  l.off_offs = o; o += s2_al(l.n_blocks * (l.G + 1) * 4);
  l.off_pairs = o; o += s2_al(l.n_blocks * S2_PAIRCAP * 4);
  l.off_cand = o; o += s2_al(l.n_blocks * S2_PAIRCAP * 64);
  l.off_ovf = o; o += s2_al(users * S2_OVF_CAP * 8);
  l.total = o + 256;
  return l;
}

bool s2_supported(int D, long Bu, int I, int k) {
  if (!(D == 64 || D == 128 || D == 256) || k < 1 || k > 32 || Bu < 1) return false;
  if (I < S2_MIN_ITEMS || sbr_cdiv(I, S2_SUPER) * 8 > S2_MAX_GROUPS) return false;
  return true;
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
  uint2* M = (uint2*)(ws + l.off_M);
  float* Lb = (float*)(ws + l.off_L);
  float* Lb2 = (float*)(ws + l.off_L2);
  int* nsel = (int*)(ws + l.off_nsel);
  int* hard = (int*)(ws + l.off_hard);
  int* ovfc = (int*)(ws + l.off_ovfc);
  int* pair_of = (int*)(ws + l.off_pairof);
  int* offs = (int*)(ws + l.off_offs);
  int* pairs = (int*)(ws + l.off_pairs);
  unsigned long long* cand = (unsigned long long*)(ws + l.off_cand);
  unsigned long long* ovf = (unsigned long long*)(ws + l.off_ovf);
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
    const size_t lds = (size_t)(l.G + 1) * 4;
    if (hipFuncSetAttribute((const void*)score_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      sbr_set_error("sbr_score_topk_f16 (two-pass): cannot raise the dynamic LDS limit of the selection kernel");
      return SBR_ERR_HIP;
    }
    const long n_full_units = plan.n_part > 0 ? (long)plan.n_wg * plan.W : (1L << 40);
    score_select_kernel<<<(unsigned int)l.n_blocks, 1024, lds, s>>>(M, Lb, l.n_units, n_full_units, (int)l.n_st, Bu, k, nsel, hard, ovfc, pair_of, offs,
                                                                      pairs, Lb2);
    SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, selection)");
  }
  // ---- pass 2
  {
    const int bps = D >= 256 ? 8 : 16;                       // selection blocks per superblock: 4,096 / 8,192 users = 2 MB of rows
    const long n_sb = sbr_cdiv(l.n_blocks, bps);
    const size_t lds = (size_t)64 * D * 2 + 64 * 4 + (size_t)S2_CHUNK * 12;
    SBR_REQUIRE(n_sb * l.G < (1L << 31), "sbr_score_topk_f16 (two-pass): grid too large");
    score_rescore_kernel<KS><<<(unsigned int)(n_sb * l.G), 256, lds, s>>>((const _Float16*)U, (const _Float16*)It, I, item_offset, Lb2, offs, pairs,
                                                                            (int)l.n_blocks, bps, (int)l.G, cand, ovfc, ovf, hard);
    SBR_CHECK_LAUNCH("sbr_score_topk_f16 (two-pass, pass 2)");
  }
  // ---- final selection
  score_finalize2_kernel<KS><<<(unsigned int)sbr_cdiv(Bu, 4), 256, 0, s>>>((const _Float16*)U, (const _Float16*)It, Bu, I, item_offset, k, u_idx,
                                                                             with_excl ? eptr : nullptr, eidx, nsel, hard, ovfc, pair_of, cand, ovf,
                                                                             out_val, out_idx);
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
