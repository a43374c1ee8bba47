// Shared between the two fused scorers (score_topk_f16_n.hip: the one-pass narrow-wave kernel; score_topk_f16_2p.hip: the two-pass
// scorer): ring / wave geometry constants, small device helpers, the lane-parallel k-th-of-32 selection, the exclusion event stream
// builder and the unit plan. Everything here has internal linkage (each translation unit compiles its own copy).
#pragma once
#include "score_topk_common.h"

#ifndef S5_NL
#define S5_NL 1                          // loader waves per workgroup (a tile period is ~2 us with 12-13 consumer waves, one wave's LDS-DMA
                                         // stream fills a 16 KB tile in 0.65 us; two loader waves measured 1 % slower with S5_PRIO 3)
#endif
#define S5_MAXW (16 - S5_NL)             // consumer waves per workgroup (+ the loader waves = 1024 threads)
#ifndef S5_PF1
#define S5_PF1 3                         // prefetch distance (K steps of 16) with one accumulator tile per LDS tile (D = 256)
#endif
#ifndef S5_PF2
#define S5_PF2 1                         // ... with two (D = 64, 128)
#endif
#ifndef S5_PRIO
#define S5_PRIO 3                        // s_setprio level of a consumer wave while it issues a tile's MFMAs (0: none; 3: -1 % in the
                                         // in-process A/B at D = 128 and 256 once no SIMD carries a fourth consumer wave)
#endif
#ifndef S5_NS
#define S5_NS 6                          // LDS ring slots of 16 KB (D = 128: 64-item tiles, D = 256: 32-item tiles)
#endif
#ifndef S5_EVABL
#define S5_EVABL 0                        // lab (timing only, wrong results): 1 = the event window is never refilled
#endif
#define S5_EV_NONE 0xFFFFFFFFu            // padding event: its tile field matches no tile
// all LDS reads of the tile have returned (the accumulators are named so that the wait stays behind the MFMAs that consume the
// fragments); device-only helpers: the host pass of hipcc rejects 64-byte "v" operands and then silently drops the kernel's stub
__device__ __forceinline__ void s5_lds_done(const f32x16& a, const f32x16& b) { asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(b) : "memory"); }
__device__ __forceinline__ void s5_pin(const f32x16& a, const f32x16& b) { asm volatile("" ::"v"(a), "v"(b)); }
// no-return LDS atomics as bare instructions: hipcc puts s_waitcnt vmcnt(0) in front of every LDS atomic of a wave that also
// issues LDS-DMA (it cannot tell the DMA destination from the atomic's word), i.e. a wait for all candidate stores in flight,
// once per tile. LDS operations of a wave execute in order; callers place the waits they need themselves.
__device__ __forceinline__ void s5_lds_add(lds_int* p, int v) { asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned int)(size_t)p), "v"(v) : "memory"); }
// the same from lane 0 only, all lanes active on entry and exit: an EXEC flip around the instruction instead of a divergent block
__device__ __forceinline__ void s5_lds_add_lane0(lds_int* p, int v) {
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"((unsigned int)(size_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void s5_lds_or(lds_int* p, unsigned int v) { asm volatile("ds_or_b32 %0, %1" ::"v"((unsigned int)(size_t)p), "v"(v) : "memory"); }
// maximum of four accumulator registers as v_max3 + v_max (fmaxf makes hipcc canonicalise every operand first: a v_max x, x each)
__device__ __forceinline__ float s5_max4(float a, float b, float c, float d) {
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  asm("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(d));
  return m;
}
__device__ __forceinline__ float s5_max2(float a, float b) {
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float s5_min2(float a, float b) {
  float m;
  asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
// Batcher's odd-even merge sort of 16 values (63 compare-exchanges), descending, on registers: every index is a compile-time
// constant after unrolling. All lanes sort their own 16 values at once.
struct S5Ce { unsigned char i, j; };
__device__ static constexpr S5Ce S5_SORT16[63] = {
    {0, 1}, {2, 3}, {4, 5}, {6, 7}, {8, 9}, {10, 11}, {12, 13}, {14, 15},
    {0, 2}, {1, 3}, {4, 6}, {5, 7}, {8, 10}, {9, 11}, {12, 14}, {13, 15},
    {1, 2}, {5, 6}, {9, 10}, {13, 14},
    {0, 4}, {1, 5}, {2, 6}, {3, 7}, {8, 12}, {9, 13}, {10, 14}, {11, 15},
    {2, 4}, {3, 5}, {10, 12}, {11, 13},
    {1, 2}, {3, 4}, {5, 6}, {9, 10}, {11, 12}, {13, 14},
    {0, 8}, {1, 9}, {2, 10}, {3, 11}, {4, 12}, {5, 13}, {6, 14}, {7, 15},
    {4, 8}, {5, 9}, {6, 10}, {7, 11},
    {2, 4}, {3, 5}, {6, 8}, {7, 9}, {10, 12}, {11, 13},
    {1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 10}, {11, 12}, {13, 14}};
// k-th largest (1 <= k <= 32, wave-uniform) of the 32 values a user holds in its two lanes l and l ^ 32 (16 each): every lane pair
// for its own user, all 32 users of the wave at once. -inf entries are ordinary values (fewer than k real ones -> -inf).
__device__ __forceinline__ float s5_kth_of_32(const float (&cm)[16], int k) {
  float a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = cm[r];
#pragma unroll
  for (int c = 0; c < 63; ++c) {
    const int i = S5_SORT16[c].i, j = S5_SORT16[c].j;
    const float hi = s5_max2(a[i], a[j]), lo = s5_min2(a[i], a[j]);
    a[i] = hi; a[j] = lo;                                    // descending: a[0] the largest
  }
  // the partner's sorted list, reversed: max(a[i], b[15 - i]) are the 16 largest of the 32 (a bitonic sequence), min(...) the rest
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float b = __shfl_xor(a[15 - i], 32, 64);
    v[i] = k <= 16 ? s5_max2(a[i], b) : s5_min2(a[i], b);
  }
  // bitonic merge, descending
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if ((i & d) == 0) {
        const float hi = s5_max2(v[i], v[i + d]), lo = s5_min2(v[i], v[i + d]);
        v[i] = hi; v[i + d] = lo;
      }
    }
  }
  const int e = (k - 1) & 15;
  float t = v[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) t = e == i ? v[i] : t;
  return t;
}
__device__ __forceinline__ void s5_pin8(const f16x8& a) { asm volatile("" ::"v"(a)); }
// ---------------------------------------------------------------------------------------------------------------------------
// Exclusion events (see the consumer prologue of the kernel): built per call from the exclusion CSR by three small kernels.
//   rows:    one thread per scored user row: the part of its sorted CSR row that falls into [item_offset, item_offset + I)
//            (two binary searches), counted into its 32-user group
//   scan:    one workgroup: group g gets room for its events rounded up to a quad + two quads of padding (the consumer reads
//            one quad ahead), exclusive prefix sum -> group_base
//   scatter: one workgroup per group: counting sort of the group's events by item tile in LDS (histogram, prefix sum,
//            scatter); events of one tile stay in arbitrary order (their bits are OR-ed); padding words are S5_EV_NONE
// event = tile << 11 | (u + 32 ((col >> 2) & 1)) << 5 | ((col >> 5) * 16 + (col & 3) + 4 ((col & 31) >> 3)):
// col = item column inside the tile, u = user inside the group — the lane and the accumulator bit of that score.
static __global__ void s5_ev_rows_kernel(long Bu, const long* __restrict__ u_idx, const long* __restrict__ indptr, const int* __restrict__ indices,
                                  int item_offset, int I, long* __restrict__ row_lo, int* __restrict__ row_cnt, int* __restrict__ grp_cnt) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Bu) return;
  const long u = u_idx ? u_idx[r] : r;
  const long b = indptr[u], e = indptr[u + 1];
  const long lim = (long)item_offset + I;
  // a row that lies inside the scored item range as a whole (the usual case: one shard = the whole catalogue) needs no search
  long lo = b, hi = e;
  if (b < e && indices[b] < item_offset) {
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (indices[mid] < item_offset) lo = mid + 1; else hi = mid; }
  }
  const long first = lo;
  if (b < e && indices[e - 1] >= lim) {
    hi = e;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (indices[mid] < lim) lo = mid + 1; else hi = mid; }
  } else {
    lo = e;
  }
  row_lo[r] = first;
  row_cnt[r] = (int)(lo - first);
  if (lo > first) atomicAdd(grp_cnt + (r >> 5), (int)(lo - first));
}

static __global__ __launch_bounds__(1024) void s5_ev_scan_kernel(int G, const int* __restrict__ grp_cnt, int* __restrict__ group_base) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (G + 1023) / 1024;
  const int g0 = t * per, g1 = g0 + per < G ? g0 + per : G;
  int sum = 0;
  for (int g = g0; g < g1; ++g) sum += ((grp_cnt[g] + 3) & ~3) + 8;
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;                                   // exclusive prefix of this thread's chunk
  for (int g = g0; g < g1; ++g) { group_base[g] = run; run += ((grp_cnt[g] + 3) & ~3) + 8; }
  if (t == 1023) group_base[G] = part[1023];
}

static __global__ __launch_bounds__(256) void s5_ev_scatter_kernel(long Bu, const int* __restrict__ indices, int item_offset, int tile_items, int n_tiles,
                                                            const long* __restrict__ row_lo, const int* __restrict__ row_cnt,
                                                            const int* __restrict__ grp_cnt, const int* __restrict__ group_base,
                                                            unsigned int* __restrict__ events, long cap) {
  extern __shared__ int hist[];                              // [n_tiles] counts, then running positions
  __shared__ int part[256];
  const int g = blockIdx.x, t = threadIdx.x;
  const int cnt = grp_cnt[g];
  const long base = group_base[g];
  const int alloc = ((cnt + 3) & ~3) + 8;
  if (base + alloc > cap) return;                            // cannot happen with a workspace of the documented size
  if (cnt > 0) {
    // the group's entries as ONE flat index range: rstart[u] = entries of the users before u (rows are ~50 entries long, the
    // group ~1,600: a loop per user would leave most of the 256 threads idle)
    __shared__ int rstart[33];
    __shared__ long rlo[32];
    if (t < 32) {
      const long r = (long)g * 32 + t;
      const int n = r < Bu ? row_cnt[r] : 0;
      rlo[t] = r < Bu ? row_lo[r] : 0;
      int incl = n;                                            // inclusive scan over the 32 lanes of this half wave
      for (int d = 1; d < 32; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (t >= d) incl += v; }
      rstart[t + 1] = incl;
      if (t == 0) rstart[0] = 0;
    }
    for (int i = t; i < n_tiles; i += 256) hist[i] = 0;
    __syncthreads();
    auto entry = [&](int f, int& u) -> int {                   // flat index -> (user u, item index relative to the shard)
      int lo_u = 0, hi_u = 31;
      while (lo_u < hi_u) { const int mid = (lo_u + hi_u + 1) >> 1; if (rstart[mid] <= f) lo_u = mid; else hi_u = mid - 1; }
      u = lo_u;
      return indices[rlo[u] + (f - rstart[u])] - item_offset;
    };
    for (int f = t; f < cnt; f += 256) { int u; atomicAdd(hist + entry(f, u) / tile_items, 1); }
    __syncthreads();
    // exclusive prefix sum over the tiles: contiguous chunk per thread
    const int per = (n_tiles + 255) / 256;
    const int i0 = t * per, i1 = i0 + per < n_tiles ? i0 + per : n_tiles;
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += hist[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int v = t >= d ? part[t - d] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = i0; i < i1; ++i) { const int c = hist[i]; hist[i] = run; run += c; }
    __syncthreads();
    for (int f = t; f < cnt; f += 256) {
      int u;
      const int rel = entry(f, u);
      const int tile = rel / tile_items, col = rel - tile * tile_items;
      const unsigned int tgt = (unsigned int)(u + 32 * ((col >> 2) & 1));
      const unsigned int bit = (unsigned int)((col >> 5) * 16 + (col & 3) + 4 * ((col & 31) >> 3));
      events[base + atomicAdd(hist + tile, 1)] = ((unsigned int)tile << 11) | (tgt << 5) | bit;
    }
  }
  for (int i = cnt + t; i < alloc; i += 256) events[base + i] = S5_EV_NONE;
}

static int s5_n_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

// How the 32-user units are dealt to workgroups (one resident per CU): W full consumer waves per workgroup on n_wg workgroups, plus —
// when the units do not divide evenly over the CUs — one PARTIAL wave on each of the first n_part workgroups: the remainder units are
// cut into P parts by item tile (see the kernel). 100k users = 3,125 units on 256 CUs: 12 full waves everywhere + 53 remainder units
// in 4 parts each on 212 workgroups (3.25 consumer waves on the fullest SIMD instead of 4).
struct S5Plan { int W, n_wg, n_part, P; };
static S5Plan s5_plan(long Bu) {
  const int G = s5_n_cu();
  const long units = sbr_cdiv(Bu, 32);
  S5Plan p;
  const long Wf = units / G;
  const long R = units - Wf * G;
  if (Wf >= 1 && Wf + 1 <= S5_MAXW && R > 0 && G / R >= 2) {
    p.W = (int)Wf; p.n_wg = G; p.P = (int)(G / R < 8 ? G / R : 8); p.n_part = (int)(R * p.P);
    return p;
  }
  // one round of whole units: the smallest W that keeps the number of rounds (workgroups per CU) at its minimum
  const long rounds = sbr_cdiv(units, (long)G * S5_MAXW);
  long w = sbr_cdiv(units, rounds * G);
  p.W = (int)(w < 1 ? 1 : (w > S5_MAXW ? S5_MAXW : w));
  p.n_wg = (int)sbr_cdiv(units, p.W); p.n_part = 0; p.P = 1;
  return p;
}

// Exclusion events of one (user list, exclusion CSR, item range, tile width) combination: group_base int[G + 1], grp_cnt int[G],
// row_cnt int[Bu], row_lo long[Bu], events uint[excl_nnz + 11 G] (per group: its events rounded up to a quad + two padding quads),
// 16-byte aligned pieces of ONE caller-owned buffer of sbr_score_topk_f16_events_bytes(Bu, excl_nnz) bytes. The exclusion mask of an
// evaluation split is the same for every evaluation (eval/eval.py:219: dataset.exclude_data), so a caller builds the stream once
// per (split, user chunk, item shard) and hands it to every later call (build_events = 0).
static long s5_al16(long b) { return (b + 15) & ~15L; }
static long s5_event_bytes(long Bu, long excl_nnz) {
  if (excl_nnz <= 0) return 0;
  const long G = sbr_cdiv(Bu, 32);
  return s5_al16((G + 1) * 4) + s5_al16(G * 4) + s5_al16(Bu * 4) + s5_al16(Bu * 8) + s5_al16((excl_nnz + 11 * G) * 4) + 16;
}

struct S5Events { const unsigned int* events; const int* group_base; };

// builds the event stream into `buf` (three launches on `s`); tile_items = 32 * NJ of the kernel that will read it
static int s5_build_events(void* buf, long buf_bytes, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                           int item_offset, int tile_items, bool build, S5Events* out, hipStream_t s) {
  SBR_REQUIRE(buf && buf_bytes >= s5_event_bytes(Bu, excl_nnz), "sbr_score_topk_f16: event buffer of %ld bytes needed (sbr_score_topk_f16_events_bytes), %ld given",
              s5_event_bytes(Bu, excl_nnz), buf_bytes);
  const long G = sbr_cdiv(Bu, 32);
  const int n_tiles_ev = sbr_cdiv(I, tile_items);
  SBR_REQUIRE((long)n_tiles_ev * 4 <= 150 * 1024 && n_tiles_ev < (1 << 21) - 1,
              "sbr_score_topk_f16: %d item tiles exceed the event builder's LDS histogram (score the catalogue in item shards)", n_tiles_ev);
  char* p = (char*)s5_al16((long)buf);
  int* gb = (int*)p; p += s5_al16((G + 1) * 4);
  int* gc = (int*)p; p += s5_al16(G * 4);
  int* rc = (int*)p; p += s5_al16(Bu * 4);
  long* rl = (long*)p; p += s5_al16(Bu * 8);
  unsigned int* ev = (unsigned int*)p;
  out->events = ev;
  out->group_base = gb;
  if (!build) return SBR_OK;
  const long cap = excl_nnz + 11 * G;
  if (hipMemsetAsync(gc, 0, G * 4, s) != hipSuccess) { sbr_set_error("sbr_score_topk_f16: memset failed"); return SBR_ERR_HIP; }
  s5_ev_rows_kernel<<<(unsigned int)sbr_cdiv(Bu, 256), 256, 0, s>>>(Bu, u_idx, eptr, eidx, item_offset, I, rl, rc, gc);
  s5_ev_scan_kernel<<<1, 1024, 0, s>>>((int)G, gc, gb);
  // (set on every build: the attribute belongs to the current device's copy of the kernel, a process-wide flag would skip the second GPU
  // of a multi-device process; building the stream happens once per evaluation split)
  if (hipFuncSetAttribute((const void*)s5_ev_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
    sbr_set_error("sbr_score_topk_f16: cannot raise the dynamic LDS limit of the event builder");
    return SBR_ERR_HIP;
  }
  s5_ev_scatter_kernel<<<(unsigned int)G, 256, (size_t)n_tiles_ev * 4, s>>>(Bu, eidx, item_offset, tile_items, n_tiles_ev, rl, rc, gc, gb, ev, cap);
  SBR_CHECK_LAUNCH("sbr_score_topk_f16 (exclusion events)");
  return SBR_OK;
}

