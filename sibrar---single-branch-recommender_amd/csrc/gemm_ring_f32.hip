// Persistent fp32 GEMM with an LDS-DMA slab ring (v_mfma_f32_32x32x2_f32, exact fp32 — same arithmetic as gemm_f32.hip).
//
// Why a second kernel: the SingleBranchNet products are tall and thin (90k x 128 x 128, 45k x 128 x 768, dW = 128 x 128 over
// 90k rows): a 64 x 128 output tile has only 4..24 K slabs of 0.85 us MFMA work each, less than one HBM round trip, so a tile
// kernel that prefetches one slab ahead through registers runs at memory LATENCY, and its load / MFMA / store phases add up
// instead of overlapping. Here
//   * a workgroup is persistent: it walks work items (output tile x K range) w = blockIdx.x, blockIdx.x + gridDim.x, ... and
//     its slab ring never drains between items: the slabs of the next tile are already landing while the current tile's
//     last MFMAs run and its stores drain;
//   * slabs go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no LDS store phase, NS - 1 slabs
//     in flight per workgroup, ONE barrier per slab;
//   * LDS images are bank-conflict free through an XOR swizzle applied on the per-lane SOURCE address (the DMA destination
//     is always lane-linear).
// Operands are addressed in 16-byte chunks; a chunk outside the matrix / K range is fetched from a zero chunk instead, so
// there is no predicated load anywhere. Eligibility (checked on the host): 16-byte aligned bases and row strides, K % 4 == 0
// for k-contiguous operands, row counts % 4 == 0 for k-major operands, gathered k-major operands with <= 512 k rows per item.
//
// LDS slab images (RK = 32 k per slab):
//   k-contiguous operand (NT A/B, NN A):  [rows][8 chunks]   position (r, p) holds chunk p ^ ((r >> 1) & 7) of row r
//       fragment read: ds_read_b128 of chunk 2*kq + half -> k = 8*kq + 4*half + {0,1,2,3} feeds four MFMAs
//   k-major operand (NN B, TN A/B):       [32 k][ROWS/4 chunks]  position (k, p) holds chunk p ^ (((k >> 2) & 1) << 3)
//       fragment read: four ds_read_b32 (k = 8*kq + 4*half + s); the two lane halves hit different 128-byte bank halves
#include "gemm_args.h"
#include <stdlib.h>

#define RK 32
#define RING_IDX_CAP 512
#define RING_BIAS_CAP 1024

__device__ __attribute__((aligned(16))) float sbr_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) float lds_f32;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) int lds_i32;

template <int N>
__device__ __forceinline__ void ring_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void ring_dma16(const float* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Per-thread issue state of one operand for the tile that is currently being ISSUED (may be one tile ahead of compute).
template <int ROWS, bool KM>
struct RingSrc {
  static constexpr int PER_T = ROWS * RK * 4 / 4096;        // 16-byte chunks per thread per slab
  const float* base[PER_T];                                  // k-contiguous: &M[row][swizzled chunk col] (null: outside)
  int col;                                                   // k-major: first column of this thread's chunk (< 0: outside)
};

// (Tried: interleaved column tiles — tile j = columns 2*lane + j, one 8-byte store per row instead of two 4-byte stores 128
// bytes apart. The B-fragment rows then sit 256 bytes apart in LDS, a two-way bank conflict that the 8-chunk swizzle cannot
// remove: 90112x128x128 39.2 vs 36.4 us, 8192x50000x128 1108 vs 1022 us. Dropped.)
// Stores the MI x 2 accumulator tiles of one wave. base: tile origin (uniform), ld: row stride, (row_l, col_l): this lane's
// first row / column inside the tile, full: the tile lies inside the matrix (uniform) — otherwise rows >= m_left / columns >=
// n_left are skipped; ci: optional row scatter (row r of the tile goes to row ci[r] of C, base then has no row offset).
// KIND 0: raw, 1: + bias, 2: relu(+ bias), 3: sbr_act(+ bias, act).
template <int MI, int KIND>
__device__ __forceinline__ void ring_store(f32x16 (&acc)[MI][2], float* __restrict__ base, long ld, int row_l, int col_l,
                                           bool full, int m_left, int n_left, const int* __restrict__ ci,
                                           const float* bj, int act) {
  auto fin = [&](float v, int j) {
    if constexpr (KIND >= 1) v += bj[j];
    if constexpr (KIND == 2) v = sbr_relu(v);
    if constexpr (KIND == 3) v = sbr_act(v, act);
    return v;
  };
  if (full && ci == nullptr) {
    float* lane_p = base + (long)row_l * ld + col_l;            // per-lane pointer; the rest of the address is uniform
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long ro = (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ld;
#pragma unroll
        for (int j = 0; j < 2; ++j) lane_p[ro + j * 32] = fin(acc[i][j][r], j);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = row_l + i * 32 + (r & 3) + 8 * (r >> 2);
      if (lr < m_left) {
        float* rp = base + (ci ? (long)ci[lr] : (long)lr) * ld + col_l;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (col_l + j * 32 < n_left) rp[j * 32] = fin(acc[i][j][r], j);
      }
    }
}

template <int MI, bool A_KM, bool B_KN, int NS>
__global__ __launch_bounds__(256, (MI == 1 ? (NS == 2 ? 3 : 2) : (NS == 2 ? 2 : 1))) void gemm_ring_kernel(GemmArgs g, int n_items) {
  constexpr int BM = 64 * MI, BN = 128;
  constexpr int A_BYTES = BM * RK * 4, B_BYTES = BN * RK * 4, STAGE = A_BYTES + B_BYTES;
  using SA = RingSrc<BM, A_KM>;
  using SB = RingSrc<BN, B_KN>;
  constexpr int PER_T = SA::PER_T + SB::PER_T;
  static_assert((NS - 2) * PER_T <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // behind the ring: TN: staged k-row indices of gathered operands; NT / NN: the bias vector (N <= RING_BIAS_CAP)
  lds_i32* idx_a = (lds_i32*)(smem + NS * STAGE);
  lds_i32* idx_b = idx_a + RING_IDX_CAP;
  lds_f32* bias_lds = (lds_f32*)(smem + NS * STAGE);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int spt = g.k_chunk / RK;                            // slabs per item (k_chunk is a multiple of RK)
  const int n_mine = blockIdx.x < n_items ? (n_items - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  const int q_total = n_mine * spt;
  if (q_total == 0) return;
  const float* zero = sbr_zero_chunk;
  if constexpr (!A_KM) {
    if (g.bias) {                                             // ordinary loads only before the DMA pipeline starts
      for (int j = t; j < g.N; j += 256) bias_lds[j] = g.bias[j];
      __syncthreads();
    }
  }

  // ---- item decoding: w -> (K range z, row panel, column panel); column panels of one row panel are consecutive -------------
  // Workgroup b always runs on XCD b % 8 and gridDim.x is a multiple of 8 (or < 8 ... any map is correct), so the items
  // w = b, b + grid, ... of one workgroup all satisfy w % 8 == b % 8. XCD x gets the CONTIGUOUS item range
  // [x*a + min(x, rem), ...): neighbouring column panels of a row panel and the shared B panel stay in one L2.
  const int per_xcd = n_items >> 3, rem_xcd = n_items & 7;
  auto decode = [&](int w_raw, int& z, int& m0, int& n0) {
    const int x = w_raw & 7;
    const int w = x * per_xcd + min(x, rem_xcd) + (w_raw >> 3);
    if (g.nt == 1 && g.splits == 1) {                         // the thin products of the training step: no divisions
      z = 0; m0 = w * BM; n0 = 0;
      return;
    }
    const int tiles = g.mt * g.nt;
    z = w / tiles;
    const int r = w - z * tiles;
    const int mi = r / g.nt;
    m0 = mi * BM;
    n0 = (r - mi * g.nt) * BN;
  };

  // ---- issue side -----------------------------------------------------------------------------------------------------------
  SA sa;
  SB sb;
  int is_item = blockIdx.x, is_slab = 0, is_kstart = 0, is_kend = 0;
  auto setup_issue = [&](int w) {
    int z, m0, n0;
    decode(w, z, m0, n0);
    is_kstart = z * g.k_chunk;
    is_kend = min(g.K, is_kstart + g.k_chunk);
    if constexpr (A_KM) {
      if (g.a_idx || g.b_idx) __syncthreads();               // every wave has issued the previous item's last slab
    }
    if constexpr (!A_KM) {
#pragma unroll
      for (int i = 0; i < SA::PER_T; ++i) {
        const int e = i * 256 + t, r = e >> 3, p = e & 7;
        const int gm = m0 + r;
        const long row = gm < g.M ? (g.a_idx ? (long)g.a_idx[gm] : (long)gm) : -1;
        sa.base[i] = row >= 0 ? g.A + row * g.lda + ((p ^ ((r >> 1) & 7)) << 2) : nullptr;
      }
    } else {
      constexpr int CPR = BM / 4;
      const int p = t % CPR;                                  // 256 % CPR == 0: the chunk column is the same for every i
      // k = e / CPR = i * (256 / CPR) + t / CPR; the swizzle bit (k >> 2) & 1 depends on i only through (256/CPR)*i, which
      // is a multiple of 8 for CPR <= 32: it is a per-thread constant
      const int kk = t / CPR;
      const int c = p ^ (((kk >> 2) & 1) << 3);
      sa.col = (m0 + c * 4 < g.M) ? m0 + c * 4 : -1;
      if (g.a_idx) {
        for (int j = t; j < is_kend - is_kstart; j += 256) idx_a[j] = g.a_idx[is_kstart + j];
      }
    }
    if constexpr (!B_KN) {
#pragma unroll
      for (int i = 0; i < SB::PER_T; ++i) {
        const int e = i * 256 + t, r = e >> 3, p = e & 7;
        const int gn = n0 + r;
        const long row = gn < g.N ? (g.b_idx ? (long)g.b_idx[gn] : (long)gn) : -1;
        sb.base[i] = row >= 0 ? g.B + row * g.ldb + ((p ^ ((r >> 1) & 7)) << 2) : nullptr;
      }
    } else {
      constexpr int CPR = BN / 4;
      const int p = t % CPR, kk = t / CPR;
      const int c = p ^ (((kk >> 2) & 1) << 3);
      sb.col = (n0 + c * 4 < g.N) ? n0 + c * 4 : -1;
      if (g.b_idx) {
        for (int j = t; j < is_kend - is_kstart; j += 256) idx_b[j] = g.b_idx[is_kstart + j];
      }
    }
    if constexpr (A_KM) {
      if (g.a_idx || g.b_idx) __syncthreads();               // staged indices visible (uniform branch)
    }
  };

  auto issue = [&](int slot) {
    unsigned char* st = smem + slot * STAGE + wave * 1024;
    const int kbase = is_kstart + is_slab * RK;
    if constexpr (!A_KM) {
#pragma unroll
      for (int i = 0; i < SA::PER_T; ++i) {
        const int p = t & 7, r = (i * 256 + t) >> 3;
        const int k0 = kbase + ((p ^ ((r >> 1) & 7)) << 2);
        const float* src = (sa.base[i] != nullptr && k0 < is_kend) ? sa.base[i] + kbase : zero;
        ring_dma16(src, st + i * 4096);
      }
    } else {
      constexpr int CPR = BM / 4, KPI = 256 / CPR;
#pragma unroll
      for (int i = 0; i < SA::PER_T; ++i) {
        const int kl = is_slab * RK + i * KPI + t / CPR;       // k row relative to the item's K range
        const int gk = is_kstart + kl;
        const float* src = zero;
        if (sa.col >= 0 && gk < is_kend) {
          const long kr = g.a_idx ? (long)idx_a[kl] : (long)gk;
          src = g.A + kr * g.lda + sa.col;
        }
        ring_dma16(src, st + i * 4096);
      }
    }
    unsigned char* sbp = st + A_BYTES;
    if constexpr (!B_KN) {
#pragma unroll
      for (int i = 0; i < SB::PER_T; ++i) {
        const int p = t & 7, r = (i * 256 + t) >> 3;
        const int k0 = kbase + ((p ^ ((r >> 1) & 7)) << 2);
        const float* src = (sb.base[i] != nullptr && k0 < is_kend) ? sb.base[i] + kbase : zero;
        ring_dma16(src, sbp + i * 4096);
      }
    } else {
      constexpr int CPR = BN / 4, KPI = 256 / CPR;
#pragma unroll
      for (int i = 0; i < SB::PER_T; ++i) {
        const int kl = is_slab * RK + i * KPI + t / CPR;
        const int gk = is_kstart + kl;
        const float* src = zero;
        if (sb.col >= 0 && gk < is_kend) {
          const long kr = g.b_idx ? (long)idx_b[kl] : (long)gk;
          src = g.B + kr * g.ldb + sb.col;
        }
        ring_dma16(src, sbp + i * 4096);
      }
    }
    if (++is_slab == spt) {
      is_slab = 0;
      is_item += gridDim.x;
      if (is_item < n_items) setup_issue(is_item);
    }
  };

  // ---- compute side -----------------------------------------------------------------------------------------------------------
  // fragment addresses inside a slab image (bytes), per lane
  int a_rd[MI], b_rd[2];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = (wm * MI + i) * 32 + l31;
    if constexpr (!A_KM) a_rd[i] = r * 128;                    // + ((2*kq + half) ^ ((r >> 1) & 7)) * 16
    else a_rd[i] = (4 * half) * (BM * 4) + ((((r >> 2) ^ (half << 3))) << 4) + (r & 3) * 4;   // + kq * 8 * BM*4 + s * BM*4
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wn * 2 + j) * 32 + l31;
    if constexpr (!B_KN) b_rd[j] = r * 128;
    else b_rd[j] = (4 * half) * (BN * 4) + ((((r >> 2) ^ (half << 3))) << 4) + (r & 3) * 4;
  }
  const int a_sw = (((wm * MI) * 32 + l31) >> 1) & 7;           // (r >> 1) & 7 is the same for every i (r differs by 32)
  const int b_sw = (((wn * 2) * 32 + l31) >> 1) & 7;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  setup_issue(is_item);
  int issued = 0;
#pragma unroll 1
  for (; issued < NS - 1 && issued < q_total; ++issued) issue(issued);

  int cp_item = blockIdx.x, cp_slab = 0;
#pragma unroll 1
  for (int q = 0; q < q_total; ++q) {
    // slab q has landed once at most the NS - 2 younger slabs are outstanding (loads return in order); near the end fewer
    // slabs follow, so wait for everything
    if (q + NS - 1 <= q_total) ring_wait_vmcnt<(NS - 2) * PER_T>();
    else ring_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                              // every wave's part of slab q landed; slab q - 1 consumed
    asm volatile("" ::: "memory");
    if (issued < q_total) {
      issue(issued % NS);                                      // == (q - 1) % NS, the slot freed by the barrier
      ++issued;
    }
    const lds_f32* As = (const lds_f32*)(smem + (q % NS) * STAGE);
    const lds_f32* Bs = (const lds_f32*)(smem + (q % NS) * STAGE + A_BYTES);
#pragma unroll
    for (int kq = 0; kq < RK / 8; ++kq) {
      float4 fa[MI], fb[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if constexpr (!A_KM) {
          const v4f v = *(const lds_v4f*)(As + ((a_rd[i] + (((2 * kq + half) ^ a_sw) << 4)) >> 2));
          fa[i] = make_float4(v.x, v.y, v.z, v.w);
        } else {
          const lds_f32* p = As + ((a_rd[i] + kq * 8 * BM * 4) >> 2);
          fa[i] = make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (!B_KN) {
          const v4f v = *(const lds_v4f*)(Bs + ((b_rd[j] + (((2 * kq + half) ^ b_sw) << 4)) >> 2));
          fb[j] = make_float4(v.x, v.y, v.z, v.w);
        } else {
          const lds_f32* p = Bs + ((b_rd[j] + kq * 8 * BN * 4) >> 2);
          fb[j] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (++cp_slab == spt) {
      // ---- epilogue of item cp_item: accumulator register r of a 32x32 tile is row (r&3) + 8*(r>>2) + 4*half, column l31.
      // All mode / bounds decisions are uniform and taken OUTSIDE the element loops; interior tiles without a row scatter
      // store through one uniform base + a per-lane offset.
      int z, m0, n0;
      decode(cp_item, z, m0, n0);
      const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
      const int row_l = wm * MI * 32 + 4 * half;               // + i*32 + (r&3) + 8*(r>>2)
      const int col_l = wn * 64 + l31;                         // + j*32
      if (g.slab) {
        float* base = g.slab + ((long)z * g.M + m0) * g.N + n0;
        ring_store<MI, 0>(acc, base, g.N, row_l, col_l, full, g.M - m0, g.N - n0, nullptr, nullptr, 0);
      } else if (g.atomic) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int gm = m0 + row_l + i * 32 + (r & 3) + 8 * (r >> 2);
            if (gm < g.M) {
              const long crow = (g.c_idx ? (long)g.c_idx[gm] : (long)gm) * g.ldc;
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int gn = n0 + col_l + j * 32;
                if (gn < g.N) atomicAdd(&g.C[crow + gn], acc[i][j][r] + ((g.bias && z == 0) ? g.bias[gn] : 0.f));
              }
            }
          }
      } else {
        float bj[2] = {0.f, 0.f};
        if (g.bias) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int gn = n0 + col_l + j * 32;
            bj[j] = gn < g.N ? bias_lds[gn] : 0.f;
          }
        }
        float* base = g.C + (g.c_idx ? 0 : (long)m0 * g.ldc) + n0;
        const int* ci = g.c_idx ? g.c_idx + m0 : nullptr;
        if (g.act == SBR_ACT_NONE) ring_store<MI, 1>(acc, base, g.ldc, row_l, col_l, full, g.M - m0, g.N - n0, ci, bj, 0);
        else if (g.act == SBR_ACT_RELU) ring_store<MI, 2>(acc, base, g.ldc, row_l, col_l, full, g.M - m0, g.N - n0, ci, bj, 0);
        else ring_store<MI, 3>(acc, base, g.ldc, row_l, col_l, full, g.M - m0, g.N - n0, ci, bj, g.act);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      cp_slab = 0;
      cp_item += gridDim.x;
    }
  }
}

template <int MI, bool A_KM, bool B_KN, int NS>
static int ring_launch(GemmArgs& g, hipStream_t s) {
  constexpr int BM = 64 * MI, BN = 128;
  g.mt = sbr_cdiv(g.M, BM);
  g.nt = sbr_cdiv(g.N, BN);
  const int n_items = g.mt * g.nt * g.splits;
  const size_t lds = (size_t)NS * (BM + BN) * RK * 4 + (A_KM ? 2 * RING_IDX_CAP * 4 : RING_BIAS_CAP * 4);
  auto kern = gemm_ring_kernel<MI, A_KM, B_KN, NS>;
  static int attr_dev = -1;
  if (sbr_attr_stale(&attr_dev)) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      sbr_set_error("sbr_gemm_f32(ring): cannot raise the dynamic LDS limit to %zu", lds);
      return SBR_ERR_HIP;
    }
  }
  const int per_cu = (int)(160 * 1024 / lds);
  int grid = 256 * per_cu;
  if (grid > n_items) grid = n_items;
  kern<<<grid, 256, lds, s>>>(g, n_items);
  SBR_CHECK_LAUNCH("sbr_gemm_f32(ring)");
  return SBR_OK;
}

static inline bool ring_al16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

int sbr_gemm_ring_launch(int mode, GemmArgs& g, hipStream_t s) {
  if (!ring_al16(g.A, g.lda) || !ring_al16(g.B, g.ldb)) return -1;
  if (g.k_chunk % RK != 0 || g.k_chunk <= 0) return -1;
  const bool a_km = mode == 2, b_kn = mode != 0;
  if (!a_km && (g.K & 3)) return -1;                           // k-contiguous rows: whole chunks only
  if (!b_kn && (g.K & 3)) return -1;
  if (a_km && (g.M & 3)) return -1;                            // k-major rows: whole chunks along m / n
  if (b_kn && (g.N & 3)) return -1;
  if (a_km && g.a_idx && g.k_chunk > RING_IDX_CAP) return -1;
  if (b_kn && g.b_idx && g.k_chunk > RING_IDX_CAP) return -1;
  if (mode == 1 && g.b_idx) return -1;
  if (mode != 2 && g.bias && g.N > RING_BIAS_CAP) return -1;
  if (mode == 2 && g.bias) return -1;
  // 64 x 128 tiles, 2-slot ring, three workgroups per CU. Measured alternatives (MI355X, us for 90112x128x128 NT /
  // 45056x128x768 gathered NT / 8192x50000x128 NT): 3 slots + 2 WG/CU 39.0 / 99.5 / 1146; 128 x 128 tiles, 4 slots, 1 WG/CU
  // 43.1 / 130.4 / 1175; this configuration 38.3 / 93.3 / 1091 — more waves per SIMD beat a deeper ring.
  // 128 x 128 tiles (2 WG/CU) once there are enough of them to fill the 512 workgroup slots more than once: 90112x128x128
  // 36.4 vs 38.6 us, 8192x50000x128 1022 vs 1118 us (103 TFLOP/s), 4096^3 124 vs 117 TFLOP/s; the 45056x128x768 projection has
  // only 352 such tiles and stays on 64 x 128 (94 vs 112 us). TN stays on 64 x 128 (the split count provides the items).
  const long big_items = (long)sbr_cdiv(g.M, 128) * sbr_cdiv(g.N, 128);
  int big = mode != 2 && big_items >= 640;
  if (big) {
    if (mode == 0) return ring_launch<2, false, false, 2>(g, s);
    if (mode == 1) return ring_launch<2, false, true, 2>(g, s);
    return ring_launch<2, true, true, 2>(g, s);
  }
  if (mode == 0) return ring_launch<1, false, false, 2>(g, s);
  if (mode == 1) return ring_launch<1, false, true, 2>(g, s);
  return ring_launch<1, true, true, 2>(g, s);
}
