// fp32 GEMM family on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an
// fmaf chain, so results stay within fp32 rounding of the reference's CPU matmuls).
//
// One kernel template serves the three products of the SingleBranchNet hot path:
//   NT  C[m,n] = act(sum_k A[ai(m),k] * B[n,k] + bias[n])  -> C[ci(m),n]      Linear forward  (polylinear.py:51)
//                                                                             + the scorer u @ i^T (sgd_alg.py:2109)
//   NN  C[m,n] = sum_k A[ai(m),k] * B[k,n]                                    dX = dZ @ W
//   TN  C[m,n] = sum_k A[ak(k),m] * B[bk(k),n]   (split-K, atomic add)        dW = dZ^T @ X
// ai/ci/ak/bk are optional int32 row-index arrays: the per-modality row gathers of
// SingleBranchNetEntity._get_modality_embeddings (sgd_alg.py:1934-1978) and the scatter back into the
// [R, C] embedding matrix are fused into the tile loads / stores instead of materialising gathered copies.
//
// Tiling: 256 threads = 4 wavefronts of 64; each wave owns MI x NI tiles of 32x32 (16 accumulator VGPRs each).
// K is consumed in BK=32 slabs staged through LDS from registers (global loads for slab t+1 are issued before the
// MFMAs of slab t). Within each 8-wide k group lane-half h supplies k = 4h + s to MFMA step s, so one 16-byte LDS
// read feeds four MFMAs.
#include "gemm_args.h"
#include <stdlib.h>

#define BK 32

template <int ROWS, bool KM>
struct TileGeom {
  // floats per thread = ROWS*BK/256
  static constexpr int P = ROWS * BK / (4 * 256);   // float4 per thread
  static constexpr int LD = KM ? (ROWS + 4) : (BK + 4);
  static constexpr int SIZE = KM ? BK * (ROWS + 4) : ROWS * (BK + 4);
};

// ---- global -> register loads -------------------------------------------------------------------------------
template <int ROWS, int P>
__device__ __forceinline__ void load_mk(float4 (&r)[P], const float* __restrict__ src, const long (&rowoff)[P],
                                        int kbase, int kend, int vec, int t) {
  const int kc = (t & 7) * 4;
  const int gk = kbase + kc;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rowoff[p] >= 0) {
      const float* q = src + rowoff[p] + gk;
      if (vec && gk + 3 < kend) {
        v = *reinterpret_cast<const float4*>(q);
      } else {
        if (gk < kend) v.x = q[0];
        if (gk + 1 < kend) v.y = q[1];
        if (gk + 2 < kend) v.z = q[2];
        if (gk + 3 < kend) v.w = q[3];
      }
    }
    r[p] = v;
  }
}

template <int ROWS, int P>
__device__ __forceinline__ void load_km(float4 (&r)[P], const float* __restrict__ src, long ld,
                                        const int* __restrict__ kidx, int row0, int nrows, int kbase, int kend,
                                        int vec, int t) {
  constexpr int TPR = ROWS / 4;          // threads per k-row
  constexpr int RPP = 256 / TPR;         // k-rows per pass
  const int c = row0 + (t % TPR) * 4;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int gk = kbase + t / TPR + RPP * p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gk < kend && c < nrows) {
      const long kr = kidx ? (long)kidx[gk] : (long)gk;
      const float* q = src + kr * ld + c;
      if (vec && c + 3 < nrows) {
        v = *reinterpret_cast<const float4*>(q);
      } else {
        v.x = q[0];
        if (c + 1 < nrows) v.y = q[1];
        if (c + 2 < nrows) v.z = q[2];
        if (c + 3 < nrows) v.w = q[3];
      }
    }
    r[p] = v;
  }
}

// ---- register -> LDS ---------------------------------------------------------------------------------------------
template <int ROWS, bool KM, int P>
__device__ __forceinline__ void store_lds(float* __restrict__ s, const float4 (&r)[P], int t) {
  if constexpr (!KM) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int row = (t >> 3) + 32 * p;
      *reinterpret_cast<float4*>(&s[row * (BK + 4) + (t & 7) * 4]) = r[p];
    }
  } else {
    constexpr int TPR = ROWS / 4;
    constexpr int RPP = 256 / TPR;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int kr = t / TPR + RPP * p;
      *reinterpret_cast<float4*>(&s[kr * (ROWS + 4) + (t % TPR) * 4]) = r[p];
    }
  }
}

// ---- LDS -> MFMA operand fragments ------------------------------------------------------------------------------------
template <int ROWS, bool KM>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ s, int row, int kq, int half) {
  if constexpr (!KM) {
    return *reinterpret_cast<const float4*>(&s[row * (BK + 4) + kq * 8 + 4 * half]);
  } else {
    const float* q = &s[(kq * 8 + 4 * half) * (ROWS + 4) + row];
    return make_float4(q[0], q[ROWS + 4], q[2 * (ROWS + 4)], q[3 * (ROWS + 4)]);
  }
}

template <int WM, int WN, int MI, int NI, bool A_KM, bool B_KN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BM = WM * MI * 32;
  constexpr int BN = WN * NI * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  using GA = TileGeom<BM, A_KM>;
  using GB = TileGeom<BN, B_KN>;
  __shared__ __attribute__((aligned(16))) float smem[GA::SIZE + GB::SIZE];
  float* As = smem;
  float* Bs = smem + GA::SIZE;

  // XCD-aware tile order: the nt column tiles of one row panel run on the same XCD (block ids b and b+8 share an
  // XCD) so that the gathered A rows are fetched into one L2 only. Speed only; any mapping is correct.
  int mt_i, nt_i;
  if (g.xcd_map) {
    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int cnt = (g.mt - x + 7) >> 3;           // panels owned by this XCD group
    if (j >= cnt * g.nt) return;
    mt_i = x + 8 * (j / g.nt);
    nt_i = j % g.nt;
  } else {
    mt_i = blockIdx.x / g.nt;
    nt_i = blockIdx.x % g.nt;
  }
  const int m0 = mt_i * BM, n0 = nt_i * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  const int kstart = blockIdx.z * g.k_chunk;
  const int kend = min(g.K, kstart + g.k_chunk);

  long a_off[GA::P];
  long b_off[GB::P];
  if constexpr (!A_KM) {
#pragma unroll
    for (int p = 0; p < GA::P; ++p) {
      const int gm = m0 + (t >> 3) + 32 * p;
      a_off[p] = gm < g.M ? (g.a_idx ? (long)g.a_idx[gm] : (long)gm) * g.lda : -1;
    }
  }
  if constexpr (!B_KN) {
#pragma unroll
    for (int p = 0; p < GB::P; ++p) {
      const int gn = n0 + (t >> 3) + 32 * p;
      b_off[p] = gn < g.N ? (g.b_idx ? (long)g.b_idx[gn] : (long)gn) * g.ldb : -1;
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[GA::P], rb[GB::P];
  // Fast path (uniform per workgroup and slab): the slab lies inside [kstart, kend), operands are 16-byte aligned and, for
  // k-major operands, the tile lies inside the matrix. Then every thread issues plain float4 loads without per-element
  // predicates (the predicated form compiles into a branch ladder that serialises the loads). Row-major (NT) operand rows
  // beyond M / N are clamped to row 0: they only feed accumulator rows / columns that are never stored.
  long a_fast[GA::P], b_fast[GB::P];
  bool a_can = g.vecA, b_can = g.vecB;
  if constexpr (!A_KM) {
#pragma unroll
    for (int p = 0; p < GA::P; ++p) a_fast[p] = (a_off[p] >= 0 ? a_off[p] : 0) + (t & 7) * 4;
  } else {
    a_can = a_can && (m0 + BM <= g.M);
  }
  if constexpr (!B_KN) {
#pragma unroll
    for (int p = 0; p < GB::P; ++p) b_fast[p] = (b_off[p] >= 0 ? b_off[p] : 0) + (t & 7) * 4;
  } else {
    b_can = b_can && (n0 + BN <= g.N);
  }
  auto fetch = [&](int kbase) {
    const bool inside = kbase + BK <= kend;
    if (inside && a_can) {
      if constexpr (!A_KM) {
#pragma unroll
        for (int p = 0; p < GA::P; ++p) ra[p] = *reinterpret_cast<const float4*>(g.A + a_fast[p] + kbase);
      } else {
        constexpr int TPR = BM / 4, RPP = 256 / TPR;
#pragma unroll
        for (int p = 0; p < GA::P; ++p) {
          const int gk = kbase + t / TPR + RPP * p;
          const long kr = g.a_idx ? (long)g.a_idx[gk] : (long)gk;
          ra[p] = *reinterpret_cast<const float4*>(g.A + kr * g.lda + m0 + (t % TPR) * 4);
        }
      }
    } else {
      if constexpr (!A_KM) load_mk<BM>(ra, g.A, a_off, kbase, kend, g.vecA, t);
      else load_km<BM>(ra, g.A, g.lda, g.a_idx, m0, g.M, kbase, kend, g.vecA, t);
    }
    if (inside && b_can) {
      if constexpr (!B_KN) {
#pragma unroll
        for (int p = 0; p < GB::P; ++p) rb[p] = *reinterpret_cast<const float4*>(g.B + b_fast[p] + kbase);
      } else {
        constexpr int TPR = BN / 4, RPP = 256 / TPR;
#pragma unroll
        for (int p = 0; p < GB::P; ++p) {
          const int gk = kbase + t / TPR + RPP * p;
          const long kr = g.b_idx ? (long)g.b_idx[gk] : (long)gk;
          rb[p] = *reinterpret_cast<const float4*>(g.B + kr * g.ldb + n0 + (t % TPR) * 4);
        }
      }
    } else {
      if constexpr (!B_KN) load_mk<BN>(rb, g.B, b_off, kbase, kend, g.vecB, t);
      else load_km<BN>(rb, g.B, g.ldb, g.b_idx, n0, g.N, kbase, kend, g.vecB, t);
    }
  };

  if (kstart < kend) fetch(kstart);
  for (int kbase = kstart; kbase < kend; kbase += BK) {
    __syncthreads();                       // previous slab fully consumed
    store_lds<BM, A_KM>(As, ra, t);
    store_lds<BN, B_KN>(Bs, rb, t);
    __syncthreads();
    if (kbase + BK < kend) fetch(kbase + BK);   // in flight under the MFMAs below
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      float4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = read_frag<BM, A_KM>(As, (wm * MI + i) * 32 + l31, kq, half);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = read_frag<BN, B_KN>(Bs, (wn * NI + j) * 32 + l31, kq, half);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: accumulator register r of a 32x32 tile is row (r&3) + 8*(r>>2) + 4*half, column lane&31
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (gm >= g.M) continue;
      const long crow = (g.c_idx ? (long)g.c_idx[gm] : (long)gm) * g.ldc;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int gn = n0 + (wn * NI + j) * 32 + l31;
        if (gn >= g.N) continue;
        float v = acc[i][j][r];
        if (g.slab) {
          g.slab[((long)blockIdx.z * g.M + gm) * g.N + gn] = v;
        } else if (g.atomic) {
          if (g.bias && blockIdx.z == 0) v += g.bias[gn];
          atomicAdd(&g.C[crow + gn], v);
        } else {
          if (g.bias) v += g.bias[gn];
          g.C[crow + gn] = sbr_act(v, g.act);
        }
      }
    }
  }
}

// out[m][n] = sum_z slab[z][m][n] in a fixed order (bitwise reproducible, unlike float atomics).
// 256 threads = 32 consecutive output elements x 8 z-groups; z-group g sums z = g, g+8, ... and the eight partial sums are
// combined through LDS in group order. One workgroup per 32 outputs keeps >= 512 workgroups busy even for a 128x128 dW.
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, long ldc, int M, int N, int splits) {
  __shared__ float part[8][32];
  const int e_local = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long total = (long)M * N;
  const long e = blockIdx.x * 32L + e_local;
  float acc = 0.f;
  if (e < total)
    for (int z = zg; z < splits; z += 8) acc += slab[z * total + e];
  part[zg][e_local] = acc;
  __syncthreads();
  if (zg == 0 && e < total) {
    float v = part[0][e_local];
#pragma unroll
    for (int g = 1; g < 8; ++g) v += part[g][e_local];
    C[(e / N) * ldc + (e % N)] = v;
  }
}

template <int WM, int WN, int MI, int NI, bool A_KM, bool B_KN>
static int launch(GemmArgs& g, int splits, hipStream_t s) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  g.mt = sbr_cdiv(g.M, BM);
  g.nt = sbr_cdiv(g.N, BN);
  // With fewer than 64 row panels the XCD-aware map would leave XCDs idle (panel p lives on XCD p % 8): use the plain map,
  // block ids then rotate over the XCDs through the split-K dimension.
  g.xcd_map = g.mt >= 64;
  const int per_xcd = sbr_cdiv(g.mt, 8) * g.nt;
  dim3 grid(g.xcd_map ? per_xcd * 8 : g.mt * g.nt, 1, splits);
  gemm_f32_kernel<WM, WN, MI, NI, A_KM, B_KN><<<grid, 256, 0, s>>>(g);
  SBR_CHECK_LAUNCH("sbr_gemm_f32");
  return SBR_OK;
}

static inline int aligned16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// mode: 0 = NT, 1 = NN, 2 = TN
extern "C" int sbr_gemm_f32(int mode, const float* A, long lda, const int* a_idx, const float* B, long ldb,
                            const int* b_idx, const float* bias, float* C, long ldc, const int* c_idx, int M, int N,
                            int K, int act, int accumulate_atomic, void* stream) {
  SBR_REQUIRE(mode >= 0 && mode <= 2, "sbr_gemm_f32: bad mode %d", mode);
  SBR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sbr_gemm_f32: negative size");
  if (M == 0 || N == 0) return SBR_OK;
  SBR_REQUIRE(A && B && C, "sbr_gemm_f32: null operand");
  SBR_REQUIRE(!(accumulate_atomic && act != SBR_ACT_NONE), "sbr_gemm_f32: activation with atomic accumulate");
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.B = B; g.ldb = ldb; g.b_idx = b_idx; g.bias = bias;
  g.C = C; g.ldc = ldc; g.c_idx = c_idx; g.M = M; g.N = N; g.K = K; g.act = act;
  g.vecA = aligned16(A, lda); g.vecB = aligned16(B, ldb);
  g.atomic = accumulate_atomic;
  g.slab = nullptr;
  g.k_chunk = ((K + BK - 1) / BK) * BK;
  if (g.k_chunk == 0) g.k_chunk = BK;
  int splits = 1;
  if (mode == 2) {
    // dW-shaped: small M x N output, K = number of rows. Split K so that >= ~512 workgroups exist; partial sums are
    // combined with float atomics into the zero-initialised output (the caller zeroes C).
    SBR_REQUIRE(accumulate_atomic, "sbr_gemm_f32: TN mode requires a zero-initialised C and accumulate_atomic=1");
    const int tiles = sbr_cdiv(M, 64) * sbr_cdiv(N, 128);
    int want = (1024 + tiles - 1) / tiles;
    int max_splits = sbr_cdiv(K, 256);
    splits = want < max_splits ? want : max_splits;
    if (splits < 1) splits = 1;
    g.k_chunk = sbr_cdiv(sbr_cdiv(K, splits), BK) * BK;
    splits = sbr_cdiv(K, g.k_chunk);
    g.splits = splits;
    return launch<2, 2, 1, 2, true, true>(g, splits, s);          // 64 x 128 tile
  }
  g.splits = 1;
  if (K >= BK) {
    const int rc = sbr_gemm_ring_launch(mode, g, s);               // persistent LDS-DMA ring (aligned operands)
    if (rc >= 0) return rc;
  }
  // 64 x 128 tiles when 128 x 128 tiles would give fewer than two workgroups per CU (measured: 45k x 128 x 768 gathered
  // projection 170 -> 138 us; 90k-row shapes are faster with the large tile)
  const int small_tile = (long)sbr_cdiv(M, 128) * sbr_cdiv(N, 128) < 512;
  if (mode == 1) {
    if (N <= 64) return launch<4, 1, 1, 2, false, true>(g, 1, s);  // 128 x 64
    if (small_tile) return launch<2, 2, 1, 2, false, true>(g, 1, s);   // 64 x 128
    return launch<2, 2, 2, 2, false, true>(g, 1, s);               // 128 x 128
  }
  if (N <= 64) return launch<4, 1, 1, 2, false, false>(g, 1, s);
  if (small_tile) return launch<2, 2, 1, 2, false, false>(g, 1, s);
  return launch<2, 2, 2, 2, false, false>(g, 1, s);
}

// ---- NT with a split-K slab reducer: few output tiles, long K ------------------------------------------------------------------
// The modality projectors at the reference's default batch (1,408 gathered rows x 768 features -> 128: 22 output tiles) leave
// 90 % of the CUs idle while each workgroup walks 24 K slabs one after the other (36 us). K is split so that ~128+ work items
// exist; the partial tiles go to slabs and a second kernel sums them in a fixed order (bitwise reproducible) and applies bias,
// activation and the output row scatter.
__global__ void splitk_reduce_epilogue_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                              const float* __restrict__ bias, int act, float* __restrict__ C, long ldc,
                                              const int* __restrict__ c_idx) {
  const long total = (long)M * N;
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= total) return;
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += slab[z * total + e];
  const int m = (int)(e / N), n = (int)(e - (long)m * N);
  if (bias) v += bias[n];
  C[(c_idx ? (long)c_idx[m] : (long)m) * ldc + n] = sbr_act(v, act);
}

static int nt_splits(int M, int N, int K) {
  const long tiles = (long)sbr_cdiv(M, 64) * sbr_cdiv(N, 128);
  if (tiles >= 96 || K < 8 * BK) return 1;                 // enough tiles to fill the chip, or too little K to share
  int splits = (int)((192 + tiles - 1) / tiles);            // aim at ~192 work items
  const int max_splits = K / (4 * BK);                      // at least 4 slabs per item
  if (splits > max_splits) splits = max_splits;
  return splits < 2 ? 1 : splits;
}

extern "C" long sbr_gemm_nt_splitk_workspace(int M, int N, int K) {
  const int splits = nt_splits(M, N, K);
  return splits > 1 ? (long)splits * M * N * (long)sizeof(float) : 0;
}

// C[ci(m), n] = act(sum_k A[ai(m), k] * B[n, k] + bias[n]) for the shapes sbr_gemm_nt_splitk_workspace() returns > 0 for
// (otherwise use sbr_gemm_f32 mode 0). workspace: that many bytes.
extern "C" int sbr_gemm_nt_splitk_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const float* bias,
                                      float* C, long ldc, const int* c_idx, int M, int N, int K, int act, void* workspace,
                                      long workspace_bytes, void* stream) {
  SBR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sbr_gemm_nt_splitk_f32: negative size");
  if (M == 0 || N == 0) return SBR_OK;
  SBR_REQUIRE(A && B && C, "sbr_gemm_nt_splitk_f32: null operand");
  int splits = nt_splits(M, N, K);
  SBR_REQUIRE(splits > 1, "sbr_gemm_nt_splitk_f32: shape %d x %d x %d is not split (use sbr_gemm_f32)", M, N, K);
  SBR_REQUIRE(workspace && workspace_bytes >= (long)splits * M * N * (long)sizeof(float), "sbr_gemm_nt_splitk_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.B = B; g.ldb = ldb; g.b_idx = nullptr; g.bias = nullptr;
  g.C = C; g.ldc = ldc; g.c_idx = nullptr; g.M = M; g.N = N; g.K = K; g.act = SBR_ACT_NONE;
  g.vecA = aligned16(A, lda); g.vecB = aligned16(B, ldb);
  g.atomic = 0;
  g.slab = (float*)workspace;
  g.k_chunk = sbr_cdiv(sbr_cdiv(K, splits), BK) * BK;
  splits = sbr_cdiv(K, g.k_chunk);
  g.splits = splits;
  int rc = sbr_gemm_ring_launch(0, g, s);
  if (rc < 0) rc = launch<2, 2, 1, 2, false, false>(g, splits, s);
  if (rc) return rc;
  const long total = (long)M * N;
  splitk_reduce_epilogue_kernel<<<sbr_cdiv(total, 256), 256, 0, s>>>(g.slab, splits, M, N, bias, act, C, ldc, c_idx);
  SBR_CHECK_LAUNCH("sbr_gemm_nt_splitk_f32/reduce");
  return SBR_OK;
}

// ---- TN with a split-K slab reducer ------------------------------------------------------------------------------------
static int tn_splits(int M, int N, int K) {
  const int tiles = sbr_cdiv(M, 64) * sbr_cdiv(N, 128);
  // work items (tiles x K ranges) to aim for: 512 for a single-panel dW (128 x 128: 45 us; 768 items 50 us, 1536 items 62 us —
  // the partial slabs grow with the split count), 1536 once there are many tiles (128 x 768 over 45056 rows: 110 vs 119 us)
  int want = ((tiles >= 8 ? 1536 : 512) + tiles - 1) / tiles;
  const int max_splits = sbr_cdiv(K, 4 * BK);         // at least 4 slabs of K per workgroup
  int splits = want < max_splits ? want : max_splits;
  const int min_splits = sbr_cdiv(K, 512);             // the ring kernel stages <= 512 gathered k-row indices per item
  if (splits < min_splits) splits = min_splits;
  return splits < 1 ? 1 : splits;
}

extern "C" long sbr_gemm_tn_f32_workspace(int M, int N, int K) {
  int splits = tn_splits(M, N, K);
  const int ss = sbr_tn_split_splits(M, N, K);                         // the bf16-split kernel cuts K its own way
  if (ss > splits) splits = ss;
  return (long)splits * M * N * (long)sizeof(float);
}

// 1 when the product runs on the bf16 matrix pipe (gemm_split_tn_kernel: M = 128 i, N = 128 j, K >= 4096 and SBR_GEMM_SPLIT / SBR_TN_SPLIT
// not 0), 0 when it stays on the fp32 ring kernel — what bench.py and the parity tests name the serving kernel by
extern "C" int sbr_gemm_tn_split_supported(int M, int N, int K) { return sbr_tn_split_splits(M, N, K) > 0; }

// slab pass of the TN product: partial tiles slab[z][M][N] for z < *splits_out (plain stores), no reduction
static int tn_slabs(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M, int N, int K,
                    void* workspace, long workspace_bytes, int* splits_out, hipStream_t s) {
  int splits = tn_splits(M, N, K);
  SBR_REQUIRE(workspace && workspace_bytes >= (long)splits * M * N * (long)sizeof(float), "sbr_gemm_tn_f32: workspace too small");
  {
    // the training step's dW shapes (M = 128, N = 128 j, thousands of rows) on the bf16 matrix pipe
    const int ss = sbr_tn_split_splits(M, N, K);
    if (ss > 0 && (long)ss * M * N * (long)sizeof(float) <= workspace_bytes) {
      const int rc = sbr_tn_split_launch(A, lda, a_idx, B, ldb, b_idx, M, N, K, (float*)workspace, splits_out, s);
      if (rc >= 0) return rc;
    }
  }
  GemmArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.B = B; g.ldb = ldb; g.b_idx = b_idx; g.bias = nullptr;
  g.C = nullptr; g.ldc = N; g.c_idx = nullptr; g.M = M; g.N = N; g.K = K; g.act = SBR_ACT_NONE;
  g.vecA = aligned16(A, lda); g.vecB = aligned16(B, ldb);
  g.atomic = 0;
  g.slab = (float*)workspace;
  g.k_chunk = sbr_cdiv(sbr_cdiv(K > 0 ? K : 1, splits), BK) * BK;
  splits = K > 0 ? sbr_cdiv(K, g.k_chunk) : 1;
  g.splits = splits;
  int rc = K >= BK ? sbr_gemm_ring_launch(2, g, s) : -1;
  if (rc < 0) rc = launch<2, 2, 1, 2, true, true>(g, splits, s);
  *splits_out = splits;
  return rc;
}

// C[m, n] = sum_k A[ak(k), m] * B[bk(k), n]  (C is overwritten). workspace: sbr_gemm_tn_f32_workspace(M, N, K) bytes.
extern "C" int sbr_gemm_tn_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx,
                               float* C, long ldc, int M, int N, int K, void* workspace, long workspace_bytes, void* stream) {
  SBR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sbr_gemm_tn_f32: negative size");
  if (M == 0 || N == 0) return SBR_OK;
  SBR_REQUIRE(A && B && C, "sbr_gemm_tn_f32: null operand");
  hipStream_t s = (hipStream_t)stream;
  int splits = 0;
  const int rc = tn_slabs(A, lda, a_idx, B, ldb, b_idx, M, N, K, workspace, workspace_bytes, &splits, s);
  if (rc) return rc;
  const long total = (long)M * N;
  splitk_reduce_kernel<<<sbr_cdiv(total, 32), 256, 0, s>>>((const float*)workspace, C, ldc, M, N, splits);
  SBR_CHECK_LAUNCH("sbr_gemm_tn_f32/reduce");
  return SBR_OK;
}

// The same product with the reduction DEFERRED: only the partial slabs are written (the workspace then belongs to this product
// until sbr_splitk_reduce_multi has consumed it); *splits_out (host) receives the number of slabs. Weight gradients are not
// needed before the optimizer, so a training step sums the slabs of all its dW products with ONE launch at the end of the
// backward pass (three reducer launches of 8 - 12 us each, latency-bound at 8 - 47 MB, become one).
extern "C" int sbr_gemm_tn_f32_slabs(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M,
                                     int N, int K, void* workspace, long workspace_bytes, int* splits_out, void* stream) {
  SBR_REQUIRE(M >= 1 && N >= 1 && K >= 1 && splits_out, "sbr_gemm_tn_f32_slabs: bad arguments");
  SBR_REQUIRE(A && B, "sbr_gemm_tn_f32_slabs: null operand");
  return tn_slabs(A, lda, a_idx, B, ldb, b_idx, M, N, K, workspace, workspace_bytes, splits_out, (hipStream_t)stream);
}

struct SplitkMulti {
  const float* slab[8];
  float* C[8];
  long ldc[8];
  int M[8], N[8], splits[8];
};

// splitk_reduce_kernel for up to 8 products at once: blockIdx.y selects the product (same fixed summation order)
__global__ void splitk_reduce_multi_kernel(SplitkMulti a) {
  __shared__ float part[8][32];
  const int q = blockIdx.y;
  const int e_local = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long total = (long)a.M[q] * a.N[q];
  const long e = blockIdx.x * 32L + e_local;
  if (blockIdx.x * 32L >= total) return;                    // uniform per workgroup
  const float* slab = a.slab[q];
  const int splits = a.splits[q];
  float acc = 0.f;
  if (e < total)
    for (int z = zg; z < splits; z += 8) acc += slab[z * total + e];
  part[zg][e_local] = acc;
  __syncthreads();
  if (zg == 0 && e < total) {
    float v = part[0][e_local];
#pragma unroll
    for (int g = 1; g < 8; ++g) v += part[g][e_local];
    a.C[q][(e / a.N[q]) * a.ldc[q] + (e % a.N[q])] = v;
  }
}

// The same with 16 bytes per lane (every N[q] a multiple of 4, slabs 16-byte aligned): a block sums 128 consecutive elements, four
// slab loads in flight per lane. Per element the summation order is the scalar kernel's (z = zg, zg + 8, ... per group, then the
// eight groups in order), so the results are bit-identical to it; 26 -> ~15 us for the step's three dW products (84 MB of slabs).
// Column-reduction workspaces finished by the same launch (sbr_splitk_reduce_multi_fin): slices n_red .. n_red + count - 1 of grid.y
struct SplitkFin {
  double* ws[8];
  float* out[8];
  int C[8];
  int count;
};

__global__ __launch_bounds__(256) void splitk_reduce_multi4_kernel(SplitkMulti a, SplitkFin f, int n_red) {
  __shared__ float4 part[8][32];
  if ((int)blockIdx.y >= n_red) {                            // a pending column sum -> its float vector (replicas left zeroed)
    const int fq = blockIdx.y - n_red;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < f.C[fq]) f.out[fq][i] = (float)sbr_colred_take(f.ws[fq], f.C[fq], i);
    return;
  }
  const int q = blockIdx.y;
  const int e_local = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long total = (long)a.M[q] * a.N[q];
  const long e = blockIdx.x * 128L + 4 * e_local;
  if (blockIdx.x * 128L >= total) return;                   // uniform per workgroup
  const float* slab = a.slab[q];
  const int splits = a.splits[q];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total) {
    int z = zg;
    for (; z + 24 < splits; z += 32) {
      const float4 v0 = *reinterpret_cast<const float4*>(slab + (long)z * total + e);
      const float4 v1 = *reinterpret_cast<const float4*>(slab + (long)(z + 8) * total + e);
      const float4 v2 = *reinterpret_cast<const float4*>(slab + (long)(z + 16) * total + e);
      const float4 v3 = *reinterpret_cast<const float4*>(slab + (long)(z + 24) * total + e);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
      acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; z < splits; z += 8) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (long)z * total + e);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[zg][e_local] = acc;
  __syncthreads();
  if (zg == 0 && e < total) {
    float4 v = part[0][e_local];
#pragma unroll
    for (int g = 1; g < 8; ++g) { const float4 p = part[g][e_local]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    float* c = a.C[q] + (e / a.N[q]) * a.ldc[q] + (e % a.N[q]);           // N % 4 == 0: the four elements share a row
    if ((((uintptr_t)c) & 15) == 0) *reinterpret_cast<float4*>(c) = v;
    else { c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
  }
}

static int splitk_reduce_multi_impl(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms,
                                    const int* Ns, const int* splits, const SplitkFin& fin, void* stream);

// slabs / outs: HOST arrays of device pointers; ldcs, Ms, Ns, splits: HOST arrays (copied into the launch)
extern "C" int sbr_splitk_reduce_multi(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms,
                                       const int* Ns, const int* splits, void* stream) {
  SplitkFin fin;
  fin.count = 0;
  return splitk_reduce_multi_impl(count, slabs, outs, ldcs, Ms, Ns, splits, fin, stream);
}

// The same launch also finishes up to 8 pending column reductions (what sbr_colred_finish does: out[i] = sum of the workspace's
// replicas of entry i, replicas left zeroed) — the two finishing launches at the end of a backward pass become one.
extern "C" int sbr_splitk_reduce_multi_fin(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms,
                                           const int* Ns, const int* splits, int fin_count, const void* const* fin_workspaces,
                                           const void* const* fin_outs, const int* fin_widths, void* stream) {
  SBR_REQUIRE(fin_count >= 0 && fin_count <= 8 && (fin_count == 0 || (fin_workspaces && fin_outs && fin_widths)),
              "sbr_splitk_reduce_multi_fin: 0..8 column reductions per call");
  SplitkFin fin;
  fin.count = fin_count;
  for (int q = 0; q < fin_count; ++q) {
    SBR_REQUIRE(fin_workspaces[q] && fin_outs[q] && fin_widths[q] >= 1, "sbr_splitk_reduce_multi_fin: null entry %d", q);
    fin.ws[q] = (double*)fin_workspaces[q]; fin.out[q] = (float*)fin_outs[q]; fin.C[q] = fin_widths[q];
  }
  return splitk_reduce_multi_impl(count, slabs, outs, ldcs, Ms, Ns, splits, fin, stream);
}

__global__ void splitk_fin_only_kernel(SplitkFin f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < f.C[blockIdx.y]) f.out[blockIdx.y][i] = (float)sbr_colred_take(f.ws[blockIdx.y], f.C[blockIdx.y], i);
}

static int splitk_reduce_multi_impl(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms,
                                    const int* Ns, const int* splits, const SplitkFin& fin, void* stream) {
  int cmax = 0;
  for (int q = 0; q < fin.count; ++q) cmax = fin.C[q] > cmax ? fin.C[q] : cmax;
  if (count == 0) {
    if (fin.count > 0) {
      splitk_fin_only_kernel<<<dim3(sbr_cdiv(cmax, 256), fin.count), 256, 0, (hipStream_t)stream>>>(fin);
      SBR_CHECK_LAUNCH("sbr_splitk_reduce_multi_fin");
    }
    return SBR_OK;
  }
  SBR_REQUIRE(count >= 1 && count <= 8 && slabs && outs && ldcs && Ms && Ns && splits, "sbr_splitk_reduce_multi: 1..8 products per call");
  SplitkMulti a;
  long most = 0;
  for (int q = 0; q < count; ++q) {
    SBR_REQUIRE(slabs[q] && outs[q] && Ms[q] >= 1 && Ns[q] >= 1 && splits[q] >= 1, "sbr_splitk_reduce_multi: bad entry %d", q);
    a.slab[q] = (const float*)slabs[q]; a.C[q] = (float*)outs[q]; a.ldc[q] = ldcs[q]; a.M[q] = Ms[q]; a.N[q] = Ns[q]; a.splits[q] = splits[q];
    const long total = (long)Ms[q] * Ns[q];
    most = total > most ? total : most;
  }
  bool vec4 = true;
  for (int q = 0; q < count; ++q) vec4 = vec4 && (Ns[q] % 4 == 0) && ((((uintptr_t)slabs[q]) & 15) == 0);
  if (vec4) {
    int gx = sbr_cdiv(most, 128);
    const int gfin = sbr_cdiv(cmax, 256);
    if (gfin > gx) gx = gfin;
    splitk_reduce_multi4_kernel<<<dim3(gx, count + fin.count), 256, 0, (hipStream_t)stream>>>(a, fin, count);
  } else {
    splitk_reduce_multi_kernel<<<dim3(sbr_cdiv(most, 32), count), 256, 0, (hipStream_t)stream>>>(a);
    if (fin.count > 0) splitk_fin_only_kernel<<<dim3(sbr_cdiv(cmax, 256), fin.count), 256, 0, (hipStream_t)stream>>>(fin);
  }
  SBR_CHECK_LAUNCH("sbr_splitk_reduce_multi");
  return SBR_OK;
}
