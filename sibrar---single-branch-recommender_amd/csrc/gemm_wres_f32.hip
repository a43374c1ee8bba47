// Weights-resident fp32 GEMM for the shared single-branch network's own products: C[M, 128] = A[M, 128] x W (W: 128 x 128).
//
// The layers of the shared MLP (algorithms/sgd_alg.py:1819-1833 -> modules/polylinear.py:51) multiply a tall activation matrix
// (R = B * N * k rows, R = 90,112 at the bench's batch) by a 128 x 128 weight, forward (NT: x W^T + b, activation) and backward
// (NN: dZ W -> dX). The general ring kernel (gemm_ring_f32.hip) streams BOTH operands through LDS slab by slab: for these shapes
// a 128 x 128 output tile is only four K slabs deep, half of the LDS-DMA traffic and every barrier is spent on re-fetching the
// same 64 KB weight, and the products run at 42-45 % of the matrix pipe and at half their HBM bound (37.5 us; 92 MB at 6 TB/s =
// 15 us, 2.95 GFLOP at 157 TFLOP/s = 19 us). Here
//   * every wave keeps ITS HALF of the weight in registers for the whole kernel: wave (wm, wn) owns output columns
//     [64 wn, 64 wn + 64) -> 2 column tiles x 16 k-quads x 4 = 128 VGPRs of B operands, loaded once;
//   * only A moves: 64-row tiles (32 KB) go HBM -> LDS by LDS-DMA into a two-stage ring, ONE barrier per tile (128 MFMAs per
//     wave) instead of one per 32-deep slab; the LDS image and the fragment reads are the ring kernel's (XOR-swizzled 16-byte
//     chunks, one ds_read_b128 feeds four MFMAs);
//   * the MFMA sequence per accumulator is the ring kernel's (k pairs (8q + s, 8q + 4 + s), q ascending, s = 0..3), so the
//     results are bit-identical to sbr_gemm_f32 on the same operands;
//   * epilogue options of the training step: bias + activation (forward); multiply by the activation derivative of a second
//     matrix Y (backward through the layer in front: dZ_prev = (dZ W) * act'(Y)) and per-column sums of what is stored (the bias
//     gradient of that layer), accumulated per workgroup over all its tiles and flushed once into a column-reduction workspace.
// Persistent: 2 workgroups per CU (64 KB of LDS, ~200 VGPRs), workgroup b walks tiles b, b + grid, ...
#include "gemm_args.h"
#include <type_traits>
#include <stdlib.h>

#define WR_N 128
#define WR_K 128
#define WR_BM 64
#define WR_COUNTED_WAIT 1
#define WR_STAGE (WR_BM * WR_K * 4)          // 32 KB: four slab images [64 rows][8 chunks of 16 B], chunk p of row r at p ^ ((r >> 1) & 7)

__device__ __attribute__((aligned(16))) float wr_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};   // source of out-of-range chunks

typedef __attribute__((address_space(3))) float wr_lds_f32;
typedef float wr_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) wr_v4f wr_lds_v4f;

struct WresArgs {
  const float* A; long lda;
  const float* W; long ldw;
  const float* bias;
  float* C; long ldc;
  long M;
  int act;
  const float* Y; long ldy;        // EPI 1: C = (A W) * act'(Y)
  double* colsum_ws;                // EPI 1: += column sums of C (replica layout of sbr_col_reduce, K = 1), may be null
};

template <int N>
__device__ __forceinline__ void wr_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: NT (W is [n][k]); MODE 1: NN (W is [k][n]). EPI 0: bias + activation; EPI 1: activation derivative of Y + column sums.
template <int MODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_wres_kernel(WresArgs g, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int n_mine = blockIdx.x < n_tiles ? (n_tiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  if (n_mine == 0) return;
  const float* zero = wr_zero_chunk;

  // ---- A tile issue: 8 chunks of 16 bytes per thread per tile (2 per slab image) ----------------------------------------------
  auto issue = [&](int tile, int stage) {
    const long m0 = (long)tile * WR_BM;
    unsigned char* st = smem + stage * WR_STAGE + wave * 1024;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = i * 256 + t, r = e >> 3, p = e & 7;
        const long gm = m0 + r;
        const float* src = gm < g.M ? g.A + gm * g.lda + sl * 32 + ((p ^ ((r >> 1) & 7)) << 2) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(st + sl * 8192 + i * 4096), 16, 0, 0);
      }
    }
  };
  issue(blockIdx.x, 0);                                        // the first tile travels while the weight is read

  // ---- this wave's half of the weight, as MFMA B operands: breg[j][q] = (W[n][8q + 4 half + s])_{s = 0..3}, n = 64 wn + 32 j + l31
  float4 breg[2][16];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = wn * 64 + j * 32 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = 8 * q + 4 * half;
      if constexpr (MODE == 0) {
        breg[j][q] = *reinterpret_cast<const float4*>(g.W + (long)n * g.ldw + k);
      } else {
        breg[j][q] = make_float4(g.W[(long)k * g.ldw + n], g.W[(long)(k + 1) * g.ldw + n], g.W[(long)(k + 2) * g.ldw + n],
                                 g.W[(long)(k + 3) * g.ldw + n]);
      }
    }
  }
  float bj[2] = {0.f, 0.f};
  if constexpr (EPI == 0) {
    if (g.bias) { bj[0] = g.bias[wn * 64 + l31]; bj[1] = g.bias[wn * 64 + 32 + l31]; }
  }

  const int a_rd = (wm * 32 + l31) * 128;                      // byte offset of this lane's row inside a slab image
  const int a_sw = ((wm * 32 + l31) >> 1) & 7;
  double cs[2] = {0.0, 0.0};                                   // EPI 1: running column sums of columns 64 wn + 32 j + l31 (this lane's rows)

  // Software pipeline over tiles: the finished values of tile t wait in `pend` and are stored BETWEEN the MFMAs of tile t + 1
  // (two stores per k-quad), so the store phase (8 us of a 40 us kernel when it ran after each tile's MFMAs: the two waves of a
  // SIMD run the same program in lockstep and do not cover each other's store phases) disappears behind the matrix pipe.
  f32x16 pend[2];
  long pend_m0 = -1;                                           // tile origin of `pend` (-1: nothing pending)
  auto store_pair = [&](int q, auto full_tag) {              // registers 2q, 2q + 1 of the pending tile (q = 0..15)
    constexpr bool FULL = decltype(full_tag)::value;
    float* cp = g.C + (pend_m0 + wm * 32 + 4 * half) * g.ldc + wn * 64 + l31;
    const int rows_left = (int)(g.M - pend_m0) - wm * 32 - 4 * half;
#pragma unroll
    for (int e = 2 * q; e < 2 * q + 2; ++e) {
      const int j = e >> 4, r = e & 15, lr = (r & 3) + 8 * (r >> 2);
      if (FULL || lr < rows_left) cp[(long)lr * g.ldc + j * 32] = pend[j][r];
    }
  };

#pragma unroll 1
  for (int it = 0; it < n_mine; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    const long m0 = (long)tile * WR_BM;
    // this thread's 8 DMA instructions of the tile are older than the 32 stores of the tile stored during the previous iteration
    // (always an interior tile: a ragged tile is the last of the whole product and is stored by the flush below)
    if (WR_COUNTED_WAIT && it >= 2 && EPI == 0) wr_wait_vmcnt<32>(); else wr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                              // every thread's part landed; the other stage has been consumed
    asm volatile("" ::: "memory");
    float4 yv[2][4];                                           // EPI 1: Y values of this lane's 32 outputs, loaded BEFORE the next DMA
    if constexpr (EPI == 1) {
      // accumulator register r of column tile j is row (r & 3) + 8 (r >> 2) + 4 half of the wave's 32 rows, column l31
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float v[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const long gm = m0 + wm * 32 + rr + 8 * r4 + 4 * half;
            v[rr] = gm < g.M ? g.Y[gm * g.ldy + wn * 64 + j * 32 + l31] : 0.f;
          }
          yv[j][r4] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (it + 1 < n_mine) issue(tile + gridDim.x, (it + 1) & 1);
    const wr_lds_f32* As = (const wr_lds_f32*)(smem + (it & 1) * WR_STAGE);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto compute = [&](auto store_tag) {                      // the tile's 128 MFMAs (+ the pending tile's 32 stores between them)
      constexpr bool STORE = decltype(store_tag)::value;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int sl = q >> 2, kq = q & 3;
        const wr_v4f a = *(const wr_lds_v4f*)(As + ((sl * 8192 + a_rd + (((2 * kq + half) ^ a_sw) << 4)) >> 2));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[j][q].x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[j][q].y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[j][q].z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[j][q].w, acc[j], 0, 0, 0);
        }
        if constexpr (STORE) store_pair(q, std::true_type{});
      }
      // pin the issue order the source spells (hipcc otherwise sinks the stores behind the last MFMA and hoists every LDS read):
      // the next k-quad's A fragment, this k-quad's 8 MFMAs, two stores of the pending tile
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q < 15) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if constexpr (STORE) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
      }
    };
    if (EPI == 0 && pend_m0 >= 0) compute(std::true_type{});   // EPI 1 has no registers for a pending tile beside Y
    else compute(std::false_type{});
    // ---- finish the tile in registers: bias + activation, or the activation derivative of Y (+ this tile's column sums). Every
    // mode decision is uniform and outside the element loops.
    const int rows_left = (int)(g.M - m0) - wm * 32 - 4 * half;
    auto finish = [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;            // 0: + bias, 1: relu(+ bias), 2: sbr_act(+ bias), 3: * act'(Y)
      float ts[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[j][r];
          if constexpr (KIND <= 2) v += bj[j];
          if constexpr (KIND == 1) v = sbr_relu(v);
          if constexpr (KIND == 2) v = sbr_act(v, g.act);
          if constexpr (KIND == 3) {
            const float4 y4 = yv[j][r >> 2];
            const float y = (r & 3) == 0 ? y4.x : ((r & 3) == 1 ? y4.y : ((r & 3) == 2 ? y4.z : y4.w));
            v = v * sbr_act_grad_from_out(y, g.act);
            if ((r & 3) + 8 * (r >> 2) < rows_left) ts[j] += v;
          }
          pend[j][r] = v;
        }
      if constexpr (KIND == 3) { cs[0] += (double)ts[0]; cs[1] += (double)ts[1]; }
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    if constexpr (EPI == 1) finish(T3{});
    else if (g.act == SBR_ACT_NONE) finish(T0{});
    else if (g.act == SBR_ACT_RELU) finish(T1{});
    else finish(T2{});
    pend_m0 = m0;
    if constexpr (EPI == 1) {                                  // stored at once (the Y values took the pending tile's registers)
      {
        if (m0 + WR_BM <= g.M) {
#pragma unroll
          for (int q = 0; q < 16; ++q) store_pair(q, std::true_type{});
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) store_pair(q, std::false_type{});
        }
      }
      pend_m0 = -1;
    }
  }
  // flush: the last tile of this workgroup (possibly the ragged last tile of the product)
  if (pend_m0 >= 0) {
    if (pend_m0 + WR_BM <= g.M) {
#pragma unroll
      for (int q = 0; q < 16; ++q) store_pair(q, std::true_type{});
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) store_pair(q, std::false_type{});
    }
  }
  if constexpr (EPI == 1) {
    if (g.colsum_ws) {
      // rows 4 half + ... of the two lane halves -> one sum per column and wave; one double atomic per column, wave and kernel
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const double o = cs[j] + __shfl_xor(cs[j], 32, 64);
        if (half == 0)
          atomicAdd(g.colsum_ws + (long)(1 + (blockIdx.x % SBR_COLRED_REP)) * WR_N + wn * 64 + j * 32 + l31, o);
      }
    }
  }
}

static bool wr_al16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// 1 when sbr_gemm_wres_f32 takes this product (else use sbr_gemm_f32): N = K = 128, 16-byte aligned operands
extern "C" int sbr_gemm_wres_supported(long M, int N, int K) { return M >= 1 && N == WR_N && K == WR_K; }

// mode 0 (NT): C = act(A W^T + bias), W [128 n][128 k]; mode 1 (NN): C = A W, W [128 k][128 n].
// Y != NULL (mode 1 only): C = (A W) * act'(Y) with `act` the activation whose OUTPUT Y is, and colsum_ws (17 * 128 doubles,
// contract of sbr_colsum / sbr_colred_finish, may be NULL) receives the pending column sums of C.
extern "C" int sbr_gemm_wres_f32(int mode, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc,
                                 long M, int N, int K, int act, const float* Y, long ldy, double* colsum_ws, void* stream) {
  SBR_REQUIRE(mode == 0 || mode == 1, "sbr_gemm_wres_f32: mode %d", mode);
  if (M == 0) return SBR_OK;
  SBR_REQUIRE(sbr_gemm_wres_supported(M, N, K), "sbr_gemm_wres_f32: shape %ld x %d x %d not supported (N = K = 128)", M, N, K);
  SBR_REQUIRE(A && W && C, "sbr_gemm_wres_f32: null operand");
  SBR_REQUIRE(wr_al16(A, lda) && wr_al16(W, ldw) && (ldc & 3) == 0, "sbr_gemm_wres_f32: operands must be 16-byte aligned");
  SBR_REQUIRE(!(Y && mode == 0) && !(colsum_ws && !Y) && !(Y && bias), "sbr_gemm_wres_f32: Y / colsum_ws belong to mode 1 without bias");
  WresArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.act = act; g.Y = Y; g.ldy = ldy;
  g.colsum_ws = colsum_ws;
  const int n_tiles = sbr_cdiv(M, WR_BM);
  int grid = 512;
  if (grid > n_tiles) grid = n_tiles;
  const size_t lds = 2 * WR_STAGE;
  hipStream_t s = (hipStream_t)stream;
#define WR_LAUNCH(MODE, EPI)                                                                                              \
  do {                                                                                                                     \
    static int attr_dev = -1;                                                                                          \
    if (sbr_attr_stale(&attr_dev)) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)gemm_wres_kernel<MODE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        sbr_set_error("sbr_gemm_wres_f32: cannot raise the dynamic LDS limit");                                            \
        return SBR_ERR_HIP;                                                                                                \
      }                                                                                                                    \
    }                                                                                                                      \
    gemm_wres_kernel<MODE, EPI><<<grid, 256, lds, s>>>(g, n_tiles);                                                        \
  } while (0)
  if (mode == 0) WR_LAUNCH(0, 0);
  else if (Y) WR_LAUNCH(1, 1);
  else WR_LAUNCH(1, 0);
#undef WR_LAUNCH
  SBR_CHECK_LAUNCH("sbr_gemm_wres_f32");
  return SBR_OK;
}
