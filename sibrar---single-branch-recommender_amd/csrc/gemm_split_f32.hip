// fp32 GEMM on the bf16 matrix pipe for the shared single-branch network's own products: C[M, 128] = A[M, 128] x W (W: 128 x 128).
//
// The layers of the shared MLP (algorithms/sgd_alg.py:1819-1833 -> modules/polylinear.py:51) multiply a tall activation matrix
// (R = B * N * k rows; 90,112 at the bench's batch) by a 128 x 128 weight, forward (NT: x W^T + b, activation) and backward
// (NN: dZ W -> dX). With v_mfma_f32_32x32x2_f32 these products are bound by the fp32 matrix pipe (157 TFLOP/s: 19 us of pipe time
// against 15 us of HBM time, 38-42 us measured).  The bf16 pipe is 16x faster per multiply-add, and an fp32 number IS the exact
// sum of three bf16 numbers:
//     x = x0 + x1 + x2,   x0 = bf16_rne(x), x1 = bf16_rne(x - x0), x2 = x - x0 - x1   (8 + 8 + 8 significand bits; both
//     subtractions are exact in fp32, and the last remainder has at most 8 significant bits, so x2 is exact)
// so   x * w = sum_{i + j <= 2} x_i w_j  +  (x1 w2 + x2 w1 + x2 w2),   |dropped| <= 2^-23 |x w|   (|x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|)
// i.e. six bf16 MFMAs (products exact, fp32 accumulate) give the product to within one fp32 rounding of each term: the same
// error class as the fp32 pipe's own accumulation, at 16 / 6 = 2.7x its rate.  The kernel below is therefore bound by HBM
// (92 MB per product), not by the matrix pipe.  tests/test_hip_kernels.py::test_split_gemm_* measures the error against an fp64
// product next to the fp32-pipe kernel's.  Non-finite inputs give NaN (inf - inf in the split) where the fp32 pipe gives inf: the
// outputs such an operand poisons are the same on both pipes (its row / column), every other output keeps its bits, and the ReLU
// epilogues keep a NaN a NaN (sbr_relu) — pinned by tests/test_hip_kernels.py::test_split_gemm_on_non_finite_operands.
//
// Layout of the work (no barrier after the set-up):
//   * one workgroup of 8 waves per CU; the weight is split ONCE per workgroup into three bf16 planes that stay in LDS (96 KB) in
//     MFMA operand order: fragment (plane p, column tile j, k step s) is 64 lanes x 16 bytes, one ds_read_b128 per lane;
//   * every wave owns whole 32-row blocks of A (block = global wave id + i * waves): it reads its rows straight from global memory
//     into registers (lane = row, 32 contiguous bytes per k step), splits them with 4.5 VALU ops per element and runs
//     4 column tiles x 8 k steps x 6 MFMAs; no operand is shared between waves, so nothing has to be synchronised;
//   * the next half block is in flight while the current one is multiplied (two 32-register raw buffers);
//   * epilogue options of the training step as in gemm_wres_f32.hip: bias + activation (forward); multiply by the activation
//     derivative of a second matrix Y and accumulate column sums (backward: dZ_prev = (dZ W) * act'(Y) and its bias gradient).
#include "gemm_split_common.h"
#include <type_traits>

#define SP_N 128
#define SP_K 128
#define SP_WAVES 8
#ifndef SPM_WAVES
#define SPM_WAVES 8                         // waves per workgroup of the K = N = 128 kernel (the projector kernel keeps SP_WAVES)
#endif
#define SP_PLANE (4 * 8 * 64 * 16)           // bytes of one bf16 plane of W in fragment order: [4 column tiles][8 k steps][64 lanes][8 bf16]

struct SplitArgs {
  const float* A; long lda;
  const float* W; long ldw;
  const float* bias;
  float* C; long ldc;
  long M;
  int act;
  const float* Y; long ldy;        // EPI 1: C = (A W) * act'(Y)
  double* colsum_ws;                // EPI 1: += column sums of C (replica layout of sbr_col_reduce, K = 1), may be null
  // EPI 2, optional (fin_mean != null): the workgroup whose arrival comes last turns the pending sums into the BatchNorm's batch
  // statistics itself (what sbr_bn_finalize_stats does with a launch of its own) and resets the replicas and the counter
  unsigned long long* fin_arrive;   // zero on entry, zero again on return
  float* fin_mean; float* fin_rstd; float* fin_running_mean; float* fin_running_var; long* fin_nbt;
  float fin_eps, fin_momentum;
};

// MODE 0: NT (W is [n][k]); MODE 1: NN (W is [k][n]). EPI 0: bias + activation; EPI 1: activation derivative of Y + column sums;
// EPI 2 (MODE 0): EPI 0 + per-column sums and sums of squares of what is stored (the batch statistics of a BatchNorm that follows,
// left pending in colsum_ws in the replica layout of sbr_col_reduce<2>: the separate statistics pass over the output is not needed).
template <int MODE, int EPI>
__global__ __launch_bounds__(64 * SPM_WAVES, 1) void gemm_split_kernel(SplitArgs g, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // Block -> wave: SIMD sid = 4 blockIdx + (wave & 3) of the chip takes blocks sid, sid + S, sid + 2 S, ... (S = SIMDs in the grid),
  // alternating between its two waves, so that the matrix pipes - the shared resource of the two waves - get equal block counts
  // (2,816 blocks over 1,024 pipes: 3 or 2 each; numbering the waves 8 blockIdx + wave would give 4 or 2).
#if SPM_WAVES == 8
  const int gw = (wave >> 2) * (gridDim.x * 4) + blockIdx.x * 4 + (wave & 3), nw = gridDim.x * SPM_WAVES;
#else
  const int gw = wave * gridDim.x + blockIdx.x, nw = gridDim.x * SPM_WAVES;
#endif

  // raw A of one half block: k steps 4 h .. 4 h + 3, per step the 8 floats k = 16 s + 8 half .. + 7 of row l31 of the block
  auto load_half = [&](int blk, int h, float4 (&raw)[4][2]) {
    long row = (long)blk * 32 + l31;
    if (row >= g.M) row = g.M - 1;                               // rows past the end are computed on a valid row and never stored
    const float* p = g.A + row * g.lda + h * 64 + half * 8;
#ifdef SP_ABL_COALESCED       /* lab (timing only, wrong results): the same bytes of the block with whole-line wave instructions */
    const float* pc = g.A + (long)blk * 32 * g.lda + h * 2048 + lane * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      raw[s][0] = *reinterpret_cast<const float4*>(pc + (2 * s) * 256);
      raw[s][1] = *reinterpret_cast<const float4*>(pc + (2 * s + 1) * 256);
    }
    (void)p;
    return;
#endif
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      raw[s][0] = *reinterpret_cast<const float4*>(p + s * 16);
      raw[s][1] = *reinterpret_cast<const float4*>(p + s * 16 + 4);
    }
  };
  float4 r0[4][2], r1[4][2];
  if (gw < n_blocks) load_half(gw, 0, r0);                       // in flight during the set-up

  // ---- set-up: the weight as three bf16 planes in fragment order. Chunk (n, kc) = the 8 values W(n, 8 kc .. 8 kc + 7) is the operand
  // of lane (n & 31) + 32 (kc & 1) in fragment (column tile n >> 5, k step kc >> 1). 2048 chunks, 4 per thread.
  constexpr int SETUP_IT = (2048 + 64 * SPM_WAVES - 1) / (64 * SPM_WAVES);
#pragma unroll
  for (int i = 0; i < SETUP_IT; ++i) {
    int n, kc;
    float4 lo, hi;
    if (SPM_WAVES != 8 && (MODE == 0 ? wave + i * SPM_WAVES >= 32 : t + i * 64 * SPM_WAVES >= 2048)) break;
    if constexpr (MODE == 0) {
      // rows of W are contiguous in k. A wave-instruction handles ONE operand fragment (column tile j, k step ks): lane L reads the
      // 8 values W(32 j + (L & 31), 16 ks + 8 (L >> 5) .. + 7) and writes its 16 bytes at lane position L of the fragment —
      // consecutive lanes, consecutive LDS addresses. (Thread order along k — 16 consecutive threads per row — makes 16 lanes write
      // 512 bytes apart: a 16-way bank conflict on every ds_write_b128 of the set-up.)
      const int f = SPM_WAVES == 8 ? wave * 4 + i : wave + i * SPM_WAVES;      // fragments 4 w .. 4 w + 3 of the 32 (eight waves)
      n = (f >> 3) * 32 + l31; kc = (f & 7) * 2 + half;
      const float* p = g.W + (long)n * g.ldw + kc * 8;
      lo = *reinterpret_cast<const float4*>(p);
      hi = *reinterpret_cast<const float4*>(p + 4);
    } else {                                                     // rows of W are contiguous in n: consecutive threads read consecutive n
      const int c = t + i * 64 * SPM_WAVES;
      n = c & 127; kc = c >> 7;
      const float* p = g.W + (long)(kc * 8) * g.ldw + n;
      lo = make_float4(p[0], p[g.ldw], p[2 * g.ldw], p[3 * g.ldw]);
      hi = make_float4(p[4 * g.ldw], p[5 * g.ldw], p[6 * g.ldw], p[7 * g.ldw]);
    }
    sp_u32x4 p0, p1, p2;
    sp_split8(lo, hi, p0, p1, p2);
    const int off = ((((n >> 5) * 8 + (kc >> 1)) * 64) + (kc & 1) * 32 + (n & 31)) * 16;
    *(sp_lds_u32x4*)(smem + off) = p0;
    *(sp_lds_u32x4*)(smem + SP_PLANE + off) = p1;
    *(sp_lds_u32x4*)(smem + 2 * SP_PLANE + off) = p2;
  }
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == 0 || EPI == 2) {
    if (g.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bj[j] = g.bias[j * 32 + l31];
    }
  }
  __syncthreads();                                               // the only barrier of the kernel

  const unsigned char* wfrag = smem + lane * 16;
  double cs[4] = {0.0, 0.0, 0.0, 0.0};                           // EPI 1 / 2: running column sums of columns 32 j + l31 over this lane's rows
  double cq[4] = {0.0, 0.0, 0.0, 0.0};                           // EPI 2: ... of their squares

  sp_f32x16 acc[4];
  auto mult_half = [&](int h, const float4 (&raw)[4][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sp_u32x4 a0, a1, a2;
      sp_split8(raw[s][0], raw[s][1], a0, a1, a2);
      const int ks = h * 4 + s;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {                           // two column tiles at a time: two independent accumulator chains
        sp_u32x4 w[2][3];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            w[jj][p] = *(const sp_lds_u32x4*)(wfrag + p * SP_PLANE + (((jp * 2 + jj) * 8 + ks) * 64) * 16);
        // smallest terms first
        acc[jp * 2 + 0] = sp_mfma(a2, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a2, w[1][0], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][2], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][2], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a1, w[0][1], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a1, w[1][1], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a1, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a1, w[1][0], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][1], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][1], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][0], acc[jp * 2 + 1]);
      }
    }
  };

#pragma unroll 1
  for (int blk = gw; blk < n_blocks; blk += nw) {
    const long m0 = (long)blk * 32;
    load_half(blk, 1, r1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    mult_half(0, r0);
    if (blk + nw < n_blocks) load_half(blk + nw, 0, r0);
    mult_half(1, r1);

    // ---- epilogue: accumulator register r of column tile j is row (r & 3) + 8 (r >> 2) + 4 half of the block, column 32 j + l31
    const int rows_left = (int)(g.M - m0) - 4 * half;
    auto finish = [&](auto kind_tag, auto full_tag) {
      constexpr int KIND = decltype(kind_tag)::value;            // 0: + bias, 1: relu(+ bias), 2: sbr_act(+ bias), 3: * act'(Y)
      constexpr bool FULL = decltype(full_tag)::value;
      float* cp = g.C + (m0 + 4 * half) * g.ldc + l31;
      const float* yp = nullptr;
      if constexpr (KIND == 3) yp = g.Y + (m0 + 4 * half) * g.ldy + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ts = 0.f, tq = 0.f;
        float yv[16];
        if constexpr (KIND == 3) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2);
            yv[r] = (FULL || lr < rows_left) ? yp[(long)lr * g.ldy + j * 32] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = (r & 3) + 8 * (r >> 2);
          float v = acc[j][r];
          if constexpr (KIND <= 2) v += bj[j];
          if constexpr (KIND == 1) v = sbr_relu(v);
          if constexpr (KIND == 2) v = sbr_act(v, g.act);
          if constexpr (KIND == 3) {
            v = v * sbr_act_grad_from_out(yv[r], g.act);
            if (FULL || lr < rows_left) ts += v;
          }
          if constexpr (EPI == 2) {
            if (FULL || lr < rows_left) { ts += v; tq += v * v; }
          }
#ifdef SP_ABL_NOSTORE       /* lab (timing only): the epilogue stores nothing */
          asm volatile("" ::"v"(v));
#else
          if (FULL || lr < rows_left) cp[(long)lr * g.ldc + j * 32] = v;
#endif
        }
        if constexpr (KIND == 3 || EPI == 2) cs[j] += (double)ts;
        if constexpr (EPI == 2) cq[j] += (double)tq;
      }
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    const bool full = m0 + 32 <= g.M;
    if constexpr (EPI == 1) {
      if (full) finish(T3{}, std::true_type{}); else finish(T3{}, std::false_type{});
    } else if (g.act == SBR_ACT_NONE) {
      if (full) finish(T0{}, std::true_type{}); else finish(T0{}, std::false_type{});
    } else if (g.act == SBR_ACT_RELU) {
      if (full) finish(T1{}, std::true_type{}); else finish(T1{}, std::false_type{});
    } else {
      if (full) finish(T2{}, std::true_type{}); else finish(T2{}, std::false_type{});
    }
  }
  if constexpr (EPI == 2) {
    // replica layout of sbr_col_reduce<2>: [1 + replica][2][128] doubles (sums, then sums of squares)
    double* rep = g.colsum_ws + (long)(1 + (gw % SBR_COLRED_REP)) * 2 * SP_N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double o = cs[j] + __shfl_xor(cs[j], 32, 64), q = cq[j] + __shfl_xor(cq[j], 32, 64);
      if (half == 0) { atomicAdd(rep + j * 32 + l31, o); atomicAdd(rep + SP_N + j * 32 + l31, q); }
    }
    if (g.fin_mean) {
      // arrival (agent-scope counter; the sums went out as agent-scope atomics, performed at the memory side: every wave waits
      // for its own to be performed, then the barrier, then the counter add — no L2 write-back needed)
      __shared__ int fin_last;
      SBR_DRAIN_VMEM();
      __syncthreads();
      if (t == 0) fin_last = atomicAdd(g.fin_arrive, 1ull) == gridDim.x - 1;
      __syncthreads();
      if (fin_last && t < SP_N) {
        // = bn_finalize_kernel (batchnorm.hip): replicas summed in replica order, left zeroed
        const int KD = 2 * SP_N;
        double sm = 0.0, sq = 0.0;
#pragma unroll
        for (int r = 1; r <= SBR_COLRED_REP; ++r) {
          sm += __hip_atomic_load(&g.colsum_ws[(long)r * KD + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sq += __hip_atomic_load(&g.colsum_ws[(long)r * KD + SP_N + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g.colsum_ws[(long)r * KD + t] = 0.0;
          g.colsum_ws[(long)r * KD + SP_N + t] = 0.0;
        }
        const double n = (double)g.M;
        const double m = sm / n;
        double var = sq / n - m * m;
        if (var < 0.0) var = 0.0;
        g.fin_mean[t] = (float)m;
        g.fin_rstd[t] = (float)(1.0 / sqrt(var + (double)g.fin_eps));
        if (g.fin_running_mean) {
          const double unbiased = g.M > 1 ? var * n / (n - 1.0) : var;
          g.fin_running_mean[t] = (1.f - g.fin_momentum) * g.fin_running_mean[t] + g.fin_momentum * (float)m;
          g.fin_running_var[t] = (1.f - g.fin_momentum) * g.fin_running_var[t] + g.fin_momentum * (float)unbiased;
        }
        if (t == 0) {
          if (g.fin_nbt) g.fin_nbt[0] += 1;
          __hip_atomic_store(g.fin_arrive, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if constexpr (EPI == 1) {
    if (g.colsum_ws) {
      // rows 4 half + ... of the two lane halves -> one sum per column and wave; one double atomic per column, wave and kernel
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double o = cs[j] + __shfl_xor(cs[j], 32, 64);
        if (half == 0) atomicAdd(g.colsum_ws + (long)(1 + (gw % SBR_COLRED_REP)) * SP_N + j * 32 + l31, o);
      }
    }
  }
}

static bool sp_al16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// 1 when sbr_gemm_split_f32 takes this product (else use sbr_gemm_f32 / sbr_gemm_wres_f32): N = K = 128
extern "C" int sbr_gemm_split_supported(long M, int N, int K) { return M >= 1 && N == SP_N && K == SP_K; }

// mode 0 (NT): C = act(A W^T + bias), W [128 n][128 k]; mode 1 (NN): C = A W, W [128 k][128 n]. fp32 operands and result; the
// multiplications run on the bf16 matrix pipe over exact three-way splits of both operands (six terms, fp32 accumulate).
// Y != NULL (mode 1 only): C = (A W) * act'(Y) with `act` the activation whose OUTPUT Y is, and colsum_ws (17 * 128 doubles,
// contract of sbr_colsum / sbr_colred_finish, may be NULL) receives the pending column sums of C.
// mode 0 with colsum_ws != NULL (17 * 2 * 128 doubles, zero on entry like every column-reduction workspace): the per-column sums
// and sums of squares of C are left pending there (sbr_bn_finalize_stats turns them into the statistics of the BatchNorm behind C).
struct SplitFin { unsigned long long* arrive; float *mean, *rstd, *running_mean, *running_var; long* nbt; float eps, momentum; };

static int gemm_split_impl(int mode, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc,
                           long M, int N, int K, int act, const float* Y, long ldy, double* colsum_ws, const SplitFin* fin, void* stream) {
  SBR_REQUIRE(mode == 0 || mode == 1, "sbr_gemm_split_f32: mode %d", mode);
  if (M == 0) return SBR_OK;
  SBR_REQUIRE(sbr_gemm_split_supported(M, N, K), "sbr_gemm_split_f32: shape %ld x %d x %d not supported (N = K = 128)", M, N, K);
  SBR_REQUIRE(A && W && C, "sbr_gemm_split_f32: null operand");
  SBR_REQUIRE(sp_al16(A, lda) && (mode == 1 || sp_al16(W, ldw)), "sbr_gemm_split_f32: operands must be 16-byte aligned");
  SBR_REQUIRE(!(Y && mode == 0) && !(colsum_ws && !Y && mode == 1) && !(Y && bias), "sbr_gemm_split_f32: Y belongs to mode 1 without bias");
  SplitArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.act = act; g.Y = Y; g.ldy = ldy;
  g.colsum_ws = colsum_ws;
  g.fin_arrive = fin ? fin->arrive : nullptr; g.fin_mean = fin ? fin->mean : nullptr; g.fin_rstd = fin ? fin->rstd : nullptr;
  g.fin_running_mean = fin ? fin->running_mean : nullptr; g.fin_running_var = fin ? fin->running_var : nullptr;
  g.fin_nbt = fin ? fin->nbt : nullptr; g.fin_eps = fin ? fin->eps : 0.f; g.fin_momentum = fin ? fin->momentum : 0.f;
  const int n_blocks = sbr_cdiv(M, 32);
  int grid = sbr_cdiv(n_blocks, SPM_WAVES);
  if (grid > 256) grid = 256;
  const size_t lds = 3 * SP_PLANE;
  hipStream_t s = (hipStream_t)stream;
#define SP_LAUNCH(MODE, EPI)                                                                                              \
  do {                                                                                                                     \
    static int attr_dev = -1;                                                                                          \
    if (sbr_attr_stale(&attr_dev)) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)gemm_split_kernel<MODE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        sbr_set_error("sbr_gemm_split_f32: cannot raise the dynamic LDS limit");                                           \
        return SBR_ERR_HIP;                                                                                                \
      }                                                                                                                    \
    }                                                                                                                      \
    gemm_split_kernel<MODE, EPI><<<grid, 64 * SPM_WAVES, lds, s>>>(g, n_blocks);                                            \
  } while (0)
  if (mode == 0 && colsum_ws) SP_LAUNCH(0, 2);
  else if (mode == 0) SP_LAUNCH(0, 0);
  else if (Y) SP_LAUNCH(1, 1);
  else SP_LAUNCH(1, 0);
#undef SP_LAUNCH
  SBR_CHECK_LAUNCH("sbr_gemm_split_f32");
  return SBR_OK;
}

extern "C" int sbr_gemm_split_f32(int mode, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc,
                                  long M, int N, int K, int act, const float* Y, long ldy, double* colsum_ws, void* stream) {
  return gemm_split_impl(mode, A, lda, W, ldw, bias, C, ldc, M, N, K, act, Y, ldy, colsum_ws, nullptr, stream);
}

// mode 0 with the statistics epilogue AND the BatchNorm finalisation (sbr_bn_finalize_stats) in the same launch: the workgroup that
// arrives last computes save_mean / save_rstd of the M rows of C, updates the running statistics (may be NULL) and
// num_batches_tracked (may be NULL) and leaves colsum_ws and *arrive (one zeroed 64-bit word, owned by the BatchNorm) zeroed.
extern "C" int sbr_gemm_split_bnstats_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, long M,
                                          int N, int K, int act, double* colsum_ws, void* arrive, float* running_mean,
                                          float* running_var, long* num_batches_tracked, float* save_mean, float* save_rstd, float eps,
                                          float momentum, void* stream) {
  SBR_REQUIRE(colsum_ws && arrive && save_mean && save_rstd && M >= 1, "sbr_gemm_split_bnstats_f32: null operand");
  SBR_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sbr_gemm_split_bnstats_f32: running_mean and running_var go together");
  SplitFin fin = {(unsigned long long*)arrive, save_mean, save_rstd, running_mean, running_var, num_batches_tracked, eps, momentum};
  return gemm_split_impl(0, A, lda, W, ldw, bias, C, ldc, M, N, K, act, nullptr, 0, colsum_ws, &fin, stream);
}

// =====================================================================================================================
// Modality projector on the bf16 matrix pipe: C[ci(m), 128] = act(A[ai(m), K] x W^T + bias), W [128 n][K], K a multiple of 128
// (algorithms/sgd_alg.py:1279-1396 FeatureEmbedding of a dense modality: nn.Linear(F, C) over the gathered feature rows).
// Same arithmetic as above (exact three-way bf16 splits of both operands, six MFMA terms, fp32 accumulate); what changes is the
// loop: the weight does not fit in LDS as a whole (128 x 768 fp32 = three bf16 planes of 288 KB), so a workgroup walks K in
// chunks of 128 and rebuilds the three planes of the chunk in LDS (96 KB) between two barriers, while every wave keeps the
// accumulators of ONE 32-row block across the chunks. The row gather (a_idx: item -> feature row) is the wave's own row
// pointer, the row scatter of the result (c_idx: slot of the shared network's input) is applied in the epilogue. The raw
// weight values of the next chunk and the next half block of A are in flight while the current ones are multiplied.
// With v_mfma_f32_32x32x2_f32 this product (45,824 x 128 x 768 at the bench's batch) is bound by the fp32 matrix pipe: 57 us of
// pipe time, 96 us measured; here it needs 6 x 1/16 of that and reads 141 MB of feature rows.
#ifndef PJ_ABL
#define PJ_ABL 0                         // lab (timing only): 2 weight planes written once, 3 the first two A chunks only, 4 = 2 + 3, 5 no A split
#endif
#ifndef PJ_ALLHALF
#define PJ_ALLHALF 0                     // 1: EVERY block is shared by the two waves of a SIMD (64 output columns each)
#endif
struct ProjArgs {
  const float* A; long lda; const int* a_idx;
  const float* W; long ldw;
  const float* bias;
  float* C; long ldc; const int* c_idx;
  long M;
  int K;
  int act;
};

#define PJ_KC 64                              // K chunk: its three bf16 planes are [4 column tiles][4 k steps][64 lanes][16 B] = 16 KB each
#define PJ_BUF (3 * 4 * 4 * 64 * 16)          // one plane set (48 KB); two of them alternate

__global__ __launch_bounds__(64 * SP_WAVES, 1) void gemm_split_proj_kernel(ProjArgs g, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int gw = (wave >> 2) * (gridDim.x * 4) + blockIdx.x * 4 + (wave & 3), nw = gridDim.x * SP_WAVES;
  const int n_it = PJ_ALLHALF ? (n_blocks + gridDim.x * 4 - 1) / (gridDim.x * 4) : (n_blocks + nw - 1) / nw;
  const int KC = g.K / PJ_KC;
  constexpr int PL = 4 * 4 * 64 * 16;                            // bytes of one plane of a chunk
  // Work items. A round hands out nw blocks of 32 rows: the first S = nw / 2 to waves 0-3 (one per SIMD), the rest to waves 4-7
  // (the SIMD's second wave). The two waves of a SIMD share its matrix pipe, so a workgroup with blocks on waves 4-7 walks its
  // K chunks at half the pace. When the last round has between S and 1.5 S blocks (the projector of the bench: 1,432 blocks, S =
  // 1,024), the blocks beyond S are split by COLUMNS between two waves 4-7 of neighbouring SIMDs (64 output columns each, both
  // read and split the same A rows): every SIMD then carries at most 1.5 blocks instead of 2.
  const int S = gridDim.x * 4;
  const int last_it = n_it - 1;
  const int rem = n_blocks - last_it * nw;                       // blocks of the last round: 1 .. nw
  const bool half_mode = rem > S && 2 * rem <= 3 * S;
  const int hs = blockIdx.x * 4 + (wave & 3);                    // half item of this wave (waves 4-7, last round, half mode)
#define PJ_ITEM(it_, blk_, jlo_, jhi_, valid_) do { \
    if (PJ_ALLHALF) { \
      blk_ = (it_) * S + blockIdx.x * 4 + (wave & 3); jlo_ = wave >> 2; jhi_ = (wave >> 2) + 1; valid_ = blk_ < n_blocks; \
    } else \
    if ((it_) == last_it && half_mode && wave >= 4) { \
      blk_ = last_it * nw + S + (hs >> 1); jlo_ = hs & 1; jhi_ = (hs & 1) + 1; valid_ = S + (hs >> 1) < rem; \
    } else { \
      blk_ = gw + (it_) * nw; jlo_ = 0; jhi_ = 2; valid_ = blk_ < n_blocks; \
    } \
  } while (0)

  // raw weight values of one chunk. A wave-instruction handles ONE operand fragment (column tile j, k step ks): lane L reads the
  // 8 values W(32 j + (L & 31), 16 ks + 8 (L >> 5) .. + 7) and, after the split, writes its 16 bytes at lane position L of the
  // fragment — consecutive lanes, consecutive LDS addresses (thread order along k makes 16 lanes write 512 bytes apart: a 16-way
  // bank conflict on every ds_write_b128). Wave w takes fragments 2 w, 2 w + 1 of the chunk's 16.
  float4 wraw[2][2];
  auto load_w = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = wave * 2 + i;
      const float* p = g.W + (long)((f >> 2) * 32 + l31) * g.ldw + (long)kc * PJ_KC + (f & 3) * 16 + half * 8;
      wraw[i][0] = *reinterpret_cast<const float4*>(p);
      wraw[i][1] = *reinterpret_cast<const float4*>(p + 4);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = wave * 2 + i;
      sp_u32x4 p0, p1, p2;
      sp_split8(wraw[i][0], wraw[i][1], p0, p1, p2);
      const int off = buf * PJ_BUF + (f * 64 + lane) * 16;
      *(sp_lds_u32x4*)(smem + off) = p0;
      *(sp_lds_u32x4*)(smem + PL + off) = p1;
      *(sp_lds_u32x4*)(smem + 2 * PL + off) = p2;
    }
  };
  // chunk q of a row: the 64 values k = 64 q .. 64 q + 63, per k step the 8 floats 16 s + 8 half .. + 7
  auto load_chunk = [&](const float* arow, int q, float4 (&raw)[4][2]) __attribute__((always_inline)) {
    const float* p = arow + q * PJ_KC + half * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      raw[s][0] = *reinterpret_cast<const float4*>(p + s * 16);
      raw[s][1] = *reinterpret_cast<const float4*>(p + s * 16 + 4);
    }
  };
  auto row_ptr = [&](int blk) -> const float* {
    long m = (long)blk * 32 + l31;
    if (m >= g.M) m = g.M - 1;                                   // rows past the end are computed on a valid row and never stored
    const long r = g.a_idx ? (long)g.a_idx[m] : m;
    return g.A + r * g.lda;
  };

  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = g.bias[j * 32 + l31];
  }
  sp_f32x16 acc[4];
  float4 r0[4][2], r1[4][2], r2[4][2];                           // three chunks of A rotate: two are in flight while one is multiplied
  auto mult_chunk = [&](int buf, const float4 (&raw)[4][2], int jlo, int jhi) __attribute__((always_inline)) {
    const unsigned char* wfrag = smem + buf * PJ_BUF + lane * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sp_u32x4 a0, a1, a2;
      if constexpr (PJ_ABL == 5) {                                 // lab (timing only): no split of the A values — their bits as they are
        a0 = __builtin_bit_cast(sp_u32x4, raw[s][0]);
        a1 = __builtin_bit_cast(sp_u32x4, raw[s][1]);
        a2 = a0 ^ a1;
      } else
      sp_split8(raw[s][0], raw[s][1], a0, a1, a2);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        if (jp < jlo || jp >= jhi) continue;                       // wave-uniform: a half item multiplies one pair of column tiles
        sp_u32x4 w[2][3];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            w[jj][p] = *(const sp_lds_u32x4*)(wfrag + p * PL + (((jp * 2 + jj) * 4 + s) * 64) * 16);
        acc[jp * 2 + 0] = sp_mfma(a2, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a2, w[1][0], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][2], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][2], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a1, w[0][1], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a1, w[1][1], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a1, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a1, w[1][0], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][1], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][1], acc[jp * 2 + 1]);
        acc[jp * 2 + 0] = sp_mfma(a0, w[0][0], acc[jp * 2 + 0]);
        acc[jp * 2 + 1] = sp_mfma(a0, w[1][0], acc[jp * 2 + 1]);
      }
    }
  };

  // Sequence of (row block, chunk) steps of this workgroup: step c = it * KC + kc uses plane set c & 1 and raw buffer c % 3.
  // While step c is multiplied the planes of step c + 1 are split and written into the other set (no MFMA waits for them: the
  // matrix pipe drains the queued MFMAs meanwhile), the raw weight values of step c + 2 and the A chunk of step c + 2 (of this
  // block or of the wave's next one) are loaded; ONE barrier per step publishes the planes of step c + 1 and retires the reads of
  // set c & 1 before it is written again two steps later.
  const int n_steps = n_it * KC;
  const float* ap0 = nullptr;                                    // row pointers of the wave's blocks with even / odd `it`
  const float* ap1 = nullptr;
  load_w(0);
  {
    int blk0, jl0, jh0;
    bool v0;
    PJ_ITEM(0, blk0, jl0, jh0, v0);
    if (v0) { ap0 = row_ptr(blk0); load_chunk(ap0, 0, r0); load_chunk(ap0, 1, r1); }
  }
  store_w(0);
  if (n_steps > 1) load_w(1 % KC);
  __syncthreads();

#define PJ_STEP(CC, cur, fill) do { \
    const int c = (CC); \
    if (c >= n_steps) break; \
    const int it = c / KC, kc = c - it * KC; \
    int blk, jlo, jhi; \
    bool valid; \
    PJ_ITEM(it, blk, jlo, jhi, valid); \
    if (kc == 0) { \
_Pragma("unroll") \
      for (int j = 0; j < 4; ++j) \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; \
    } \
    if (valid) { \
 \
      const int c2 = c + 2, it2 = c2 / KC, kc2 = c2 - it2 * KC; \
      int blk2, jlo2, jhi2; \
      bool valid2; \
      PJ_ITEM(it2, blk2, jlo2, jhi2, valid2); \
      if (PJ_ABL != 3 && PJ_ABL != 4 && c2 < n_steps && valid2) { \
        if (kc2 == 0) { if (it2 & 1) ap1 = row_ptr(blk2); else ap0 = row_ptr(blk2); } \
        load_chunk((it2 & 1) ? ap1 : ap0, kc2, fill); \
      } \
      mult_chunk(c & 1, cur, jlo, jhi); \
    } \
    if (PJ_ABL != 2 && PJ_ABL != 4 && c + 1 < n_steps) { \
      store_w((c + 1) & 1); \
      if (c + 2 < n_steps) load_w((kc + 2) % KC); \
    } \
    if (valid && kc == KC - 1) { \
 \
      const long m0 = (long)blk * 32; \
      const int rows_left = (int)(g.M - m0) - 4 * half; \
      long orow[16]; \
_Pragma("unroll") \
      for (int r = 0; r < 16; ++r) { \
        const int lr = (r & 3) + 8 * (r >> 2); \
        const long m = m0 + 4 * half + lr; \
        orow[r] = (g.c_idx && lr < rows_left) ? (long)g.c_idx[m] : m; \
      } \
_Pragma("unroll") \
      for (int j = 0; j < 4; ++j) { \
        if ((j >> 1) < jlo || (j >> 1) >= jhi) continue; \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) { \
          const int lr = (r & 3) + 8 * (r >> 2); \
          float v = acc[j][r] + bj[j]; \
          v = g.act == SBR_ACT_NONE ? v : (g.act == SBR_ACT_RELU ? sbr_relu(v) : sbr_act(v, g.act)); \
          if (lr < rows_left) g.C[orow[r] * g.ldc + j * 32 + l31] = v; \
        } \
      } \
    } \
    __syncthreads(); \
  } while (0)
#pragma unroll 1
  for (int c0 = 0; c0 < n_steps; c0 += 3) {
    PJ_STEP(c0, r0, r2);
    PJ_STEP(c0 + 1, r1, r0);
    PJ_STEP(c0 + 2, r2, r1);
  }
}

// 1 when sbr_gemm_split_proj_f32 takes this product: N = 128, K a multiple of 64 of at least 256
extern "C" int sbr_gemm_split_proj_supported(long M, int N, int K) { return M >= 1 && N == SP_N && K >= 2 * SP_K && K % PJ_KC == 0; }

// C[ci(m), 0..127] = act(A[ai(m), 0..K-1] x W^T + bias): A rows and W rows 16-byte aligned; a_idx / c_idx / bias may be NULL.
extern "C" int sbr_gemm_split_proj_f32(const float* A, long lda, const int* a_idx, const float* W, long ldw, const float* bias, float* C,
                                       long ldc, const int* c_idx, long M, int N, int K, int act, void* stream) {
  if (M == 0) return SBR_OK;
  SBR_REQUIRE(sbr_gemm_split_proj_supported(M, N, K), "sbr_gemm_split_proj_f32: shape %ld x %d x %d not supported (N = 128, K = 64 j >= 256)", M, N, K);
  SBR_REQUIRE(A && W && C, "sbr_gemm_split_proj_f32: null operand");
  SBR_REQUIRE(sp_al16(A, lda) && sp_al16(W, ldw), "sbr_gemm_split_proj_f32: operands must be 16-byte aligned");
  ProjArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.c_idx = c_idx; g.M = M; g.K = K; g.act = act;
  const int n_blocks = sbr_cdiv(M, 32);
  int grid = sbr_cdiv(n_blocks, 4);                              // one block per SIMD first: the second wave of a SIMD only adds
  if (grid > 256) grid = 256;                                    // matrix-pipe time to it (see the work-item note in the kernel)
  const size_t lds = 2 * PJ_BUF;
  static int attr_dev = -1;
  if (sbr_attr_stale(&attr_dev)) {
    if (hipFuncSetAttribute((const void*)gemm_split_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      sbr_set_error("sbr_gemm_split_proj_f32: cannot raise the dynamic LDS limit");
      return SBR_ERR_HIP;
    }
  }
  gemm_split_proj_kernel<<<grid, 64 * SP_WAVES, lds, (hipStream_t)stream>>>(g, n_blocks);
  SBR_CHECK_LAUNCH("sbr_gemm_split_proj_f32");
  return SBR_OK;
}
