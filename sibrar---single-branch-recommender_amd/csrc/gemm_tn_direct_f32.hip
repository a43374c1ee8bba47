// Weight-gradient products of the training step without LDS: slab[z][128][N] = sum over a K range of A[ak(k), 0..127]^T B[bk(k), n]
// (autograd of nn.Linear w.r.t. its weight, modules/polylinear.py:51 and sgd_alg.py:1279-1396: dW = dZ^T X, dZ [R, 128], X [R, N]
// possibly gathered by row). M = 128, N a multiple of 128, both operands row-major in k ("TN").
//
// v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand: A(row = lane % 32, k = lane / 32), B(k = lane / 32, col = lane % 32).
// For a TN product both operands are read along their rows, so a lane can load them straight from global memory in the layout the
// MFMA wants — no LDS staging, no barrier, no transposition:
//   * A: lane (r, kk) reads the float4 A[row(k0 + kk)][4 r .. 4 r + 3] — 32 lanes cover one 512-byte row of dZ, a wave-instruction two
//     rows. Component c of that float4 is the A operand of the MFMA that produces the output rows i = 4 r' + c (r' = MFMA row):
//     four MFMAs per k pair cover all 128 output rows (the row permutation costs nothing: it is undone in the slab store);
//   * B the same way: lane (r, kk) reads the float4 B[row(k0 + kk)][col0 + 4 r .. + 3]; component d gives the output columns 4 r'' + d.
// A workgroup owns one 128-column block of the output and one K range; each of its four waves (one per SIMD) takes a quarter of the
// range and keeps the WHOLE 128 x 128 partial sum in 16 x 16 accumulator registers: 16 independent MFMAs (1,024 matrix-pipe cycles)
// per two 16-byte loads, every operand byte fetched once. Four 8-row steps of operands rotate in registers (three in flight
// while one is multiplied). The four quarter sums are added through LDS at the end (fixed order: bitwise
// reproducible), and ONE slab per workgroup is written with 16-byte stores: 256 slabs of 64 KB for a 128 x 128 product, 42 x 6 for 128 x 768 —
// a third of what the ring kernel's (tile x K range) items leave behind, which is what the slab reducer then reads.
// Gather indices are wave-uniform (16 consecutive k per step): scalar loads through the constant address space.
#include "gemm_args.h"

typedef float td_f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) int* td_idx_ptr;

struct TnDirectArgs {
  const float* A; long lda; const int* a_idx;
  const float* B; long ldb; const int* b_idx;
  int N, K;
  float* slab;                 // [splits][128][N]
  int k_chunk;                 // rows per workgroup (multiple of 32)
  int ncb;                     // N / 128
};

#ifndef TD_ABL
#define TD_ABL 0
#endif
#define TD_STEP 8               // k rows per pipeline step of a wave (TD_PAIRS MFMA k pairs: 64 MFMAs, 4,096 matrix-pipe cycles)
#define TD_PAIRS (TD_STEP / 2)

template <bool GA, bool GB>
__global__ __launch_bounds__(256, 1) void gemm_tn_direct_kernel(TnDirectArgs g) {
  extern __shared__ __attribute__((aligned(16))) float4 td_lds[];         // one 128 x 128 tile in register order: 64 KB
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kk = lane >> 5;
  const int cb = blockIdx.x % g.ncb, z = blockIdx.x / g.ncb;
  const int kbeg_wg = z * g.k_chunk;
  const int kend_wg = kbeg_wg + g.k_chunk < g.K ? kbeg_wg + g.k_chunk : g.K;
  const int kq = g.k_chunk >> 2;                                           // rows per wave: multiple of TD_STEP
  const int kbeg = kbeg_wg + wave * kq;
  const int kend = kbeg + kq < kend_wg ? kbeg + kq : kend_wg;
  const long colB = (long)cb * 128 + 4 * r;

  // acc[4 c + d]: output rows 4 r' + c (r' = MFMA row), columns 4 r'' + d (r'' = MFMA column)
  td_f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // One pipeline step = the operands of the k pairs k0 .. k0 + TD_STEP - 1. Loads are UNCONDITIONAL (rows past the end of the
  // operands are clamped to the last row, their products are masked out in `mult`): a load inside a branch makes hipcc wait with
  // vmcnt(0) in front of the MFMAs — i.e. for the prefetch it has just issued — and the step's memory latency lands on every step.
  const int klast = g.K - 1;
  auto load = [&](int k0, float4 (&a)[TD_PAIRS], float4 (&b)[TD_PAIRS]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < TD_PAIRS; ++p) {
      const int ka = k0 + 2 * p;                               // wave-uniform
      const int k0c = ka < klast ? ka : klast, k1c = ka + 1 < klast ? ka + 1 : klast;
      long ra0 = k0c, ra1 = k1c, rb0 = k0c, rb1 = k1c;
      if constexpr (GA) { ra0 = ((td_idx_ptr)g.a_idx)[k0c]; ra1 = ((td_idx_ptr)g.a_idx)[k1c]; }
      if constexpr (GB) { rb0 = ((td_idx_ptr)g.b_idx)[k0c]; rb1 = ((td_idx_ptr)g.b_idx)[k1c]; }
      const long ra = kk ? ra1 : ra0, rb = kk ? rb1 : rb0;
#if TD_ABL == 1
      if (k0 != kbeg) continue;                                // lab: operands of the first step only (pure MFMA loop)
#endif
      a[p] = *reinterpret_cast<const float4*>(g.A + ra * g.lda + 4 * r);
      b[p] = *reinterpret_cast<const float4*>(g.B + rb * g.ldb + colB);
    }
  };
  auto mult = [&](int k0, const float4 (&a)[TD_PAIRS], const float4 (&b)[TD_PAIRS]) __attribute__((always_inline)) {
    const bool tail = k0 + TD_STEP > kend;                     // wave-uniform: some rows of this step lie past the range
#pragma unroll
    for (int p = 0; p < TD_PAIRS; ++p) {
      float av[4] = {a[p].x, a[p].y, a[p].z, a[p].w};
      const float bv[4] = {b[p].x, b[p].y, b[p].z, b[p].w};
      if (tail) {
        const bool valid = k0 + 2 * p + kk < kend;
#pragma unroll
        for (int c = 0; c < 4; ++c) av[c] = valid ? av[c] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[4 * c + d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[d], acc[4 * c + d], 0, 0, 0);
    }
  };

  // Four operand steps rotate in registers: three are in flight while one is multiplied. The product is close to its HBM bound
  // (128 x 128 over 90,112 rows: 92 MB against 18.8 us of matrix-pipe time = 4.9 TB/s), so by Little's law the chip needs ~10 MB
  // = 38 KB per CU in flight; one step per wave is 8 KB, 32 KB per CU with nothing else to hide the latency behind (one wave per
  // SIMD): measured 47.8 us with one step ahead.
  float4 a0[TD_PAIRS], a1[TD_PAIRS], a2[TD_PAIRS], a3[TD_PAIRS], b0[TD_PAIRS], b1[TD_PAIRS], b2[TD_PAIRS], b3[TD_PAIRS];
  if (kbeg < kend) {
    load(kbeg, a0, b0);
    load(kbeg + TD_STEP, a1, b1);
    load(kbeg + 2 * TD_STEP, a2, b2);
#pragma unroll 1
    for (int k0 = kbeg; k0 < kend; k0 += 4 * TD_STEP) {
      // every mult is unconditional (a step past the range multiplies zeros): under a branch LLVM sinks the loads of its operands
      // into the branch, next to their use, and the prefetch is gone
      load(k0 + 3 * TD_STEP, a3, b3);
      __builtin_amdgcn_sched_barrier(0);
      mult(k0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load(k0 + 4 * TD_STEP, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mult(k0 + TD_STEP, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      load(k0 + 5 * TD_STEP, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mult(k0 + 2 * TD_STEP, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      load(k0 + 6 * TD_STEP, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      mult(k0 + 3 * TD_STEP, a3, b3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- waves 1, 2, 3 hand their tiles to wave 0 through LDS, one after the other (fixed order: bitwise reproducible); entry
  // (c, q) of a lane = the float4 over d. Then one slab per workgroup.
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          td_lds[(c * 16 + q) * 64 + lane] = make_float4(acc[4 * c][q], acc[4 * c + 1][q], acc[4 * c + 2][q], acc[4 * c + 3][q]);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float4 v = td_lds[(c * 16 + q) * 64 + lane];
          acc[4 * c][q] += v.x; acc[4 * c + 1][q] += v.y; acc[4 * c + 2][q] += v.z; acc[4 * c + 3][q] += v.w;
        }
    }
    __syncthreads();
  }
  if (wave == 0) {
    // accumulator register q of MFMA (c, d), lane (r, kk): output row 4 ((q & 3) + 8 (q >> 2) + 4 kk) + c, column col0 + 4 r + d
    float* out = g.slab + ((long)z * 128) * g.N + colB;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 4 * ((q & 3) + 8 * (q >> 2) + 4 * kk) + c;
        *reinterpret_cast<float4*>(out + (long)i * g.N) = make_float4(acc[4 * c][q], acc[4 * c + 1][q], acc[4 * c + 2][q], acc[4 * c + 3][q]);
      }
  }
}

static bool td_al16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// K ranges (= slabs) the direct kernel would use for this shape, 0 when it does not take it: M = 128, N = 128 j, long K, A rows
// 16-byte aligned. One workgroup per (128-column block, K range), about one per CU.
int sbr_tn_direct_splits(const float* A, long lda, int M, int N, int K) {
  // OPT-IN (SBR_TN_DIRECT=1): measured slower than the ring kernel on the step's shapes — 128 x 128 over 90,112 rows 49 us
  // against 36 us, 128 x 768 over 45,824 gathered rows 111 us against 99 us. With the loads taken out of the loop the same
  // kernel still needs 46 / 92 us: 16 independent fp32 MFMAs back to back on every SIMD of the chip run at ~75 % of the nominal
  // rate (clock), the LDS hand-over of three 64 KB tiles plus accumulator moves cost ~8 us per workgroup, and the remaining
  // 20 us of the large product are HBM latency that three steps of prefetch (24 KB per wave) do not cover. Kept for the record
  // and for its test; the ring kernel stays the product path.
  if (!(getenv("SBR_TN_DIRECT") && atoi(getenv("SBR_TN_DIRECT")) == 1)) return 0;
  if (M != 128 || N < 128 || N % 128 != 0 || K < 8192 || (A && !td_al16(A, lda))) return 0;      // (A == NULL: shape query; B is checked at the launch)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  const int ncb = N / 128;
  int splits = n_cu / ncb;
  if (splits < 1) splits = 1;
  int k_chunk = sbr_cdiv(sbr_cdiv(K, splits), 4 * TD_STEP) * (4 * TD_STEP);
  return sbr_cdiv(K, k_chunk);
}

// slab[z][128][N] for z < sbr_tn_direct_splits(...) (plain stores). Returns -1 when the shape is not eligible.
int sbr_tn_direct_launch(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M, int N, int K,
                         float* slab, int* splits_out, hipStream_t s) {
  const int splits = sbr_tn_direct_splits(A, lda, M, N, K);
  if (splits <= 0) return -1;
  TnDirectArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.B = B; g.ldb = ldb; g.b_idx = b_idx; g.N = N; g.K = K; g.slab = slab;
  g.ncb = N / 128;
  if (!td_al16(B, ldb)) return -1;
  g.k_chunk = sbr_cdiv(sbr_cdiv(K, splits), 4 * TD_STEP) * (4 * TD_STEP);
  const size_t lds = 4 * 16 * 64 * sizeof(float4);
#define TD_LAUNCH(GA, GB)                                                                                                 \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)gemm_tn_direct_kernel<GA, GB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        sbr_set_error("sbr_gemm_tn_f32: cannot raise the dynamic LDS limit of the direct kernel");                         \
        return SBR_ERR_HIP;                                                                                                \
      }                                                                                                                    \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    gemm_tn_direct_kernel<GA, GB><<<g.ncb * splits, 256, lds, s>>>(g);                                                     \
  } while (0)
  if (a_idx && b_idx) TD_LAUNCH(true, true);
  else if (a_idx) TD_LAUNCH(true, false);
  else if (b_idx) TD_LAUNCH(false, true);
  else TD_LAUNCH(false, false);
#undef TD_LAUNCH
  SBR_CHECK_LAUNCH("sbr_gemm_tn_f32 (direct)");
  *splits_out = splits;
  return SBR_OK;
}
