// fp32 GEMM on the bf16 matrix pipe for WIDE layers: C[ci(m), N] = act(A[ai(m), K] x W + bias) with N a multiple of 256 and K a
// multiple of 32 — the shared single-branch network and the modality projectors of the larger configurations (modules/polylinear.py:51
// with hidden widths 256 / 512: conf/single/algorithms/sbnet_onion18_huge_no-user_conf.yml:39-54; BASELINE configs[3] with C = D = 256),
// forward (NT: W is [N][K], nn.Linear's layout) and input gradient (NN: dX = dZ W with W [K][N], the same buffer read the other way).
//
// Arithmetic as in gemm_split_f32.hip: both fp32 operands are split exactly into three bf16 numbers, the six leading partial
// products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (dropped terms < 2^-23 of each product). What differs from the
// N = 128 kernels is the shape of the work: a workgroup (8 waves) owns a tile of 256 rows x 256 columns, every wave one 32-row block
// of it with all 256 columns (8 accumulator tiles = 128 registers). The split of an A element (4.5 vector instructions) now feeds
// 256 output columns instead of 128, and the weight planes of a K chunk (32 deep: 3 x 16 KB per set, two sets alternate) are built
// once per 256 rows: per chunk a wave issues 96 MFMAs against ~90 instructions of operand splitting — the N = 128 projector kernel
// has 48 against the same ~90 and sits at a quarter of the matrix pipe. K is walked in chunks of 32 with ONE barrier per chunk; the
// raw weight values of chunk c + 2 and the A chunk of step c + 1 are in flight while chunk c is multiplied; consecutive tiles of a
// workgroup continue the same pipeline (no drain between tiles). Tiles are numbered so that the column blocks of one row group run
// on neighbouring workgroups at the same time (the A rows are read from HBM once and from L2 by the others).
// Non-finite operands: as in gemm_split_f32.hip (NaN where the fp32 pipe gives +-inf; the same outputs are poisoned).
#include "gemm_split_common.h"

#define SW_WAVES 8
#define SW_N 256                              // output columns of a workgroup tile
#define SW_KC 32                              // K chunk
#define SW_PL (8 * 2 * 64 * 16)               // bytes of one bf16 plane of a chunk: [8 column tiles][2 k steps][64 lanes][16 B]
#define SW_BUF (3 * SW_PL)                    // one plane set (48 KB)

struct WideArgs {
  const float* A; long lda; const int* a_idx;
  const float* W; long ldw;
  const float* bias;
  float* C; long ldc; const int* c_idx;
  long M;
  int N, K;
  int act;
  int w_kn;                                   // 0: W is [N][K] (NT); 1: W is [K][N] (NN)
};

__global__ __launch_bounds__(64 * SW_WAVES, 1) void gemm_split_wide_kernel(WideArgs g, int n_groups, int n_cb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int KC = g.K / SW_KC;
  const int n_items = n_groups * n_cb;
  const int my_items = ((int)blockIdx.x < n_items) ? (n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int n_steps = my_items * KC;
  if (n_steps == 0) return;
  // item q of this workgroup: tile number blockIdx.x + q * gridDim.x -> (row group, column block)
  auto item_of = [&](int q, int& grp, int& cb) {
    const int it = (int)blockIdx.x + q * (int)gridDim.x;
    grp = it / n_cb;
    cb = it - grp * n_cb;
  };

  // raw weight values of one chunk (256 columns x 32 k), 16 floats per thread, loaded COALESCED in both layouts:
  //   NN (W [K][N]): wave w takes fragments 2 w, 2 w + 1 of the chunk's 16 (fragment f: column tile f >> 1, k step f & 1); lane L holds
  //       the 8 values W(k = 32 kc + 16 (f & 1) + 8 (L >> 5) .. + 7, col = 256 cb + 32 (f >> 1) + (L & 31)): 32 consecutive columns per
  //       load instruction; after the split the lane writes its 16 bytes at lane position L of the fragment (ds_write_b128).
  //   NT (W [N][K]): a row of the chunk is 128 contiguous bytes; thread t takes the float4 at k = 4 (t & 7) .. + 3 of the rows
  //       (t >> 3) + 64 i, i < 4: a wave-instruction reads 8 whole row chunks. Its four bf16 values per plane are HALF of a fragment
  //       slot (slot = lane position (row & 31) + 32 ((t >> 1) & 1) of fragment (row >> 5, (t >> 2) & 1), half t & 1): ds_write_b64.
  //       (Lanes = rows, each reading 32 bytes of its own 2 KB-strided row, made every load instruction touch 64 cache lines: the NT
  //       product ran 25 % slower than the NN product of the same shape.)
  float4 wraw[2][2];
  auto load_w = [&](int cb, int kc) {
    if (!g.w_kn) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long col = (long)cb * SW_N + (t >> 3) + 64 * i;
        const float* p = g.W + col * g.ldw + (long)kc * SW_KC + (t & 7) * 4;
        wraw[i >> 1][i & 1] = *reinterpret_cast<const float4*>(p);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = wave * 2 + i;
      const long col = (long)cb * SW_N + (f >> 1) * 32 + l31;
      const long k0 = (long)kc * SW_KC + (f & 1) * 16 + half * 8;
      const float* p = g.W + k0 * g.ldw + col;
      wraw[i][0] = make_float4(p[0], p[g.ldw], p[2 * g.ldw], p[3 * g.ldw]);
      wraw[i][1] = make_float4(p[4 * g.ldw], p[5 * g.ldw], p[6 * g.ldw], p[7 * g.ldw]);
    }
  };
  auto store_w = [&](int buf) {
    if (!g.w_kn) {
      typedef unsigned sw_u32x2 __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(3))) sw_u32x2 sw_lds_u32x2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = wraw[i >> 1][i & 1];
        unsigned a0, a1, a2, b0, b1, b2;
        sp_split2(v.x, v.y, a0, a1, a2);
        sp_split2(v.z, v.w, b0, b1, b2);
        const int row = (t >> 3) + 64 * i;
        const int frag = (row >> 5) * 2 + ((t >> 2) & 1);
        const int off = buf * SW_BUF + (frag * 64 + (row & 31) + 32 * ((t >> 1) & 1)) * 16 + (t & 1) * 8;
        sw_u32x2 q0 = {a0, b0}, q1 = {a1, b1}, q2 = {a2, b2};
        *(sw_lds_u32x2*)(smem + off) = q0;
        *(sw_lds_u32x2*)(smem + SW_PL + off) = q1;
        *(sw_lds_u32x2*)(smem + 2 * SW_PL + off) = q2;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = wave * 2 + i;
      sp_u32x4 p0, p1, p2;
      sp_split8(wraw[i][0], wraw[i][1], p0, p1, p2);
      const int off = buf * SW_BUF + (f * 64 + lane) * 16;
      *(sp_lds_u32x4*)(smem + off) = p0;
      *(sp_lds_u32x4*)(smem + SW_PL + off) = p1;
      *(sp_lds_u32x4*)(smem + 2 * SW_PL + off) = p2;
    }
  };
  // chunk kc of the wave's row: per k step the 8 floats k = 32 kc + 16 s + 8 half .. + 7
  auto load_chunk = [&](const float* arow, int kc, float4 (&raw)[2][2]) __attribute__((always_inline)) {
    const float* p = arow + kc * SW_KC + half * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      raw[s][0] = *reinterpret_cast<const float4*>(p + s * 16);
      raw[s][1] = *reinterpret_cast<const float4*>(p + s * 16 + 4);
    }
  };
  auto row_ptr = [&](int grp) -> const float* {
    long m = ((long)grp * SW_WAVES + wave) * 32 + l31;
    if (m >= g.M) m = g.M - 1;                                   // rows past the end are computed on a valid row and never stored
    const long r = g.a_idx ? (long)g.a_idx[m] : m;
    return g.A + r * g.lda;
  };

  sp_f32x16 acc[8];
  float4 r0[2][2];                                               // the wave's A chunk: refilled for the next step as soon as both k steps are split
  // One step = 16 units (k step s, column tile j) of 6 MFMAs on one accumulator tile. The three weight fragments of unit u + 1 are
  // requested BEFORE the MFMAs of unit u are issued and land in a second register set: left to itself hipcc issues every read one
  // or two instructions ahead of the MFMA that needs it (no registers to hoist into at 256) and the wave sits out the LDS latency
  // ~30 times per step (PMC: half of the wave cycles in s_waitcnt, matrix pipe 38 % busy). (Two tiles per unit — two independent
  // accumulator chains — would need 48 fragment registers: spills.)
  auto mult_chunk = [&](int buf, float4 (&raw)[2][2], const float* next_row, int next_kc) __attribute__((always_inline)) {
    const unsigned char* wfrag = smem + buf * SW_BUF + lane * 16;
    sp_u32x4 w[2][3];                                            // [set][plane]
    auto fetch = [&](int set, int s, int j) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) w[set][p] = *(const sp_lds_u32x4*)(wfrag + p * SW_PL + ((j * 2 + s) * 64) * 16);
    };
    fetch(0, 0, 0);
    sp_u32x4 a0, a1, a2;
    sp_split8(raw[0][0], raw[0][1], a0, a1, a2);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = u & 7, set = u & 1;
      if (u + 1 < 16) fetch(set ^ 1, (u + 1) >> 3, (u + 1) & 7);
      __builtin_amdgcn_sched_barrier(0);                         // the reads of unit u + 1 stay in front of the MFMAs of unit u
      // smallest terms first
      acc[j] = sp_mfma(a2, w[set][0], acc[j]);
      acc[j] = sp_mfma(a0, w[set][2], acc[j]);
      acc[j] = sp_mfma(a1, w[set][1], acc[j]);
      acc[j] = sp_mfma(a1, w[set][0], acc[j]);
      acc[j] = sp_mfma(a0, w[set][1], acc[j]);
      acc[j] = sp_mfma(a0, w[set][0], acc[j]);
      if (u == 7) {
        sp_split8(raw[1][0], raw[1][1], a0, a1, a2);             // the operands of k step 1 (behind the last MFMAs of k step 0)
        if (next_row) load_chunk(next_row, next_kc, raw);        // both k steps are split: the A chunk of the next step takes the registers
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Step c = q * KC + kc (item q of this workgroup, chunk kc) uses plane set c & 1 and raw A buffer c & 1. While step c is multiplied
  // the planes of step c + 1 are split and written into the other set, the raw weight values of step c + 2 and the A chunk of step
  // c + 1 are loaded; ONE barrier per step publishes the planes of step c + 1 and retires the reads of set c & 1.
  int grp_c, cb_c;                                               // item of the current step
  item_of(0, grp_c, cb_c);
  const float* ap = row_ptr(grp_c);
  load_w(cb_c, 0);
  load_chunk(ap, 0, r0);
  store_w(0);
  if (n_steps > 1) {
    int g1 = grp_c, c1 = cb_c;
    if (KC == 1) item_of(1, g1, c1);
    load_w(c1, 1 % KC);
  }
  __syncthreads();

#define SW_STEP(CC) do { \
    const int c = (CC); \
    if (c >= n_steps) break; \
    const int q = c / KC, kc = c - q * KC; \
    if (kc == 0) { \
      item_of(q, grp_c, cb_c); \
_Pragma("unroll") \
      for (int j = 0; j < 8; ++j) \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; \
    } \
    const float* nrow = nullptr; \
    int nkc = 0; \
    if (c + 1 < n_steps) {                                        /* A chunk of step c + 1 (of this tile or of the next one) */ \
      const int q1 = (c + 1) / KC, kc1 = (c + 1) - q1 * KC; \
      if (kc1 == 0) { int g1, c1; item_of(q1, g1, c1); ap = row_ptr(g1); } \
      nrow = ap; nkc = kc1; \
    } \
    mult_chunk(c & 1, r0, nrow, nkc); \
    if (c + 1 < n_steps) { \
      store_w((c + 1) & 1); \
      if (c + 2 < n_steps) { \
        const int q2 = (c + 2) / KC, kc2 = (c + 2) - q2 * KC; \
        int g2, c2; \
        item_of(q2, g2, c2); \
        load_w(c2, kc2); \
      } \
    } \
    if (kc == KC - 1) { \
      const long m0 = ((long)grp_c * SW_WAVES + wave) * 32; \
      const int rows_left = (int)(g.M - m0) - 4 * half; \
      float bj[8]; \
_Pragma("unroll") \
      for (int j = 0; j < 8; ++j) bj[j] = g.bias ? g.bias[(long)cb_c * SW_N + j * 32 + l31] : 0.f; \
_Pragma("unroll") \
      for (int r = 0; r < 16; ++r) { \
        const int lr = (r & 3) + 8 * (r >> 2); \
        if (lr < rows_left) { \
          const long m = m0 + 4 * half + lr; \
          const long orow = g.c_idx ? (long)g.c_idx[m] : m; \
          float* cp = g.C + orow * g.ldc + (long)cb_c * SW_N + l31; \
_Pragma("unroll") \
          for (int j = 0; j < 8; ++j) { \
            float v = acc[j][r] + bj[j]; \
            v = g.act == SBR_ACT_NONE ? v : (g.act == SBR_ACT_RELU ? sbr_relu(v) : sbr_act(v, g.act)); \
            cp[j * 32] = v; \
          } \
        } \
      } \
    } \
    __syncthreads(); \
  } while (0)
#pragma unroll 1
  for (int c0 = 0; c0 < n_steps; c0 += 2) {
    SW_STEP(c0);
    SW_STEP(c0 + 1);
  }
}

static bool sw_al16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// 1 when sbr_gemm_split_wide_f32 takes this product: N a multiple of 256, K a multiple of 32 of at least 64
extern "C" int sbr_gemm_split_wide_supported(long M, int N, int K) { return M >= 1 && N >= SW_N && N % SW_N == 0 && K >= 2 * SW_KC && K % SW_KC == 0; }

// mode 0 (NT): C[ci(m), 0..N-1] = act(A[ai(m), 0..K-1] x W^T + bias), W [N][K]; mode 1 (NN): the same with W [K][N] (C = A W).
// A rows 16-byte aligned (lda % 4 == 0), W rows 16-byte aligned in mode 0; a_idx / c_idx / bias may be NULL.
extern "C" int sbr_gemm_split_wide_f32(int mode, const float* A, long lda, const int* a_idx, const float* W, long ldw, const float* bias,
                                       float* C, long ldc, const int* c_idx, long M, int N, int K, int act, void* stream) {
  SBR_REQUIRE(mode == 0 || mode == 1, "sbr_gemm_split_wide_f32: mode %d", mode);
  if (M == 0) return SBR_OK;
  SBR_REQUIRE(sbr_gemm_split_wide_supported(M, N, K), "sbr_gemm_split_wide_f32: shape %ld x %d x %d not supported (N = 256 i, K = 32 j >= 64)", M, N, K);
  SBR_REQUIRE(A && W && C, "sbr_gemm_split_wide_f32: null operand");
  SBR_REQUIRE(sw_al16(A, lda) && (mode == 1 || sw_al16(W, ldw)), "sbr_gemm_split_wide_f32: operands must be 16-byte aligned");
  WideArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.c_idx = c_idx;
  g.M = M; g.N = N; g.K = K; g.act = act; g.w_kn = mode;
  const int n_groups = sbr_cdiv(M, 32 * SW_WAVES), n_cb = N / SW_N;
  int grid = n_groups * n_cb;
  if (grid > 256) grid = 256;
  const size_t lds = 2 * SW_BUF;
  static int attr_dev = -1;
  if (sbr_attr_stale(&attr_dev)) {
    if (hipFuncSetAttribute((const void*)gemm_split_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      sbr_set_error("sbr_gemm_split_wide_f32: cannot raise the dynamic LDS limit");
      return SBR_ERR_HIP;
    }
  }
  gemm_split_wide_kernel<<<grid, 64 * SW_WAVES, lds, (hipStream_t)stream>>>(g, n_groups, n_cb);
  SBR_CHECK_LAUNCH("sbr_gemm_split_wide_f32");
  return SBR_OK;
}
