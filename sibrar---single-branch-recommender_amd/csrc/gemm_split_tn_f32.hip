// Weight-gradient products of the training step on the bf16 matrix pipe:
//     slab[z][M][N] = sum over the rows r of K range z of  A[ak(r), 0..M-1]^T  B[bk(r), 0..N-1]
// (autograd of nn.Linear w.r.t. its weight, modules/polylinear.py:51 and sgd_alg.py:1279-1396: dW = dZ^T X with dZ [R, 128] and
// X [R, N], either operand gathered by row). M and N multiples of 128 (M = 128 in the c2 step, 512 in c3), the reduction runs over
// the ROWS of both operands ("TN").
//
// Arithmetic: as in gemm_split_f32.hip — both fp32 operands are split exactly into three bf16 numbers, the six leading partial
// products go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped terms are < 2^-23 of each product).  With
// v_mfma_f32_32x32x2_f32 these products are what was left on the fp32 pipe: 19 / 57 us of pipe time for 128 x 128 over 90,112 rows
// and 128 x 768 over 45,824 rows (36 / 99 us measured with the ring kernel); here they need 6/16 of that and the kernel is bound by
// the operand stream (92 / 164 MB).
//
// The MFMA wants, per lane, 8 CONSECUTIVE k of one output row / column — for a TN product that is a column walk through a row-major
// operand.  The transposition happens in the loads: a thread owns two adjacent operand columns and reads 8 consecutive rows of them
// with 8 dwordx2 loads (a wave-instruction covers one whole 512-byte row of the operand's 128 columns; the row pointer — gathered or
// not — is wave-uniform and lives in SGPRs), splits the 2 x 8 values and writes the three bf16 planes of each column with one 16-byte
// LDS store, at the lane position of the MFMA fragment it belongs to.  Even columns fill fragment tiles 0-1, odd columns tiles 2-3
// (tile T, position i <-> column 64 (T & 1) + 2 i + (T >> 1)): stores and fragment reads (plain ds_read_b128) are conflict-free, and a
// wave that multiplies tiles T and T + 2 holds adjacent output columns in its two accumulator tiles (8-byte slab stores).
//
// Work: one workgroup of 8 waves per CU owns a 128 x 128 output tile (column block j of N) over one K range z; it walks the range
// in chunks of 32 rows (two MFMA k steps).  Both operands of a chunk are 6 planes of 8 KB (48 KB); two plane sets alternate: while
// chunk c is multiplied, chunk c + 1 is split and written into the other set and the raw values of chunks c + 2 .. c + 4 are in
// flight (TS_NBUF = 3 raw buffers of 16 registers rotate; 4 measured slower); ONE barrier per chunk.  Waves 0-3 take k step 0 of every chunk, waves 4-7
// k step 1; within a group each wave owns a 64 x 64 quarter of the tile (4 accumulator tiles, 12 fragment reads per 24 MFMAs).
// The two groups' sums are added through LDS at the end in a fixed order and ONE slab per workgroup is stored (16 bytes per lane,
// all eight waves): 256 slabs of 64 KB
// for a 128 x 128 product, 42 x 6 for 128 x 768 — the ring kernel left 512 / 1536 partial tiles for the slab reducer.
// Row pointers of the range (gathered or not; the pointer of a zero row past the end of K) are staged in LDS once.
// Column blocks of the same K range run on the same XCD (its L2 then serves the re-reads of dZ).
#include "gemm_split_common.h"

#ifndef TS_ABL
#define TS_ABL 0                  // lab (timing only): 1 no global loads in the loop, 2 no split / plane stores in the loop, 3 = 1 + 2, 4 no loop at all (set-up, prologue and epilogue only)
#endif
#ifndef TS_SALU
#define TS_SALU 0                 // 0: row pointer lists staged in LDS; 1 (experiment, measured SLOWER: 29.8 / 73.5 us against 26.4 / 64.4 us on the
                                  // same box): row pointers by SALU arithmetic + scalar index loads one step ahead — the set-up loses its
                                  // memory round trip (7.6 instead of 8.7 us without any chunk), but the scalar loads share lgkmcnt with the
                                  // LDS traffic: every fragment / barrier wait of the next step also waits for them
#endif
#define TS_KC 32                  // rows per chunk
#define TS_PL (4 * 2 * 64 * 16)   // one bf16 plane of one operand of a chunk: [4 tiles of 32 columns][2 k steps][64 lanes][16 B] = 8 KB
#define TS_OP (3 * TS_PL)         // the three planes of an operand
#define TS_BUF (2 * TS_OP)        // a plane set: A then B (48 KB)
#ifndef TS_NBUF
#define TS_NBUF 3                 // raw chunk buffers (3 or 4): chunks c + 2 .. c + TS_NBUF + 1 are in flight while chunk c is multiplied
#endif
#define TS_PAD (TS_NBUF == 4 ? 8 : 9)   // chunks of zero-row padding behind a range: 3 (5) of the rounded-up loop + 5 (4) of prefetch
#define TS_MAXROWS (2048 + TS_PAD * TS_KC)   // rows of a K range + padding (the two row-pointer lists are staged in LDS: 37 KB)

struct TnSplitArgs {
  const float* A; long lda; const int* a_idx;
  const float* B; long ldb; const int* b_idx;
  float* slab;                    // [nz][M][N]
  int M, N, K;
  int nz, nj, nm;                 // K ranges, column blocks (N / 128), row blocks (M / 128)
  int chunks;                     // ceil(K / 32)
};

__device__ __attribute__((aligned(256))) float ts_zero_row[128];
typedef const __attribute__((address_space(1))) sp_f32x2* ts_gptr;      // row pointers come back from LDS as integers: say that they are global

// WIDE: more than one 128 x 128 output tile per K range. The two forms differ only in the workgroup map — and in their symbol,
// which keeps the step's 128 x 128 and 128 x 768 products apart in a kernel trace.
// bid / W: this workgroup's index among the W (a multiple of 8) workgroups of its product — the whole grid of
// gemm_split_tn_kernel
template <bool WIDE>
__device__ __forceinline__ void ts_body(const TnSplitArgs& g, const int bid, const int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // workgroup -> (K range z, column block j): consecutive logical indices share z; the logical index is contiguous per XCD
  const int L = (bid & 7) * (W >> 3) + (bid >> 3);
  const int tiles = WIDE ? g.nj * g.nm : 1;                        // 128 x 128 output tiles per K range: consecutive L share z
  if (L >= g.nz * tiles) return;
  const int z = WIDE ? L / tiles : L, tile = WIDE ? L - z * tiles : 0;
  const int mi = WIDE ? tile / g.nj : 0, j = WIDE ? tile - mi * g.nj : 0;
  const int cb = g.chunks / g.nz, cr = g.chunks - cb * g.nz;
  const int c_begin = z * cb + (z < cr ? z : cr);
  const int n_chunks = cb + (z < cr ? 1 : 0);                      // >= 1 (host: nz <= chunks), <= TS_MAXROWS / 32 - TS_PAD
  const long row0 = (long)c_begin * TS_KC;

#if !TS_SALU
  // ---- row pointers of the range (operand B: of column block j), the zero row for rows past the end
  unsigned long long* ptrs = reinterpret_cast<unsigned long long*>(smem + 2 * TS_BUF);            // [2][TS_MAXROWS]
  {
    // all index loads first (rows clamped into the operand), then the pointer stores: one memory round trip for the whole list
    constexpr int PER = (TS_MAXROWS + 511) / 512;
    const int n_list = (n_chunks + TS_PAD) * TS_KC, n_in = n_chunks * TS_KC;
    long ra[PER], rb[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const long row = row0 + t + 512 * q;
      ra[q] = rb[q] = row > g.K - 1 ? g.K - 1 : row;
    }
    if (g.a_idx) {
#pragma unroll
      for (int q = 0; q < PER; ++q) ra[q] = g.a_idx[ra[q]];
    }
    if (g.b_idx) {
#pragma unroll
      for (int q = 0; q < PER; ++q) rb[q] = g.b_idx[rb[q]];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int r = t + 512 * q;
      const bool in = r < n_in && row0 + r < g.K;
      if (r < n_list) {
        ptrs[r] = (unsigned long long)(in ? g.A + ra[q] * g.lda + (long)mi * 128 : ts_zero_row);
        ptrs[TS_MAXROWS + r] = (unsigned long long)(in ? g.B + rb[q] * g.ldb + (long)j * 128 : ts_zero_row);
      }
    }
  }
  __syncthreads();
#endif

  // ---- loader role: waves 0-3 operand A, waves 4-7 operand B; wave & 3 = the group of 8 rows of the chunk; a thread owns the
  // columns 2 lane and 2 lane + 1 of its operand's 128
  const int op = wave >> 2, rg = wave & 3;
  // destination of the split values: fragment (tile = half [+ 2 for the odd column], k step = rg >> 1), lane position l31 + 32 (rg & 1)
  const int st_off = op * TS_OP + ((((half) * 2 + (rg >> 1)) * 64) + l31 + 32 * (rg & 1)) * 16;

  sp_f32x2 r0[8], r1[8], r2[8], r3[8];
#if TS_SALU
  // Row pointers without LDS or VALU: the wave's eight rows of a chunk are wave-uniform, so their pointers are SALU arithmetic on
  // SGPRs — base + row * ld for a plain operand, base + idx[row] * ld with the indices of the NEXT chunk to request fetched by
  // scalar loads one step ahead (through the constant address space: the index lists are read-only for the launch); rows past
  // the end of the range or of K get the zero row (s_cselect). No pointer list is staged, the set-up needs no memory round trip.
  typedef const __attribute__((address_space(4))) int* ts_cidx;
  const float* obase = op ? g.B + (long)j * 128 : g.A + (long)mi * 128;
  const long old_ = op ? g.ldb : g.lda;
  const int* gidx = op ? g.b_idx : g.a_idx;
  const bool has_idx = gidx != nullptr;
  typedef int ts_i8 __attribute__((ext_vector_type(8), aligned(4)));
  typedef const __attribute__((address_space(4))) ts_i8* ts_cidx8;
  const ts_cidx sidx = (ts_cidx)gidx + row0;                        // first entry of the range in the index list
  const long rows_left = (long)g.K - row0;
  const int valid_rows = (int)(rows_left < (long)n_chunks * TS_KC ? rows_left : (long)n_chunks * TS_KC);      // >= 1
  int nid[8];
  // the eight table rows of the wave's row group of chunk C: one s_load_dwordx8 (a group that reaches past the end of the range:
  // clamped single loads — the last chunk of the last range only; plain operands: the row numbers themselves)
#define TS_IDX(C, ids) do { \
    const int r0_ = (C) * TS_KC + rg * 8; \
    if (!has_idx) { \
_Pragma("unroll") \
      for (int q_ = 0; q_ < 8; ++q_) ids[q_] = (int)row0 + r0_ + q_; \
    } else if (r0_ + 8 <= valid_rows) { \
      const ts_i8 v_ = *(ts_cidx8)(sidx + r0_); \
_Pragma("unroll") \
      for (int q_ = 0; q_ < 8; ++q_) ids[q_] = v_[q_]; \
    } else { \
_Pragma("unroll") \
      for (int q_ = 0; q_ < 8; ++q_) ids[q_] = sidx[r0_ + q_ < valid_rows ? r0_ + q_ : valid_rows - 1]; \
    } \
  } while (0)
#define TS_LOADI(C, raw, ids) do { \
_Pragma("unroll") \
    for (int q_ = 0; q_ < 8; ++q_) { \
      const int r_ = (C) * TS_KC + rg * 8 + q_; \
      const bool in_ = r_ < valid_rows; \
      const float* b_ = in_ ? obase : (const float*)ts_zero_row; \
      const long o_ = in_ ? (long)ids[q_] * old_ : 0; \
      raw[q_] = ((ts_gptr)(b_ + o_))[lane]; \
    } \
  } while (0)
#define TS_LOAD(C, raw) do { TS_LOADI(C, raw, nid); TS_IDX((C) + 1, nid); } while (0)
#else
  const unsigned char* optr = smem + 2 * TS_BUF + (op * TS_MAXROWS + rg * 8) * 8;
#define TS_LOAD(C, raw) do { \
    const sp_lds_u32x4* ip_ = (const sp_lds_u32x4*)(optr + (C) * (TS_KC * 8)); \
    const sp_u32x4 v0_ = ip_[0], v1_ = ip_[1], v2_ = ip_[2], v3_ = ip_[3];      /* eight row pointers, the same for every lane */ \
    TS_LOAD2(v0_, 0, raw); TS_LOAD2(v1_, 2, raw); TS_LOAD2(v2_, 4, raw); TS_LOAD2(v3_, 6, raw); \
  } while (0)
#define TS_LOAD2(v, q, raw) do { \
    const ts_gptr pa_ = (ts_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v[0])); \
    const ts_gptr pb_ = (ts_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v[2])); \
    raw[q] = pa_[lane]; \
    raw[q + 1] = pb_[lane]; \
  } while (0)
#endif
#define TS_STORE(buf, raw) do { \
_Pragma("unroll") \
    for (int h_ = 0; h_ < 2; ++h_) { \
      sp_u32x4 p0_, p1_, p2_; \
      sp_split8(make_float4(raw[0][h_], raw[1][h_], raw[2][h_], raw[3][h_]), make_float4(raw[4][h_], raw[5][h_], raw[6][h_], raw[7][h_]), p0_, p1_, p2_); \
      unsigned char* d_ = smem + (buf) * TS_BUF + st_off + h_ * (2 * 2 * 64 * 16); \
      *(sp_lds_u32x4*)(d_) = p0_; \
      *(sp_lds_u32x4*)(d_ + TS_PL) = p1_; \
      *(sp_lds_u32x4*)(d_ + 2 * TS_PL) = p2_; \
    } \
  } while (0)

  // ---- multiplier role: k step ks = wave >> 2; quarter (mh, nh) of the tile
  const int ks = wave >> 2, mh = (wave >> 1) & 1, nh = wave & 1;
  const int fa_off = ((mh * 2 + ks) * 64 + lane) * 16;                        // A fragment of tile mh (+ 2 tiles for mh + 2)
  const int fb_off = TS_OP + ((nh * 2 + ks) * 64 + lane) * 16;
  sp_f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#define TS_MULT(buf) do { \
    const unsigned char* f_ = smem + (buf) * TS_BUF; \
    sp_u32x4 a_[2][3], b_[2][3]; \
_Pragma("unroll") \
    for (int x_ = 0; x_ < 2; ++x_) \
_Pragma("unroll") \
      for (int p_ = 0; p_ < 3; ++p_) { \
        a_[x_][p_] = *(const sp_lds_u32x4*)(f_ + fa_off + p_ * TS_PL + x_ * (2 * 2 * 64 * 16)); \
        b_[x_][p_] = *(const sp_lds_u32x4*)(f_ + fb_off + p_ * TS_PL + x_ * (2 * 2 * 64 * 16)); \
      } \
    TS_TERM(2, 0); TS_TERM(0, 2); TS_TERM(1, 1); TS_TERM(1, 0); TS_TERM(0, 1); TS_TERM(0, 0); \
  } while (0)
#define TS_TERM(pa, pb) do { \
    acc[0] = sp_mfma(a_[0][pa], b_[0][pb], acc[0]); \
    acc[1] = sp_mfma(a_[0][pa], b_[1][pb], acc[1]); \
    acc[2] = sp_mfma(a_[1][pa], b_[0][pb], acc[2]); \
    acc[3] = sp_mfma(a_[1][pa], b_[1][pb], acc[3]); \
  } while (0)

  // ---- prologue: chunks 0 .. 3 in flight, chunk 0 split into set 0, chunk 4 requested
#if TS_SALU
  {
    int i0[8], i1[8];                                              // two index chunks per round of scalar loads
    TS_IDX(0, i0); TS_IDX(1, i1);
    TS_LOADI(0, r0, i0);
    TS_LOADI(1, r1, i1);
    TS_IDX(2, i0); TS_IDX(3, i1);
    TS_LOADI(2, r2, i0);
    if (TS_NBUF == 4) { TS_LOADI(3, r3, i1); TS_IDX(4, nid); } else { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) nid[q_] = i1[q_]; }
  }
  TS_STORE(0, r0);
  TS_LOAD(TS_NBUF, r0);
#else
  TS_LOAD(0, r0);
  TS_LOAD(1, r1);
  TS_LOAD(2, r2);
  if (TS_NBUF == 4) TS_LOAD(3, r3);
  TS_STORE(0, r0);
  TS_LOAD(TS_NBUF, r0);
#endif
  __syncthreads();

  // step c: multiply set c & 1; split chunk c + 1 (raw buffer (c + 1) % 4) into the other set; request chunk c + 5 into that buffer.
  // The loop body has NO branch: the trip count is rounded up to the period of the two rotations (4) and chunks past the end of the
  // range are chunks of the zero row (the pointer list is padded) — they cost up to three steps of MFMAs on zeros. With a branch
  // around the loads or the step, hipcc's wait-count pass can no longer tell how many loads are behind the one it needs and
  // waits vmcnt(0) in front of every split, i.e. for the prefetch of the next two chunks.
  // The two waves of a SIMD (w and w + 4: one of each k-step group) take the two halves of a step in opposite order, so that one
  // issues MFMAs while the other one splits and writes (TS_SKEW=0: both multiply first).
#ifndef TS_SKEW
#define TS_SKEW 0
#endif
#ifndef TS_SGB
#define TS_SGB 4                  // VALU instructions the scheduler is asked to place behind every MFMA of a step (0: its own order)
#endif
#define TS_PIPE() do { \
    if (TS_SGB > 0) { \
_Pragma("unroll") \
      for (int i_ = 0; i_ < 24; ++i_) { \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x002, TS_SGB, 0); \
      } \
    } \
  } while (0)
#define TS_STEP(CC, nxt) do { \
    const int c_ = (CC); \
    if (TS_SKEW && ks == 1) { \
      if (TS_ABL != 2 && TS_ABL != 3) TS_STORE((c_ + 1) & 1, nxt); \
      TS_MULT(c_ & 1); \
    } else { \
      TS_MULT(c_ & 1); \
      if (TS_ABL != 2 && TS_ABL != 3) TS_STORE((c_ + 1) & 1, nxt); \
    } \
    if (TS_ABL != 1 && TS_ABL != 3) TS_LOAD(c_ + TS_NBUF + 1, nxt); \
    TS_PIPE(); \
    __syncthreads(); \
  } while (0)
#pragma unroll 1
  for (int c0 = 0; c0 < (TS_ABL == 4 ? 0 : n_chunks); c0 += (TS_NBUF == 4 ? 4 : 6)) {
    if (TS_NBUF == 4) {
      TS_STEP(c0, r1);
      TS_STEP(c0 + 1, r2);
      TS_STEP(c0 + 2, r3);
      TS_STEP(c0 + 3, r0);
    } else {
      TS_STEP(c0, r1);
      TS_STEP(c0 + 1, r2);
      TS_STEP(c0 + 2, r0);
      TS_STEP(c0 + 3, r1);
      TS_STEP(c0 + 4, r2);
      TS_STEP(c0 + 5, r0);
    }
  }

  // ---- epilogue: both k-step groups write their quarter tiles into a row-major 128 x 128 image each (the plane sets and the
  // pointer lists are free: the last step ended with a barrier), then all 512 threads add the two images (fixed order) and store the
  // slab with 16 bytes per lane, a 512-byte row per half wave.  (Stores straight from the accumulator layout — 8 bytes per lane, by
  // half of the waves — are issue-bound: ~4 us for the 64 KB of a workgroup.)
  {
    float* img = reinterpret_cast<float*>(smem) + ks * (128 * 128);
    // accumulator (x, y), register q, lane: output row 64 mh + 2 (8 (q >> 2) + 4 half + (q & 3)) + x, column 64 nh + 2 l31 + y
    float* w = img + (64 * mh + 8 * half) * 128 + 64 * nh + 2 * l31;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        sp_f32x2 v;
        v[0] = acc[x * 2][q];
        v[1] = acc[x * 2 + 1][q];
        *reinterpret_cast<sp_f32x2*>(w + (2 * ((q & 3) + 8 * (q >> 2)) + x) * 128) = v;
      }
  }
  __syncthreads();
  {
    const float4* i0 = reinterpret_cast<const float4*>(smem);
    const float4* i1 = i0 + 128 * 128 / 4;
    float* out = g.slab + ((long)z * g.M + (long)mi * 128) * g.N + (long)j * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = t + 512 * i;                                   // float4 f of the image: row f >> 5, columns 4 (f & 31) ..
      const float4 a = i0[f], b = i1[f];
      *reinterpret_cast<float4*>(out + (long)(f >> 5) * g.N + 4 * (f & 31)) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
  }
}

template <bool WIDE>
__global__ __launch_bounds__(512, 1) void gemm_split_tn_kernel(TnSplitArgs g) {
  ts_body<WIDE>(g, blockIdx.x, gridDim.x);
}

// Several products in ONE launch (the dW products of a backward pass, launched together when the pass is over: their operands
// live in the step's arena until then): workgroups start[p] .. start[p + 1] - 1 (each range a multiple of 8 long, so the XCD
// round-robin of the whole grid is the round-robin of every slice) belong to product p. Two launch floors and two ramp-up /
// drain phases per step less than three launches of gemm_split_tn_kernel.
static bool ts_enabled() {          // read per call: tests and A/B runs switch it within one process
  const char* a = getenv("SBR_GEMM_SPLIT");
  const char* b = getenv("SBR_TN_SPLIT");
  return !((a && atoi(a) == 0) || (b && atoi(b) == 0));
}

// K ranges (= slabs) the kernel would write for this shape; 0: the shape stays on the fp32 pipe
int sbr_tn_split_splits(int M, int N, int K) {
  if (!ts_enabled() || M < 128 || M % 128 != 0 || N < 128 || N % 128 != 0 || K < 4096) return 0;      // (the launch also wants 8-byte aligned rows)
  const int nj = (N / 128) * (M / 128), chunks = sbr_cdiv(K, TS_KC);                                    // output tiles per K range
  int nz = 256 / nj;
  const int least = sbr_cdiv(chunks, TS_MAXROWS / TS_KC - TS_PAD);
  if (nz < least) nz = least;
  if (nz < 1) nz = 1;
  if (nz > chunks) nz = chunks;
  return nz;
}

// slab[z][M][N] for z < sbr_tn_split_splits(M, N, K) (plain stores). Returns -1 when the shape is not eligible.
int sbr_tn_split_launch(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M, int N, int K,
                        float* slab, int* splits_out, hipStream_t s) {
  const int nz = sbr_tn_split_splits(M, N, K);
  if (nz <= 0) return -1;
  if (((uintptr_t)A | (uintptr_t)B) % 8 != 0 || (uintptr_t)slab % 16 != 0 || lda % 2 != 0 || ldb % 2 != 0) return -1;
  TnSplitArgs g;
  g.A = A; g.lda = lda; g.a_idx = a_idx; g.B = B; g.ldb = ldb; g.b_idx = b_idx; g.slab = slab; g.N = N; g.K = K;
  g.M = M; g.nz = nz; g.nj = N / 128; g.nm = M / 128; g.chunks = sbr_cdiv(K, TS_KC);
  const size_t lds = TS_SALU ? 2 * 128 * 128 * sizeof(float) : 2 * TS_BUF + 2 * TS_MAXROWS * sizeof(unsigned long long);   // TS_SALU: the epilogue images (128 KB)
  const int grid = sbr_cdiv(nz * g.nj * g.nm, 8) * 8;
#define TS_LAUNCH(WIDE)                                                                                                   \
  do {                                                                                                                     \
    static int attr_dev = -1;                                                                                          \
    if (sbr_attr_stale(&attr_dev)) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)gemm_split_tn_kernel<WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        sbr_set_error("sbr_gemm_tn_f32: cannot raise the dynamic LDS limit of the bf16-split kernel");                     \
        return SBR_ERR_HIP;                                                                                                \
      }                                                                                                                    \
    }                                                                                                                      \
    gemm_split_tn_kernel<WIDE><<<grid, 512, lds, s>>>(g);                                                                  \
  } while (0)
  if (g.nj * g.nm > 1) TS_LAUNCH(true);
  else TS_LAUNCH(false);
#undef TS_LAUNCH
  SBR_CHECK_LAUNCH("sbr_gemm_tn_f32 (bf16 split)");
  *splits_out = nz;
  return SBR_OK;
}
