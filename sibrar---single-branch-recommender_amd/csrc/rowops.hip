// Row-wise (HBM-bound) kernels of the SingleBranchNet hot path: index resolution, embedding lookups / bags,
// the CSR "interactions" projector, activation-gradient gathers, column sums, L2 normalisation, dropout,
// modality aggregation and the per-slot user x item dot product.
// All of them move rows of C..D floats; lanes run along the feature dimension so that every wave instruction
// touches one contiguous 256-byte (or wider) segment of a row.
#include "common.h"

#define SBR_MAX_SEG 16

struct SegTable {
  int n_seg;
  int offs[SBR_MAX_SEG + 1];
  const int* maps[SBR_MAX_SEG];
  int lens[SBR_MAX_SEG];            // ids covered by maps[s] (identity map: number of rows)
};

// rows[j] = map_seg(j)[ idx[slots[j] / k] ]  — Feature.__getitem__'s id -> row map (data/Feature.py:146) applied to the
// flattened, k-times repeated index tensor of SingleBranchNetEntity._get_modality_embeddings (sgd_alg.py:1944-1946).
__global__ void resolve_rows_kernel(const long* __restrict__ idx, int k, const int* __restrict__ slots, int n,
                                    SegTable st, int* __restrict__ rows, int* __restrict__ err) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int seg = 0;
  while (seg + 1 < st.n_seg && j >= st.offs[seg + 1]) ++seg;
  const long id = idx[slots[j] / k];
  int r = -1;
  if (id >= 0 && id < st.lens[seg]) r = st.maps[seg] ? st.maps[seg][id] : (int)id;
  if (r < 0) {                          // id not present in this feature's split: flag it (the host raises KeyError at its next
    atomicExch(err, 1);                 // check) and fall back to row 0 so that no kernel ever indexes out of bounds
    r = 0;
  }
  rows[j] = r;
}

extern "C" int sbr_resolve_rows(const long* idx, int k, const int* slots, int n, int n_seg, const int* seg_offsets,
                                const int* const* rowmaps, const int* rowmap_lens, int* rows_out, int* err_flag, void* stream) {
  SBR_REQUIRE(n_seg >= 1 && n_seg <= SBR_MAX_SEG, "sbr_resolve_rows: n_seg %d out of range", n_seg);
  SBR_REQUIRE(k >= 1, "sbr_resolve_rows: k must be >= 1");
  if (n == 0) return SBR_OK;
  SegTable st;
  st.n_seg = n_seg;
  for (int i = 0; i <= n_seg; ++i) st.offs[i] = seg_offsets[i];
  SBR_REQUIRE(rowmaps && rowmap_lens && err_flag, "sbr_resolve_rows: null operand");
  for (int i = 0; i < n_seg; ++i) { st.maps[i] = rowmaps[i]; st.lens[i] = rowmap_lens[i]; }
  resolve_rows_kernel<<<sbr_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(idx, k, slots, n, st, rows_out, err_flag);
  SBR_CHECK_LAUNCH("sbr_resolve_rows");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stable counting sort of the modality draw (sgd_alg.py:1934-1957 groups the flattened index tensor by sampled modality
// with boolean masks): slots_out[seg_off[m] + r] = index of the r-th slot (ascending) whose modality is m; the tail of every
// segment (capacity - count, the padding of a graph-mode plan) is filled with the sentinel slot R.
// Two launches: (1) per-block histograms of contiguous 4096-slot chunks, (2) every block adds up the histograms in front of
// it, ranks its own slots with a block-wide exclusive scan over per-thread counts (thread t owns 16 consecutive slots, so the
// order stays ascending) and writes them; the padding tails are filled by all blocks together.
// ---------------------------------------------------------------------------------------------------------------
#define SBR_PART_MAX 8
#define PART_EPT_ONE 16                   // slots per thread when ONE block covers the batch (<= 4,096 slots: one launch)
#define PART_EPT 4                        // ... otherwise: one 32-bit load, 1,024 slots per block (16 per thread = 22 blocks at the bench's 90k slots:
                                          // latency-bound, 15 + 6 us for the two kernels)
#define PART_CHUNK (256 * PART_EPT)
struct PartSeg { int n_mod; int offs[SBR_PART_MAX + 1]; };
// the EPT modality positions lo .. lo + EPT - 1 of a thread (-1 past the end); EPT = 4: one aligned 32-bit load
template <int EPT>
__device__ __forceinline__ void part_load(const signed char* __restrict__ pos, long lo, long R, bool aligned, signed char (&out)[EPT]) {
  if (EPT == 4 && aligned && lo + 4 <= R) {
    const int w = *reinterpret_cast<const int*>(pos + lo);
    out[0] = (signed char)(w & 0xFF); out[1] = (signed char)((w >> 8) & 0xFF); out[2] = (signed char)((w >> 16) & 0xFF); out[3] = (signed char)((w >> 24) & 0xFF);
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = (lo + e < R) ? pos[lo + e] : (signed char)-1;
  }
}

template <int EPT>
__global__ __launch_bounds__(256) void partition_count_kernel(const signed char* __restrict__ pos, long R, int n_mod,
                                                              int* __restrict__ hist) {
  __shared__ int h[SBR_PART_MAX];
  if (threadIdx.x < SBR_PART_MAX) h[threadIdx.x] = 0;
  __syncthreads();
  const long lo = (long)blockIdx.x * (256 * EPT) + threadIdx.x * EPT;
  int local[SBR_PART_MAX];
#pragma unroll
  for (int m = 0; m < SBR_PART_MAX; ++m) local[m] = 0;
  signed char mine[EPT];
  part_load<EPT>(pos, lo, R, (((uintptr_t)pos) & 3) == 0, mine);
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
#pragma unroll
    for (int q = 0; q < SBR_PART_MAX; ++q) local[q] += (mine[e] == q);
  }
#pragma unroll
  for (int m = 0; m < SBR_PART_MAX; ++m) {
    const int s = (int)sbr_wave_sum((float)local[m]);       // <= 1024 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0 && m < n_mod) atomicAdd(&h[m], s);
  }
  __syncthreads();
  if (threadIdx.x < SBR_PART_MAX) hist[blockIdx.x * SBR_PART_MAX + threadIdx.x] = h[threadIdx.x];
}

template <int EPT>
__global__ __launch_bounds__(256) void partition_scatter_kernel(const signed char* __restrict__ pos, long R, PartSeg sg,
                                                                const int* __restrict__ hist, int* __restrict__ slots_out) {
  __shared__ int before[SBR_PART_MAX], total[SBR_PART_MAX];
  __shared__ int wave_tot[SBR_PART_MAX][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < SBR_PART_MAX) { before[t] = 0; total[t] = 0; }
  __syncthreads();
  if (hist) {                       // hist == nullptr: one block covers all slots and takes the totals from its own scan below
    // per-block histograms of ALL blocks -> counts in front of this block and totals, all threads at once
    for (int i = t; i < (int)gridDim.x * SBR_PART_MAX; i += 256) {
      const int v = hist[i];
      if (v) {
        atomicAdd(&total[i & (SBR_PART_MAX - 1)], v);
        if (i / SBR_PART_MAX < (int)blockIdx.x) atomicAdd(&before[i & (SBR_PART_MAX - 1)], v);
      }
    }
  }
  const long lo = (long)blockIdx.x * (256 * EPT) + t * EPT;
  signed char mine[EPT];
  int local[SBR_PART_MAX];
#pragma unroll
  for (int m = 0; m < SBR_PART_MAX; ++m) local[m] = 0;
  part_load<EPT>(pos, lo, R, (((uintptr_t)pos) & 3) == 0, mine);
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
#pragma unroll
    for (int q = 0; q < SBR_PART_MAX; ++q) local[q] += (mine[e] == q);
  }
  // exclusive scan of the per-thread counts over the block, per modality: wave scan + wave totals
  int excl[SBR_PART_MAX];
#pragma unroll
  for (int m = 0; m < SBR_PART_MAX; ++m) {
    int incl = local[m];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    excl[m] = incl - local[m];
    if (lane == 63) wave_tot[m][wave] = incl;
  }
  __syncthreads();
  if (!hist && t < SBR_PART_MAX) total[t] = wave_tot[t][0] + wave_tot[t][1] + wave_tot[t][2] + wave_tot[t][3];
#pragma unroll
  for (int m = 0; m < SBR_PART_MAX; ++m) {
    int w = (m < sg.n_mod ? sg.offs[m] : 0) + before[m] + excl[m];
    for (int k = 0; k < wave; ++k) w += wave_tot[m][k];
    excl[m] = w;                                              // first output position of this thread for modality m
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
#pragma unroll
    for (int q = 0; q < SBR_PART_MAX; ++q)
      if (mine[e] == q) slots_out[excl[q]++] = (int)(lo + e);
  }
  // padding tails [offs[m] + total[m], offs[m+1]) <- R, shared by all blocks
  if (!hist) __syncthreads();      // single-block mode: total[] was written after the barrier above
  for (int m = 0; m < sg.n_mod; ++m) {
    const int beg = sg.offs[m] + total[m], end = sg.offs[m + 1];
    for (int e = beg + blockIdx.x * 256 + t; e < end; e += gridDim.x * 256) slots_out[e] = (int)R;
  }
}

extern "C" long sbr_partition_slots_workspace(long R) { return (long)sbr_cdiv(R > 0 ? R : 1, PART_CHUNK) * SBR_PART_MAX * (long)sizeof(int); }

extern "C" int sbr_partition_slots(const signed char* pos, long R, int n_mod, const int* seg_offsets, int* slots_out,
                                   void* workspace, long workspace_bytes, void* stream) {
  SBR_REQUIRE(n_mod >= 1 && n_mod <= SBR_PART_MAX, "sbr_partition_slots: n_mod %d out of range", n_mod);
  SBR_REQUIRE(R >= 0 && R < 2147483647L, "sbr_partition_slots: R out of range");
  SBR_REQUIRE(pos && seg_offsets && slots_out, "sbr_partition_slots: null operand");
  SBR_REQUIRE(workspace && workspace_bytes >= sbr_partition_slots_workspace(R), "sbr_partition_slots: workspace too small");
  PartSeg sg;
  sg.n_mod = n_mod;
  for (int i = 0; i <= n_mod; ++i) sg.offs[i] = seg_offsets[i];
  if (sg.offs[n_mod] == 0) return SBR_OK;
  const int nb = sbr_cdiv(R > 0 ? R : 1, PART_CHUNK);
  hipStream_t s = (hipStream_t)stream;
  if (R <= 256 * PART_EPT_ONE) {   // small batches: one launch (the kernel chain, not the work, bounds them)
    partition_scatter_kernel<PART_EPT_ONE><<<1, 256, 0, s>>>(pos, R, sg, nullptr, slots_out);
  } else {
    partition_count_kernel<PART_EPT><<<nb, 256, 0, s>>>(pos, R, n_mod, (int*)workspace);
    partition_scatter_kernel<PART_EPT><<<nb, 256, 0, s>>>(pos, R, sg, (const int*)workspace, slots_out);
  }
  SBR_CHECK_LAUNCH("sbr_partition_slots");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// nn.Embedding lookup (sgd_alg.py:1331,1386): out[oi(j), :] = W[rows[j], :]
// ---------------------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ W, long ldw, const int* __restrict__ rows,
                                   float* __restrict__ out, long ldo, const int* __restrict__ out_idx, long n, int D) {
  const long total = n * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / D;
    const int c = (int)(e - j * D);
    const long o = out_idx ? out_idx[j] : j;
    out[o * ldo + c] = W[(long)rows[j] * ldw + c];
  }
}

// 16 bytes per thread, one thread per (row, 4-column chunk), no grid-stride loop: the scalar kernel above keeps one dependent
// 4-byte load in flight per thread and runs at memory latency (45k rows x 128: 14 us = 2 TB/s of useful traffic)
__global__ void gather_rows4_kernel(const float* __restrict__ W, long ldw, const int* __restrict__ rows,
                                    float* __restrict__ out, long ldo, const int* __restrict__ out_idx, long n, int D4) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= n * D4) return;
  const long j = e / D4;
  const int c = (int)(e - j * D4) * 4;
  const long o = out_idx ? out_idx[j] : j;
  *reinterpret_cast<float4*>(out + o * ldo + c) = *reinterpret_cast<const float4*>(W + (long)rows[j] * ldw + c);
}

extern "C" int sbr_gather_rows(const float* W, long ldw, const int* rows, float* out, long ldo, const int* out_idx,
                               long n, int D, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(W && rows && out, "sbr_gather_rows: null operand");
  if ((D & 3) == 0 && (ldw & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)W) | ((uintptr_t)out)) & 15) == 0 &&
      n * (D / 4) < (1L << 31) * 256) {
    const long total4 = n * (D / 4);
    gather_rows4_kernel<<<(unsigned)sbr_cdiv(total4, 256), 256, 0, (hipStream_t)stream>>>(W, ldw, rows, out, ldo, out_idx, n, D / 4);
    SBR_CHECK_LAUNCH("sbr_gather_rows");
    return SBR_OK;
  }
  int blocks = sbr_cdiv(n * D, 256);
  if (blocks > 4096) blocks = 4096;
  gather_rows_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(W, ldw, rows, out, ldo, out_idx, n, D);
  SBR_CHECK_LAUNCH("sbr_gather_rows");
  return SBR_OK;
}

// A plain embedding-lookup side in ONE launch: id -> table row (sbr_resolve_rows with one segment and k = 1) and the row gather
// (sbr_gather_rows) — out[j, :] = W[row(idx[j]), :], rows_out[j] = row(idx[j]) kept for the backward pass. Every one of the D / 4
// threads of a row resolves the id itself (one cached load each); ids without a row set *err and read row 0, as sbr_resolve_rows.
__global__ void lookup_rows4_kernel(const long* __restrict__ idx, long n, const int* __restrict__ rowmap, int map_len,
                                    const float* __restrict__ W, long ldw, int* __restrict__ rows_out, float* __restrict__ out, long ldo,
                                    int D4, int* __restrict__ err) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= n * D4) return;
  const long j = e / D4;
  const int c4 = (int)(e - j * D4);
  const long id = idx[j];
  int r = -1;
  if (id >= 0 && id < map_len) r = rowmap ? rowmap[id] : (int)id;
  if (r < 0) {
    if (c4 == 0) atomicExch(err, 1);
    r = 0;
  }
  if (c4 == 0) rows_out[j] = r;
  *reinterpret_cast<float4*>(out + j * ldo + 4 * c4) = *reinterpret_cast<const float4*>(W + (long)r * ldw + 4 * c4);
}

// 1 when sbr_lookup_rows takes this layout (else: sbr_resolve_rows + sbr_gather_rows)
extern "C" int sbr_lookup_rows_supported(const float* W, long ldw, const float* out, long ldo, int D) {
  return (D & 3) == 0 && D >= 4 && (ldw & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)W) | ((uintptr_t)out)) & 15) == 0;
}

extern "C" int sbr_lookup_rows(const long* idx, long n, const int* rowmap, int rowmap_len, const float* W, long ldw, int* rows_out,
                               float* out, long ldo, int D, int* err_flag, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(idx && W && rows_out && out && err_flag, "sbr_lookup_rows: null operand");
  SBR_REQUIRE(sbr_lookup_rows_supported(W, ldw, out, ldo, D), "sbr_lookup_rows: D=%d / alignment not supported", D);
  const long total4 = n * (D / 4);
  SBR_REQUIRE(total4 < (1L << 31) * 256, "sbr_lookup_rows: too many rows");
  lookup_rows4_kernel<<<(unsigned)sbr_cdiv(total4, 256), 256, 0, (hipStream_t)stream>>>(idx, n, rowmap, rowmap_len, W, ldw, rows_out, out,
                                                                                        ldo, D / 4, err_flag);
  SBR_CHECK_LAUNCH("sbr_lookup_rows");
  return SBR_OK;
}

// backward of the lookup: dW[rows[j], :] += scale * dOut[ii(j), :]   (float atomics into a zero-initialised dense gradient;
// the reference's dense nn.Embedding gradient, consumed by a dense optimizer — trainer.py:62-68)
__global__ void scatter_add_rows_kernel(const float* __restrict__ dOut, long ldo, const int* __restrict__ in_idx,
                                        const int* __restrict__ rows, float* __restrict__ dW, long ldw, long n, int D) {
  const long total = n * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / D;
    const int c = (int)(e - j * D);
    const long i = in_idx ? in_idx[j] : j;
    atomicAdd(&dW[(long)rows[j] * ldw + c], dOut[i * ldo + c]);
  }
}

// D a power of two <= 256: a thread owns ONE element — row (256 >> LOG2D) * block + (t >> LOG2D), column t & (D - 1) — so a
// wave-instruction adds 256 contiguous bytes of one destination row (the shape the memory-side atomic units take at full rate,
// MI355X_MICROARCH.md "Global float atomics") and no thread divides: the grid-stride kernel above spends a 64-bit division per
// element and keeps 5-6 dependent atomics per thread (45,824 rows x 128: 41 us = 0.57 TB/s of added bytes).
__global__ __launch_bounds__(256) void scatter_add_rows_pow2_kernel(const float* __restrict__ dOut, long ldo,
                                                                    const int* __restrict__ in_idx, const int* __restrict__ rows,
                                                                    float* __restrict__ dW, long ldw, long n, int log2d) {
  const int t = threadIdx.x;
  const long j = ((long)blockIdx.x << (8 - log2d)) + (t >> log2d);
  if (j >= n) return;
  const int c = t & ((1 << log2d) - 1);
  const long i = in_idx ? in_idx[j] : j;
  atomicAdd(&dW[(long)rows[j] * ldw + c], dOut[i * ldo + c]);
}

// D = 128, many rows: persistent waves, each takes rows w, w + W, ... four at a time — the four (source row, table row) index pairs
// first, then the eight 256-byte halves of the four gradient rows, then eight atomic instructions back to back (a wave-instruction
// adds 256 contiguous bytes of one destination row). The one-element-per-thread kernel above issues ONE atomic per wave behind a
// chain of three dependent loads and a wave launch; this one keeps eight in flight per wave.
__global__ __launch_bounds__(256) void scatter_add_rows128_kernel(const float* __restrict__ dOut, long ldo, const int* __restrict__ in_idx,
                                                                  const int* __restrict__ rows, float* __restrict__ dW, long ldw, long n) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long j0 = wid * 4; j0 < n; j0 += nw * 4) {
    long src[4], dst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long j = j0 + q < n ? j0 + q : n - 1;
      src[q] = in_idx ? in_idx[j] : j;
      dst[q] = rows[j];
    }
    float v[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q][0] = dOut[src[q] * ldo + lane];
      v[q][1] = dOut[src[q] * ldo + 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (a gradient row of zeros adds nothing: a graph-mode step pads every slot list with ~770 sentinel slots that all name ONE
      // table row and carry a zero gradient row, engine._EntityRun.plan — 1,540 atomic instructions on the same four lines)
      if (j0 + q < n && __ballot(v[q][0] != 0.f || v[q][1] != 0.f) != 0) {          // wave-uniform
        atomicAdd(&dW[dst[q] * ldw + lane], v[q][0]);
        atomicAdd(&dW[dst[q] * ldw + 64 + lane], v[q][1]);
      }
    }
  }
}

extern "C" int sbr_scatter_add_rows(const float* dOut, long ldo, const int* in_idx, const int* rows, float* dW,
                                    long ldw, long n, int D, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dOut && rows && dW, "sbr_scatter_add_rows: null operand");
  if (D == 128 && n >= 4096) {
    int blocks = (int)sbr_cdiv(n, 16);                           // >= 4 rows per wave
    if (blocks > 2048) blocks = 2048;                            // 8 workgroups of 4 waves per CU
    scatter_add_rows128_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dOut, ldo, in_idx, rows, dW, ldw, n);
    SBR_CHECK_LAUNCH("sbr_scatter_add_rows");
    return SBR_OK;
  }
  if (D >= 16 && D <= 256 && (D & (D - 1)) == 0) {
    int log2d = 0;
    while ((1 << log2d) < D) ++log2d;
    const long rows_per_block = 256 >> log2d;
    scatter_add_rows_pow2_kernel<<<sbr_cdiv(n, rows_per_block), 256, 0, (hipStream_t)stream>>>(dOut, ldo, in_idx, rows, dW, ldw, n, log2d);
    SBR_CHECK_LAUNCH("sbr_scatter_add_rows");
    return SBR_OK;
  }
  // (Tried: 16-byte loads + four atomics per thread. Each atomic instruction of a wave then touches every fourth float —
  // four times the L2 atomic transactions of the lane-contiguous scalar kernel: the step went from 0.86 to 0.98 ms.)
  int blocks = sbr_cdiv(n * D, 256);
  if (blocks > 4096) blocks = 4096;
  scatter_add_rows_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dOut, ldo, in_idx, rows, dW, ldw, n, D);
  SBR_CHECK_LAUNCH("sbr_scatter_add_rows");
  return SBR_OK;
}

// The same gradient from row lists sorted by table row (data-parallel exchange of lookup gradients: every rank holds the same
// gathered lists and must produce the same bits, so no float atomics): position j of the sorted order names source row
// p = perm[j] — block p / blk at dOut + (p / blk) * block_stride, row p % blk of it. The thread of a segment head adds the
// segment's rows in sorted (= stable source) order and owns the destination row.
__global__ void scatter_add_rows_sorted_kernel(const float* __restrict__ dOut, long ldo, long blk, long block_stride,
                                               const long* __restrict__ perm, const int* __restrict__ rows_sorted,
                                               float* __restrict__ dW, long ldw, long n, int D) {
  const long total = n * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / D;
    const int c = (int)(e - j * D);
    const int r = rows_sorted[j];
    if (j > 0 && rows_sorted[j - 1] == r) continue;
    float acc = 0.f;
    for (long q = j; q < n && rows_sorted[q] == r; ++q) {
      const long p = perm[q];
      acc += dOut[(p / blk) * block_stride + (p % blk) * ldo + c];
    }
    dW[(long)r * ldw + c] += acc;
  }
}

extern "C" int sbr_scatter_add_rows_sorted(const float* dOut, long ldo, long blk, long block_stride, const long* perm,
                                           const int* rows_sorted, float* dW, long ldw, long n, int D, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dOut && perm && rows_sorted && dW, "sbr_scatter_add_rows_sorted: null operand");
  SBR_REQUIRE(blk >= 1, "sbr_scatter_add_rows_sorted: block length %ld", blk);
  int blocks = sbr_cdiv(n * D, 256);
  if (blocks > 4096) blocks = 4096;
  scatter_add_rows_sorted_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dOut, ldo, blk, block_stride, perm, rows_sorted, dW,
                                                                          ldw, n, D);
  SBR_CHECK_LAUNCH("sbr_scatter_add_rows_sorted");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// nn.EmbeddingBag(mode='mean', padding_idx=pad) over padded tag lists (sgd_alg.py:1336-1337; Feature.py:254-255)
// one wave per output row; lanes run along D
// ---------------------------------------------------------------------------------------------------------------
__global__ void bag_mean_fwd_kernel(const float* __restrict__ W, long ldw, const int* __restrict__ tags, int T, int pad,
                                    const int* __restrict__ rows, float* __restrict__ out, long ldo,
                                    const int* __restrict__ out_idx, long n, int D) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const int* tg = tags + (long)rows[j] * T;
  const long o = (out_idx ? (long)out_idx[j] : j) * ldo;
  int cnt = 0;
  for (int q = 0; q < T; ++q) cnt += (tg[q] != pad);
  const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
  for (int c = lane; c < D; c += 64) {
    float acc = 0.f;
    for (int q = 0; q < T; ++q) {
      const int tq = tg[q];
      if (tq != pad) acc += W[(long)tq * ldw + c];
    }
    out[o + c] = acc * inv;
  }
}

__global__ void bag_mean_bwd_kernel(const float* __restrict__ dOut, long ldo, const int* __restrict__ in_idx,
                                    const int* __restrict__ tags, int T, int pad, const int* __restrict__ rows,
                                    float* __restrict__ dW, long ldw, long n, int D) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const int* tg = tags + (long)rows[j] * T;
  const long i = (in_idx ? (long)in_idx[j] : j) * ldo;
  int cnt = 0;
  for (int q = 0; q < T; ++q) cnt += (tg[q] != pad);
  if (cnt == 0) return;
  const float inv = 1.f / (float)cnt;
  for (int c = lane; c < D; c += 64) {
    const float g = dOut[i + c] * inv;
    for (int q = 0; q < T; ++q) {
      const int tq = tg[q];
      if (tq != pad) atomicAdd(&dW[(long)tq * ldw + c], g);
    }
  }
}

extern "C" int sbr_bag_mean_fwd(const float* W, long ldw, const int* tags, int T, int pad, const int* rows, float* out,
                                long ldo, const int* out_idx, long n, int D, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(W && tags && rows && out, "sbr_bag_mean_fwd: null operand");
  bag_mean_fwd_kernel<<<sbr_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(W, ldw, tags, T, pad, rows, out, ldo, out_idx, n, D);
  SBR_CHECK_LAUNCH("sbr_bag_mean_fwd");
  return SBR_OK;
}

extern "C" int sbr_bag_mean_bwd(const float* dOut, long ldo, const int* in_idx, const int* tags, int T, int pad,
                                const int* rows, float* dW, long ldw, long n, int D, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dOut && tags && rows && dW, "sbr_bag_mean_bwd: null operand");
  bag_mean_bwd_kernel<<<sbr_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(dOut, ldo, in_idx, tags, T, pad, rows, dW, ldw, n, D);
  SBR_CHECK_LAUNCH("sbr_bag_mean_bwd");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Linear over a CSR "interactions" row without densifying it (reference: Feature.py:149-150 toarray() +
// Linear(n_cols -> C), sgd_alg.py:1380, polylinear.py:51):   out[oi(j), :] = act(bias + sum_{q in row} val_q * Wt[col_q, :])
// Wt is the projector weight stored column-major ([n_cols, C] rows of C floats), so every nnz reads one contiguous row.
// ---------------------------------------------------------------------------------------------------------------
__global__ void csr_project_fwd_kernel(const long* __restrict__ indptr, const int* __restrict__ indices,
                                       const float* __restrict__ vals, const float* __restrict__ Wt, long ldw,
                                       const float* __restrict__ bias, const int* __restrict__ rows,
                                       float* __restrict__ out, long ldo, const int* __restrict__ out_idx, long n,
                                       int C, int act) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const long r = rows[j];
  const long beg = indptr[r], end = indptr[r + 1];
  const long o = (out_idx ? (long)out_idx[j] : j) * ldo;
  for (int c = lane; c < C; c += 64) {
    float acc = 0.f;
    for (long q = beg; q < end; ++q) {
      const float w = Wt[(long)indices[q] * ldw + c];
      acc += vals ? vals[q] * w : w;
    }
    if (bias) acc += bias[c];
    out[o + c] = sbr_act(acc, act);
  }
}

// dWt[col_q, :] += val_q * dZ[j, :]
__global__ void csr_project_bwd_kernel(const long* __restrict__ indptr, const int* __restrict__ indices,
                                       const float* __restrict__ vals, const float* __restrict__ dZ, long ldz,
                                       const int* __restrict__ rows, float* __restrict__ dWt, long ldw, long n, int C) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const long r = rows[j];
  const long beg = indptr[r], end = indptr[r + 1];
  for (int c = lane; c < C; c += 64) {
    const float g = dZ[j * ldz + c];
    for (long q = beg; q < end; ++q) atomicAdd(&dWt[(long)indices[q] * ldw + c], vals ? vals[q] * g : g);
  }
}

// Workgroup-per-row variants (C % 4 == 0, 16-byte aligned weight rows). Interaction rows are long-tailed (a popular item
// has thousands of nnz, the median a few dozen): one wave walking one row serialises the tail, so a row is spread over the
// whole workgroup. Forward: 256 threads = G nnz-groups x (C/4) lanes, each lane accumulates a float4 of the output row over
// every G-th nnz (four nnz in flight per lane), groups are combined through LDS. Backward: the four waves take 64-nnz blocks
// round robin; a block's column indices are loaded once (one per lane) and broadcast, and every nnz becomes one
// 256-byte-contiguous float atomic per 64 columns (the full-rate form, MI355X_MICROARCH.md "float atomic add").
// ACC (the backward pass in its gather form, sbr_csr_project_bwd_gather): the same sum over the TRANSPOSED matrix — row j = feature
// column j, its entries the entities that have the feature, Wt = the per-entity gradient rows — added to out[j, :]; rows may be null.
template <bool ACC>
__global__ __launch_bounds__(256) void csr_project_fwd_wg_kernel(
    const long* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ Wt, long ldw, const float* __restrict__ bias, const int* __restrict__ rows,
    float* __restrict__ out, long ldo, const int* __restrict__ out_idx, int C, int act) {
  __shared__ float4 part[256];
  const long j = blockIdx.x;
  const int t = threadIdx.x;
  const int LPG = C >> 2;                 // lanes per nnz group
  const int G = 256 / LPG;                // nnz groups (>= 1: C <= 1024)
  const int g = t / LPG, l = t - g * LPG;
  const long r = rows ? (long)rows[j] : j;
  const long beg = indptr[r], end = indptr[r + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    long q = beg + g;
    for (; q + 3L * G < end; q += 4L * G) {
      const int i0 = indices[q], i1 = indices[q + G], i2 = indices[q + 2L * G], i3 = indices[q + 3L * G];
      const float4 w0 = *reinterpret_cast<const float4*>(Wt + (long)i0 * ldw + 4 * l);
      const float4 w1 = *reinterpret_cast<const float4*>(Wt + (long)i1 * ldw + 4 * l);
      const float4 w2 = *reinterpret_cast<const float4*>(Wt + (long)i2 * ldw + 4 * l);
      const float4 w3 = *reinterpret_cast<const float4*>(Wt + (long)i3 * ldw + 4 * l);
      const float v0 = vals ? vals[q] : 1.f, v1 = vals ? vals[q + G] : 1.f, v2 = vals ? vals[q + 2L * G] : 1.f,
                  v3 = vals ? vals[q + 3L * G] : 1.f;
      acc.x += v0 * w0.x + v1 * w1.x + v2 * w2.x + v3 * w3.x;
      acc.y += v0 * w0.y + v1 * w1.y + v2 * w2.y + v3 * w3.y;
      acc.z += v0 * w0.z + v1 * w1.z + v2 * w2.z + v3 * w3.z;
      acc.w += v0 * w0.w + v1 * w1.w + v2 * w2.w + v3 * w3.w;
    }
    for (; q < end; q += G) {
      const float4 w = *reinterpret_cast<const float4*>(Wt + (long)indices[q] * ldw + 4 * l);
      const float v = vals ? vals[q] : 1.f;
      acc.x += v * w.x; acc.y += v * w.y; acc.z += v * w.z; acc.w += v * w.w;
    }
  }
  part[t] = acc;
  __syncthreads();
  if (t < LPG) {
    float4 s = part[t];
    for (int k = 1; k < G; ++k) {
      const float4 p = part[k * LPG + t];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    float* o = out + (out_idx ? (long)out_idx[j] : j) * ldo + 4 * t;
    if constexpr (ACC) {
      if (beg < end) {
        float4 cur = *reinterpret_cast<float4*>(o);
        cur.x += s.x; cur.y += s.y; cur.z += s.z; cur.w += s.w;
        *reinterpret_cast<float4*>(o) = cur;
      }
      return;
    }
    if (bias) { s.x += bias[4 * t]; s.y += bias[4 * t + 1]; s.z += bias[4 * t + 2]; s.w += bias[4 * t + 3]; }
    o[0] = sbr_act(s.x, act); o[1] = sbr_act(s.y, act); o[2] = sbr_act(s.z, act); o[3] = sbr_act(s.w, act);
  }
}

template <int NC>   // NC = ceil(C / 64) column chunks per lane
__global__ __launch_bounds__(256) void csr_project_bwd_wg_kernel(
    const long* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ dZ, long ldz, const int* __restrict__ rows, float* __restrict__ dWt, long ldw, int C) {
  const long j = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = rows[j];
  const long beg = indptr[r], end = indptr[r + 1];
  float g[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) g[i] = (lane + 64 * i < C) ? dZ[j * ldz + lane + 64 * i] : 0.f;
  for (long base = beg + 64L * wave; base < end; base += 256) {
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    const int my_idx = lane < cnt ? indices[base + lane] : 0;
    const float my_val = (vals && lane < cnt) ? vals[base + lane] : 1.f;
    for (int k = 0; k < cnt; ++k) {
      const long row = (long)__shfl(my_idx, k, 64) * ldw;
      const float v = __shfl(my_val, k, 64);
#pragma unroll
      for (int i = 0; i < NC; ++i)
        if (lane + 64 * i < C) atomicAdd(&dWt[row + lane + 64 * i], v * g[i]);
    }
  }
}

extern "C" int sbr_csr_project_fwd(const long* indptr, const int* indices, const float* vals, const float* Wt, long ldw,
                                   const float* bias, const int* rows, float* out, long ldo, const int* out_idx, long n,
                                   int C, int act, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(indptr && indices && Wt && rows && out, "sbr_csr_project_fwd: null operand");
  if ((C & 3) == 0 && C <= 1024 && (ldw & 3) == 0 && (((uintptr_t)Wt) & 15) == 0) {
    csr_project_fwd_wg_kernel<false><<<(unsigned)n, 256, 0, (hipStream_t)stream>>>(indptr, indices, vals, Wt, ldw, bias, rows, out, ldo,
                                                                           out_idx, C, act);
  } else {
    csr_project_fwd_kernel<<<sbr_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(indptr, indices, vals, Wt, ldw, bias, rows, out,
                                                                            ldo, out_idx, n, C, act);
  }
  SBR_CHECK_LAUNCH("sbr_csr_project_fwd");
  return SBR_OK;
}

extern "C" int sbr_csr_project_bwd(const long* indptr, const int* indices, const float* vals, const float* dZ, long ldz,
                                   const int* rows, float* dWt, long ldw, long n, int C, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(indptr && indices && dZ && rows && dWt, "sbr_csr_project_bwd: null operand");
  hipStream_t s = (hipStream_t)stream;
  const int nc = sbr_cdiv(C, 64);
  switch (nc) {
    case 1: csr_project_bwd_wg_kernel<1><<<(unsigned)n, 256, 0, s>>>(indptr, indices, vals, dZ, ldz, rows, dWt, ldw, C); break;
    case 2: csr_project_bwd_wg_kernel<2><<<(unsigned)n, 256, 0, s>>>(indptr, indices, vals, dZ, ldz, rows, dWt, ldw, C); break;
    case 3: case 4: csr_project_bwd_wg_kernel<4><<<(unsigned)n, 256, 0, s>>>(indptr, indices, vals, dZ, ldz, rows, dWt, ldw, C); break;
    case 5: case 6: case 7: case 8:
      csr_project_bwd_wg_kernel<8><<<(unsigned)n, 256, 0, s>>>(indptr, indices, vals, dZ, ldz, rows, dWt, ldw, C); break;
    default:
      csr_project_bwd_kernel<<<sbr_cdiv(n, 4), 256, 0, s>>>(indptr, indices, vals, dZ, ldz, rows, dWt, ldw, n, C);
  }
  SBR_CHECK_LAUNCH("sbr_csr_project_bwd");
  return SBR_OK;
}

__global__ void rowops_zero_kernel(float4* __restrict__ p, long n4) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) p[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The same gradient in GATHER form: dWt = X^T dZ with X the [slots, n_cols] matrix of the slots' feature rows. The scatter form
// above issues one float atomic per (slot, nnz, column) — Onion18 at batch 4096: 30,805 slots x ~24 nnz x 512 columns = 0.38 G
// atomics, 1.25 ms, the largest kernel of that step. Here the slot gradients are first added up per ENTITY (dZe[rows[j], :] +=
// dZ[j, :]: one atomic per slot and column), and every feature column then sums the rows of the entities that have it — the
// forward kernel run over the transposed matrix, no atomics, one writer per row of dWt, a fixed summation order. Entities outside
// the batch contribute zero rows (read, not skipped: at these batch sizes nearly every entity is in the batch).
// (t_indptr, t_indices, t_vals): the CSR form of the TRANSPOSED feature matrix [n_cols, n_entities]; dZe: workspace
// [n_entities, C] (ldz_e floats per row), overwritten.
// dz_idx (may be null): slot j's gradient row is dZ[dz_idx[j], :]. An EmbeddingBag(mean) over padded tag lists is the same product
// with X[entity, tag] = 1 / (tags of the entity): sbr_bag_mean_bwd in gather form is this entry point over that matrix's transpose.
extern "C" int sbr_csr_project_bwd_gather(const long* t_indptr, const int* t_indices, const float* t_vals, const float* dZ, long ldz,
                                          const int* dz_idx, const int* rows, long n, float* dZe, long lde, long n_entities,
                                          float* dWt, long ldw, long n_cols, int C, void* stream) {
  if (n == 0 || n_cols == 0) return SBR_OK;
  SBR_REQUIRE(t_indptr && t_indices && dZ && rows && dZe && dWt, "sbr_csr_project_bwd_gather: null operand");
  SBR_REQUIRE((C & 3) == 0 && C >= 4 && C <= 1024 && lde == C && (ldw & 3) == 0 && (((uintptr_t)dZe | (uintptr_t)dWt) & 15) == 0 && n_entities >= 1,
              "sbr_csr_project_bwd_gather: needs C %% 4 == 0, C <= 1024, a dense 16-byte aligned workspace (lde = C) and 16-byte aligned gradient rows");
  hipStream_t s = (hipStream_t)stream;
  const long n4 = n_entities * C / 4;
  int zb = (int)sbr_cdiv(n4, 256);
  rowops_zero_kernel<<<zb > 4096 ? 4096 : zb, 256, 0, s>>>((float4*)dZe, n4);
  SBR_CHECK_LAUNCH("sbr_csr_project_bwd_gather (zero)");
  const int rc = sbr_scatter_add_rows(dZ, ldz, dz_idx, rows, dZe, lde, n, C, stream);
  if (rc) return rc;
  csr_project_fwd_wg_kernel<true><<<(unsigned)n_cols, 256, 0, s>>>(t_indptr, t_indices, t_vals, dZe, lde, nullptr, nullptr, dWt, ldw, nullptr, C, 0);
  SBR_CHECK_LAUNCH("sbr_csr_project_bwd_gather");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// dZ[j, :] = dY[ii(j), :] * act'(Y[ii(j), :])       (gather fused with the activation derivative)
// ---------------------------------------------------------------------------------------------------------------
__global__ void act_grad_gather_kernel(const float* __restrict__ dY, const float* __restrict__ Y, long ld,
                                       const int* __restrict__ in_idx, float* __restrict__ dZ, long ldz, long n, int C,
                                       int act) {
  const long total = n * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / C;
    const int c = (int)(e - j * C);
    const long i = (in_idx ? (long)in_idx[j] : j) * ld + c;
    dZ[j * ldz + c] = dY[i] * sbr_act_grad_from_out(Y[i], act);
  }
}

__global__ void act_grad_gather4_kernel(const float* __restrict__ dY, const float* __restrict__ Y, long ld,
                                        const int* __restrict__ in_idx, float* __restrict__ dZ, long ldz, long n, int C4,
                                        int act) {
  const long total = n * C4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / C4;
    const int c = (int)(e - j * C4) * 4;
    const long i = (in_idx ? (long)in_idx[j] : j) * ld + c;
    const float4 g = *reinterpret_cast<const float4*>(dY + i);
    const float4 y = *reinterpret_cast<const float4*>(Y + i);
    float4 o;
    o.x = g.x * sbr_act_grad_from_out(y.x, act); o.y = g.y * sbr_act_grad_from_out(y.y, act);
    o.z = g.z * sbr_act_grad_from_out(y.z, act); o.w = g.w * sbr_act_grad_from_out(y.w, act);
    *reinterpret_cast<float4*>(dZ + j * ldz + c) = o;
  }
}

extern "C" int sbr_act_grad_gather(const float* dY, const float* Y, long ld, const int* in_idx, float* dZ, long ldz,
                                   long n, int C, int act, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dY && Y && dZ, "sbr_act_grad_gather: null operand");
  if ((C & 3) == 0 && (ld & 3) == 0 && (ldz & 3) == 0 && ((((uintptr_t)dY) | ((uintptr_t)Y) | ((uintptr_t)dZ)) & 15) == 0) {
    int blocks4 = sbr_cdiv(n * (C / 4), 256);
    if (blocks4 > 8192) blocks4 = 8192;
    act_grad_gather4_kernel<<<blocks4, 256, 0, (hipStream_t)stream>>>(dY, Y, ld, in_idx, dZ, ldz, n, C / 4, act);
    SBR_CHECK_LAUNCH("sbr_act_grad_gather");
    return SBR_OK;
  }
  int blocks = sbr_cdiv(n * C, 256);
  if (blocks > 4096) blocks = 4096;
  act_grad_gather_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dY, Y, ld, in_idx, dZ, ldz, n, C, act);
  SBR_CHECK_LAUNCH("sbr_act_grad_gather");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[c] = sum_j X[j, c]; double accumulation, one atomic per block and column
// ---------------------------------------------------------------------------------------------------------------
__global__ void colsum_kernel(const float* __restrict__ X, long ld, long n, int C, double* __restrict__ acc) {
  // block = 256 threads as 4 row-groups x 64 columns
  const int cg = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  double s = 0.0;
  if (cg < C)
    for (long j = blockIdx.x * 4L + rg; j < n; j += gridDim.x * 4L) s += (double)X[j * ld + cg];
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && cg < C) atomicAdd(&acc[cg], sm[threadIdx.x] + sm[threadIdx.x + 64] + sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
}

__global__ void d2f_kernel(const double* __restrict__ a, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)a[i];
}

// workspace: C doubles, zeroed by this call
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ X, long ld, long n, int C,
                                                      double* __restrict__ acc) {
  sbr_col_reduce<1>(n, C, acc, [&](long j, int cg, float4* v) { v[0] = *reinterpret_cast<const float4*>(X + j * ld + 4 * cg); });
}

__global__ void colsum_final_kernel(double* __restrict__ ws, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) out[i] = (float)sbr_colred_take(ws, C, i);
}

// workspace: 17*C doubles, ZERO on first use; every call leaves it zeroed again (no memset per call)
extern "C" int sbr_colsum(const float* X, long ld, long n, int C, float* out, double* workspace, void* stream) {
  SBR_REQUIRE(out && workspace, "sbr_colsum: null operand");
  hipStream_t s = (hipStream_t)stream;
  if (n > 0 && sbr_col_reduce_ok(X, ld, C)) {
    colsum4_kernel<<<sbr_col_reduce_blocks(n, C), 256, 0, s>>>(X, ld, n, C, workspace);
  } else if (n > 0) {          // generic path: atomics straight into replica 1
    int bx = sbr_cdiv(n, 64);
    if (bx > 512) bx = 512;
    colsum_kernel<<<dim3(bx, sbr_cdiv(C, 64)), 256, 0, s>>>(X, ld, n, C, workspace + C);
  }
  colsum_final_kernel<<<sbr_cdiv(C, 256), 256, 0, s>>>(workspace, C, out);
  SBR_CHECK_LAUNCH("sbr_colsum");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// F.normalize(x, p=2, dim=-1, eps) (sgd_alg.py:1873-1874): y = x / max(||x||, eps). One wave per row.
// ---------------------------------------------------------------------------------------------------------------
__global__ void l2norm_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ inv_norm,
                                  long n, int C, float eps) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = X[j * C + c]; ss += v * v; }
  ss = sbr_wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  for (int c = lane; c < C; c += 64) Y[j * C + c] = X[j * C + c] * inv;
  if (lane == 0) inv_norm[j] = inv;
}

// dx = inv * (dy - y * (y . dy))  when ||x|| > eps, else dy / eps
__global__ void l2norm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                  const float* __restrict__ inv_norm, float* __restrict__ dX, long n, int C, float eps) {
  const long j = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const float inv = inv_norm[j];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += Y[j * C + c] * dY[j * C + c];
  dot = sbr_wave_sum(dot);
  const bool clamped = inv >= 1.f / eps;
  for (int c = lane; c < C; c += 64) {
    const float g = dY[j * C + c];
    dX[j * C + c] = clamped ? g * inv : inv * (g - Y[j * C + c] * dot);
  }
}

extern "C" int sbr_l2norm_fwd(const float* X, float* Y, float* inv_norm, long n, int C, float eps, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(X && Y && inv_norm, "sbr_l2norm_fwd: null operand");
  l2norm_fwd_kernel<<<sbr_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(X, Y, inv_norm, n, C, eps);
  SBR_CHECK_LAUNCH("sbr_l2norm_fwd");
  return SBR_OK;
}

extern "C" int sbr_l2norm_bwd(const float* dY, const float* Y, const float* inv_norm, float* dX, long n, int C, float eps,
                              void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dY && Y && inv_norm && dX, "sbr_l2norm_bwd: null operand");
  l2norm_bwd_kernel<<<sbr_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(dY, Y, inv_norm, dX, n, C, eps);
  SBR_CHECK_LAUNCH("sbr_l2norm_bwd");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// nn.Dropout(p) (sgd_alg.py:1815, polylinear.py:48) with a counter-based generator: element e is kept iff
// hash(seed, e) >= p * 2^32. The backward pass recomputes the mask from (seed, e); nothing is stored.
// (The reference draws its mask from torch's CPU generator; a GPU mask cannot reproduce that stream — parity tests
// run with dropout disabled, SURVEY.md §7 "Hard parts".)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int sbr_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (unsigned int)(z >> 32);
}

__global__ void dropout_kernel(const float* __restrict__ X, float* __restrict__ Y, long total, float p,
                               unsigned long long seed, const long* __restrict__ seed_dev) {
  if (seed_dev) seed += (unsigned long long)seed_dev[0];     // step seed kept in device memory (hipGraph replay)
  const unsigned int thr = (unsigned int)fminf(p * 4294967296.f, 4294967295.f);
  const float scale = 1.f / (1.f - p);
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    Y[e] = sbr_mix(seed * 0x100000001B3ULL + (unsigned long long)e) >= thr ? X[e] * scale : 0.f;
}

// forward and backward are the same map (y = x * m / (1-p)); pass dY as X for the backward
extern "C" int sbr_dropout(const float* X, float* Y, long total, float p, unsigned long long seed, void* stream) {
  if (total == 0) return SBR_OK;
  SBR_REQUIRE(X && Y, "sbr_dropout: null operand");
  SBR_REQUIRE(p >= 0.f && p < 1.f, "sbr_dropout: p must be in [0, 1)");
  int blocks = sbr_cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  dropout_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(X, Y, total, p, seed, nullptr);
  SBR_CHECK_LAUNCH("sbr_dropout");
  return SBR_OK;
}

// the same map with the step's seed read from device memory at run time: mask = f(seed_dev[0] + seed_offset, element). A
// captured hipGraph replays this launch unchanged while the host refreshes seed_dev[0] before every replay.
extern "C" int sbr_dropout_dev(const float* X, float* Y, long total, float p, const long* seed_dev, long seed_offset,
                               void* stream) {
  if (total == 0) return SBR_OK;
  SBR_REQUIRE(X && Y && seed_dev, "sbr_dropout_dev: null operand");
  SBR_REQUIRE(p >= 0.f && p < 1.f, "sbr_dropout_dev: p must be in [0, 1)");
  int blocks = sbr_cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  dropout_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(X, Y, total, p, (unsigned long long)seed_offset, seed_dev);
  SBR_CHECK_LAUNCH("sbr_dropout_dev");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// aggregation over the k sampled modalities (sgd_alg.py:27-31, 1861): E [S, k, D] -> out [S, D]; mode 0 mean, 1 max
// ---------------------------------------------------------------------------------------------------------------
__global__ void aggregate_fwd_kernel(const float* __restrict__ E, float* __restrict__ out, unsigned char* __restrict__ arg,
                                     long S, int k, int D, int mode) {
  const long total = S * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long s = e / D;
    const int c = (int)(e - s * D);
    const float* p = E + s * k * D + c;
    if (mode == 0) {
      float acc = 0.f;
      for (int q = 0; q < k; ++q) acc += p[(long)q * D];
      out[e] = acc / (float)k;
    } else {
      float best = p[0];
      int bi = 0;
      for (int q = 1; q < k; ++q) {
        const float v = p[(long)q * D];
        if (v > best) { best = v; bi = q; }
      }
      out[e] = best;
      arg[e] = (unsigned char)bi;
    }
  }
}

__global__ void aggregate_bwd_kernel(const float* __restrict__ dOut, const unsigned char* __restrict__ arg,
                                     float* __restrict__ dE, long S, int k, int D, int mode) {
  const long total = S * k * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long s = e / ((long)k * D);
    const int rem = (int)(e - s * k * D);
    const int q = rem / D, c = rem - q * D;
    const float g = dOut[s * D + c];
    dE[e] = mode == 0 ? g / (float)k : (arg[s * D + c] == q ? g : 0.f);
  }
}

extern "C" int sbr_aggregate_fwd(const float* E, float* out, unsigned char* argmax, long S, int k, int D, int mode,
                                 void* stream) {
  if (S == 0) return SBR_OK;
  SBR_REQUIRE(E && out && (mode == 0 || argmax), "sbr_aggregate_fwd: null operand");
  SBR_REQUIRE(k >= 1 && k <= 255, "sbr_aggregate_fwd: k out of range");
  int blocks = sbr_cdiv(S * D, 256);
  if (blocks > 4096) blocks = 4096;
  aggregate_fwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(E, out, argmax, S, k, D, mode);
  SBR_CHECK_LAUNCH("sbr_aggregate_fwd");
  return SBR_OK;
}

extern "C" int sbr_aggregate_bwd(const float* dOut, const unsigned char* argmax, float* dE, long S, int k, int D, int mode,
                                 void* stream) {
  if (S == 0) return SBR_OK;
  SBR_REQUIRE(dOut && dE && (mode == 0 || argmax), "sbr_aggregate_bwd: null operand");
  int blocks = sbr_cdiv(S * k * D, 256);
  if (blocks > 4096) blocks = 4096;
  aggregate_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dOut, argmax, dE, S, k, D, mode);
  SBR_CHECK_LAUNCH("sbr_aggregate_bwd");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// the training scorer einsum('be,bce->bc') (sgd_alg.py:2114): logits[b, n] = u[b, :] . i[b, n, :]
// one wave per (b, n) slot, wavefront shuffle reduction
// ---------------------------------------------------------------------------------------------------------------
__global__ void score_dot_fwd_kernel(const float* __restrict__ U, const float* __restrict__ I, float* __restrict__ out,
                                     long B, int N, int D) {
  const long s = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= B * N) return;
  const int lane = threadIdx.x & 63;
  const long b = s / N;
  float acc = 0.f;
  for (int c = lane; c < D; c += 64) acc += U[b * D + c] * I[s * D + c];
  acc = sbr_wave_sum(acc);
  if (lane == 0) out[s] = acc;
}

// dU[b, :] = sum_n g[b, n] * I[b, n, :] ;  dI[b, n, :] = g[b, n] * U[b, :]      one block per b
__global__ void score_dot_bwd_kernel(const float* __restrict__ G, const float* __restrict__ U,
                                     const float* __restrict__ I, float* __restrict__ dU, float* __restrict__ dI, int N,
                                     int D) {
  const long b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float u = U[b * D + c];
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
      const float g = G[b * N + n];
      acc += g * I[(b * N + n) * D + c];
      if (dI) dI[(b * N + n) * D + c] = g * u;
    }
    if (dU) dU[b * D + c] = acc;
  }
}

// D % 4 == 0, D <= 256: D/4 lanes per slot read one float4 each, 64 / (D/4) slots per wave (power-of-two groups)
template <int LPS>
__global__ void score_dot_fwd4_kernel(const float* __restrict__ U, const float* __restrict__ I, float* __restrict__ out,
                                      long B, int N, int D) {
  constexpr int SPW = 64 / LPS;                               // slots per wave
  const int lane = threadIdx.x & 63, l = lane % LPS, sub = lane / LPS;
  const long wave = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const long s = wave * SPW + sub;
  float acc = 0.f;
  if (s < B * N && 4 * l < D) {
    const long b = s / N;
    const float4 u = *reinterpret_cast<const float4*>(U + b * D + 4 * l);
    const float4 v = *reinterpret_cast<const float4*>(I + s * D + 4 * l);
    acc = u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
#pragma unroll
  for (int o = LPS >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (l == 0 && s < B * N) out[s] = acc;
}

extern "C" int sbr_score_dot_fwd(const float* U, const float* I, float* out, long B, int N, int D, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(U && I && out, "sbr_score_dot_fwd: null operand");
  if ((D & 3) == 0 && D <= 256 && ((((uintptr_t)U) | ((uintptr_t)I)) & 15) == 0) {
    hipStream_t s = (hipStream_t)stream;
    const int lps = D <= 64 ? 16 : (D <= 128 ? 32 : 64);
    const long waves = sbr_cdiv(B * N, 64 / lps);
    const int blocks = sbr_cdiv(waves, 4);
    if (lps == 16) score_dot_fwd4_kernel<16><<<blocks, 256, 0, s>>>(U, I, out, B, N, D);
    else if (lps == 32) score_dot_fwd4_kernel<32><<<blocks, 256, 0, s>>>(U, I, out, B, N, D);
    else score_dot_fwd4_kernel<64><<<blocks, 256, 0, s>>>(U, I, out, B, N, D);
    SBR_CHECK_LAUNCH("sbr_score_dot_fwd");
    return SBR_OK;
  }
  score_dot_fwd_kernel<<<sbr_cdiv(B * N, 4), 256, 0, (hipStream_t)stream>>>(U, I, out, B, N, D);
  SBR_CHECK_LAUNCH("sbr_score_dot_fwd");
  return SBR_OK;
}

extern "C" int sbr_score_dot_bwd(const float* G, const float* U, const float* I, float* dU, float* dI, long B, int N,
                                 int D, void* stream) {
  if (B == 0) return SBR_OK;
  SBR_REQUIRE(G && U && I, "sbr_score_dot_bwd: null operand");
  const int threads = D >= 256 ? 256 : (D > 64 ? 128 : 64);
  score_dot_bwd_kernel<<<(unsigned)B, threads, 0, (hipStream_t)stream>>>(G, U, I, dU, dI, N, D);
  SBR_CHECK_LAUNCH("sbr_score_dot_bwd");
  return SBR_OK;
}

// SGDBaseline (sgd_alg.py:110-119): out[b, n] = user_bias[u[b]] + item_bias[i[b, n]] + global_bias
__global__ void bias_score_kernel(const float* __restrict__ ub, const float* __restrict__ ib, const float* __restrict__ gb,
                                  const long* __restrict__ u, const long* __restrict__ i, float* __restrict__ out, long B,
                                  int N) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= B * N) return;
  out[e] = ub[u[e / N]] + ib[i[e]] + gb[0];
}

extern "C" int sbr_bias_score_fwd(const float* user_bias, const float* item_bias, const float* global_bias, const long* u,
                                  const long* i, float* out, long B, int N, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(user_bias && item_bias && global_bias && u && i && out, "sbr_bias_score_fwd: null operand");
  bias_score_kernel<<<sbr_cdiv(B * N, 256), 256, 0, (hipStream_t)stream>>>(user_bias, item_bias, global_bias, u, i, out, B, N);
  SBR_CHECK_LAUNCH("sbr_bias_score_fwd");
  return SBR_OK;
}

// Dense interaction vectors of a batch of entities (InteractionRecDataset._get_interaction_vectors, data/dataset.py:306-319:
// ``matrix[indices].toarray()``) for the models that consume them densely (DropoutNet's preference network, sgd_alg.py:1693-1725):
// out[j, :] = row ent[j] of the CSR matrix, or all zeros when ent[j] < 0 (an entity whose preferences are dropped).
__global__ void csr_rows_to_dense_kernel(const long* __restrict__ indptr, const int* __restrict__ indices,
                                         const float* __restrict__ data, const long* __restrict__ ent, float* __restrict__ out,
                                         long ldo, int dim) {
  const long j = blockIdx.x;
  float* o = out + j * ldo;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) o[c] = 0.f;
  __syncthreads();
  const long e = ent[j];
  if (e < 0) return;
  for (long p = indptr[e] + threadIdx.x; p < indptr[e + 1]; p += blockDim.x) o[indices[p]] = data ? data[p] : 1.f;
}

extern "C" int sbr_csr_rows_to_dense(const long* indptr, const int* indices, const float* data, const long* ent, long n, int dim,
                                     float* out, long ldo, void* stream) {
  if (n == 0 || dim == 0) return SBR_OK;
  SBR_REQUIRE(indptr && indices && ent && out, "sbr_csr_rows_to_dense: null operand");
  SBR_REQUIRE(ldo >= dim, "sbr_csr_rows_to_dense: row stride %ld < %d columns", ldo, dim);
  csr_rows_to_dense_kernel<<<(unsigned)n, 256, 0, (hipStream_t)stream>>>(indptr, indices, data, ent, out, ldo, dim);
  SBR_CHECK_LAUNCH("sbr_csr_rows_to_dense");
  return SBR_OK;
}

// Bias terms of the factorisation models (SGDMatrixFactorization.combine, sgd_alg.py:186-194; SGDBaseline, :110-119), forward
// and backward: out[b, n] = base[b, n] + user_bias[u[b]] + item_bias[i[b, n]] + global_bias, every term optional. u == null:
// row b itself; i == null: column n itself (all-pairs scoring against an already gathered bias vector).
__global__ void bias_score_add_kernel(const float* __restrict__ ub, const float* __restrict__ ib, const float* __restrict__ gb,
                                      const long* __restrict__ u, const long* __restrict__ i, const float* __restrict__ base,
                                      float* __restrict__ out, long B, int N) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= B * N) return;
  const long b = e / N;
  float v = base ? base[e] : 0.f;
  if (ub) v += ub[u ? u[b] : b];
  if (ib) v += ib[i ? i[e] : e - b * N];
  if (gb) v += gb[0];
  out[e] = v;
}

extern "C" int sbr_bias_score_add_fwd(const float* user_bias, const float* item_bias, const float* global_bias, const long* u,
                                      const long* i, const float* base, float* out, long B, int N, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(out, "sbr_bias_score_add_fwd: null output");
  bias_score_add_kernel<<<sbr_cdiv(B * N, 256), 256, 0, (hipStream_t)stream>>>(user_bias, item_bias, global_bias, u, i, base, out, B, N);
  SBR_CHECK_LAUNCH("sbr_bias_score_add_fwd");
  return SBR_OK;
}

// gradients of the bias tables (ACCUMULATED: zero-initialised by the caller); the gradient of `base` is g itself
__global__ void bias_score_bwd_kernel(const float* __restrict__ g, const long* __restrict__ u, const long* __restrict__ i,
                                      float* __restrict__ d_ub, float* __restrict__ d_ib, float* __restrict__ d_gb, long B, int N) {
  __shared__ float part[4];
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float v = 0.f;
  if (e < B * N) {
    v = g[e];
    const long b = e / N;
    if (d_ub) atomicAdd(&d_ub[u ? u[b] : b], v);
    if (d_ib) atomicAdd(&d_ib[i ? i[e] : e - b * N], v);
  }
  if (d_gb) {                                                 // block sum -> one atomic per block
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(d_gb, part[0] + part[1] + part[2] + part[3]);
  }
}

extern "C" int sbr_bias_score_bwd(const float* g, const long* u, const long* i, float* d_user_bias, float* d_item_bias,
                                  float* d_global_bias, long B, int N, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(g, "sbr_bias_score_bwd: null operand");
  bias_score_bwd_kernel<<<sbr_cdiv(B * N, 256), 256, 0, (hipStream_t)stream>>>(g, u, i, d_user_bias, d_item_bias, d_global_bias, B, N);
  SBR_CHECK_LAUNCH("sbr_bias_score_bwd");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// membership test of (row, col) pairs in a CSR matrix with sorted column indices — the `v in positives` test of the
// negative-sampling collate (data/dataloader.py:184-191), evaluated for all still-colliding slots of a batch at once.
// ---------------------------------------------------------------------------------------------------------------
__global__ void csr_contains_kernel(const long* __restrict__ indptr, const int* __restrict__ indices,
                                    const long* __restrict__ rows, const long* __restrict__ cols, long n,
                                    unsigned char* __restrict__ out) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long r = rows[j];
  const int c = (int)cols[j];
  long lo = indptr[r], hi = indptr[r + 1];
  const long end = hi;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (indices[mid] < c) lo = mid + 1; else hi = mid;
  }
  out[j] = (lo < end && indices[lo] == c) ? 1 : 0;
}

extern "C" int sbr_csr_contains(const long* indptr, const int* indices, const long* rows, const long* cols, long n,
                                unsigned char* out, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(indptr && indices && rows && cols && out, "sbr_csr_contains: null operand");
  csr_contains_kernel<<<sbr_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(indptr, indices, rows, cols, n, out);
  SBR_CHECK_LAUNCH("sbr_csr_contains");
  return SBR_OK;
}
