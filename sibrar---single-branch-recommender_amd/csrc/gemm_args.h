// Argument block shared by the two fp32 GEMM kernels (gemm_f32.hip: register-staged tiles, any shape / alignment;
// gemm_ring_f32.hip: persistent LDS-DMA ring for 16-byte aligned operands).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A; long lda; const int* a_idx;
  const float* B; long ldb; const int* b_idx;
  const float* bias;
  float* C; long ldc; const int* c_idx;
  int M, N, K;
  int act;
  int k_chunk;
  int atomic;
  float* slab;      // TN split-K: partial tiles are stored to slab[z][M][N] (plain stores) and summed by a second kernel
  int vecA, vecB;
  int mt, nt;
  int xcd_map;      // 1: XCD-aware panel map (many row panels); 0: plain map (few tiles, split-K spreads the XCDs)
  int splits;       // grid z of the tile kernel / number of K ranges of the ring kernel
};


// gemm_ring_f32.hip. mode: 0 NT, 1 NN, 2 TN. Returns -1 when the shape / alignment is not eligible (caller falls back to
// the tile kernel), otherwise SBR_OK / an error code. g.k_chunk and g.splits must be set; g.slab / g.atomic as for the tile
// kernel.
int sbr_gemm_ring_launch(int mode, GemmArgs& g, hipStream_t s);

// gemm_split_tn_f32.hip: weight-gradient products (M = 128 i, N = 128 j, long K) on the bf16 matrix pipe (exact three-way split of both operands, six MFMA terms), one slab
// per workgroup. sbr_tn_split_splits: slabs it would write (0: shape not eligible or SBR_GEMM_SPLIT=0 / SBR_TN_SPLIT=0).
int sbr_tn_split_splits(int M, int N, int K);
int sbr_tn_split_launch(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M, int N, int K,
                        float* slab, int* splits_out, hipStream_t s);
