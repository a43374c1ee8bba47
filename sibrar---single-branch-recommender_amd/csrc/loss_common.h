// Pieces of the recommendation losses shared by loss.hip and the fused scorer + loss kernel of fused_tail.hip: the loss kinds, the
// float64 BCE-with-logits terms (train/rec_losses.py: labels are float64, so BCE and BPR evaluate in float64) and the block sum.
#pragma once
#include "common.h"

#define LOSS_BCE 0
#define LOSS_BPR 1
#define LOSS_SSM 2

__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  v = sbr_wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
  return t;   // valid on thread 0
}

// softplus-form BCE-with-logits term: max(x,0) - x*y + log1p(exp(-|x|))
__device__ __forceinline__ double bce_term(double x, double y) { return fmax(x, 0.0) - x * y + log1p(exp(-fabs(x))); }
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

