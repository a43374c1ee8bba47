// BatchNorm1d over the row dimension of an [R, D] activation matrix, fused with the following activation
// (modules/polylinear.py:61-65, 68; algorithms/sgd_alg.py:1837). torch defaults: eps 1e-5, momentum 0.1, the running
// variance is the unbiased batch variance, normalisation uses the biased one.
//
// Train forward:  pass 1 column sums (sum x, sum x^2) in double -> ws[2*D] ; pass 2 normalise + affine + activation.
// Train backward: pass 1 column sums of dz and dz*xhat        -> ws[2*D] ; pass 2 dx = w*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)).
// HBM-bound: forward reads X twice and writes Y once (12 B/element), backward reads X, Y, dY twice and writes dX.
#include "common.h"

// grid: (row chunks, column groups of 64); block 256 = 4 row lanes x 64 columns
__global__ void bn_stats_kernel(const float* __restrict__ X, long n, int D, double* __restrict__ ws) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  double s = 0.0, ss = 0.0;
  if (c < D)
    for (long j = blockIdx.x * 4L + rg; j < n; j += gridDim.x * 4L) {
      const double v = (double)X[j * D + c];
      s += v;
      ss += v * v;
    }
  __shared__ double sm[2][256];
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = ss;
  __syncthreads();
  if (rg == 0 && c < D) {
    const int t = threadIdx.x;
    atomicAdd(&ws[c], sm[0][t] + sm[0][t + 64] + sm[0][t + 128] + sm[0][t + 192]);
    atomicAdd(&ws[D + c], sm[1][t] + sm[1][t + 64] + sm[1][t + 128] + sm[1][t + 192]);
  }
}

__global__ __launch_bounds__(256) void bn_stats4_kernel(const float* __restrict__ X, long n, int D, double* __restrict__ ws) {
  sbr_col_reduce<2>(n, D, ws, [&](long j, int cg, float4* v) {
    const float4 x = *reinterpret_cast<const float4*>(X + j * D + 4 * cg);
    v[0] = x;
    v[1] = make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w);
  });
}

// one thread per column: batch mean / rstd, running-stat update
__global__ void bn_finalize_kernel(double* __restrict__ ws, long n, int D, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, long* __restrict__ num_batches) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches) num_batches[0] += 1;
  if (c >= D) return;
  const double m = sbr_colred_take(ws, 2 * D, c) / (double)n;                 // also re-zeroes the replicas
  double var = sbr_colred_take(ws, 2 * D, D + c) / (double)n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_apply_kernel(const float* __restrict__ X, float* __restrict__ Y, long n, int D,
                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ w, const float* __restrict__ b, int act) {
  const long total = n * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    Y[e] = sbr_act((X[e] - mean[c]) * rstd[c] * w[c] + b[c], act);
  }
}

// eval mode: running statistics
__global__ void bn_eval_kernel(const float* __restrict__ X, float* __restrict__ Y, long n, int D,
                               const float* __restrict__ rm, const float* __restrict__ rv, const float* __restrict__ w,
                               const float* __restrict__ b, float eps, int act) {
  const long total = n * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    Y[e] = sbr_act((X[e] - rm[c]) / sqrtf(rv[c] + eps) * w[c] + b[c], act);
  }
}

static int grid1d(long total) {
  int b = sbr_cdiv(total, 256);
  return b > 4096 ? 4096 : (b < 1 ? 1 : b);
}

// workspace ws: 34*D doubles (2*D totals + 16 replicas, see sbr_col_reduce in common.h), ZERO on first use; every call
// leaves the replica part zeroed again (no memset per call)
static int bn_train_stats(const float* X, long n, int D, float* running_mean, float* running_var, long* num_batches_tracked,
                          float* save_mean, float* save_rstd, double* ws, float eps, float momentum, hipStream_t s) {
  if (sbr_col_reduce_ok(X, D, D)) {
    bn_stats4_kernel<<<sbr_col_reduce_blocks(n, D), 256, 0, s>>>(X, n, D, ws);
  } else {                       // generic path: atomics straight into replica 1
    int bx = sbr_cdiv(n, 64);
    if (bx > 512) bx = 512;
    bn_stats_kernel<<<dim3(bx, sbr_cdiv(D, 64)), 256, 0, s>>>(X, n, D, ws + 2 * D);
  }
  SBR_CHECK_LAUNCH("sbr_bn_train_fwd/stats");
  bn_finalize_kernel<<<sbr_cdiv(D, 256), 256, 0, s>>>(ws, n, D, eps, momentum, save_mean, save_rstd, running_mean,
                                                      running_var, num_batches_tracked);
  SBR_CHECK_LAUNCH("sbr_bn_train_fwd/finalize");
  return SBR_OK;
}

extern "C" int sbr_bn_train_fwd(const float* X, float* Y, long n, int D, const float* weight, const float* bias,
                                float* running_mean, float* running_var, long* num_batches_tracked, float* save_mean,
                                float* save_rstd, double* ws, float eps, float momentum, int act, void* stream) {
  SBR_REQUIRE(X && Y && weight && bias && save_mean && save_rstd && ws, "sbr_bn_train_fwd: null operand");
  SBR_REQUIRE(n >= 1, "sbr_bn_train_fwd: BatchNorm needs at least one row");
  hipStream_t s = (hipStream_t)stream;
  const int rc = bn_train_stats(X, n, D, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, ws, eps, momentum, s);
  if (rc != SBR_OK) return rc;
  bn_apply_kernel<<<grid1d(n * D), 256, 0, s>>>(X, Y, n, D, save_mean, save_rstd, weight, bias, act);
  SBR_CHECK_LAUNCH("sbr_bn_train_fwd/apply");
  return SBR_OK;
}

// statistics half of sbr_bn_train_fwd (batch mean / rstd, running-statistics update) without the normalising pass: the
// consumer applies the normalisation itself (fused_tail.hip: sbr_bn_score_fwd)
extern "C" int sbr_bn_train_stats(const float* X, long n, int D, float* running_mean, float* running_var,
                                  long* num_batches_tracked, float* save_mean, float* save_rstd, double* ws, float eps,
                                  float momentum, void* stream) {
  SBR_REQUIRE(X && save_mean && save_rstd && ws, "sbr_bn_train_stats: null operand");
  SBR_REQUIRE(n >= 1, "sbr_bn_train_stats: BatchNorm needs at least one row");
  return bn_train_stats(X, n, D, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, ws, eps, momentum,
                        (hipStream_t)stream);
}

// second half of sbr_bn_train_stats for statistics that a producer kernel has already left pending in `ws` (the NT GEMM in front
// of the BatchNorm: sbr_gemm_split_f32 mode 0 with colsum_ws): batch mean / rstd of the n rows, running-statistics update
extern "C" int sbr_bn_finalize_stats(long n, int D, float* running_mean, float* running_var, long* num_batches_tracked,
                                     float* save_mean, float* save_rstd, double* ws, float eps, float momentum, void* stream) {
  SBR_REQUIRE(save_mean && save_rstd && ws && n >= 1 && D >= 1, "sbr_bn_finalize_stats: bad arguments");
  bn_finalize_kernel<<<sbr_cdiv(D, 256), 256, 0, (hipStream_t)stream>>>(ws, n, D, eps, momentum, save_mean, save_rstd, running_mean,
                                                                        running_var, num_batches_tracked);
  SBR_CHECK_LAUNCH("sbr_bn_finalize_stats");
  return SBR_OK;
}

extern "C" int sbr_bn_eval_fwd(const float* X, float* Y, long n, int D, const float* weight, const float* bias,
                               const float* running_mean, const float* running_var, float eps, int act, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(X && Y && weight && bias && running_mean && running_var, "sbr_bn_eval_fwd: null operand");
  bn_eval_kernel<<<grid1d(n * D), 256, 0, (hipStream_t)stream>>>(X, Y, n, D, running_mean, running_var, weight, bias, eps, act);
  SBR_CHECK_LAUNCH("sbr_bn_eval_fwd");
  return SBR_OK;
}

// ---- backward ------------------------------------------------------------------------------------------------------
// dz = dY * act'(Y);  ws[c] = sum dz ; ws[D + c] = sum dz * xhat
__global__ void bn_bwd_stats_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ X,
                                    long n, int D, const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                    double* __restrict__ ws) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  double s = 0.0, sx = 0.0;
  if (c < D) {
    const float m = mean[c], r = rstd[c];
    for (long j = blockIdx.x * 4L + rg; j < n; j += gridDim.x * 4L) {
      const long e = j * D + c;
      const float dz = dY[e] * sbr_act_grad_from_out(Y[e], act);
      s += (double)dz;
      sx += (double)(dz * ((X[e] - m) * r));
    }
  }
  __shared__ double sm[2][256];
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = sx;
  __syncthreads();
  if (rg == 0 && c < D) {
    const int t = threadIdx.x;
    atomicAdd(&ws[c], sm[0][t] + sm[0][t + 64] + sm[0][t + 128] + sm[0][t + 192]);
    atomicAdd(&ws[D + c], sm[1][t] + sm[1][t + 64] + sm[1][t + 128] + sm[1][t + 192]);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_stats4_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                            const float* __restrict__ X, long n, int D,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int act, double* __restrict__ ws) {
  const int cg0 = threadIdx.x % (D >> 2);
  const float4 m = *reinterpret_cast<const float4*>(mean + 4 * cg0), r = *reinterpret_cast<const float4*>(rstd + 4 * cg0);
  sbr_col_reduce<2>(n, D, ws, [&](long j, int cg, float4* v) {
    const long e = j * D + 4 * cg;
    const float4 g = *reinterpret_cast<const float4*>(dY + e), y = *reinterpret_cast<const float4*>(Y + e),
                 x = *reinterpret_cast<const float4*>(X + e);
    float4 dz;
    dz.x = g.x * sbr_act_grad_from_out(y.x, act); dz.y = g.y * sbr_act_grad_from_out(y.y, act);
    dz.z = g.z * sbr_act_grad_from_out(y.z, act); dz.w = g.w * sbr_act_grad_from_out(y.w, act);
    v[0] = dz;
    v[1] = make_float4(dz.x * ((x.x - m.x) * r.x), dz.y * ((x.y - m.y) * r.y), dz.z * ((x.z - m.z) * r.z),
                       dz.w * ((x.w - m.w) * r.w));
  });
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ X,
                                    float* __restrict__ dX, long n, int D, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ w, int act,
                                    const double* __restrict__ ws, float* __restrict__ dW, float* __restrict__ dB) {
  const long total = n * D;
  const double inv_n = 1.0 / (double)n;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const float dz = dY[e] * sbr_act_grad_from_out(Y[e], act);
    const float xhat = (X[e] - mean[c]) * rstd[c];
    const float mdz = (float)(ws[c] * inv_n), mdzx = (float)(ws[D + c] * inv_n);
    dX[e] = w[c] * rstd[c] * (dz - mdz - xhat * mdzx);
    if (e < D) {       // first row's threads also publish the affine gradients
      dB[c] = (float)ws[c];
      dW[c] = (float)ws[D + c];
    }
  }
}

extern "C" int sbr_bn_train_bwd(const float* dY, const float* Y, const float* X, float* dX, long n, int D,
                                const float* weight, const float* save_mean, const float* save_rstd, float* dWeight,
                                float* dBias, double* ws, int act, void* stream) {
  SBR_REQUIRE(dY && Y && X && dX && weight && save_mean && save_rstd && dWeight && dBias && ws, "sbr_bn_train_bwd: null operand");
  SBR_REQUIRE(n >= 1, "sbr_bn_train_bwd: empty batch");
  hipStream_t s = (hipStream_t)stream;
  if (sbr_col_reduce_ok(X, D, D) && ((((uintptr_t)dY) | ((uintptr_t)Y) | ((uintptr_t)save_mean) | ((uintptr_t)save_rstd)) & 15) == 0) {
    bn_bwd_stats4_kernel<<<sbr_col_reduce_blocks(n, D), 256, 0, s>>>(dY, Y, X, n, D, save_mean, save_rstd, act, ws);
  } else {                       // generic path: atomics straight into replica 1
    int bx = sbr_cdiv(n, 64);
    if (bx > 512) bx = 512;
    bn_bwd_stats_kernel<<<dim3(bx, sbr_cdiv(D, 64)), 256, 0, s>>>(dY, Y, X, n, D, save_mean, save_rstd, act, ws + 2 * D);
  }
  sbr_colred_final_kernel<<<sbr_cdiv(2 * D, 256), 256, 0, s>>>(ws, 2 * D);
  SBR_CHECK_LAUNCH("sbr_bn_train_bwd/stats");
  bn_bwd_apply_kernel<<<grid1d(n * D), 256, 0, s>>>(dY, Y, X, dX, n, D, save_mean, save_rstd, weight, act, ws, dWeight, dBias);
  SBR_CHECK_LAUNCH("sbr_bn_train_bwd/apply");
  return SBR_OK;
}
