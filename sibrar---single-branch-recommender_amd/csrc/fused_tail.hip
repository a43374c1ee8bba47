// Fused tail of the training step: trailing BatchNorm1d + per-slot scorer, and column sums folded into their producers.
//
// With one modality per slot (k = 1: no embedding regularisation, the default of the shipped sbnet configs) the item
// representation is the output of the entity's trailing BatchNorm1d (algorithms/sgd_alg.py:1834-1837, 1871-1877) and its only
// consumer is the scorer einsum('be,bce->bc') (sgd_alg.py:2114). Unfused, the [R, D] representation (R = B * N slots) is
// written by bn_apply, read by the scorer, its gradient written by the scorer's backward and read twice by the BatchNorm
// backward: 8 passes of R * D * 4 bytes. Here the normalised rows and their gradient dy[s, :] = dlogits[s] * u[b(s), :] are
// never stored:
//   forward   sbr_bn_train_stats (column statistics, running-stat update)  ->  sbr_bn_score_fwd (reads Z once)
//   backward  pass A: dU and the BatchNorm column sums from Z, U, dlogits (reads Z once);
//             pass B: dX = w rstd (dy - mean(dy) - xhat mean(dy xhat)) (reads Z, writes dX) + column sums of dX — the
//             bias gradient of the Linear in front of the BatchNorm (modules/polylinear.py:51) — folded in.
// Every expression keeps the operand order of the unfused kernels (batchnorm.hip, rowops.hip: score_dot_*), so logits, dU and
// dX are bitwise what the unfused sequence produces; the column sums differ only in fp32 summation order inside a thread.
//
// sbr_act_grad_gather_colsum / sbr_colred_finish: the activation-derivative (+ row gather) kernel also accumulates the column
// sums of its output (the bias gradient of its layer) into a column-reduction workspace; all pending workspaces of a step
// are turned into float vectors by ONE launch at the end of the backward pass.
#include "common.h"

// ---- forward: logits[s] = sum_d U[b, d] * ((Z[s, d] - mean[d]) * rstd[d] * w[d] + beta[d]) ---------------------------------
template <int LPS>
__global__ void bn_score_fwd4_kernel(const float* __restrict__ Z, const float* __restrict__ U, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ beta,
                                     float* __restrict__ out, long B, int N, int D) {
  constexpr int SPW = 64 / LPS;                               // slots per wave
  const int lane = threadIdx.x & 63, l = lane % LPS, sub = lane / LPS;
  const long wave = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const long s = wave * SPW + sub;
  float acc = 0.f;
  if (s < B * N && 4 * l < D) {
    const long b = s / N;
    const float4 u = *reinterpret_cast<const float4*>(U + b * D + 4 * l);
    const float4 z = *reinterpret_cast<const float4*>(Z + s * D + 4 * l);
    const float4 m = *reinterpret_cast<const float4*>(mean + 4 * l), r = *reinterpret_cast<const float4*>(rstd + 4 * l);
    const float4 g = *reinterpret_cast<const float4*>(w + 4 * l), be = *reinterpret_cast<const float4*>(beta + 4 * l);
    const float y0 = (z.x - m.x) * r.x * g.x + be.x, y1 = (z.y - m.y) * r.y * g.y + be.y;
    const float y2 = (z.z - m.z) * r.z * g.z + be.z, y3 = (z.w - m.w) * r.w * g.w + be.w;
    acc = u.x * y0 + u.y * y1 + u.z * y2 + u.w * y3;
  }
#pragma unroll
  for (int o = LPS >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (l == 0 && s < B * N) out[s] = acc;
}

static bool tail_ok(const void* a, const void* b, const void* c, int D) {
  return (D & 3) == 0 && D >= 4 && D <= 256 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

extern "C" int sbr_bn_score_supported(int D) { return (D & 3) == 0 && D >= 4 && D <= 256 && (256 % (D >> 2)) == 0; }

extern "C" int sbr_bn_score_fwd(const float* Z, const float* U, const float* mean, const float* rstd, const float* weight,
                                const float* bias, float* logits, long B, int N, int D, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(Z && U && mean && rstd && weight && bias && logits, "sbr_bn_score_fwd: null operand");
  SBR_REQUIRE(sbr_bn_score_supported(D) && tail_ok(Z, U, mean, D) && tail_ok(rstd, weight, bias, D),
              "sbr_bn_score_fwd: D=%d / alignment not supported (use sbr_bn_train_fwd + sbr_score_dot_fwd)", D);
  hipStream_t s = (hipStream_t)stream;
  const int lps = D <= 64 ? 16 : (D <= 128 ? 32 : 64);
  const long waves = sbr_cdiv(B * N, 64 / lps);
  const int blocks = sbr_cdiv(waves, 4);
  if (lps == 16) bn_score_fwd4_kernel<16><<<blocks, 256, 0, s>>>(Z, U, mean, rstd, weight, bias, logits, B, N, D);
  else if (lps == 32) bn_score_fwd4_kernel<32><<<blocks, 256, 0, s>>>(Z, U, mean, rstd, weight, bias, logits, B, N, D);
  else bn_score_fwd4_kernel<64><<<blocks, 256, 0, s>>>(Z, U, mean, rstd, weight, bias, logits, B, N, D);
  SBR_CHECK_LAUNCH("sbr_bn_score_fwd");
  return SBR_OK;
}

// ---- backward, pass A --------------------------------------------------------------------------------------------------
// block = 256 threads = RL row lanes x D/4 column groups; a row lane walks whole users (N consecutive rows of Z):
//   dU[b, :] = sum_n G[b, n] * y[s, :]                                  (registers, no cross-thread reduction)
//   ws[c] += sum_s dy[s, c],  ws[D + c] += sum_s dy[s, c] * xhat[s, c]   (dy = G[s] * U[b, :]; block-reduced like sbr_col_reduce)
__global__ __launch_bounds__(256) void bn_score_bwd_stats_kernel(const float* __restrict__ G, const float* __restrict__ U,
                                                                 const float* __restrict__ Z, float* __restrict__ dU, long B, int N,
                                                                 int D, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ w, const float* __restrict__ beta,
                                                                 double* __restrict__ ws) {
  __shared__ float4 sm[2][256];
  const int C4 = D >> 2, RL = 256 / C4;
  const int t = threadIdx.x, cg = t % C4, rl = t / C4;
  const long chunk = (B + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * chunk, hi = (lo + chunk < B) ? lo + chunk : B;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (rl < RL) {
    const float4 m = *reinterpret_cast<const float4*>(mean + 4 * cg), r = *reinterpret_cast<const float4*>(rstd + 4 * cg);
    const float4 g = *reinterpret_cast<const float4*>(w + 4 * cg), be = *reinterpret_cast<const float4*>(beta + 4 * cg);
    for (long b = lo + rl; b < hi; b += RL) {
      const float4 u = *reinterpret_cast<const float4*>(U + b * D + 4 * cg);
      float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* gp = G + b * N;
      const float* zp = Z + (b * N) * D + 4 * cg;
#pragma unroll 4
      for (int n = 0; n < N; ++n) {
        const float gn = gp[n];
        const float4 z = *reinterpret_cast<const float4*>(zp + (long)n * D);
        const float x0 = (z.x - m.x) * r.x, x1 = (z.y - m.y) * r.y, x2 = (z.z - m.z) * r.z, x3 = (z.w - m.w) * r.w;
        du.x += gn * (x0 * g.x + be.x); du.y += gn * (x1 * g.y + be.y);
        du.z += gn * (x2 * g.z + be.z); du.w += gn * (x3 * g.w + be.w);
        const float d0 = gn * u.x, d1 = gn * u.y, d2 = gn * u.z, d3 = gn * u.w;
        a0.x += d0; a0.y += d1; a0.z += d2; a0.w += d3;
        a1.x += d0 * x0; a1.y += d1 * x1; a1.z += d2 * x2; a1.w += d3 * x3;
      }
      if (dU) *reinterpret_cast<float4*>(dU + b * D + 4 * cg) = du;
    }
  }
  sm[0][t] = a0;
  sm[1][t] = a1;
  __syncthreads();
  if (t < C4) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int q = 0; q < RL; ++q) {
        const float4 p = sm[k][q * C4 + t];
        s0 += (double)p.x; s1 += (double)p.y; s2 += (double)p.z; s3 += (double)p.w;
      }
      double* o = ws + (long)(1 + (blockIdx.x % SBR_COLRED_REP)) * 2 * D + (long)k * D + 4 * t;
      atomicAdd(o, s0); atomicAdd(o + 1, s1); atomicAdd(o + 2, s2); atomicAdd(o + 3, s3);
    }
  }
}

// ---- backward, pass B: dX (+ its column sums into ws2 when given) ------------------------------------------------------------
template <bool COLSUM>
__global__ __launch_bounds__(256) void bn_score_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ U,
                                                                 const float* __restrict__ Z, float* __restrict__ dX, long R, int N,
                                                                 int D, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ w, const double* __restrict__ ws,
                                                                 float* __restrict__ dW, float* __restrict__ dBeta,
                                                                 double* __restrict__ ws2) {
  const int C4 = D >> 2;
  const int cg0 = threadIdx.x % C4;
  const float4 m = *reinterpret_cast<const float4*>(mean + 4 * cg0), r = *reinterpret_cast<const float4*>(rstd + 4 * cg0);
  const float4 g = *reinterpret_cast<const float4*>(w + 4 * cg0);
  const double inv_n = 1.0 / (double)R;
  float4 mdz, mdzx;
  mdz.x = (float)(ws[4 * cg0] * inv_n); mdz.y = (float)(ws[4 * cg0 + 1] * inv_n);
  mdz.z = (float)(ws[4 * cg0 + 2] * inv_n); mdz.w = (float)(ws[4 * cg0 + 3] * inv_n);
  mdzx.x = (float)(ws[D + 4 * cg0] * inv_n); mdzx.y = (float)(ws[D + 4 * cg0 + 1] * inv_n);
  mdzx.z = (float)(ws[D + 4 * cg0 + 2] * inv_n); mdzx.w = (float)(ws[D + 4 * cg0 + 3] * inv_n);
  if (blockIdx.x == 0 && threadIdx.x < C4) {                 // the affine gradients of the BatchNorm
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dBeta[4 * cg0 + q] = (float)ws[4 * cg0 + q];
      dW[4 * cg0 + q] = (float)ws[D + 4 * cg0 + q];
    }
  }
  auto row = [&](long j, int cg, float4* v) {
    const long b = j / N;
    const float gn = G[j];
    const float4 u = *reinterpret_cast<const float4*>(U + b * D + 4 * cg);
    const float4 z = *reinterpret_cast<const float4*>(Z + j * D + 4 * cg);
    float4 o;
    o.x = g.x * r.x * (gn * u.x - mdz.x - (z.x - m.x) * r.x * mdzx.x);
    o.y = g.y * r.y * (gn * u.y - mdz.y - (z.y - m.y) * r.y * mdzx.y);
    o.z = g.z * r.z * (gn * u.z - mdz.z - (z.z - m.z) * r.z * mdzx.z);
    o.w = g.w * r.w * (gn * u.w - mdz.w - (z.w - m.w) * r.w * mdzx.w);
    *reinterpret_cast<float4*>(dX + j * D + 4 * cg) = o;
    v[0] = o;
  };
  if constexpr (COLSUM) {
    sbr_col_reduce<1>(R, D, ws2, row);
  } else {
    const int RL = 256 / C4, rl = threadIdx.x / C4;
    const long chunk = (R + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * chunk, hi = (lo + chunk < R) ? lo + chunk : R;
    float4 v[1];
    if (rl < RL)
      for (long j = lo + rl; j < hi; j += RL) row(j, cg0, v);
  }
}

extern "C" int sbr_bn_score_bwd_stats(const float* G, const float* U, const float* Z, float* dU, long B, int N, int D,
                                      const float* weight, const float* bias, const float* save_mean, const float* save_rstd,
                                      double* ws, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(G && U && Z && weight && bias && save_mean && save_rstd && ws, "sbr_bn_score_bwd_stats: null operand");
  SBR_REQUIRE(sbr_bn_score_supported(D) && tail_ok(Z, U, save_mean, D) && tail_ok(save_rstd, weight, bias, D) && tail_ok(dU, dU, dU, D),
              "sbr_bn_score_bwd_stats: D=%d / alignment not supported", D);
  hipStream_t s = (hipStream_t)stream;
  const int RL = 256 / (D >> 2);
  long blocks = (B + 2L * RL - 1) / (2L * RL);               // >= 2 users per row lane
  { const long cap = getenv("SBR_COLRED_BLOCKS") ? atol(getenv("SBR_COLRED_BLOCKS")) : 512; if (blocks > cap) blocks = cap; }
  if (blocks < 1) blocks = 1;
  bn_score_bwd_stats_kernel<<<(int)blocks, 256, 0, s>>>(G, U, Z, dU, B, N, D, save_mean, save_rstd, weight, bias, ws);
  SBR_CHECK_LAUNCH("sbr_bn_score_bwd_stats");
  sbr_colred_final_kernel<<<sbr_cdiv(2 * D, 256), 256, 0, s>>>(ws, 2 * D);
  SBR_CHECK_LAUNCH("sbr_bn_score_bwd_stats/final");
  return SBR_OK;
}

extern "C" int sbr_bn_score_bwd_apply(const float* G, const float* U, const float* Z, float* dX, long B, int N, int D,
                                      const float* weight, const float* save_mean, const float* save_rstd, const double* ws,
                                      float* dWeight, float* dBias, double* ws_colsum, void* stream) {
  if (B * N == 0) return SBR_OK;
  SBR_REQUIRE(G && U && Z && dX && weight && save_mean && save_rstd && ws && dWeight && dBias, "sbr_bn_score_bwd_apply: null operand");
  SBR_REQUIRE(sbr_bn_score_supported(D) && tail_ok(Z, U, save_mean, D) && tail_ok(save_rstd, weight, dX, D),
              "sbr_bn_score_bwd_apply: D=%d / alignment not supported", D);
  hipStream_t s = (hipStream_t)stream;
  const long R = B * N;
  const int blocks = sbr_col_reduce_blocks(R, D);
  if (ws_colsum)
    bn_score_bwd_apply_kernel<true><<<blocks, 256, 0, s>>>(G, U, Z, dX, R, N, D, save_mean, save_rstd, weight, ws, dWeight, dBias, ws_colsum);
  else
    bn_score_bwd_apply_kernel<false><<<blocks, 256, 0, s>>>(G, U, Z, dX, R, N, D, save_mean, save_rstd, weight, ws, dWeight, dBias, nullptr);
  SBR_CHECK_LAUNCH("sbr_bn_score_bwd_apply");
  return SBR_OK;
}

// ---- activation derivative (+ row gather) with the column sums of its output ---------------------------------------------
__global__ __launch_bounds__(256) void act_grad_colsum4_kernel(const float* __restrict__ dY, const float* __restrict__ Y, long ld,
                                                               const int* __restrict__ in_idx, float* __restrict__ dZ, long ldz,
                                                               long n, int C, int act, double* __restrict__ ws) {
  sbr_col_reduce<1>(n, C, ws, [&](long j, int cg, float4* v) {
    const long i = (in_idx ? (long)in_idx[j] : j) * ld + 4 * cg;
    const float4 g = *reinterpret_cast<const float4*>(dY + i);
    const float4 y = *reinterpret_cast<const float4*>(Y + i);
    float4 o;
    o.x = g.x * sbr_act_grad_from_out(y.x, act); o.y = g.y * sbr_act_grad_from_out(y.y, act);
    o.z = g.z * sbr_act_grad_from_out(y.z, act); o.w = g.w * sbr_act_grad_from_out(y.w, act);
    *reinterpret_cast<float4*>(dZ + j * ldz + 4 * cg) = o;
    v[0] = o;
  });
}

extern "C" int sbr_act_grad_colsum_supported(int C) { return (C & 3) == 0 && C >= 4 && C <= 1024 && (256 % (C >> 2)) == 0; }

extern "C" int sbr_act_grad_gather_colsum(const float* dY, const float* Y, long ld, const int* in_idx, float* dZ, long ldz,
                                          long n, int C, int act, double* ws, void* stream) {
  if (n == 0) return SBR_OK;
  SBR_REQUIRE(dY && Y && dZ && ws, "sbr_act_grad_gather_colsum: null operand");
  SBR_REQUIRE(sbr_act_grad_colsum_supported(C) && (ld & 3) == 0 && (ldz & 3) == 0 &&
                  ((((uintptr_t)dY) | ((uintptr_t)Y) | ((uintptr_t)dZ)) & 15) == 0,
              "sbr_act_grad_gather_colsum: C=%d / alignment not supported (use sbr_act_grad_gather + sbr_colsum)", C);
  act_grad_colsum4_kernel<<<sbr_col_reduce_blocks(n, C), 256, 0, (hipStream_t)stream>>>(dY, Y, ld, in_idx, dZ, ldz, n, C, act, ws);
  SBR_CHECK_LAUNCH("sbr_act_grad_gather_colsum");
  return SBR_OK;
}

// ---- one launch turns up to 8 pending column-reduction workspaces (K = 1) into float vectors -------------------------------
struct ColredFin {
  double* ws[8];
  float* out[8];
  int C[8];
};

__global__ void colred_finish_kernel(ColredFin f) {
  const int q = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < f.C[q]) f.out[q][i] = (float)sbr_colred_take(f.ws[q], f.C[q], i);
}

extern "C" int sbr_colred_finish(int count, const void* const* workspaces, const void* const* outs, const int* widths, void* stream) {
  if (count == 0) return SBR_OK;
  SBR_REQUIRE(count >= 1 && count <= 8 && workspaces && outs && widths, "sbr_colred_finish: 1..8 reductions per call");
  ColredFin f;
  int cmax = 0;
  for (int q = 0; q < count; ++q) {
    SBR_REQUIRE(workspaces[q] && outs[q] && widths[q] >= 1, "sbr_colred_finish: null entry %d", q);
    f.ws[q] = (double*)workspaces[q];
    f.out[q] = (float*)outs[q];
    f.C[q] = widths[q];
    cmax = widths[q] > cmax ? widths[q] : cmax;
  }
  colred_finish_kernel<<<dim3(sbr_cdiv(cmax, 256), count), 256, 0, (hipStream_t)stream>>>(f);
  SBR_CHECK_LAUNCH("sbr_colred_finish");
  return SBR_OK;
}
