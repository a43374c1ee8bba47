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
#include "loss_common.h"

// The three kernels that walk the rows of Z (scorer forward, backward pass A, and the fused forward + loss + pass A kernel) go
// through these functions — explicit fused multiply-adds, no contraction left to the compiler — so that they produce the same bits
// whatever code surrounds them (left to itself hipcc contracts a * b + c differently from one kernel to the next).
__device__ __forceinline__ float4 bns_xhat(const float4 z, const float4 m, const float4 r) {
#pragma clang fp contract(off)
  return make_float4((z.x - m.x) * r.x, (z.y - m.y) * r.y, (z.z - m.z) * r.z, (z.w - m.w) * r.w);
}
// this thread's four columns of the logit: sum_c u[c] * (xhat[c] * g[c] + beta[c])
__device__ __forceinline__ float bns_dot(const float4 xh, const float4 g, const float4 be, const float4 u) {
#pragma clang fp contract(off)
  const float y0 = fmaf(xh.x, g.x, be.x), y1 = fmaf(xh.y, g.y, be.y), y2 = fmaf(xh.z, g.z, be.z), y3 = fmaf(xh.w, g.w, be.w);
  return fmaf(u.w, y3, fmaf(u.z, y2, fmaf(u.y, y1, u.x * y0)));
}
// one slot row of backward pass A: du += gn * y, a0 += dy, a1 += dy * xhat with dy = gn * u
__device__ __forceinline__ void bns_pass_a(const float gn, const float4 xh, const float4 g, const float4 be, const float4 u, float4& du,
                                           float4& a0, float4& a1) {
#pragma clang fp contract(off)
  du.x = fmaf(gn, fmaf(xh.x, g.x, be.x), du.x); du.y = fmaf(gn, fmaf(xh.y, g.y, be.y), du.y);
  du.z = fmaf(gn, fmaf(xh.z, g.z, be.z), du.z); du.w = fmaf(gn, fmaf(xh.w, g.w, be.w), du.w);
  const float d0 = gn * u.x, d1 = gn * u.y, d2 = gn * u.z, d3 = gn * u.w;
  a0.x += d0; a0.y += d1; a0.z += d2; a0.w += d3;
  a1.x = fmaf(d0, xh.x, a1.x); a1.y = fmaf(d1, xh.y, a1.y); a1.z = fmaf(d2, xh.z, a1.z); a1.w = fmaf(d3, xh.w, a1.w);
}

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
    acc = bns_dot(bns_xhat(z, m, r), g, be, u);
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
        bns_pass_a(gn, bns_xhat(z, m, r), g, be, u, du, a0, a1);
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

// ---- forward + loss + backward pass A in ONE launch ---------------------------------------------------------------------------
// sbr_bn_score_fwd, the recommendation loss (loss.hip: rec_loss_kernel) and sbr_bn_score_bwd_stats (+ its replica finisher) walk
// the same rows: a user's N slot rows of Z and its row of U give the N logits, the logits give the loss term and dlogits of the
// user, and those with the SAME Z rows give dU and the BatchNorm column sums. Here a row lane (D / 4 threads, as in pass A) keeps
// the user's N normalised rows in registers between the two uses: Z is read once instead of twice, logits / dlogits never make a
// round trip through memory between kernels, and four launches (forward 16 us, loss 12.5 us, pass A 14.7 us, finisher 5.9 us at
// the bench's shape) become one. Arithmetic: every expression and every reduction order of the three kernels is kept — partial dot
// per thread and xor butterfly as in bn_score_fwd4_kernel, the loss formulas of rec_loss_kernel element by element (thread j of a
// row lane owns logit j; BPR's positive-column gradient is summed in column order), du / column sums as in pass A.
// The last block to arrive (agent-scope counter in lws[0]) adds the block partial sums of the loss in block order, writes loss_out
// and the packed (total, rec, reg = 0) scalars, turns the column-sum replicas into totals (leaving the replicas zeroed) and resets
// the counter. Needs N <= D / 4 and N <= NMAX.
// (N is a template parameter: with a run-time N every "n < N" of the unrolled slot loops is a wave-uniform predicate kept in SGPRs,
// 40 - 110 of which spilled)
template <int NMAX, int KIND>
__global__ __launch_bounds__(256) void bn_score_loss_kernel(const float* __restrict__ Z, const float* __restrict__ U,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, const float* __restrict__ beta,
                                                            const double* __restrict__ labels, double scale, float shift,
                                                            float* __restrict__ logits, float* __restrict__ dlogits,
                                                            float* __restrict__ dU, long B, int D, double* __restrict__ ws,
                                                            double* __restrict__ lws, double* __restrict__ loss_out,
                                                            double* __restrict__ out3) {
  constexpr int N = NMAX;
  __shared__ float4 sm[2][256];
  __shared__ double lsm[4];
  __shared__ int last_flag;
  const int C4 = D >> 2, RL = 256 / C4;
  const int t = threadIdx.x, cg = t % C4, rl = t / C4;
  const int lane = t & 63, lane0 = lane - (lane % C4);          // first lane of this row lane inside its wave
  const long chunk = (B + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * chunk, hi = (lo + chunk < B) ? lo + chunk : B;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  double lacc = 0.0;
  const double up = scale;
  if (rl < RL) {
    const float4 m = *reinterpret_cast<const float4*>(mean + 4 * cg), r = *reinterpret_cast<const float4*>(rstd + 4 * cg);
    const float4 g = *reinterpret_cast<const float4*>(w + 4 * cg), be = *reinterpret_cast<const float4*>(beta + 4 * cg);
    for (long b = lo + rl; b < hi; b += RL) {
      const float4 u = *reinterpret_cast<const float4*>(U + b * D + 4 * cg);
      const float* zp = Z + (b * N) * D + 4 * cg;
      float4 xh[NMAX];
      float x[NMAX];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {
          xh[n] = bns_xhat(*reinterpret_cast<const float4*>(zp + (long)n * D), m, r);
        }
      }
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        x[n] = 0.f;
        if (n < N) x[n] = bns_dot(xh[n], g, be, u);
      }
      // xor butterfly over the row lane, all N sums per step (N cross-lane reads in flight behind one wait: reduced one after the
      // other, the 5 x N dependent ds_bpermute round trips were most of the kernel's time)
      for (int o = C4 >> 1; o > 0; o >>= 1) {
        float tx[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) tx[n] = __shfl_xor(x[n], o, 64);
#pragma unroll
        for (int n = 0; n < NMAX; ++n) x[n] += tx[n];            // x[n]: the logit of slot (b, n), in every thread of the row lane
      }
      // ---- loss term and dlogits: thread cg owns logit cg
      float xme = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) xme = (n < N && n == cg) ? x[n] : xme;
      float dl = 0.f;
      double term = 0.0;
      if constexpr (KIND == LOSS_BCE) {
        if (cg < N) {
          const double xv = (double)xme, y = labels[b * N + cg];
          term = bce_term(xv, y);
          dl = (float)(up * (sigmoid_d(xv) - y));
        }
      } else if constexpr (KIND == LOSS_BPR) {
        const double y = labels[b * N];
        double gd = 0.0;
        if (cg >= 1 && cg < N) {
          const double d = (double)(x[0] - xme);
          term = bce_term(d, y);
          gd = up * (sigmoid_d(d) - y);
          dl = (float)(-gd);
        }
        double gpos = 0.0;                                       // sum over the negatives in column order, as rec_loss_kernel
#pragma unroll
        for (int j = 1; j < NMAX; ++j) {
          if (j < N) {
            const int hi32 = __shfl(__double2hiint(gd), lane0 + j, 64), lo32 = __shfl(__double2loint(gd), lane0 + j, 64);
            gpos += __hiloint2double(hi32, lo32);
          }
        }
        if (cg == 0) dl = (float)gpos;
      } else {
        float mx = x[0];
#pragma unroll
        for (int j = 1; j < NMAX; ++j) if (j < N) mx = fmaxf(mx, x[j] + shift);
        float se = expf(x[0] - mx);
#pragma unroll
        for (int j = 1; j < NMAX; ++j) if (j < N) se += expf(x[j] + shift - mx);
        const float lse = mx + logf(se);
        if (cg == 0) term = (double)(lse - x[0]);
        if (cg < N) {
          const float p = expf(xme + (cg ? shift : 0.f) - lse);
          dl = (float)(up * (double)(p - (cg == 0 ? 1.f : 0.f)));
        }
      }
      lacc += term;
      if (cg < N) {
        dlogits[b * N + cg] = dl;
        if (logits) logits[b * N + cg] = xme;
      }
      // ---- backward pass A on the rows still in registers
      float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {
          bns_pass_a(__shfl(dl, lane0 + n, 64), xh[n], g, be, u, du, a0, a1);
        }
      }
      *reinterpret_cast<float4*>(dU + b * D + 4 * cg) = du;
    }
  }
  sm[0][t] = a0;
  sm[1][t] = a1;
  const double lsum = block_sum_d(lacc, lsm);                    // (contains the barrier that publishes sm)
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
  // ---- arrival: the last block finishes the loss and the column sums. Everything another block has to see went out as an
  // agent-scope atomic (performed at the memory side), so "release" only has to WAIT for this block's atomics (an explicit
  // s_waitcnt vmcnt(0) per wave: a workgroup-scope fence does not emit it): __threadfence() writes the XCD's whole L2 back — per
  // block, with the dU / dlogits lines of every block in it — and made this kernel's time grow with the number of blocks (120 us at
  // 1,024 blocks, 75 at 512).
  SBR_DRAIN_VMEM();
  __syncthreads();                                               // every thread's atomics have been performed
  if (t == 0) {
    __hip_atomic_store(&lws[1 + blockIdx.x], lsum * scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SBR_DRAIN_VMEM();                                            // the partial sum has been performed before the counter moves
    const unsigned long long before = atomicAdd(reinterpret_cast<unsigned long long*>(lws), 1ull);
    last_flag = before == gridDim.x - 1;
  }
  __syncthreads();
#ifdef BSL_NO_FINAL
  return;
#endif
  if (!last_flag) return;
  // (the last block reads what the others published with agent-scope atomic loads only)
  // (all 256 threads fetch the block partial sums — one thread walking up to 512 dependent-latency loads took 120 us — and add them
  // in a fixed pattern: thread t takes partials t, t + 256, then the block sum)
  double part = 0.0;
  for (unsigned i = t; i < gridDim.x; i += 256) part += __hip_atomic_load(&lws[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();                                               // lsm is reused
  const double sum = block_sum_d(part, lsm);
  if (t == 0) {
    loss_out[0] = sum;
    if (out3) { out3[0] = sum; out3[1] = sum; out3[2] = 0.0; }
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(lws), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int KD = 2 * D;
  for (int i = t; i < KD; i += 256) {
    double s = 0.0;
#pragma unroll
    for (int q = 1; q <= SBR_COLRED_REP; ++q) {
      s += __hip_atomic_load(&ws[(long)q * KD + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ws[(long)q * KD + i] = 0.0;
    }
    ws[i] = s;
  }
}

#ifndef BSL_MAX_BLOCKS
#define BSL_MAX_BLOCKS 512
#endif
#ifndef BSL_USERS
#define BSL_USERS 2                                      // users per row lane the grid aims for
#endif
extern "C" int sbr_bn_score_loss_supported(int D, int N) { return sbr_bn_score_supported(D) && N >= 1 && N <= (D >> 2) && N <= 16; }
extern "C" long sbr_bn_score_loss_workspace(void) { return (BSL_MAX_BLOCKS + 1) * (long)sizeof(double); }

// logits (may be NULL), dlogits [B, N], dU [B, D], loss_out [1], out3 (may be NULL) = (loss, loss, 0); ws: the BatchNorm's
// column-reduction workspace as for sbr_bn_score_bwd_stats (totals in ws[0 .. 2 D) afterwards); lws: sbr_bn_score_loss_workspace()
// bytes, zeroed ONCE by the caller and left zeroed by every call (calls sharing it must not overlap). kind / labels / scale / shift
// as for sbr_rec_loss_fwd_bwd.
extern "C" int sbr_bn_score_loss_fwd_bwd(const float* Z, const float* U, const float* save_mean, const float* save_rstd,
                                         const float* weight, const float* bias, int kind, const double* labels, double scale,
                                         float shift, float* logits, float* dlogits, float* dU, double* loss_out, double* out3, long B,
                                         int N, int D, double* ws, void* lws, long lws_bytes, void* stream) {
  SBR_REQUIRE(kind >= 0 && kind <= 2, "sbr_bn_score_loss_fwd_bwd: unknown loss kind %d", kind);
  SBR_REQUIRE(B >= 1, "sbr_bn_score_loss_fwd_bwd: empty batch");
  SBR_REQUIRE(Z && U && save_mean && save_rstd && weight && bias && dlogits && dU && loss_out && ws && (kind == LOSS_SSM || labels),
              "sbr_bn_score_loss_fwd_bwd: null operand");
  SBR_REQUIRE(sbr_bn_score_loss_supported(D, N) && tail_ok(Z, U, save_mean, D) && tail_ok(save_rstd, weight, bias, D) && tail_ok(dU, dU, dU, D),
              "sbr_bn_score_loss_fwd_bwd: D=%d N=%d / alignment not supported", D, N);
  SBR_REQUIRE(lws && lws_bytes >= sbr_bn_score_loss_workspace(), "sbr_bn_score_loss_fwd_bwd: workspace too small");
  const int RL = 256 / (D >> 2);
  long blocks = (B + (long)BSL_USERS * RL - 1) / ((long)BSL_USERS * RL);
  if (blocks > BSL_MAX_BLOCKS) blocks = BSL_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipStream_t s = (hipStream_t)stream;
#define BSL_LAUNCH(NM, KD)                                                                                                  \
  bn_score_loss_kernel<NM, KD><<<(int)blocks, 256, 0, s>>>(Z, U, save_mean, save_rstd, weight, bias, labels, scale, shift, logits, \
                                                           dlogits, dU, B, D, ws, (double*)lws, loss_out, out3)
#define BSL_KINDS(NM)                                                                                                       \
  case NM:                                                                                                                 \
    if (kind == LOSS_BCE) BSL_LAUNCH(NM, LOSS_BCE);                                                                        \
    else if (kind == LOSS_BPR) BSL_LAUNCH(NM, LOSS_BPR);                                                                   \
    else BSL_LAUNCH(NM, LOSS_SSM);                                                                                         \
    break;
  switch (N) {
    BSL_KINDS(1) BSL_KINDS(2) BSL_KINDS(3) BSL_KINDS(4) BSL_KINDS(5) BSL_KINDS(6) BSL_KINDS(7) BSL_KINDS(8)
    BSL_KINDS(9) BSL_KINDS(10) BSL_KINDS(11) BSL_KINDS(12) BSL_KINDS(13) BSL_KINDS(14) BSL_KINDS(15) BSL_KINDS(16)
    default: break;
  }
#undef BSL_KINDS
#undef BSL_LAUNCH
  SBR_CHECK_LAUNCH("sbr_bn_score_loss_fwd_bwd");
  return SBR_OK;
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
  if (blocks > 512) blocks = 512;
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
