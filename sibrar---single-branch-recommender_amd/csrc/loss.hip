// Recommendation losses (train/rec_losses.py) and the InfoNCE regulariser (train/regularization_losses.py),
// forward and gradient w.r.t. the logits / embeddings.
//
// dtype notes that follow the reference: labels are float64 (data/dataloader.py:196), therefore BCE and BPR evaluate
// BCEWithLogits in float64 and return a float64 scalar; sampled-softmax and InfoNCE stay in float32 (we accumulate their
// sums in double and round once).
#include "loss_common.h"

// loss scalars are zeroed by a one-thread kernel, not by hipMemsetAsync: inside a replayed hipGraph the 8-byte memset node was
// observed to stop taking effect while another host thread issued copies (the scalar then kept a stale value for every later
// replay); a kernel node has no such dependence on the runtime's fill path.
__global__ void zero_f64_kernel(double* __restrict__ p) { p[0] = 0.0; }

// one thread per batch row
// MODE 3 = MODE 2 without a zeroed loss_out: every block leaves its partial sum in ws[1 + block], the block whose agent-scope
// counter add (ws[0]) comes last sums the partials in block order (the same bits on every run), writes loss_out (and out3 =
// (loss, loss, 0) when given: the packed total / rec / reg scalars of a step without regularisation losses) and resets the counter.
template <int MODE>   // 0: loss, 1: gradient, 2: both (upstream gradient 1: the fused training step), 3: both, self-contained sum
__global__ void rec_loss_kernel(int kind, const float* __restrict__ logits, const double* __restrict__ labels, long B, int N,
                                double scale, float shift, double* __restrict__ loss_out, const void* __restrict__ gout,
                                int gout_is_double, float* __restrict__ dlogits, double* __restrict__ ws = nullptr,
                                double* __restrict__ out3 = nullptr) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double acc = 0.0;
  constexpr bool FWD = MODE != 1, BWD = MODE != 0;
  double up = scale;
  if (MODE == 1) up = (gout_is_double ? ((const double*)gout)[0] : (double)((const float*)gout)[0]) * scale;
  if (b < B) {
    const float* x = logits + b * N;
    if (kind == LOSS_BCE) {
      for (int j = 0; j < N; ++j) {
        const double xv = (double)x[j], y = labels[b * N + j];
        if (FWD) acc += bce_term(xv, y);
        if (BWD) dlogits[b * N + j] = (float)(up * (sigmoid_d(xv) - y));
      }
    } else if (kind == LOSS_BPR) {
      // rec_losses.py:73-81: diff = pos - neg (float32), target = label of the positive column
      const double y = labels[b * N];
      double gpos = 0.0;
      for (int j = 1; j < N; ++j) {
        const double d = (double)(x[0] - x[j]);
        if (FWD) acc += bce_term(d, y);
        if (BWD) {
          const double gd = up * (sigmoid_d(d) - y);
          gpos += gd;
          dlogits[b * N + j] = (float)(-gd);
        }
      }
      if (BWD) dlogits[b * N] = (float)gpos;
    } else {
      // rec_losses.py:101-108: -x_pos + logsumexp(x) with the negatives shifted by log(n_items / n_neg) ('uniform')
      float mx = x[0];
      for (int j = 1; j < N; ++j) mx = fmaxf(mx, x[j] + shift);
      float se = expf(x[0] - mx);
      for (int j = 1; j < N; ++j) se += expf(x[j] + shift - mx);
      const float lse = mx + logf(se);
      if (FWD) acc += (double)(lse - x[0]);
      if (BWD) {
        for (int j = 0; j < N; ++j) {
          const float p = expf(x[j] + (j ? shift : 0.f) - lse);
          dlogits[b * N + j] = (float)(up * (double)(p - (j == 0 ? 1.f : 0.f)));
        }
      }
    }
  }
  if (FWD) {
    __shared__ double sm[4];
    const double t = block_sum_d(acc, sm);
    if (MODE == 3) {
      __shared__ int last;
      if (threadIdx.x == 0) {
        __hip_atomic_store(&ws[1 + blockIdx.x], t * scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SBR_DRAIN_VMEM();                                        // the store has been performed (no L2 write-back: see fused_tail.hip)
        const unsigned long long before = atomicAdd(reinterpret_cast<unsigned long long*>(ws), 1ull);
        last = before == gridDim.x - 1;                          // every other block's partial is out
      }
      __syncthreads();
      if (last) {
        // all threads fetch the partial sums (one thread walking them pays a memory round trip per partial) and add them in a
        // fixed pattern: thread i takes partials i, i + 256, ..., then the block sum
        double part = 0.0;
        for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)
          part += __hip_atomic_load(&ws[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                         // sm is reused
        const double sum = block_sum_d(part, sm);
        if (threadIdx.x == 0) {
          loss_out[0] = sum;
          if (out3) { out3[0] = sum; out3[1] = sum; out3[2] = 0.0; }
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(ws), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else if (threadIdx.x == 0) {
      atomicAdd(loss_out, t * scale);
    }
  }
}

// scale = 1 / count for 'mean' (count = B*N for bce, B*(N-1) for bpr, B for sampled softmax), 1 for 'sum'
extern "C" int sbr_rec_loss_fwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                                double* loss_out, void* stream) {
  SBR_REQUIRE(kind >= 0 && kind <= 2, "sbr_rec_loss_fwd: unknown loss kind %d", kind);
  SBR_REQUIRE(logits && loss_out && (kind == LOSS_SSM || labels), "sbr_rec_loss_fwd: null operand");
  hipStream_t s = (hipStream_t)stream;
  zero_f64_kernel<<<1, 1, 0, s>>>(loss_out);
  if (B == 0) return SBR_OK;
  rec_loss_kernel<0><<<sbr_cdiv(B, 256), 256, 0, s>>>(kind, logits, labels, B, N, scale, shift, loss_out, nullptr, 0, nullptr);
  SBR_CHECK_LAUNCH("sbr_rec_loss_fwd");
  return SBR_OK;
}

extern "C" int sbr_rec_loss_bwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                                const void* grad_out, int grad_out_is_double, float* dlogits, void* stream) {
  SBR_REQUIRE(kind >= 0 && kind <= 2, "sbr_rec_loss_bwd: unknown loss kind %d", kind);
  SBR_REQUIRE(logits && grad_out && dlogits && (kind == LOSS_SSM || labels), "sbr_rec_loss_bwd: null operand");
  if (B == 0) return SBR_OK;
  rec_loss_kernel<1><<<sbr_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(kind, logits, labels, B, N, scale, shift, nullptr,
                                                                           grad_out, grad_out_is_double, dlogits);
  SBR_CHECK_LAUNCH("sbr_rec_loss_bwd");
  return SBR_OK;
}

// loss and d loss / d logits in one pass (upstream gradient 1): what a training step needs (trainer.py:213-221)
extern "C" int sbr_rec_loss_fwd_bwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                                    double* loss_out, float* dlogits, void* stream) {
  SBR_REQUIRE(kind >= 0 && kind <= 2, "sbr_rec_loss_fwd_bwd: unknown loss kind %d", kind);
  SBR_REQUIRE(logits && loss_out && dlogits && (kind == LOSS_SSM || labels), "sbr_rec_loss_fwd_bwd: null operand");
  hipStream_t s = (hipStream_t)stream;
  zero_f64_kernel<<<1, 1, 0, s>>>(loss_out);
  if (B == 0) return SBR_OK;
  rec_loss_kernel<2><<<sbr_cdiv(B, 256), 256, 0, s>>>(kind, logits, labels, B, N, scale, shift, loss_out, nullptr, 0, dlogits);
  SBR_CHECK_LAUNCH("sbr_rec_loss_fwd_bwd");
  return SBR_OK;
}

// The same with ONE launch: no zeroing launch in front (the block partial sums go through ``ws``, sbr_rec_loss_workspace(B) bytes that
// the caller zeroes ONCE and then leaves alone: the kernel resets what it uses), a fixed summation order, and optionally the packed
// loss scalars out3 = (loss, loss, 0) of a step without regularisation losses (what sbr_pack_losses would write). Calls that share a
// workspace must not overlap in time.
extern "C" long sbr_rec_loss_workspace(long B) { return (sbr_cdiv(B > 0 ? B : 1, 256) + 1) * (long)sizeof(double); }

extern "C" int sbr_rec_loss_fwd_bwd_ws(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                                       double* loss_out, float* dlogits, double* out3, void* ws, long ws_bytes, void* stream) {
  SBR_REQUIRE(kind >= 0 && kind <= 2, "sbr_rec_loss_fwd_bwd_ws: unknown loss kind %d", kind);
  SBR_REQUIRE(logits && loss_out && dlogits && (kind == LOSS_SSM || labels), "sbr_rec_loss_fwd_bwd_ws: null operand");
  SBR_REQUIRE(B >= 1, "sbr_rec_loss_fwd_bwd_ws: empty batch (use sbr_rec_loss_fwd_bwd)");
  SBR_REQUIRE(ws && ws_bytes >= sbr_rec_loss_workspace(B), "sbr_rec_loss_fwd_bwd_ws: workspace too small");
  rec_loss_kernel<3><<<sbr_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(kind, logits, labels, B, N, scale, shift, loss_out, nullptr, 0,
                                                                           dlogits, (double*)ws, out3);
  SBR_CHECK_LAUNCH("sbr_rec_loss_fwd_bwd_ws");
  return SBR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// InfoNCE (regularization_losses.py:14-43): groups g of N rows; logits L = A_g B_g^T / tau; symmetric cross entropy
// against the diagonal. One workgroup per group; L lives in LDS ([N][N+1] floats).
// A / B rows are addressed as base + (g*N + i) * ld, so the two modality slices e[..., 0, :] / e[..., 1, :] of the
// [S, 2, D] embedding tensor are read in place (ld = 2*D).
// ---------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void infonce_kernel(const float* __restrict__ A, const float* __restrict__ Bm, long ld, int N, int D,
                               float inv_tau, double scale, double* __restrict__ loss_out, const float* __restrict__ gout,
                               float* __restrict__ dA, float* __restrict__ dB, long ldg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LN = N + 1;
  float* L = sm;                 // [N][N+1]
  float* lse_r = sm + N * LN;    // [N]
  float* lse_c = lse_r + N;      // [N]
  const long g = blockIdx.x;
  const float* a = A + g * N * ld;
  const float* b = Bm + g * N * ld;
  for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
    const int i = p / N, j = p - i * N;
    float acc = 0.f;
    for (int c = 0; c < D; ++c) acc += a[i * ld + c] * b[j * ld + c];
    L[i * LN + j] = acc * inv_tau;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) {
    const bool col = i >= N;
    const int r = col ? i - N : i;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, col ? L[j * LN + r] : L[r * LN + j]);
    float se = 0.f;
    for (int j = 0; j < N; ++j) se += expf((col ? L[j * LN + r] : L[r * LN + j]) - mx);
    (col ? lse_c : lse_r)[r] = mx + logf(se);
  }
  __syncthreads();
  if (!BWD) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc += (double)(lse_r[i] - L[i * LN + i]) + (double)(lse_c[i] - L[i * LN + i]);
    __shared__ double red[4];
    const double t = block_sum_d(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss_out, t * scale);
  } else {
    const float up = gout[0] * (float)scale * inv_tau;
    // G[i][j] = up * (softmax_row + softmax_col - 2*delta)
    for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
      const int i = p / N, j = p - i * N;
      const float l = L[i * LN + j];
      L[i * LN + j] = up * (expf(l - lse_r[i]) + expf(l - lse_c[j]) - (i == j ? 2.f : 0.f));
    }
    __syncthreads();
    float* da = dA + g * N * ldg;
    float* db = dB + g * N * ldg;
    for (int p = threadIdx.x; p < N * D; p += blockDim.x) {
      const int i = p / D, c = p - i * D;
      float sa = 0.f, sb = 0.f;
      for (int j = 0; j < N; ++j) {
        sa += L[i * LN + j] * b[j * ld + c];      // dA[i] = sum_j G[i][j] B[j]
        sb += L[j * LN + i] * a[j * ld + c];      // dB[i] = sum_j G[j][i] A[j]
      }
      da[i * ldg + c] = sa;
      db[i * ldg + c] = sb;
    }
  }
}

// Small groups (the per-user groups of the training step: N = 1 + negatives rows, regularization_losses.py:14-43 called from
// sgd_alg.py:1994-2002): ONE WAVE per group, four groups per workgroup. The generic kernel above gives every thread one of the N x N
// dot products and lets it walk two rows from global memory, 4 bytes at a time, a kilobyte apart from its neighbours' — 121 of 256
// threads busy for N = 11, 0.2 ms forward + 0.2 ms backward for 46 MB of embeddings at Onion18's batch 4096. Here a wave copies its
// group's rows of A and B into LDS with 16-byte loads (row stride D + 4 floats: conflict-free 16-byte reads of different rows),
// lanes take the (i, j) pairs for the logits, rows / columns for the log-sum-exps, and COLUMNS for the backward products
// dA = G B, dB = G^T A (G from LDS, broadcast). N <= 16, D <= 256, D % 4 == 0, 16-byte aligned rows.
#define INF_S_MAXN 16
#define INF_S_GPW(BWD) ((BWD) ? 1 : 4)       // groups per wave: the forward pass amortises its loss atomic, the backward pass wants the waves
#define INF_S_MAXD 256
template <bool BWD>
__global__ __launch_bounds__(256) void infonce_small_kernel(const float* __restrict__ A, const float* __restrict__ Bm, long ld, long G,
                                                            int N, int D, float inv_tau, double scale, double* __restrict__ loss_out,
                                                            const float* __restrict__ gout, float* __restrict__ dA,
                                                            float* __restrict__ dB, long ldg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int LD = D + 4, LN = N + 1;
  double loss_acc = 0.0;                                        // forward: this wave's groups
  // a forward wave takes INF_S_GPW groups one after the other (its LDS region is its own: wave barriers only inside the loop): the forward
  // pass then adds ONE double per workgroup to the loss instead of one per group — 4,096 atomics on one address were 40 us
  for (int it = 0; it < INF_S_GPW(BWD); ++it) {
  const long g = ((long)blockIdx.x * INF_S_GPW(BWD) + it) * 4 + wave;
  if (g >= G) break;                                            // wave-uniform
  float* a = sm + (size_t)wave * (2 * N * LD + N * LN + 2 * N + 2);
  float* b = a + N * LD;
  float* L = b + N * LD;                                        // [N][N + 1]
  float* lse_r = L + N * LN;
  float* lse_c = lse_r + N;
  const float* ga = A + g * N * ld;
  const float* gb = Bm + g * N * ld;
  const int D4 = D >> 2;
  for (int p = lane; p < N * D4; p += 64) {
    const int i = p / D4, c4 = p - i * D4;
    *reinterpret_cast<float4*>(a + i * LD + 4 * c4) = *reinterpret_cast<const float4*>(ga + i * ld + 4 * c4);
    *reinterpret_cast<float4*>(b + i * LD + 4 * c4) = *reinterpret_cast<const float4*>(gb + i * ld + 4 * c4);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int p = lane; p < N * N; p += 64) {
    const int i = p / N, j = p - i * N;
    const float4* ai = reinterpret_cast<const float4*>(a + i * LD);
    const float4* bj = reinterpret_cast<const float4*>(b + j * LD);
    float acc = 0.f;
    for (int c = 0; c < D4; ++c) {
      const float4 x = ai[c], y = bj[c];
      acc += x.x * y.x; acc += x.y * y.y; acc += x.z * y.z; acc += x.w * y.w;      // the column order of the generic kernel
    }
    L[i * LN + j] = acc * inv_tau;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 2 * N) {
    const bool col = lane >= N;
    const int r = col ? lane - N : lane;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, col ? L[j * LN + r] : L[r * LN + j]);
    float se = 0.f;
    for (int j = 0; j < N; ++j) se += expf((col ? L[j * LN + r] : L[r * LN + j]) - mx);
    (col ? lse_c : lse_r)[r] = mx + logf(se);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if constexpr (!BWD) {
    loss_acc += lane < N ? (double)(lse_r[lane] - L[lane * LN + lane]) + (double)(lse_c[lane] - L[lane * LN + lane]) : 0.0;
  } else {
    const float up = gout[0] * (float)scale * inv_tau;
    for (int p = lane; p < N * N; p += 64) {
      const int i = p / N, j = p - i * N;
      const float l = L[i * LN + j];
      L[i * LN + j] = up * (expf(l - lse_r[i]) + expf(l - lse_c[j]) - (i == j ? 2.f : 0.f));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* da = dA + g * N * ldg;
    float* db = dB + g * N * ldg;
    // lane = a group of four columns; D / 4 <= 64 lanes cover a row
    if (lane < D4) {
      for (int i = 0; i < N; ++i) {
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < N; ++j) {
          const float gij = L[i * LN + j], gji = L[j * LN + i];
          const float4 y = *reinterpret_cast<const float4*>(b + j * LD + 4 * lane);
          const float4 x = *reinterpret_cast<const float4*>(a + j * LD + 4 * lane);
          sa.x += gij * y.x; sa.y += gij * y.y; sa.z += gij * y.z; sa.w += gij * y.w;     // dA[i] = sum_j G[i][j] B[j]
          sb.x += gji * x.x; sb.y += gji * x.y; sb.z += gji * x.z; sb.w += gji * x.w;     // dB[i] = sum_j G[j][i] A[j]
        }
        *reinterpret_cast<float4*>(da + i * ldg + 4 * lane) = sa;
        *reinterpret_cast<float4*>(db + i * ldg + 4 * lane) = sb;
      }
    }
  }
  // the next group reuses this wave's LDS region
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if constexpr (!BWD) {
    __shared__ double red[4];
    const double t = block_sum_d(loss_acc, red);                // every wave of the workgroup arrives here
    if (threadIdx.x == 0) atomicAdd(loss_out, t * scale);
  }
}
static bool infonce_small_ok(const float* A, const float* B, long ld, int N, int D, const float* dA, const float* dB, long ldg) {
  return N <= INF_S_MAXN && D <= INF_S_MAXD && (D & 3) == 0 && (ld & 3) == 0 && (ldg & 3) == 0 &&
         ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)dA) | ((uintptr_t)dB)) & 15) == 0;
}
static int infonce_small_lds(int N, int D) { return 4 * (2 * N * (D + 4) + N * (N + 1) + 2 * N + 2) * (int)sizeof(float); }

#define INFONCE_MAX_N 176

static int infonce_lds(int N) { return (N * (N + 1) + 2 * N) * (int)sizeof(float); }

extern "C" int sbr_infonce_max_n(void) { return INFONCE_MAX_N; }

// scale = 1/(G*N) for 'mean', 1 for 'sum'
extern "C" int sbr_infonce_fwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                               double* loss_out, void* stream) {
  SBR_REQUIRE(A && B && loss_out, "sbr_infonce_fwd: null operand");
  SBR_REQUIRE(N >= 1 && N <= INFONCE_MAX_N, "sbr_infonce_fwd: N=%d outside [1, %d]", N, INFONCE_MAX_N);
  hipStream_t s = (hipStream_t)stream;
  zero_f64_kernel<<<1, 1, 0, s>>>(loss_out);
  if (G == 0) return SBR_OK;
  if (infonce_small_ok(A, B, ld, N, D, nullptr, nullptr, 0)) {
    const int ls = infonce_small_lds(N, D);
    if (ls > 64 * 1024) (void)hipFuncSetAttribute((const void*)infonce_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ls);
    infonce_small_kernel<false><<<(unsigned)sbr_cdiv(G, 4 * INF_S_GPW(false)), 256, ls, s>>>(A, B, ld, G, N, D, 1.f / tau, scale, loss_out, nullptr, nullptr, nullptr, 0);
    SBR_CHECK_LAUNCH("sbr_infonce_fwd (small groups)");
    return SBR_OK;
  }
  const int lds = infonce_lds(N);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)infonce_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  infonce_kernel<false><<<(unsigned)G, 256, lds, s>>>(A, B, ld, N, D, 1.f / tau, scale, loss_out, nullptr, nullptr, nullptr, 0);
  SBR_CHECK_LAUNCH("sbr_infonce_fwd");
  return SBR_OK;
}

extern "C" int sbr_infonce_bwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                               const float* grad_out, float* dA, float* dB, long ldg, void* stream) {
  SBR_REQUIRE(A && B && grad_out && dA && dB, "sbr_infonce_bwd: null operand");
  SBR_REQUIRE(N >= 1 && N <= INFONCE_MAX_N, "sbr_infonce_bwd: N=%d outside [1, %d]", N, INFONCE_MAX_N);
  if (G == 0) return SBR_OK;
  if (infonce_small_ok(A, B, ld, N, D, dA, dB, ldg)) {
    const int ls = infonce_small_lds(N, D);
    if (ls > 64 * 1024) (void)hipFuncSetAttribute((const void*)infonce_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ls);
    infonce_small_kernel<true><<<(unsigned)sbr_cdiv(G, 4 * INF_S_GPW(true)), 256, ls, (hipStream_t)stream>>>(A, B, ld, G, N, D, 1.f / tau, scale, nullptr, grad_out, dA, dB, ldg);
    SBR_CHECK_LAUNCH("sbr_infonce_bwd (small groups)");
    return SBR_OK;
  }
  const int lds = infonce_lds(N);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)infonce_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  infonce_kernel<true><<<(unsigned)G, 256, lds, (hipStream_t)stream>>>(A, B, ld, N, D, 1.f / tau, scale, nullptr, grad_out, dA, dB, ldg);
  SBR_CHECK_LAUNCH("sbr_infonce_bwd");
  return SBR_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// InfoNCE for large groups (in-batch user contrast: ONE group of B rows, regularization_losses.py:14-43 called from
// sgd_alg.py:1994-2002 with a [B, 2, D] tensor): the N x N logit matrix goes through the fp32 MFMA GEMMs instead of LDS.
//   T = B_g A_g^T  -> column log-sum-exps (rows of the transpose)        S = A_g B_g^T -> row log-sum-exps, diagonal
//   loss += scale * sum_i (lse_r[i] + lse_c[i] - 2 S_ii / tau)
//   backward: S <- up * (softmax_rows + softmax_cols - 2 I), dA = S B_g (NN), dB = S^T A_g (TN, split-K slab reducer)
// workspace: [N*N] logits | [N] lse_r | [N] lse_c | split-K slabs of the TN product. Groups are processed one after the other.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int sbr_gemm_f32(int mode, const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx,
                            const float* bias, float* C, long ldc, const int* c_idx, int M, int N, int K, int act,
                            int accumulate_atomic, void* stream);
extern "C" long sbr_gemm_tn_f32_workspace(int M, int N, int K);
extern "C" int sbr_gemm_tn_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, float* C,
                               long ldc, int M, int N, int K, void* workspace, long workspace_bytes, void* stream);

// one wave per row: lse[i] = log sum_j exp(S[i][j] * inv_tau)
__global__ void lse_rows_kernel(const float* __restrict__ S, int N, float inv_tau, float* __restrict__ lse) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float* r = S + (long)row * N;
  float mx = -INFINITY;
  for (int j = lane; j < N; j += 64) mx = fmaxf(mx, r[j] * inv_tau);
  mx = sbr_wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < N; j += 64) se += expf(r[j] * inv_tau - mx);
  se = sbr_wave_sum(se);
  if (lane == 0) lse[row] = mx + logf(se);
}

__global__ void infonce_gemm_loss_kernel(const float* __restrict__ S, int N, float inv_tau, const float* __restrict__ lse_r,
                                         const float* __restrict__ lse_c, double scale, double* __restrict__ loss_out) {
  double acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const double d = (double)(S[(long)i * N + i] * inv_tau);
    acc += ((double)lse_r[i] - d) + ((double)lse_c[i] - d);
  }
  __shared__ double red[4];
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss_out, t * scale);
}

// S[i][j] <- up * (exp(l - lse_r[i]) + exp(l - lse_c[j]) - 2 [i == j]),  l = S[i][j] / tau,  up = gout * scale / tau
__global__ void infonce_gemm_ds_kernel(float* __restrict__ S, int N, float inv_tau, const float* __restrict__ lse_r,
                                       const float* __restrict__ lse_c, const float* __restrict__ gout, float scale) {
  const float up = gout[0] * scale * inv_tau;
  const long total = (long)N * N;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int i = (int)(p / N), j = (int)(p - (long)i * N);
    const float l = S[p] * inv_tau;
    S[p] = up * (expf(l - lse_r[i]) + expf(l - lse_c[j]) - (i == j ? 2.f : 0.f));
  }
}

extern "C" long sbr_infonce_gemm_workspace(int N, int D) {
  return ((long)N * N + 2L * N) * (long)sizeof(float) + sbr_gemm_tn_f32_workspace(N, D, N) + 256;
}

static int infonce_gemm_stats(const float* a, const float* b, long ld, int N, int D, float inv_tau, float* S, float* lse_r,
                              float* lse_c, hipStream_t s) {
  // column statistics from the transposed product, then the product itself (left in S)
  int rc = sbr_gemm_f32(0, b, ld, nullptr, a, ld, nullptr, nullptr, S, N, nullptr, N, N, D, SBR_ACT_NONE, 0, s);
  if (rc) return rc;
  lse_rows_kernel<<<sbr_cdiv(N, 4), 256, 0, s>>>(S, N, inv_tau, lse_c);
  SBR_CHECK_LAUNCH("sbr_infonce_gemm/lse_c");
  rc = sbr_gemm_f32(0, a, ld, nullptr, b, ld, nullptr, nullptr, S, N, nullptr, N, N, D, SBR_ACT_NONE, 0, s);
  if (rc) return rc;
  lse_rows_kernel<<<sbr_cdiv(N, 4), 256, 0, s>>>(S, N, inv_tau, lse_r);
  SBR_CHECK_LAUNCH("sbr_infonce_gemm/lse_r");
  return SBR_OK;
}

extern "C" int sbr_infonce_gemm_fwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                                    double* loss_out, void* workspace, long workspace_bytes, void* stream) {
  SBR_REQUIRE(A && B && loss_out, "sbr_infonce_gemm_fwd: null operand");
  SBR_REQUIRE(N >= 1 && D >= 1, "sbr_infonce_gemm_fwd: bad shape");
  SBR_REQUIRE(workspace && workspace_bytes >= sbr_infonce_gemm_workspace(N, D), "sbr_infonce_gemm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  zero_f64_kernel<<<1, 1, 0, s>>>(loss_out);
  float* S = (float*)workspace;
  float* lse_r = S + (long)N * N;
  float* lse_c = lse_r + N;
  for (long g = 0; g < G; ++g) {
    const float* a = A + g * N * ld;
    const float* b = B + g * N * ld;
    int rc = infonce_gemm_stats(a, b, ld, N, D, 1.f / tau, S, lse_r, lse_c, s);
    if (rc) return rc;
    infonce_gemm_loss_kernel<<<sbr_cdiv(N, 256) > 64 ? 64 : sbr_cdiv(N, 256), 256, 0, s>>>(S, N, 1.f / tau, lse_r, lse_c, scale, loss_out);
    SBR_CHECK_LAUNCH("sbr_infonce_gemm_fwd");
  }
  return SBR_OK;
}

extern "C" int sbr_infonce_gemm_bwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                                    const float* grad_out, float* dA, float* dB, long ldg, void* workspace,
                                    long workspace_bytes, void* stream) {
  SBR_REQUIRE(A && B && grad_out && dA && dB, "sbr_infonce_gemm_bwd: null operand");
  SBR_REQUIRE(N >= 1 && D >= 1, "sbr_infonce_gemm_bwd: bad shape");
  SBR_REQUIRE(workspace && workspace_bytes >= sbr_infonce_gemm_workspace(N, D), "sbr_infonce_gemm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* S = (float*)workspace;
  float* lse_r = S + (long)N * N;
  float* lse_c = lse_r + N;
  char* tn_ws = (char*)(((uintptr_t)(lse_c + N) + 255) & ~(uintptr_t)255);
  const long tn_bytes = sbr_gemm_tn_f32_workspace(N, D, N);
  for (long g = 0; g < G; ++g) {
    const float* a = A + g * N * ld;
    const float* b = B + g * N * ld;
    int rc = infonce_gemm_stats(a, b, ld, N, D, 1.f / tau, S, lse_r, lse_c, s);
    if (rc) return rc;
    const long total = (long)N * N;
    const int blocks = (int)(sbr_cdiv(total, 256) > 4096 ? 4096 : sbr_cdiv(total, 256));
    infonce_gemm_ds_kernel<<<blocks, 256, 0, s>>>(S, N, 1.f / tau, lse_r, lse_c, grad_out, (float)scale);
    SBR_CHECK_LAUNCH("sbr_infonce_gemm_bwd/ds");
    // dA[i] = sum_j G[i][j] B[j]   (NN: [N x N] @ [N x D]);   dB[i] = sum_j G[j][i] A[j]   (TN over the N rows)
    rc = sbr_gemm_f32(1, S, N, nullptr, b, ld, nullptr, nullptr, dA + g * N * ldg, ldg, nullptr, N, D, N, SBR_ACT_NONE, 0, s);
    if (rc) return rc;
    rc = sbr_gemm_tn_f32(S, N, nullptr, a, ld, nullptr, dB + g * N * ldg, ldg, N, D, N, tn_ws, tn_bytes, s);
    if (rc) return rc;
  }
  return SBR_OK;
}


// out[0] = rec + reg, out[1] = rec, out[2] = reg with reg = w_a * reg_a + w_b * reg_b (trainer.py:213-216: total loss = rec loss +
// weighted regularisation losses). One tiny launch instead of five elementwise ones; null reg pointers count as zero.
__global__ void pack_losses_kernel(const double* __restrict__ rec, const double* __restrict__ reg_a, double w_a,
                                   const double* __restrict__ reg_b, double w_b, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double reg = (reg_a ? w_a * reg_a[0] : 0.0) + (reg_b ? w_b * reg_b[0] : 0.0);
    out[0] = rec[0] + reg;
    out[1] = rec[0];
    out[2] = reg;
  }
}

extern "C" int sbr_pack_losses(const double* rec, const double* reg_a, double w_a, const double* reg_b, double w_b, double* out3,
                               void* stream) {
  SBR_REQUIRE(rec && out3, "sbr_pack_losses: null operand");
  pack_losses_kernel<<<1, 64, 0, (hipStream_t)stream>>>(rec, reg_a, w_a, reg_b, w_b, out3);
  SBR_CHECK_LAUNCH("sbr_pack_losses");
  return SBR_OK;
}
