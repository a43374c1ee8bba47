// Dense optimizer steps over one flat fp32 parameter buffer (train/trainer.py:62-68 builds
// torch.optim.{AdamW,Adam,Adagrad}(model.parameters(), lr, weight_decay) — every element of every table is updated
// every step, so the step is a pure HBM stream: AdamW moves 28 B per parameter (read p,g,m,v; write p,m,v)).
// The arithmetic follows torch's single-tensor rules in fp32, in the same operation order.
#include "common.h"

// One element of one Adam / AdamW step. The dense kernel and the row-wise (deferred) kernels below go through this one inlined
// function with run-time operands, so that a deferred row replays exactly the instruction sequence the dense kernel would have
// executed for it (same contractions, same rounding).
// omb1 / omb2 / decay: 1 - beta1, 1 - beta2, 1 - lr * wd evaluated in DOUBLE on the host and rounded to fp32 once, as torch does
// with its Python-float scalars (1 - 0.999 -> 0.001f; evaluated in fp32 it is 0.99998713e-3: a 6e-6 relative bias of every Adam
// step, found by the 257 M-parameter c4 test against the fp64 update rule)
struct AdamHyper { float lr, b1, b2, eps, wd; int decoupled; float omb1, omb2, decay; };
static AdamHyper adam_hyper(double lr, double b1, double b2, double eps, double wd, int decoupled) {
  AdamHyper h = {(float)lr, (float)b1, (float)b2, (float)eps, (float)wd, decoupled, (float)(1.0 - b1), (float)(1.0 - b2),
                 (float)(1.0 - lr * wd)};
  return h;
}

__device__ __forceinline__ void adam_element(float& pe, float ge, float& me, float& ve, const AdamHyper& h, float step_size,
                                             float bc2_sqrt) {
  // no fused multiply-add contraction here: left to itself the compiler contracts differently in different callers (mul + add
  // in the dense kernel, v_pk_fma in the flush kernel), and a deferred row must round exactly like a densely updated one
#pragma clang fp contract(off)
  if (h.decoupled) pe *= h.decay;                 // AdamW: p.mul_(1 - lr * wd)
  else ge += h.wd * pe;                            // Adam: grad = grad.add(p, alpha=wd)
  me = me + (ge - me) * h.omb1;                    // exp_avg.lerp_(grad, 1 - beta1)
  ve = ve * h.b2 + h.omb2 * ge * ge;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(ve) / bc2_sqrt + h.eps;
  pe = pe - step_size * (me / denom);
}

// ZERO: the gradient element is reset to +0 once it has been consumed (optimizer.zero_grad() of train/trainer.py:222 folded into the
// step). Only elements that are not +0 already are written, so the mostly-zero gradients of the embedding tables (8 % of the user
// table's rows see a gradient in a step of the bench) cost next to nothing: the separate fill wrote all 77 MB of the c2 gradient.
// cp_n > 0: the launch also copies cp_n doubles cp_src -> cp_dst (the loss scalars of the step, which live in a buffer the next
// replay of the captured step overwrites: one 4.9 us copy launch per step less).
template <bool ZERO>
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, AdamHyper h, float step_size, float bc2_sqrt, const double* __restrict__ cp_src = nullptr,
                             double* __restrict__ cp_dst = nullptr, int cp_n = 0) {
  if (blockIdx.x == 0 && (int)threadIdx.x < cp_n) cp_dst[threadIdx.x] = cp_src[threadIdx.x];
  if ((n & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
    // 16 bytes per lane; the same per-element function, bit for bit
    for (long e4 = blockIdx.x * (long)blockDim.x + threadIdx.x; e4 < (n >> 2); e4 += (long)gridDim.x * blockDim.x) {
      const long e = e4 << 2;
      float4 pe = *reinterpret_cast<const float4*>(p + e), me = *reinterpret_cast<const float4*>(m + e), ve = *reinterpret_cast<const float4*>(v + e);
      const float4 ge = *reinterpret_cast<const float4*>(g + e);
      adam_element(pe.x, ge.x, me.x, ve.x, h, step_size, bc2_sqrt);
      adam_element(pe.y, ge.y, me.y, ve.y, h, step_size, bc2_sqrt);
      adam_element(pe.z, ge.z, me.z, ve.z, h, step_size, bc2_sqrt);
      adam_element(pe.w, ge.w, me.w, ve.w, h, step_size, bc2_sqrt);
      *reinterpret_cast<float4*>(p + e) = pe;
      *reinterpret_cast<float4*>(m + e) = me;
      *reinterpret_cast<float4*>(v + e) = ve;
      if (ZERO && (__float_as_uint(ge.x) | __float_as_uint(ge.y) | __float_as_uint(ge.z) | __float_as_uint(ge.w)) != 0u)
        *reinterpret_cast<float4*>(g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float pe = p[e], me = m[e], ve = v[e];
    const float ge = g[e];
    adam_element(pe, ge, me, ve, h, step_size, bc2_sqrt);
    p[e] = pe;
    m[e] = me;
    v[e] = ve;
    if (ZERO && __float_as_uint(ge) != 0u) g[e] = 0.f;
  }
}

// ---- deferred row-wise Adam for lookup tables --------------------------------------------------------------------------------
// The reference's dense optimizer (trainer.py:62-68) updates EVERY row of an embedding table EVERY step — rows without a
// gradient still decay their moments and their weights — which makes the optimizer the largest kernel of the step (c2: 358 of its
// 541 MB per step are the user table, of whose rows a batch touches 8 %; c4: 7.2 GB of state per step). A row that gets no
// gradient at steps s0+1 .. t-1 can take those updates later, in order, with g = 0: the arithmetic is the same sequence of fp32
// operations, so the result is bit-identical to the dense optimizer's. State per table, per SUB-ROW (64 consecutive elements of a
// row = the work of one wave; a row of D elements has ceil(D / 64) of them): last[q] = step up to which sub-row q is current;
// sched[s] = (lr / bc1(s), sqrt(bc2(s))) of every step so far (evaluated on the host in double as for the dense kernel). A step
// brings the rows its batch READS up to t-1 before the forward pass (catch-up), updates the rows that RECEIVED gradient with step t
// after the backward pass (inside the dense launch of the other parameters: sbr_adam_step_rows), and a flush replays everything
// before any other reader (evaluation, state_dict, checkpoints) looks at the table.
// Duplicate rows in a batch: the first wave to raise claim[q] to the launch's token owns the sub-row.
// The replay is a dependent chain per element; what made the first version slow (52 us for 8,192 rows) was not its arithmetic but
// one L2 round trip per replayed step for sched[s]: a wave now fetches the schedule of up to 64 steps with ONE load (lane i holds
// step s0 + 1 + i) and broadcasts an entry per step with v_readlane; the loop is unrolled by four so that the square roots and
// divisions of neighbouring steps (which depend on the moments only, not on the parameter) overlap.
// (s_over, over): the schedule entry of step s_over comes from the arguments instead of the table (-1: none) — the launch that
// records the entry of its own step replays that step for the rows of its sweep
// A zero-gradient step of an element whose first moment is exactly zero — a row that has never received a gradient, or one idle for
// ~900 steps (0.9^n underflows) — is `v *= beta2; p *= 1 - lr wd` bit for bit under AdamW: lerp(m, 0) = +0, v * beta2 + 0 = v * beta2,
// and p - step * (+0 / denom) = p for any finite v (a NaN / inf second moment takes the general path). When that holds for the whole
// wave the replay skips the square root and the two divisions: 2 vector instructions per element-step instead of ~60. The sweep of a
// cold table (c4: a row is touched every ~4,000 steps) and of the rows a short run never reaches costs next to nothing then.
__device__ __forceinline__ bool adam_wave_is_idle(float me, float ve, const AdamHyper& h) {
  return h.decoupled && __all(me == 0.f && ve <= 3.4028234663852886e38f);
}
__device__ __forceinline__ void adam_idle_step(float& pe, float& me, float& ve, const AdamHyper& h) {
#pragma clang fp contract(off)
  pe *= h.decay;
  me = 0.f;
  ve = ve * h.b2;
}
__device__ __forceinline__ void adam_replay(float& pe, float& me, float& ve, int s_from, int s_to, const float2* __restrict__ sched,
                                            const AdamHyper& h, float zero, int lane, int s_over = -1,
                                            float2 over = make_float2(1.f, 1.f)) {
  for (int base = s_from; base <= s_to; base += 64) {                 // s_from / s_to are wave-uniform
    const int cnt = s_to - base + 1 < 64 ? s_to - base + 1 : 64;
    if (adam_wave_is_idle(me, ve, h)) {                               // stays true: m stays +0, v stays finite
      for (int i = 0; i < cnt; ++i) adam_idle_step(pe, me, ve, h);
      continue;
    }
    float2 mine = lane < cnt && base + lane != s_over ? sched[base + lane] : make_float2(1.f, 1.f);
    if (base + lane == s_over) mine = over;
    const int sx = __float_as_int(mine.x), sy = __float_as_int(mine.y);
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
      if (adam_wave_is_idle(me, ve, h)) break;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        adam_element(pe, zero, me, ve, h, __int_as_float(__builtin_amdgcn_readlane(sx, i + q)), __int_as_float(__builtin_amdgcn_readlane(sy, i + q)));
    }
    for (; i < cnt; ++i) {
      if (adam_wave_is_idle(me, ve, h)) adam_idle_step(pe, me, ve, h);
      else adam_element(pe, zero, me, ve, h, __int_as_float(__builtin_amdgcn_readlane(sx, i)), __int_as_float(__builtin_amdgcn_readlane(sy, i)));
    }
  }
}

// One wave = one sub-row (64 elements) of one row named by ids[j]. mode 0: catch-up to step t - 1. mode 1: catch-up to t - 1, then
// step t with the gradient, which is zeroed.
__device__ __forceinline__ void adam_subrow(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int D,
                                            int n_sub, const long* __restrict__ ids64, const int* __restrict__ ids32,
                                            const int* __restrict__ rowmap, long w, int* __restrict__ claim, int* __restrict__ last,
                                            const float2* __restrict__ sched, int t, const AdamHyper& h, float step_size, float bc2_sqrt,
                                            float zero, int mode, int lane) {
  const long j = w / n_sub;
  const int sub = (int)(w - j * n_sub);
  const long id = ids64 ? ids64[j] : (long)ids32[j];
  const int r = rowmap ? rowmap[id] : (int)id;
  if (r < 0) return;                                          // id without a row: flagged by the lookup kernel
  const long q = (long)r * n_sub + sub;
  const int token = 2 * t - 1 + mode;
  int old = 0;
  if (lane == 0) old = atomicMax(&claim[q], token);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old >= token) return;                                   // another wave of this launch owns the sub-row
  const int s0 = __builtin_amdgcn_readfirstlane(last[q]);
  const int c = sub * 64 + lane;
  const long e = (long)r * D + (c < D ? c : D - 1);
  float pe = p[e], me = m[e], ve = v[e];
  adam_replay(pe, me, ve, s0 + 1, t - 1, sched, h, zero, lane);
  if (mode == 1) adam_element(pe, g[e], me, ve, h, step_size, bc2_sqrt);
  if (c < D) {
    p[e] = pe;
    m[e] = me;
    v[e] = ve;
    if (mode == 1) g[e] = 0.f;
  }
  if (lane == 0) last[q] = mode == 1 ? t : t - 1;
}

__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int D, int n_sub, const long* __restrict__ ids64,
                                                        const int* __restrict__ ids32, const int* __restrict__ rowmap, long n,
                                                        int* __restrict__ claim, int* __restrict__ last,
                                                        float2* __restrict__ sched, int t, AdamHyper h, float step_size,
                                                        float bc2_sqrt, const float* __restrict__ zero_src, int mode) {
  const long w = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (mode == 1 && blockIdx.x == 0 && threadIdx.x == 0) sched[t] = make_float2(step_size, bc2_sqrt);
  if (w >= n * n_sub) return;
  adam_subrow(p, g, m, v, D, n_sub, ids64, ids32, rowmap, w, claim, last, sched, t, h, step_size, bc2_sqrt, zero_src[0], mode, threadIdx.x & 63);
}

// every sub-row of the table up to step t (before another reader looks at the table)
__global__ __launch_bounds__(256) void adam_rows_flush_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                              int D, int n_sub, long n_rows, int* __restrict__ last,
                                                              const float2* __restrict__ sched, int t, AdamHyper h,
                                                              const float* __restrict__ zero_src) {
  const long q = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= n_rows * n_sub) return;
  const int lane = threadIdx.x & 63;
  const int s0 = __builtin_amdgcn_readfirstlane(last[q]);
  if (s0 >= t) return;
  const long r = q / n_sub;
  const int c = (int)(q - r * n_sub) * 64 + lane;
  const long e = r * D + (c < D ? c : D - 1);
  float pe = p[e], me = m[e], ve = v[e];
  adam_replay(pe, me, ve, s0 + 1, t, sched, h, zero_src[0], lane);
  if (c < D) {
    p[e] = pe;
    m[e] = me;
    v[e] = ve;
  }
  if (lane == 0) last[q] = t;
}

// The optimizer launch of a step whose flat buffers hold one deferred table in [lo, hi): ONE launch for optimizer.step() +
// zero_grad(). Three kinds of workgroups:
//   rows   [row_blocks]: step t for the sub-rows that received gradient (mode 1 of adam_subrow, four waves each);
//   sweep  [sweep_blocks]: the sub-rows [sweep_lo, sweep_lo + n_sweep) (mod the table) that are NOT in this step's batch are brought
//          up to step t. The sweep visits every sub-row once in W steps, so no row is ever more than W steps behind: the catch-up in
//          front of a forward pass replays at most W steps per row (not ~n_rows / batch), and a flush W steps per row, not a whole
//          epoch's. The replay arithmetic is what it is (one zero-gradient step per row and step, as in the dense optimizer); here it
//          runs beside the HBM stream of the dense part. A sub-row of this step's batch is recognised by the claim its catch-up
//          left (token 2 t - 1, or the 2 t of its update wave): the sweep leaves it to the update wave.
//   dense: the dense kernel (gradient reset and loss read-out of adamw_kernel<true>) over [0, lo) and [hi, n).
// Dense and row/sweep workgroups alternate in the grid so that the arithmetic of the one overlaps the memory stream of the other.
__global__ __launch_bounds__(256) void adam_step_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, long n, long lo, long hi, int D, int n_sub,
                                                             const long* __restrict__ ids64, const int* __restrict__ ids32,
                                                             const int* __restrict__ rowmap, long n_ids, int* __restrict__ claim,
                                                             int* __restrict__ last, float2* __restrict__ sched, int t, AdamHyper h,
                                                             float step_size, float bc2_sqrt, const float* __restrict__ zero_src,
                                                             int row_blocks, int sweep_blocks, long sweep_lo, long n_sweep,
                                                             const double* __restrict__ cp_src, double* __restrict__ cp_dst, int cp_n) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) sched[t] = make_float2(step_size, bc2_sqrt);        // read by later launches only
    if ((int)threadIdx.x < cp_n) cp_dst[threadIdx.x] = cp_src[threadIdx.x];
  }
  const int n_alu = row_blocks + sweep_blocks, n_dense = (int)gridDim.x - n_alu;
  const int pair = n_alu < n_dense ? n_alu : n_dense;
  const int b = (int)blockIdx.x;
  bool dense;
  int idx;
  if (b < 2 * pair) { dense = (b & 1) == 0; idx = b >> 1; }
  else { dense = n_dense > n_alu; idx = b - pair; }
  if (!dense) {
    const int lane = threadIdx.x & 63;
    if (idx < row_blocks) {
      const long w = idx * 4L + (threadIdx.x >> 6);
      if (w < n_ids * n_sub)
        adam_subrow(p + lo, g + lo, m + lo, v + lo, D, n_sub, ids64, ids32, rowmap, w, claim, last, sched, t, h, step_size, bc2_sqrt, zero_src[0], 1, lane);
      return;
    }
    const long w = (idx - row_blocks) * 4L + (threadIdx.x >> 6);
    if (w >= n_sweep) return;
    const long n_q = (hi - lo) / D * n_sub;
    long q = sweep_lo + w;
    q = q >= n_q ? q - n_q : q;
    int old = 0;
    if (lane == 0) old = atomicMax(&claim[q], 2 * t - 1);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old >= 2 * t - 1) return;                               // a sub-row of this step's batch: its update wave owns it
    const int s0 = __builtin_amdgcn_readfirstlane(last[q]);
    if (s0 >= t) return;
    const long r = q / n_sub;
    const int c = (int)(q - r * n_sub) * 64 + lane;
    const long e = lo + r * D + (c < D ? c : D - 1);
    float pe = p[e], me = m[e], ve = v[e];
    adam_replay(pe, me, ve, s0 + 1, t, sched, h, zero_src[0], lane, t, make_float2(step_size, bc2_sqrt));
    if (c < D) {
      p[e] = pe;
      m[e] = me;
      v[e] = ve;
    }
    if (lane == 0) last[q] = t;
    return;
  }
  const long span = hi - lo, n_rest = n - span;
  if (((lo | hi | n) & 3) == 0) {
    // 16 bytes per lane (the flat buffers' segments are 256-byte aligned: a quad never straddles the table): a quarter of the memory
    // instructions of the element loop below for the same bytes; the arithmetic per element is the same function, bit for bit
    const long n4 = n_rest >> 2;
    for (long d4 = idx * (long)blockDim.x + threadIdx.x; d4 < n4; d4 += (long)n_dense * blockDim.x) {
      const long d = d4 << 2;
      const long e = d < lo ? d : d + span;
      float4 pe = *reinterpret_cast<const float4*>(p + e), me = *reinterpret_cast<const float4*>(m + e), ve = *reinterpret_cast<const float4*>(v + e);
      const float4 ge = *reinterpret_cast<const float4*>(g + e);
      adam_element(pe.x, ge.x, me.x, ve.x, h, step_size, bc2_sqrt);
      adam_element(pe.y, ge.y, me.y, ve.y, h, step_size, bc2_sqrt);
      adam_element(pe.z, ge.z, me.z, ve.z, h, step_size, bc2_sqrt);
      adam_element(pe.w, ge.w, me.w, ve.w, h, step_size, bc2_sqrt);
      *reinterpret_cast<float4*>(p + e) = pe;
      *reinterpret_cast<float4*>(m + e) = me;
      *reinterpret_cast<float4*>(v + e) = ve;
      if ((__float_as_uint(ge.x) | __float_as_uint(ge.y) | __float_as_uint(ge.z) | __float_as_uint(ge.w)) != 0u)
        *reinterpret_cast<float4*>(g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  for (long d = idx * (long)blockDim.x + threadIdx.x; d < n_rest; d += (long)n_dense * blockDim.x) {
    const long e = d < lo ? d : d + span;
    float pe = p[e], me = m[e], ve = v[e];
    const float ge = g[e];
    adam_element(pe, ge, me, ve, h, step_size, bc2_sqrt);
    p[e] = pe;
    m[e] = me;
    v[e] = ve;
    if (__float_as_uint(ge) != 0u) g[e] = 0.f;
  }
}

__device__ float sbr_adam_zero = 0.f;

__global__ void adagrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s, long n, float lr,
                               float eps, float wd) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float pe = p[e];
    const float ge = g[e] + wd * pe;
    const float se = s[e] + ge * ge;
    p[e] = pe - lr * (ge / (sqrtf(se) + eps));
    s[e] = se;
  }
}

static int grid_for(long n) {
  int b = sbr_cdiv(n, 256);
  return b > 8192 ? 8192 : (b < 1 ? 1 : b);
}

// kind: 0 AdamW, 1 Adam. step is the 1-based step count (bias corrections are evaluated on the host in double).
extern "C" int sbr_adam_step(int kind, float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                             double eps, double wd, long step, void* stream) {
  SBR_REQUIRE(kind == 0 || kind == 1, "sbr_adam_step: unknown kind %d", kind);
  SBR_REQUIRE(p && g && m && v, "sbr_adam_step: null operand");
  SBR_REQUIRE(step >= 1, "sbr_adam_step: step must be >= 1");
  if (n == 0) return SBR_OK;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, kind == 0);
  adamw_kernel<false><<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, const_cast<float*>(g), m, v, n, h, (float)(lr / bc1), (float)sqrt(bc2));
  SBR_CHECK_LAUNCH("sbr_adam_step");
  return SBR_OK;
}

// sbr_adam_step followed by zeroing the gradient (optimizer.step(); optimizer.zero_grad(), train/trainer.py:221-222) in one launch
extern "C" int sbr_adam_step_zero_grad(int kind, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2,
                                       double eps, double wd, long step, const double* copy_src, double* copy_dst, int copy_n,
                                       void* stream) {
  SBR_REQUIRE(kind == 0 || kind == 1, "sbr_adam_step_zero_grad: unknown kind %d", kind);
  SBR_REQUIRE(p && g && m && v, "sbr_adam_step_zero_grad: null operand");
  SBR_REQUIRE(step >= 1, "sbr_adam_step_zero_grad: step must be >= 1");
  SBR_REQUIRE(copy_n >= 0 && copy_n <= 256 && (copy_n == 0 || (copy_src && copy_dst)), "sbr_adam_step_zero_grad: bad copy request");
  SBR_REQUIRE(n >= 1 || copy_n == 0, "sbr_adam_step_zero_grad: a copy needs a non-empty step");
  if (n == 0) return SBR_OK;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, kind == 0);
  adamw_kernel<true><<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, h, (float)(lr / bc1), (float)sqrt(bc2), copy_src, copy_dst,
                                                                   copy_n);
  SBR_CHECK_LAUNCH("sbr_adam_step_zero_grad");
  return SBR_OK;
}

// Deferred row-wise Adam / AdamW over one [n_rows, D] lookup table (see adam_subrow). mode 0: bring the rows named by
// ids (int64 ids64 or int32 ids32, optionally mapped through rowmap) up to step - 1; mode 1: the same, then apply `step` with
// their gradient rows, zero those gradient rows and record the step's scalars in sched[step]; mode 2: flush all rows to `step`.
// claim / last: int32 [n_rows * ceil(D / 64)] (one entry per 64-element sub-row), zero-initialised by the caller;
// sched: float2 [>= step + 1].
extern "C" int sbr_adam_rows(int kind, int mode, float* p, float* g, float* m, float* v, long n_rows, int D, const long* ids64,
                             const int* ids32, const int* rowmap, long n, int* claim, int* last, void* sched, double lr, double b1,
                             double b2, double eps, double wd, long step, void* stream) {
  SBR_REQUIRE(kind == 0 || kind == 1, "sbr_adam_rows: unknown kind %d", kind);
  SBR_REQUIRE(mode >= 0 && mode <= 2, "sbr_adam_rows: unknown mode %d", mode);
  SBR_REQUIRE(p && m && v && last && sched && D >= 1, "sbr_adam_rows: null operand");
  SBR_REQUIRE(step >= 1 && step < (1L << 30), "sbr_adam_rows: step %ld out of range", step);
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, kind == 0);
  float* zero = nullptr;
  if (hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(sbr_adam_zero)) != hipSuccess) {
    sbr_set_error("sbr_adam_rows: hipGetSymbolAddress failed");
    return SBR_ERR_HIP;
  }
  hipStream_t s = (hipStream_t)stream;
  const int n_sub = (D + 63) / 64;
  if (mode == 2) {
    if (n_rows == 0) return SBR_OK;
    adam_rows_flush_kernel<<<sbr_cdiv(n_rows * n_sub, 4), 256, 0, s>>>(p, m, v, D, n_sub, n_rows, last, (const float2*)sched, (int)step, h, zero);
    SBR_CHECK_LAUNCH("sbr_adam_rows (flush)");
    return SBR_OK;
  }
  SBR_REQUIRE(claim && (ids64 || ids32), "sbr_adam_rows: null operand");
  SBR_REQUIRE(mode == 0 || g, "sbr_adam_rows: the update needs the gradient");
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const long blocks = n > 0 ? sbr_cdiv(n * n_sub, 4) : 1;     // mode 1 with no rows still records sched[step]
  adam_rows_kernel<<<blocks, 256, 0, s>>>(p, g, m, v, D, n_sub, ids64, ids32, rowmap, n, claim, last, (float2*)sched, (int)step, h,
                                          (float)(lr / bc1), (float)sqrt(bc2), zero, mode);
  SBR_CHECK_LAUNCH("sbr_adam_rows");
  return SBR_OK;
}

// optimizer.step() + zero_grad() of a step with ONE deferred table in one launch: the flat buffers p / g / m / v of n elements hold
// the table in [lo, hi) (hi - lo = n_rows * D); its rows named by ids get mode 1 of sbr_adam_rows, every other element the dense step
// of sbr_adam_step_zero_grad (gradient reset, copy_n doubles copy_src -> copy_dst). Untouched rows of the table are not read —
// except the n_sweep sub-rows from sweep_lo on (cyclic), which are brought up to `step` (see the kernel); n_sweep > 0 REQUIRES that the
// step's catch-up (sbr_adam_rows mode 0, same ids, same step) ran before: its claims are how the sweep tells the batch's rows.
extern "C" int sbr_adam_step_rows(int kind, float* p, float* g, float* m, float* v, long n, long lo, long hi, int D, const long* ids64,
                                  const int* ids32, const int* rowmap, long n_ids, int* claim, int* last, void* sched, double lr,
                                  double b1, double b2, double eps, double wd, long step, long sweep_lo, long n_sweep,
                                  const double* copy_src, double* copy_dst, int copy_n, void* stream) {
  SBR_REQUIRE(kind == 0 || kind == 1, "sbr_adam_step_rows: unknown kind %d", kind);
  SBR_REQUIRE(p && g && m && v && claim && last && sched && (ids64 || ids32 || n_ids == 0), "sbr_adam_step_rows: null operand");
  SBR_REQUIRE(0 <= lo && lo <= hi && hi <= n && D >= 1 && (hi - lo) % D == 0, "sbr_adam_step_rows: bad table range [%ld, %ld) of %ld, D = %d", lo, hi, n, D);
  SBR_REQUIRE(step >= 1 && step < (1L << 30), "sbr_adam_step_rows: step %ld out of range", step);
  SBR_REQUIRE(copy_n >= 0 && copy_n <= 256 && (copy_n == 0 || (copy_src && copy_dst)), "sbr_adam_step_rows: bad copy request");
  {
    const long n_q = (hi - lo) / D * ((D + 63) / 64);
    SBR_REQUIRE(n_sweep >= 0 && n_sweep <= n_q && (n_sweep == 0 || (sweep_lo >= 0 && sweep_lo < n_q)),
                "sbr_adam_step_rows: bad sweep [%ld, +%ld) of %ld sub-rows", sweep_lo, n_sweep, n_q);
  }
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, kind == 0);
  float* zero = nullptr;
  if (hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(sbr_adam_zero)) != hipSuccess) {
    sbr_set_error("sbr_adam_step_rows: hipGetSymbolAddress failed");
    return SBR_ERR_HIP;
  }
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const int n_sub = (D + 63) / 64;
  const int row_blocks = n_ids > 0 ? sbr_cdiv(n_ids * n_sub, 4) : 0;
  const int sweep_blocks = sbr_cdiv(n_sweep, 4);
  const int dense_blocks = grid_for(n - (hi - lo));             // >= 1: block 0 records sched[step] and makes the copy in any case
  adam_step_rows_kernel<<<row_blocks + sweep_blocks + dense_blocks, 256, 0, (hipStream_t)stream>>>(
      p, g, m, v, n, lo, hi, D, n_sub, ids64, ids32, rowmap, n_ids, claim, last, (float2*)sched, (int)step, h, (float)(lr / bc1), (float)sqrt(bc2), zero,
      row_blocks, sweep_blocks, sweep_lo, n_sweep, copy_src, copy_dst, copy_n);
  SBR_CHECK_LAUNCH("sbr_adam_step_rows");
  return SBR_OK;
}

extern "C" int sbr_adagrad_step(float* p, const float* g, float* state_sum, long n, double lr, double eps, double wd,
                                void* stream) {
  SBR_REQUIRE(p && g && state_sum, "sbr_adagrad_step: null operand");
  if (n == 0) return SBR_OK;
  adagrad_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, state_sum, n, (float)lr, (float)eps, (float)wd);
  SBR_CHECK_LAUNCH("sbr_adagrad_step");
  return SBR_OK;
}
