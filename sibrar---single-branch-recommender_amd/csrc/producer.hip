// Native batch producer: the host side of one training step — collate with bit-exact negative sampling, modality draw, launch
// plan, packed upload — on ONE C++ thread that never touches the Python interpreter.
//
// What it replaces (reference: the main thread of train/trainer.py:204 iterating its DataLoader):
//   * data/dataloader.py:154-198  NegativeSamplingDataLoader._neg_sampling_collate_fn (uniform_recbole): all B * n_neg slots drawn
//     with np.random.choice(items_in_split, n) on the global legacy MT19937 stream, colliding slots redrawn until none hits a
//     positive of its user. Same draws in the same order (MtStream below == sbr_host_mt19937_randint); the `v in positives`
//     test runs on the resident interaction CSR — large rounds with the GPU kernel of sbr_csr_contains on the producer's own
//     stream, small ones by binary search on the host copy;
//   * algorithms/sgd_alg.py:1904-1927 + utilities/utils.py:60-90  one or two modalities per index slot from the entity's
//     np.random.default_rng(sampling_seed): PCG64 (XSL-RR 128/64) with numpy's 32-bit Lemire bounded draws, Floyd's algorithm and
//     the Fisher-Yates pass for k = 2 — the Python formulation lives in sampling.sample_modality_ids;
//   * engine.FusedTrainStep.prepare: per-modality counts padded to the graph's bucket grid, one packed pinned staging buffer
//     [users | users[0] | items | items[0] | user draw | item draw | dropout seed] and ONE host-to-device copy per batch.
// The three Python threads of round 1 (collate -> prepare -> launch) held the GIL for ~0.75 ms of every 0.87 ms step at
// B = 8192 and for longer than the GPU work at B = 256; with this producer the launch thread is the only Python on the path.
// Generator states are handed in when an epoch starts and handed back when it ends (or is stopped), so numpy's global stream
// and the entities' generators continue exactly where the reference's would.
#include <chrono>
#include <condition_variable>
#include <stdio.h>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <math.h>
#include "common.h"

extern "C" int sbr_csr_contains(const long* indptr, const int* indices, const long* rows, const long* cols, long n,
                                unsigned char* out, void* stream);

namespace {

#define MT_N 624
#define MT_M 397

struct Mt {
  uint32_t key[MT_N];
  int p;
  uint32_t rng, mask;
  uint32_t out[MT_N];                                        // tempered outputs of key[tp0 .. MT_N): see fill()
  int tp = MT_N;                                             // key position out[] was tempered for up to MT_N (MT_N: none)
  void regen() {
    uint32_t* mt = key;
    int i;
    uint32_t y;
    for (i = 0; i < MT_N - MT_M; ++i) {
      y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; i < MT_N - 1; ++i) {
      y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  void set_range(long n) {
    rng = (uint32_t)(n - 1);
    mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  }
  inline long next() {                                      // RandomState.randint(0, n): masked rejection
    if (rng == 0) return 0;
    for (;;) {
      if (p == MT_N) { regen(); p = 0; }
      uint32_t y = key[p++];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      const uint32_t v = y & mask;
      if (v <= rng) return (long)v;
    }
  }
  // n draws of randint(0, rng + 1) into dst (through map when given): the same words in the same order as n calls of next(), but
  // the tempering runs over the whole remaining state block at once (a loop the compiler vectorises) and the rejection loop only
  // masks and compares — the per-call form spent 0.4 ms on the 81,920 negatives of a batch of 8,192
  void fill(long n, long* dst, const long* map) {
    if (rng == 0) { for (long i = 0; i < n; ++i) dst[i] = map ? map[0] : 0; return; }
    long i = 0;
    while (i < n) {
      if (p == MT_N) { regen(); p = 0; }
      const int lo = p;
      for (int q = lo; q < MT_N; ++q) {
        uint32_t y = key[q];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        out[q] = y & mask;
      }
      int q = lo;
      const uint32_t r = rng;
      for (; q < MT_N && i < n; ++q) {
        const uint32_t v = out[q];
        dst[i] = map ? map[v <= r ? v : 0] : (long)v;        // (a rejected word is overwritten by the next accepted one)
        i += v <= r;
      }
      p = q;
    }
  }
};

// numpy.random.PCG64 (pcg64.h: step, then XSL-RR output) with the bit generator's 32-bit half buffer (low half first)
struct Pcg64 {
  unsigned __int128 state, inc;
  int has_uint32;
  uint32_t uinteger;
  inline uint64_t next64() {
    const unsigned __int128 mult = (((unsigned __int128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;
    state = state * mult + inc;
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((-rot) & 63));
  }
  inline uint32_t next32() {
    if (has_uint32) { has_uint32 = 0; return uinteger; }
    const uint64_t n = next64();
    has_uint32 = 1;
    uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
  }
  // Generator.integers(0, rng + 1) for rng < 2^32 - 1: distributions.c buffered_bounded_lemire_uint32
  inline uint32_t bounded(uint32_t rng) {
    if (rng == 0) return 0;                                  // nothing is drawn for a one-value range
    if (rng == 0xFFFFFFFFu) return next32();
    const uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)next32() * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
      const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
      while (leftover < threshold) {
        m = (uint64_t)next32() * rng_excl;
        leftover = (uint32_t)m;
      }
    }
    return (uint32_t)(m >> 32);
  }
};

struct Entity {
  int enabled = 0, n_mod = 0, k = 0, central = -1;          // central >= 0: column 0 is `central`, column 1 one of the others
  long slots_per_row = 1;                                    // index slots per batch row (1 user side, 1 + n_neg item side)
  Pcg64 rng{};
};

// positions [n_slots, k] (int8) like sampling.sample_modalities; counts[m] over all n_slots * k entries
static void draw_modalities(Entity& e, long n_slots, int8_t* out, long* counts) {
  for (int m = 0; m < e.n_mod; ++m) counts[m] = 0;
  Pcg64& g = e.rng;
  if (e.central >= 0) {                                      // sgd_alg.py:1921-1924: [central, one of the others]
    const uint32_t r = (uint32_t)(e.n_mod - 2);
    for (long s = 0; s < n_slots; ++s) {
      const int pick = (int)g.bounded(r);
      const int other = pick < e.central ? pick : pick + 1;
      out[2 * s] = (int8_t)e.central;
      out[2 * s + 1] = (int8_t)other;
      ++counts[e.central];
      ++counts[other];
    }
  } else if (e.k == 1) {
    const uint32_t r = (uint32_t)(e.n_mod - 1);
    for (long s = 0; s < n_slots; ++s) {
      const int v = (int)g.bounded(r);
      out[s] = (int8_t)v;
      ++counts[v];
    }
  } else {                                                   // k == 2: Floyd (j = n - 2, n - 1) + Fisher-Yates step i = 1
    const uint32_t r0 = (uint32_t)(e.n_mod - 2), r1 = (uint32_t)(e.n_mod - 1);
    for (long s = 0; s < n_slots; ++s) {
      const int w0 = (int)g.bounded(r0), w1 = (int)g.bounded(r1), w2 = (int)g.bounded(1u);
      const int v0 = w0, v1 = (w1 == v0) ? e.n_mod - 1 : w1;
      const bool swap = w2 == 0;
      const int a = swap ? v1 : v0, b = swap ? v0 : v1;
      out[2 * s] = (int8_t)a;
      out[2 * s + 1] = (int8_t)b;
      ++counts[a];
      ++counts[b];
    }
  }
}

// engine._EntityRun.plan: padded capacities of the per-modality slot lists (graph mode)
static void pad_counts(long* counts, int n_mod, long R) {
  long bucket = (long)(5.0 * sqrt((double)R));             // int(5.0 * math.sqrt(R)): truncation, like the Python formulation
  bucket = (bucket + 63) / 64 * 64;
  if (bucket < 64) bucket = 64;
  const long off = (R / n_mod + bucket / 2) % bucket;
  for (int m = 0; m < n_mod; ++m) {
    if (counts[m] <= 0) continue;
    long c = counts[m] - off;
    if (c < 0) c = 0;
    counts[m] = (c + bucket - 1) / bucket * bucket + off;
  }
}

#define DESC_WORDS 32
// descriptor of one produced batch (longs): 0 slot, 1 B, 2 packed bytes, 3..8 segment offsets (u, i, user draw, item draw, seed,
// end), 9 user R, 10 item R, 11.. user counts[8], 19.. item counts[8], 27 batch number
struct Producer {
  int device = 0;
  long B = 0;
  int n_neg = 0;
  long n_cand = 0;
  std::vector<long> items_in_split;                          // empty: identity
  const long* h_indptr = nullptr;
  const int* h_indices = nullptr;
  const long* d_indptr = nullptr;
  const int* d_indices = nullptr;
  long host_below = 1024;
  int pad = 1;
  int n_slots = 0;
  long slot_bytes = 0;
  std::vector<void*> slot_dev;
  std::vector<unsigned char*> slot_host;                     // pinned staging, one per slot
  std::vector<hipEvent_t> ready, consumed;
  std::vector<char> used;
  Entity ent[2];                                             // 0 user, 1 item
  Mt mt{};
  long seed_base = 0, n_prepared = 0;
  // epoch
  const long* rows_e = nullptr;
  const long* cols_e = nullptr;
  long n_inter = 0, first = 0, stride = 0, n_batches = 0;
  // scratch
  long *q_users = nullptr, *q_items = nullptr;               // pinned [B * n_neg]
  unsigned char* q_flags = nullptr;                          // pinned
  long *dq_users = nullptr, *dq_items = nullptr;             // device
  unsigned char* dq_flags = nullptr;
  std::vector<long> values, todo;
  hipStream_t stream = nullptr;
  // thread + queue
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::vector<long>> queue;
  bool running = false, stop = false, done = false;
  long produced = 0;
  std::string error;
  // stage clocks of the producer thread (seconds, summed over the epoch; printed at sbr_producer_stop when SBR_PRODUCER_TIMING=1)
  double t_wait = 0, t_draw = 0, t_member = 0, t_pack = 0, t_mod = 0, t_upload = 0;
};

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool hip_ok(Producer* p, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  p->error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

// flags[q] = items[q] in row users[q]; large rounds on the GPU (device CSR), small ones on the host copy
static bool membership(Producer* p, long n) {
  if (n >= p->host_below && p->d_indptr) {
    if (!hip_ok(p, hipMemcpyAsync(p->dq_users, p->q_users, n * 8, hipMemcpyHostToDevice, p->stream), "producer: H2D")) return false;
    if (!hip_ok(p, hipMemcpyAsync(p->dq_items, p->q_items, n * 8, hipMemcpyHostToDevice, p->stream), "producer: H2D")) return false;
    if (sbr_csr_contains(p->d_indptr, p->d_indices, p->dq_users, p->dq_items, n, p->dq_flags, p->stream) != SBR_OK) {
      p->error = "producer: sbr_csr_contains failed";
      return false;
    }
    if (!hip_ok(p, hipMemcpyAsync(p->q_flags, p->dq_flags, n, hipMemcpyDeviceToHost, p->stream), "producer: D2H")) return false;
    return hip_ok(p, hipStreamSynchronize(p->stream), "producer: sync");
  }
  for (long q = 0; q < n; ++q) {
    const long u = p->q_users[q], v = p->q_items[q];
    long lo = p->h_indptr[u], hi = p->h_indptr[u + 1];
    const long end = hi;
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (p->h_indices[mid] < v) lo = mid + 1; else hi = mid;
    }
    p->q_flags[q] = (lo < end && p->h_indices[lo] == v) ? 1 : 0;
  }
  return true;
}

static bool produce_one(Producer* p, long bno, std::vector<long>& desc) {
  const int slot = (int)(bno % p->n_slots);
  double t0 = now_s();
  if (p->used[slot]) {
    if (!hip_ok(p, hipEventSynchronize(p->consumed[slot]), "producer: wait for the consumer")) return false;
  }
  { const double t1 = now_s(); p->t_wait += t1 - t0; t0 = t1; }
  const long lo = bno * p->stride + p->first;
  long B = p->B;
  if (lo + B > p->n_inter) B = p->n_inter - lo;
  if (B <= 0) { p->error = "producer: empty batch"; return false; }
  const int N = 1 + p->n_neg;
  const long total = B * p->n_neg;
  const long* users = p->rows_e + lo;
  const long* pos_items = p->cols_e + lo;
  // ---- collate (data/dataloader.py:154-198)
  long* values = p->values.data();
  long* todo = p->todo.data();
  const bool ident = p->items_in_split.empty();
  p->mt.fill(total, values, ident ? nullptr : p->items_in_split.data());
  long m = total;
  for (int j = 0; j < p->n_neg; ++j) {                       // slot s = j * B + b belongs to user b (no division per slot)
    const long base = (long)j * B;
    for (long b = 0; b < B; ++b) { p->q_users[base + b] = users[b]; p->q_items[base + b] = values[base + b]; todo[base + b] = base + b; }
  }
  { const double t1 = now_s(); p->t_draw += t1 - t0; t0 = t1; }
  bool first_round = true;
  while (m > 0) {
    if (!first_round) {
      for (long q = 0; q < m; ++q) {                         // one randint(n_cand, m) call: m sequential draws
        const long r = p->mt.next();
        const long v = ident ? r : p->items_in_split[r];
        values[todo[q]] = v;
        p->q_users[q] = users[todo[q] % B];
        p->q_items[q] = v;
      }
    }
    first_round = false;
    if (!membership(p, m)) return false;
    long m2 = 0;
    for (long q = 0; q < m; ++q)
      if (p->q_flags[q]) todo[m2++] = todo[q];
    m = m2;
  }
  { const double t1 = now_s(); p->t_member += t1 - t0; t0 = t1; }
  // ---- packed staging buffer: [u | u[0] | items | items[0] | user draw | item draw | seed], 16-byte aligned segments
  unsigned char* h = p->slot_host[slot];
  long off[6];
  off[0] = 0;
  off[1] = (off[0] + (B + 1) * 8 + 15) & ~15L;
  const long Ru = p->ent[0].enabled ? B * p->ent[0].k : 0;
  const long Ri = B * N * p->ent[1].k;
  off[2] = (off[1] + (B * N + 1) * 8 + 15) & ~15L;
  off[3] = (off[2] + Ru + 15) & ~15L;
  off[4] = (off[3] + Ri + 15) & ~15L;
  off[5] = (off[4] + 8 + 15) & ~15L;
  if (off[5] > p->slot_bytes) { p->error = "producer: slot too small"; return false; }
  long* hu = (long*)(h + off[0]);
  for (long b = 0; b < B; ++b) hu[b] = users[b];
  hu[B] = users[0];
  long* hi_ = (long*)(h + off[1]);
  for (long b = 0; b < B; ++b) hi_[b * N] = pos_items[b];
  for (int j = 0; j < p->n_neg; ++j) {
    const long* v = values + (long)j * B;
    for (long b = 0; b < B; ++b) hi_[b * N + 1 + j] = v[b];
  }
  hi_[B * N] = hi_[0];
  { const double t1 = now_s(); p->t_pack += t1 - t0; t0 = t1; }
  desc.assign(DESC_WORDS, 0);
  long* cu = desc.data() + 11;
  long* ci = desc.data() + 19;
  // modality draws: user side first (SingleBranchNet.forward, sgd_alg.py:2121-2122)
  if (p->ent[0].enabled) {
    draw_modalities(p->ent[0], B, (int8_t*)(h + off[2]), cu);
    if (p->pad) pad_counts(cu, p->ent[0].n_mod, Ru);
  }
  draw_modalities(p->ent[1], B * N, (int8_t*)(h + off[3]), ci);
  if (p->pad) pad_counts(ci, p->ent[1].n_mod, Ri);
  { const double t1 = now_s(); p->t_mod += t1 - t0; t0 = t1; }
  p->n_prepared += 1;
  *(long*)(h + off[4]) = (p->seed_base + 2 * p->n_prepared) & 0x3FFFFFFFFFFFFFFFL;
  if (!hip_ok(p, hipMemcpyAsync(p->slot_dev[slot], h, off[5], hipMemcpyHostToDevice, p->stream), "producer: upload")) return false;
  if (!hip_ok(p, hipEventRecord(p->ready[slot], p->stream), "producer: event")) return false;
  p->used[slot] = 1;
  p->t_upload += now_s() - t0;
  desc[0] = slot; desc[1] = B; desc[2] = off[5];
  for (int q = 0; q < 6; ++q) desc[3 + q] = off[q];
  desc[9] = Ru; desc[10] = Ri; desc[27] = bno;
  return true;
}

static void run(Producer* p) {
  if (hipSetDevice(p->device) != hipSuccess) {
    std::lock_guard<std::mutex> lk(p->mu);
    p->error = "producer: hipSetDevice failed";
    p->done = true;
    p->cv.notify_all();
    return;
  }
  for (long b = 0; b < p->n_batches; ++b) {
    {
      std::unique_lock<std::mutex> lk(p->mu);
      // at most n_slots - 2 batches ahead of the consumer: a slot is rewritten only after its `consumed` event, which the
      // consumer records when it has queued the step that reads it
      p->cv.wait(lk, [&] { return p->stop || (long)p->queue.size() < (long)p->n_slots - 2; });
      if (p->stop) break;
    }
    std::vector<long> desc;
    const bool ok = produce_one(p, b, desc);
    std::lock_guard<std::mutex> lk(p->mu);
    if (!ok) break;
    p->queue.push_back(std::move(desc));
    p->produced = b + 1;
    p->cv.notify_all();
  }
  std::lock_guard<std::mutex> lk(p->mu);
  p->done = true;
  p->cv.notify_all();
}

}  // namespace

extern "C" {

// slot_dev: n_slots device buffers of slot_bytes each (allocated by the caller); d_indptr / d_indices: device copy of the
// interaction CSR (may be NULL: host search only). Scratch (pinned staging, query buffers, stream, events) is owned by the handle.
void* sbr_producer_create(int device, long B, int n_neg, long n_cand, const long* items_in_split, const long* h_indptr,
                          const int* h_indices, const long* d_indptr, const int* d_indices, long host_below, int pad, int n_slots,
                          long slot_bytes, void* const* slot_dev) {
  if (B < 1 || n_neg < 0 || n_cand < 1 || n_cand - 1 > 0xFFFFFFFFL || !h_indptr || !h_indices || n_slots < 3 || !slot_dev) {
    sbr_set_error("sbr_producer_create: bad arguments");
    return nullptr;
  }
  Producer* p = new Producer();
  p->device = device; p->B = B; p->n_neg = n_neg; p->n_cand = n_cand;
  if (items_in_split) p->items_in_split.assign(items_in_split, items_in_split + n_cand);
  p->h_indptr = h_indptr; p->h_indices = h_indices; p->d_indptr = d_indptr; p->d_indices = d_indices;
  p->host_below = host_below; p->pad = pad; p->n_slots = n_slots; p->slot_bytes = slot_bytes;
  p->mt.set_range(n_cand);
  bool ok = hipSetDevice(device) == hipSuccess;
  ok = ok && hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, -1) == hipSuccess;
  const long total = B * (n_neg > 0 ? n_neg : 1);
  ok = ok && hipHostMalloc((void**)&p->q_users, total * 8) == hipSuccess && hipHostMalloc((void**)&p->q_items, total * 8) == hipSuccess &&
       hipHostMalloc((void**)&p->q_flags, total) == hipSuccess;
  if (ok && d_indptr)
    ok = hipMalloc((void**)&p->dq_users, total * 8) == hipSuccess && hipMalloc((void**)&p->dq_items, total * 8) == hipSuccess &&
         hipMalloc((void**)&p->dq_flags, total) == hipSuccess;
  p->values.resize(total);
  p->todo.resize(total);
  for (int s = 0; ok && s < n_slots; ++s) {
    unsigned char* h = nullptr;
    hipEvent_t a, b;
    ok = hipHostMalloc((void**)&h, slot_bytes) == hipSuccess && hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess;
    if (!ok) break;
    p->slot_dev.push_back(slot_dev[s]);
    p->slot_host.push_back(h);
    p->ready.push_back(a);
    p->consumed.push_back(b);
    p->used.push_back(0);
  }
  if (!ok) {
    sbr_set_error("sbr_producer_create: HIP resource allocation failed");
    delete p;
    return nullptr;
  }
  return p;
}

// which: 0 user side, 1 item side. k = 1 | 2; central >= 0: central-modality regularisation (k = 2).
int sbr_producer_set_entity(void* handle, int which, int enabled, int n_mod, int k, int central) {
  SBR_REQUIRE(handle && (which == 0 || which == 1), "sbr_producer_set_entity: bad handle / side");
  SBR_REQUIRE(!enabled || (n_mod >= 1 && n_mod <= 8 && (k == 1 || k == 2) && k <= n_mod && central < n_mod && (central < 0 || (k == 2 && n_mod >= 2))),
              "sbr_producer_set_entity: n_mod=%d k=%d central=%d not supported", n_mod, k, central);
  Producer* p = (Producer*)handle;
  p->ent[which].enabled = enabled; p->ent[which].n_mod = n_mod; p->ent[which].k = k; p->ent[which].central = central;
  return SBR_OK;
}

// Starts the producer thread for one epoch. rows_e / cols_e: the epoch's (user, item) pairs in visiting order (host, must
// outlive the epoch); batch b covers pairs [b * stride + first, + B). mt_key[624] / mt_pos: numpy's global legacy state;
// pcg[2][6]: per side (state hi, state lo, inc hi, inc lo, has_uint32, uinteger) of the entity's PCG64 generator.
int sbr_producer_start(void* handle, const long* rows_e, const long* cols_e, long n_inter, long first, long stride, long n_batches,
                       const unsigned int* mt_key, int mt_pos, const unsigned long long* pcg, long seed_base, long n_prepared) {
  SBR_REQUIRE(handle && rows_e && cols_e && mt_key && pcg, "sbr_producer_start: null operand");
  Producer* p = (Producer*)handle;
  SBR_REQUIRE(!p->running, "sbr_producer_start: an epoch is still running (sbr_producer_stop first)");
  SBR_REQUIRE(p->ent[1].enabled, "sbr_producer_start: the item side must be configured (sbr_producer_set_entity)");
  SBR_REQUIRE(mt_pos >= 0 && mt_pos <= MT_N && n_batches >= 0 && stride >= 1, "sbr_producer_start: bad state / geometry");
  memcpy(p->mt.key, mt_key, sizeof(p->mt.key));
  p->mt.p = mt_pos;
  for (int w = 0; w < 2; ++w) {
    Pcg64& g = p->ent[w].rng;
    g.state = (((unsigned __int128)pcg[6 * w]) << 64) | pcg[6 * w + 1];
    g.inc = (((unsigned __int128)pcg[6 * w + 2]) << 64) | pcg[6 * w + 3];
    g.has_uint32 = (int)pcg[6 * w + 4];
    g.uinteger = (uint32_t)pcg[6 * w + 5];
  }
  p->rows_e = rows_e; p->cols_e = cols_e; p->n_inter = n_inter; p->first = first; p->stride = stride; p->n_batches = n_batches;
  p->seed_base = seed_base; p->n_prepared = n_prepared;
  p->queue.clear();
  p->stop = false; p->done = false; p->produced = 0; p->error.clear();
  for (auto& u : p->used) u = 0;
  p->t_wait = p->t_draw = p->t_member = p->t_pack = p->t_mod = p->t_upload = 0;
  p->running = true;
  p->th = std::thread(run, p);
  return SBR_OK;
}

// Blocks (without the interpreter lock: ctypes releases it) until the next batch is ready. desc: 32 longs (see Producer).
// Returns 0: a batch, 1: the epoch is over, other: error.
int sbr_producer_next(void* handle, long* desc) {
  SBR_REQUIRE(handle && desc, "sbr_producer_next: null operand");
  Producer* p = (Producer*)handle;
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv.wait(lk, [&] { return !p->queue.empty() || p->done; });
  if (!p->queue.empty()) {
    memcpy(desc, p->queue.front().data(), DESC_WORDS * sizeof(long));
    p->queue.pop_front();
    p->cv.notify_all();
    return SBR_OK;
  }
  if (!p->error.empty()) {
    sbr_set_error("%s", p->error.c_str());
    return SBR_ERR_HIP;
  }
  return 1;
}

// the consumer's stream waits for the upload of `slot` / records that everything queued so far has read it
int sbr_producer_wait(void* handle, int slot, void* stream) {
  SBR_REQUIRE(handle, "sbr_producer_wait: null handle");
  Producer* p = (Producer*)handle;
  SBR_REQUIRE(slot >= 0 && slot < p->n_slots, "sbr_producer_wait: slot %d", slot);
  if (hipStreamWaitEvent((hipStream_t)stream, p->ready[slot], 0) != hipSuccess) {
    sbr_set_error("sbr_producer_wait: hipStreamWaitEvent failed");
    return SBR_ERR_HIP;
  }
  return SBR_OK;
}

int sbr_producer_release(void* handle, int slot, void* stream) {
  SBR_REQUIRE(handle, "sbr_producer_release: null handle");
  Producer* p = (Producer*)handle;
  SBR_REQUIRE(slot >= 0 && slot < p->n_slots, "sbr_producer_release: slot %d", slot);
  if (hipEventRecord(p->consumed[slot], (hipStream_t)stream) != hipSuccess) {
    sbr_set_error("sbr_producer_release: hipEventRecord failed");
    return SBR_ERR_HIP;
  }
  return SBR_OK;
}

// Stops the epoch (no-op when none runs) and hands the generator states back as they stand after the last PRODUCED batch.
// out_counters: [batches produced, n_prepared].
int sbr_producer_stop(void* handle, unsigned int* mt_key, int* mt_pos, unsigned long long* pcg, long* out_counters) {
  SBR_REQUIRE(handle, "sbr_producer_stop: null handle");
  Producer* p = (Producer*)handle;
  if (p->running) {
    {
      std::lock_guard<std::mutex> lk(p->mu);
      p->stop = true;
      p->cv.notify_all();
    }
    p->th.join();
    p->running = false;
    if (getenv("SBR_PRODUCER_TIMING") && atoi(getenv("SBR_PRODUCER_TIMING")) == 1 && p->produced > 0) {
      const double n = (double)p->produced, ms = 1e3 / n;
      fprintf(stderr, "[sbr producer] %ld batches: wait for a free slot %.3f | draws %.3f | membership rounds %.3f | pack %.3f | modality "
                      "draws %.3f | upload %.3f ms per batch\n", p->produced, p->t_wait * ms, p->t_draw * ms, p->t_member * ms,
              p->t_pack * ms, p->t_mod * ms, p->t_upload * ms);
    }
  }
  if (mt_key) memcpy(mt_key, p->mt.key, sizeof(p->mt.key));
  if (mt_pos) *mt_pos = p->mt.p;
  if (pcg)
    for (int w = 0; w < 2; ++w) {
      const Pcg64& g = p->ent[w].rng;
      pcg[6 * w] = (unsigned long long)(g.state >> 64); pcg[6 * w + 1] = (unsigned long long)g.state;
      pcg[6 * w + 2] = (unsigned long long)(g.inc >> 64); pcg[6 * w + 3] = (unsigned long long)g.inc;
      pcg[6 * w + 4] = (unsigned long long)g.has_uint32; pcg[6 * w + 5] = g.uinteger;
    }
  if (out_counters) { out_counters[0] = p->produced; out_counters[1] = p->n_prepared; }
  if (!p->error.empty()) {
    sbr_set_error("%s", p->error.c_str());
    return SBR_ERR_HIP;
  }
  return SBR_OK;
}

int sbr_producer_destroy(void* handle) {
  if (!handle) return SBR_OK;
  Producer* p = (Producer*)handle;
  sbr_producer_stop(handle, nullptr, nullptr, nullptr, nullptr);
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  for (auto e : p->ready) (void)hipEventDestroy(e);
  for (auto e : p->consumed) (void)hipEventDestroy(e);
  for (auto h : p->slot_host) (void)hipHostFree(h);
  if (p->q_users) (void)hipHostFree(p->q_users);
  if (p->q_items) (void)hipHostFree(p->q_items);
  if (p->q_flags) (void)hipHostFree(p->q_flags);
  if (p->dq_users) (void)hipFree(p->dq_users);
  if (p->dq_items) (void)hipFree(p->dq_items);
  if (p->dq_flags) (void)hipFree(p->dq_flags);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return SBR_OK;
}

// ---- the two generator replicas on their own (host only): what tests pin against numpy -----------------------------------------
// positions int8 [n_slots, k] drawn like sampling.sample_modalities; pcg: (state hi, state lo, inc hi, inc lo, has_uint32,
// uinteger), updated in place; counts[n_mod] out (may be NULL).
int sbr_host_pcg64_modalities(unsigned long long* pcg, long n_slots, int n_mod, int k, int central, signed char* out, long* counts) {
  SBR_REQUIRE(pcg && out && n_mod >= 1 && n_mod <= 127 && (k == 1 || k == 2) && k <= n_mod && central < n_mod,
              "sbr_host_pcg64_modalities: bad arguments");
  Entity e;
  e.enabled = 1; e.n_mod = n_mod; e.k = k; e.central = central;
  e.rng.state = (((unsigned __int128)pcg[0]) << 64) | pcg[1];
  e.rng.inc = (((unsigned __int128)pcg[2]) << 64) | pcg[3];
  e.rng.has_uint32 = (int)pcg[4];
  e.rng.uinteger = (uint32_t)pcg[5];
  std::vector<long> tmp(n_mod > 8 ? n_mod : 8);
  draw_modalities(e, n_slots, (int8_t*)out, counts ? counts : tmp.data());
  pcg[0] = (unsigned long long)(e.rng.state >> 64); pcg[1] = (unsigned long long)e.rng.state;
  pcg[4] = (unsigned long long)e.rng.has_uint32; pcg[5] = e.rng.uinteger;
  return SBR_OK;
}

// engine._EntityRun.plan's padding of per-modality counts (counts updated in place)
int sbr_host_pad_counts(long* counts, int n_mod, long R) {
  SBR_REQUIRE(counts && n_mod >= 1, "sbr_host_pad_counts: bad arguments");
  pad_counts(counts, n_mod, R);
  return SBR_OK;
}

}  // extern "C"
