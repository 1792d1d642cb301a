// C-ABI of the engine (include/mpvss_hip.h): context, workspace, kernel orchestration and the
// host-side Fiat-Shamir transcript.  No CPU fallback exists: without a HIP device every compute
// entry point fails with MPVSS_E_DEVICE.
#include "../../include/mpvss_hip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host_modq.h"
#include "host_scalar.h"
#include "modp_kernels.h"
#include "sha256.h"
#include "verdict_kernels.h"

namespace {

constexpr size_t EB = MPVSS_MODP_BYTES;          // element / scalar bytes
constexpr size_t TABW = MODP_TABLE_WORDS;        // words per 16-entry window table
// shares per pass (bounds the table workspace: 2 x 1.2 GiB).  MPVSS_MAX_CHUNK lowers it so that the tests can
// drive the multi-chunk path with small inputs.
static size_t max_chunk_init() {
  const char* e = getenv("MPVSS_MAX_CHUNK");
  if (e) {
    const long v = atol(e);
    if (v >= 16) return (size_t)v;
  }
  return (size_t)1 << 18;
}
static const size_t MAX_CHUNK = max_chunk_init();

// ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); launches of streams that share a queue run
// one after the other.  The block pipeline wants 8 (measured: 16 or 32 make every kernel 2-3x slower,
// profiles/r02_ec_streams_ab.txt).  The runtime reads the variable when it initialises, so it has to be in the
// environment before the process's first HIP call: that is the HOST APPLICATION's business (its launcher, or an explicit
// mpvss_process_init() at the top of main) -- the library does not touch the environment on its own.

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct EcWork {   // device workspace of the elliptic-curve entry points (one per context and one per block slot)
  DevBuf a, b, c, d, e, pos, cm, cmenc, x, o1, o2, ok, gen, chal, pts, fdst;
  DevBuf tab1, tab2, tab3;       // window tables of per-share bases: [n][8][cached words]
  DevBuf p1, p2;                 // results of the double-scalar multiplications in internal coordinates
  DevBuf flags;                  // ints: [0] forward-difference gate, [1] first bad response, [2] first bad challenge
  DevBuf pg;                     // r G of a lone block, computed beside the X path (a1 = r G + c X is then one table multiplication and an addition)
  DevBuf hand, wtab;             // hand-over space of the quad-lane stepping pipeline, window tables of its seed kernel (a box that has the chip to itself)
  std::vector<DevBuf*> all() {
    return {&a, &b, &c, &d, &e, &pos, &cm, &cmenc, &x, &o1, &o2, &ok, &gen, &chal, &pts, &fdst, &tab1, &tab2, &tab3, &p1, &p2,
            &flags, &hand, &wtab, &pg};
  }
};

// Registered public keys: per-key tables for y^r in HBM plus a device copy of the keys themselves.
struct mpvss_keyset {
  DevBuf table, keys;
  size_t n = 0;
};

struct mpvss_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* consts = nullptr;
  void* consts_q = nullptr;     // the same rows for q' = (q-1)/2 (scalar-ring kernels), uploaded on first use
  // staging of the stand-alone scalar-ring calls (capi_scalar.inc): their own stream, a ring of small pinned + device buffers
  struct ScalarEntry { void* pin = nullptr; size_t cap = 0; DevBuf dev; hipEvent_t done = nullptr; bool in_use = false; };
  static constexpr unsigned SCALAR_RING = 16;
  ScalarEntry scalar_ring[SCALAR_RING];
  unsigned scalar_seq = 0;
  hipStream_t scalar_stream = nullptr;
  void* pair_tables = nullptr;   // constant digit matrices of the pair-layout kernels (bn_pair.h), one device copy per context
  std::string err;
  mutable std::mutex err_mu;     // guards `err` alone: mpvss_last_error may run beside calls of other threads
  std::mutex pipe_mu;                  // one library box pipeline (mpvss_*_verify_many*) per context at a time; taken BEFORE mu
  std::mutex mu;
  std::condition_variable slot_cv;     // signalled (under mu) whenever a block slot is released: callers waiting for room in the ring
  // mpvss_modp_deal / mpvss_ec_deal: the one-call dealer's inputs and intermediate secrets (keys, P(i), witnesses, responses,
  // positions, the staged polynomial) in buffers of their own -- the call releases `mu` while it waits for and hashes its
  // blocks, and the shared workspace of the synchronous entry points is anybody's then.  Every deal in flight borrows one set
  // from a pool (grow-only, most recently returned first), so that several host threads can deal on one context at once.
  struct DealBufs { DevBuf keys, p, w, r, pos, coef, c; void* pin = nullptr; size_t pin_cap = 0; bool in_use = false; };
  // Pinned buffers of the one-box entry point for callers that hand over HOST memory: the caller's thread copies its inputs in here
  // with the context lock RELEASED (48 MB at the headline shape: 2-5 ms that other callers' enqueues and absorbs no longer wait for),
  // the block's H2D copies read it, the block's absorb gives it back.  One per call in flight, grow-only.
  struct HostStage { void* pin = nullptr; size_t cap = 0; bool in_use = false; };
  std::vector<HostStage*> stage_pool;
  std::vector<DealBufs*> deal_pool;
  DealBufs* deal_acquire() {
    for (size_t i = deal_pool.size(); i-- > 0;)
      if (!deal_pool[i]->in_use) { deal_pool[i]->in_use = true; return deal_pool[i]; }
    DealBufs* d = new (std::nothrow) DealBufs();
    if (!d) return nullptr;
    d->in_use = true;
    deal_pool.push_back(d);
    return d;
  }
  // Device workspace of one call in flight (grow-only buffers, the stream pair and the events that order them).
  // work0 serves the ordinary entry points; every verify-block slot has its own, so that several boxes can be in
  // flight on the GPU at once (the serial phases of one box overlap the wide phases of the next).
  struct Work {
    DevBuf in_a, in_b, in_c, in_d, in_e, pos, cm, xbe, out1, out2, tab1, tab2, tabg, cbuf;
    DevBuf fd_flag, fd_state, fd_xm, fd_xinv, fd_pre, fd_tot, fd_totinv, fd_root, fd_hand_t, fd_hand_s, fd_gather;   // forward differences
    DevBuf tab3, gr_m;   // X tables of a1; gr_m: g^r_i in Montgomery form
    DevBuf verd;         // per-share verdict bytes of verify_share batches (K7)
    DevBuf csched;       // sliding-window schedule of the box's challenge
    DevBuf row_m;        // Montgomery-form results of the row-layout kernel of small batches (dual_exp_any)
    const uint8_t* cm_bytes_dev = nullptr;   // device copy of the commitments' bytes of the current call
    struct RootJob {                         // pinned: the one real inversion of the seed phase, done by the host
      uint8_t in_be[256], out_be[256];
      int ok;
      int one;                               // pinned constant 1 (initial value of the forward-difference flag)
      uint8_t challenge[256];                // pinned copy of the call's challenge (the caller's buffer is not kept)
      uint16_t csched[160];                  // pinned: sliding-window schedule of that challenge (sliding_schedule)
    };
    RootJob* root = nullptr;
    bool fd_used = false;                    // eval_x took the forward-difference path in the call being enqueued
    hipStream_t sa = nullptr, sb = nullptr;  // stream pair: sb runs a2 beside the serial phases of the X path on sa
    hipEvent_t ev_fork = nullptr, ev_gr = nullptr;
    hipEvent_t ev_a2 = nullptr;
    bool ready = false;
    std::vector<DevBuf*> all() {
      return {&in_a, &in_b, &in_c, &in_d, &in_e, &pos, &cm, &xbe, &out1, &out2, &tab1, &tab2, &tabg, &cbuf, &fd_flag,
              &fd_state, &fd_xm, &fd_xinv, &fd_pre, &fd_tot, &fd_totinv, &fd_root, &fd_hand_t, &fd_hand_s, &fd_gather, &tab3, &gr_m, &verd,
              &csched, &row_m};
    }
  };
  Work work0;
  Work* w = &work0;                               // workspace of the call being enqueued
  hipStream_t stream_b = nullptr;                 // second stream of the current workspace
  int prio_high = 0, prio_low = 0;
  DevBuf comb[2];            // fixed-base comb tables of g = 4 (index 0) and G = 2 (index 1), built on first use
  bool comb_ready[2] = {false, false};
  DevBuf qbounds;            // q and q - 1 as big-endian bytes (bounds of the well-formedness verdicts), built on first use
  DevBuf comb16[2];          // their wide versions (16-bit windows, 2.5 GB each), built on first large batch
  bool comb16_ready[2] = {false, false};
  // pinned host staging
  void* pin = nullptr;
  size_t pin_cap = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double kernel_ms[4] = {-1, -1, -1, -1};   // 0 X path, 1 comb_dual_exp (a1), 2 table builds, 3 dual_exp (a2 / exp)
  int kernel_launches[4] = {0, 0, 0, 0};
  struct Span { int id; hipEvent_t a, b; };
  struct SpanSet {
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
  };
  SpanSet main_spans;
  SpanSet* sp = &main_spans;   // where TIMED_LAUNCH records
  // Up to NSLOT verify blocks may be in flight (compute of block k+1.. is enqueued before block k is absorbed):
  // each has its own pinned staging, timing events and completion event.
  struct BlockSlot {
    void* pin = nullptr;
    size_t cap = 0;
    size_t n = 0;
    bool busy = false;
    bool absorbing = false;        // a host thread is waiting for / hashing this block (context lock released)
    bool claimed = false;          // taken by mpvss_block_claim, its absorb call has not started yet
    bool owned = false;            // enqueued by a caller that absorbs it itself by position (the one-call entry points, the library's
                                   // box pipeline): the FIFO entry points (absorb without a ticket, mpvss_block_claim) pass it over
    unsigned pos = 0;              // the block's number in this context's enqueue order (value of `head` when it was committed)
    std::atomic<unsigned long long>* gpu_done_ctr = nullptr;   // the owning box pipeline's count of blocks whose GPU work is done
    unsigned ring_pos = 0;         // ring position the block occupies (pos % NSLOT)
    bool dealer = false;           // kind 2: a dealer's block (flag [1] = polynomial values, [2] = witnesses; t may be 0)
    bool fd_used = false;          // the block's X path was the forward-difference one: its final flags are in the staging
    unsigned fd_chunks = 0;        // chunks of the block that took the forward-difference path (one flag each)
    bool check_positions = false;
    unsigned nbox = 1;             // kind 0: boxes of one shape in this block (a group enqueued by mpvss_modp_verify_many), n / nbox shares each
    int kind = 0;                  // 0: block of verify_distribution_shares, 1: batch of verify_share proofs (W_B),
                                   // 2: block of a curve group's verify_distribution_shares, 3: a curve group's verify_share batch
    int group = 0;                 // kind 2: MPVSS_GROUP_*
    size_t t = 0;                  // kind 2: number of commitments (their decode flags sit in the staging)
    EcWork ecw;
    double enqueue_ms = 0;         // host time spent enqueueing this block's GPU work
    hipEvent_t done = nullptr;
    SpanSet spans;
    double kernel_ms[4] = {0, 0, 0, 0};
    Work work;
  };
  // Blocks in flight form a ring of NSLOT positions in enqueue order (head: number of blocks enqueued so far = the next
  // block's number; tail: no block below it is still waiting for a FIFO consumer).  A position borrows a slot from a free
  // stack -- the most recently released one first, so a caller that keeps k blocks in flight touches k slots' workspaces,
  // not NSLOT of them.  Who absorbs a block: whoever enqueued it, by its number (`owned`: the one-call entry points and the
  // library's box pipeline -- any number of host threads may do that on one context at the same time), or, for blocks of
  // the explicit block API, the FIFO entry points, oldest first.
  static constexpr unsigned NSLOT = MPVSS_BLOCK_SLOTS;
  BlockSlot slot[NSLOT];
  BlockSlot none_slot, full_slot;          // what ring_slot() / head_slot() answer when there is no block / no room
  int ring[NSLOT];
  int free_stack[NSLOT];
  unsigned free_top = NSLOT;
  unsigned head = 0, tail = 0;
  mpvss_ctx() {
    for (unsigned i = 0; i < NSLOT; ++i) { ring[i] = -1; free_stack[i] = (int)(NSLOT - 1 - i); }
    full_slot.busy = true;
  }
  // a run of many boxes is under way (mpvss_*_verify_many with more than two boxes, or the caller said so through
  // MPVSS_PIPELINED=1): a block then never takes the configuration meant for a call that has the GPU to itself, not even
  // the first ones of the run
  // The key caches' table memory outlives a key set: allocating 19 GB costs 0.5-1.1 s in a busy process (hipMalloc maps the pages;
  // measured, MPVSS_TRACE_KEYSET) against 61 ms for building the tables of 65536 keys, so the buffer of a set the caches drop is
  // kept for the next set they build (one buffer; freed when both caches are switched off, when the context goes, or when any
  // workspace allocation of the context fails for lack of memory).
  void* spare_table = nullptr;
  size_t spare_table_cap = 0;
  void drop_spare_table() {
    if (spare_table) (void)hipFree(spare_table);
    spare_table = nullptr;
    spare_table_cap = 0;
  }
  int key_cache_min_boxes = 0;   // mpvss_ctx_set_key_cache: verify_many registers key arrays that this many large boxes of a call share (0: off)
  // Key tables ACROSS calls (mpvss_ctx_set_key_cache_lru): the one-box entry point looks the SHA-256 of a host key array up here; an
  // array seen `kc_min_sightings` times gets its tables built once and every later box against it takes the registered-key path.
  // Least recently used sets without blocks in flight make room; `users` counts the blocks that still read an entry's tables.
  struct KeyCacheEntry {
    uint8_t digest[32];
    size_t n = 0;
    mpvss_keyset* ks = nullptr;
    unsigned users = 0, sightings = 0;
    unsigned long long last_use = 0;
  };
  std::vector<KeyCacheEntry> kc;
  std::atomic<int> kc_max_sets{0};
  int kc_min_sightings = 2;
  unsigned long long kc_clock = 0;
  bool pipelined_env = false;    // MPVSS_PIPELINED=1
  int pipelines_running = 0;     // library box pipelines under way on this context (several host threads may each run one)
  bool pipelined_hint() const { return pipelined_env || pipelines_running > 0; }
  bool busy_with_others() const { return pipelined_hint() || NSLOT - free_top >= 2; }
  // many blocks in flight: work counts, not the latency of one block's chains (the row layout's shorter, dearer operations lose there:
  // 16 one-box callers of BASELINE config C2 0.50 M share verifications/s with the pair layout and forward differences, 0.45 M without)
  bool crowded() const { return pipelined_hint() || NSLOT - free_top >= 8; }
  BlockSlot& head_slot() {
    if (ring[head % NSLOT] >= 0 || free_top == 0) return full_slot;
    return slot[free_stack[free_top - 1]];
  }
  void commit_head(BlockSlot& sl) {          // the block in `sl` (from head_slot()) is fully enqueued
    sl.ring_pos = head % NSLOT;
    sl.pos = head;
    sl.owned = sl.claimed = sl.absorbing = false;
    sl.gpu_done_ctr = nullptr;
    ring[sl.ring_pos] = (int)(&sl - slot);
    --free_top;
    sl.busy = true;
    ++head;
  }
  BlockSlot& ring_slot(unsigned position) {  // the block with this number, if it is still in the ring
    const int i = ring[position % NSLOT];
    return (i < 0 || slot[i].pos != position) ? none_slot : slot[i];
  }
  void release(BlockSlot& sl) {
    sl.busy = sl.owned = sl.claimed = false;
    ring[sl.ring_pos] = -1;
    free_stack[free_top++] = (int)(&sl - slot);
    slot_cv.notify_all();
  }
  // the last `count` blocks this thread enqueued (it still holds the context lock) are its own: it absorbs them by number
  unsigned own_last(unsigned count, std::atomic<unsigned long long>* gpu_done_ctr = nullptr) {
    for (unsigned p = head - count; p != head; ++p) {
      ring_slot(p).owned = true;
      ring_slot(p).gpu_done_ctr = gpu_done_ctr;
    }
    return head - count;
  }
  // the oldest block that waits for a FIFO consumer (none_slot when there is none)
  BlockSlot& fifo_front() {
    auto waiting = [&](unsigned p) { BlockSlot& s = ring_slot(p); return s.busy && !s.absorbing && !s.claimed && !s.owned; };
    while (tail != head && !waiting(tail)) ++tail;      // (what is passed over never waits for a FIFO consumer again)
    return tail == head ? none_slot : ring_slot(tail);
  }
  // blocks whose consumer is known and at work (owned, claimed or being absorbed): room in the ring WILL appear
  bool consumers_at_work() const {
    for (unsigned i = 0; i < NSLOT; ++i)
      if (slot[i].busy && (slot[i].owned || slot[i].claimed || slot[i].absorbing)) return true;
    return false;
  }
  // Curve groups: X paths of several boxes of one mpvss_ec_verify_many call computed by the same launches (capi_ec.inc)
  struct XBatch {
    DevBuf cmenc, cm, okcm, pos, pts, xenc, state, flags, hand, wtab;
    hipStream_t s = nullptr;
    hipEvent_t done = nullptr;
    void* pin = nullptr;
    size_t pin_cap = 0;
    std::vector<DevBuf*> all() { return {&cmenc, &cm, &okcm, &pos, &pts, &xenc, &state, &flags, &hand, &wtab}; }
  };
  std::vector<XBatch*> ec_xb;
  // blocks whose X went through the forward-difference path / of those, blocks that fell back to Horner's rule on
  // the device (positions not consecutive, an X that is 0 mod q, a pipeline stage that gave up)
  unsigned long long fd_blocks = 0, fd_fallbacks = 0;
  DevBuf ec_comb[2];             // fixed-base combs of the curve groups' generators (built on first use)
  bool ec_comb_ready[2] = {false, false};
  // host-side accounting of the block pipeline (mpvss_pipeline_stats_get): sums over absorbed blocks
  struct PipeStats {
    double enqueue_ms = 0, wait_ms = 0, hash_ms = 0;
    double kernel_ms[4] = {0, 0, 0, 0};
    unsigned long long kernel_launches[4] = {0, 0, 0, 0};
    unsigned long long blocks = 0;
  } pstats;
  EcWork ecwork;
};

namespace {

int fd_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int fail(mpvss_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  if (ctx) {
    char buf[512];
    if (e != hipSuccess)
      snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(buf, sizeof(buf), "%s", what);
    std::lock_guard<std::mutex> g(ctx->err_mu);
    ctx->err = buf;
  }
  return code;
}

#define HIPCHK(ctx, call)                                                        \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) return fail((ctx), MPVSS_E_DEVICE, #call, e_);         \
  } while (0)

#define LAUNCHCHK(ctx, call)                                                     \
  do {                                                                           \
    int e_ = (call);                                                             \
    if (e_ != 0) return fail((ctx), MPVSS_E_DEVICE, #call, (hipError_t)e_);      \
  } while (0)

#define RET_IF(x)             \
  do {                        \
    int rc_ = (x);            \
    if (rc_ != 0) return rc_; \
  } while (0)

int ensure(mpvss_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream_b) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_b));
    HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess && ctx->spare_table) {      // the key caches' spare table buffer gives way to a workspace
    (void)hipGetLastError();
    ctx->drop_spare_table();
    e = hipMalloc(&b.p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();     // the runtime's last-error slot is sticky: the launchers' `return hipGetLastError()` must not see this
    b.p = nullptr;
    return fail(ctx, MPVSS_E_NOMEM, "hipMalloc(workspace)", e);
  }
  b.cap = bytes;
  return 0;
}

// An OPTIONAL buffer (hand-over space of the curve groups' stage pipelines, the seeds' window tables): the caller has a
// slower path that does without it, so a failed allocation is not an error of the call -- nothing is recorded in the
// context and the runtime's sticky last error is drained, otherwise the next launcher would return hipErrorOutOfMemory
// for a launch that succeeded.  MPVSS_TEST_OPTIONAL_NOMEM=1 makes every such allocation fail THROUGH hipMalloc (a size no
// device has), so that the test sees the real error state (tests/test_gpu_ec_fd.py).
bool try_ensure(mpvss_ctx* ctx, DevBuf& b, size_t bytes) {
  static const int force_fail = fd_env("MPVSS_TEST_OPTIONAL_NOMEM", 0);
  if (force_fail) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)1 << 46);
    if (e == hipSuccess) (void)hipFree(p);
    (void)hipGetLastError();
    return false;
  }
  if (bytes <= b.cap) return true;
  if (b.p) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || (ctx->stream_b && hipStreamSynchronize(ctx->stream_b) != hipSuccess) ||
        hipFree(b.p) != hipSuccess) {
      (void)hipGetLastError();
      return false;                // the buffer stays as it is (too small for this call: the caller takes its other path)
    }
    b.p = nullptr;
    b.cap = 0;
  }
  if (hipMalloc(&b.p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    b.p = nullptr;
    return false;
  }
  b.cap = bytes;
  return true;
}

int ensure_pinned(mpvss_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pin_cap) return 0;
  if (ctx->pin) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipHostFree(ctx->pin));
    ctx->pin = nullptr;
    ctx->pin_cap = 0;
  }
  hipError_t e = hipHostMalloc(&ctx->pin, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(staging)", e);
  ctx->pin_cap = bytes;
  return 0;
}

// Bring an input array onto the device if the caller handed host memory.
int stage_in(mpvss_ctx* ctx, int space, const void* src, size_t bytes, DevBuf& buf, const void** dev) {
  if (space == MPVSS_DEVICE) {
    *dev = src;
    return 0;
  }
  RET_IF(ensure(ctx, buf, bytes));
  HIPCHK(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = buf.p;
  return 0;
}

// kernel timing spans (hipEvents on the engine's stream)
int span_begin(mpvss_ctx* ctx, int id) {
  mpvss_ctx::SpanSet& ss = *ctx->sp;
  for (int k = 0; k < 2; ++k) {
    if (ss.ev_used == ss.ev_pool.size()) {
      hipEvent_t e;
      HIPCHK(ctx, hipEventCreate(&e));
      ss.ev_pool.push_back(e);
    }
    ++ss.ev_used;
  }
  mpvss_ctx::Span s{id, ss.ev_pool[ss.ev_used - 2], ss.ev_pool[ss.ev_used - 1]};
  HIPCHK(ctx, hipEventRecord(s.a, ctx->stream));
  ss.spans.push_back(s);
  return 0;
}
int span_end(mpvss_ctx* ctx) {
  HIPCHK(ctx, hipEventRecord(ctx->sp->spans.back().b, ctx->stream));
  return 0;
}
void spans_reset(mpvss_ctx* ctx) {
  ctx->sp->spans.clear();
  ctx->sp->ev_used = 0;
  if (ctx->sp == &ctx->main_spans)
    for (double& m : ctx->kernel_ms) m = -1;
}
int spans_sum(mpvss_ctx* ctx, mpvss_ctx::SpanSet& ss, double out[4]) {
  for (int i = 0; i < 4; ++i) { out[i] = 0; ctx->kernel_launches[i] = 0; }
  for (auto& s : ss.spans) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, s.a, s.b));
    out[s.id] += ms;
    ++ctx->kernel_launches[s.id];
  }
  return 0;
}
int spans_collect(mpvss_ctx* ctx) {
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return spans_sum(ctx, ctx->main_spans, ctx->kernel_ms);
}

struct Timed {
  mpvss_ctx* ctx;
  int rc;
  Timed(mpvss_ctx* c, int id) : ctx(c) { rc = span_begin(c, id); }
  int end() { return span_end(ctx); }
};

#define TIMED_LAUNCH(ctx, id, call)     \
  do {                                  \
    RET_IF(span_begin((ctx), (id)));    \
    LAUNCHCHK((ctx), (call));           \
    RET_IF(span_end((ctx)));            \
  } while (0)

// minimal-length big-endian view of a 256-byte element (modp.rs:150-152: zero -> one 0x00 byte)
inline void frame_update(mpvss::Sha256& h, const uint8_t* e256) {
  size_t skip = 0;
  while (skip < EB - 1 && e256[skip] == 0) ++skip;
  const uint64_t len = EB - skip;
  uint8_t pre[8];
  for (int i = 0; i < 8; ++i) pre[i] = (uint8_t)(len >> (56 - 8 * i));
  h.update(pre, 8);                 // dleq.rs:58-61
  h.update(e256 + skip, (size_t)len);
}

// The transcript entries of shares [i0, i1): framed(X_i) | framed(Y_i) | framed(a1_i) | framed(a2_i) in share order (dleq.rs:87-99).
// The frames of a run of shares are laid out contiguously in a cache-resident buffer and hashed with ONE update: the compression
// function then runs over ~1000 blocks per call with its state in registers, instead of eight small updates per share (each with a
// partial-block copy and a one-block call) -- 5 % less time per box on the SHA-NI path, same bytes into the hash.
inline void frame_shares(mpvss::Sha256& h, const uint8_t* hX, const uint8_t* hY, const uint8_t* h1, const uint8_t* h2, size_t i0, size_t i1) {
  constexpr size_t RUN = 32;
  uint8_t buf[RUN * 4 * (EB + 8)];
  for (size_t i = i0; i < i1; i += RUN) {
    const size_t e = i + RUN < i1 ? i + RUN : i1;
    uint8_t* w = buf;
    for (size_t k = i; k < e; ++k)
      for (const uint8_t* a : {hX, hY, h1, h2}) {
        const uint8_t* el = a + k * EB;
        size_t skip = 0;
        while (skip < EB - 1 && el[skip] == 0) ++skip;      // minimal-length big-endian magnitude, zero -> one 0x00 byte (modp.rs:150-152)
        const uint64_t len = EB - skip;
        for (int b = 0; b < 8; ++b) w[b] = (uint8_t)(len >> (56 - 8 * b));     // dleq.rs:58-61
        memcpy(w + 8, el + skip, (size_t)len);
        w += 8 + len;
      }
    h.update(buf, (size_t)(w - buf));
  }
}

// the minimal-length bytes alone (no length prefix): SHA256(element_to_bytes(e)) of reconstruct, participant.rs:512
inline void frame_min_bytes_update(mpvss::Sha256& h, const uint8_t* e256) {
  size_t skip = 0;
  while (skip < EB - 1 && e256[skip] == 0) ++skip;
  h.update(e256 + skip, EB - skip);
}

// hash_to_scalar(digest) == c  (modp.rs:142-148; the 256-bit hash is already < (q-1)/2)
inline bool challenge_matches(const uint8_t digest[32], const uint8_t c256[256]) {
  uint8_t hh[32];
  mpvss::sha256(digest, 32, hh);
  for (size_t i = 0; i < EB - 32; ++i)
    if (c256[i] != 0) return false;
  return memcmp(hh, c256 + EB - 32, 32) == 0;
}

inline bool fits_256_bits(const uint8_t* c256) {
  for (size_t i = 0; i < EB - 32; ++i)
    if (c256[i] != 0) return false;
  return true;
}

int check_positions_host(mpvss_ctx* ctx, const int64_t* pos, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (pos[i] < 0) return fail(ctx, MPVSS_E_INVALID, "negative position (the reference panics: negative exponent)");
  return 0;
}

const uint8_t* g_bytes(int g) {
  static uint8_t b[3][EB];
  static bool init = false;
  if (!init) {
    memset(b, 0, sizeof(b));
    b[0][EB - 1] = 4;  // subgroup generator g = 2^2 (modp.rs:65-66)
    b[1][EB - 1] = 2;  // main generator G (modp.rs:64)
    b[2][EB - 1] = 1;
    init = true;
  }
  return b[g];
}

// the shared 16-entry table of one base (host bytes) into ctx->w->tabg
int shared_table(mpvss_ctx* ctx, const uint8_t* base_host, const uint32_t** tab) {
  RET_IF(ensure(ctx, ctx->w->tabg, TABW * 4 + EB));
  uint8_t* dbase = (uint8_t*)ctx->w->tabg.p + TABW * 4;
  HIPCHK(ctx, hipMemcpyAsync(dbase, base_host, EB, hipMemcpyHostToDevice, ctx->stream));
  TIMED_LAUNCH(ctx, 2, modp_launch_build_table(dbase, 1, (uint32_t*)ctx->w->tabg.p, ctx->consts, ctx->stream));
  *tab = (const uint32_t*)ctx->w->tabg.p;
  return 0;
}

// which of the two well-known generators a 256-byte base is: 0 -> g = 4, 1 -> G = 2, -1 -> neither
int generator_id(const uint8_t* base_host) {
  for (size_t i = 0; i < EB - 1; ++i)
    if (base_host[i] != 0) return -1;
  if (base_host[EB - 1] == 4) return 0;
  if (base_host[EB - 1] == 2) return 1;
  return -1;
}

// fixed-base comb table of generator `gid` (modp.rs:64-66), built once per context (about 25 ms).  Batches of
// MPVSS_COMB16_MIN (default 8192) numbers or more use the wide comb instead: 16-bit windows, 128 products per
// exponentiation instead of 512, 2.5 GB of HBM per generator, built once per context from the narrow one.
int comb_table(mpvss_ctx* ctx, int gid, const uint32_t** comb, size_t batch = 0) {
  if (!ctx->comb_ready[gid]) {
    RET_IF(ensure(ctx, ctx->comb[gid], (size_t)MODP_COMB_WORDS * 4 + EB));
    uint8_t* dbase = (uint8_t*)ctx->comb[gid].p + (size_t)MODP_COMB_WORDS * 4;
    HIPCHK(ctx, hipMemcpyAsync(dbase, g_bytes(gid), EB, hipMemcpyHostToDevice, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_comb_build(dbase, (uint32_t*)ctx->comb[gid].p, ctx->consts, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // once per context: the table is shared by every stream
    ctx->comb_ready[gid] = true;
  }
  *comb = (const uint32_t*)ctx->comb[gid].p;
  static const size_t wide_min = (size_t)fd_env("MPVSS_COMB16_MIN", 8192);
  if (wide_min > 0 && batch >= wide_min) {
    if (!ctx->comb16_ready[gid]) {
      RET_IF(ensure(ctx, ctx->comb16[gid], MODP_COMB16_WORDS * 4));
      LAUNCHCHK(ctx, modp_launch_comb16_build(*comb, (uint32_t*)ctx->comb16[gid].p, ctx->consts, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->comb16_ready[gid] = true;
    }
    *comb = (const uint32_t*)ctx->comb16[gid].p;
  }
  return 0;
}

// window width of a table returned by comb_table()
int comb_bits_of(const mpvss_ctx* ctx, const uint32_t* comb) {
  return (comb != nullptr && (comb == ctx->comb16[0].p || comb == ctx->comb16[1].p)) ? 16 : 4;
}

// per-number tables of `count` bases (device bytes) into buf
// Sliding-window (width 4, odd digits) schedule of a 256-bit exponent given as 256 big-endian bytes: out[0] = number of
// windows, then (bit position of the window's lowest bit, digit) from the top window down.  At most 64 windows.
// The challenge of a box is ONE exponent for all its shares (dleq.rs:75-81), so the kernels can follow a schedule made
// here instead of 64 fixed windows per share: about 51 products per base, and tables of the odd powers only.
void sliding_schedule(const uint8_t* c_be256, uint16_t* out) {
  auto bit = [&](int i) { return (c_be256[255 - (i >> 3)] >> (i & 7)) & 1; };
  uint16_t n = 0;
  int i = 255;
  while (i >= 0) {
    if (!bit(i)) { --i; continue; }
    int l = i - 3 < 0 ? 0 : i - 3;
    while (!bit(l)) ++l;
    unsigned d = 0;
    for (int j = i; j >= l; --j) d = (d << 1) | (unsigned)bit(j);
    out[1 + 2 * n] = (uint16_t)l;
    out[2 + 2 * n] = (uint16_t)d;
    ++n;
    i = l - 1;
  }
  out[0] = n;
}

int number_tables(mpvss_ctx* ctx, const uint8_t* bases_dev, size_t count, DevBuf& buf, const uint32_t** tab) {
  RET_IF(ensure(ctx, buf, count * TABW * 4));
  TIMED_LAUNCH(ctx, 2, modp_launch_build_table(bases_dev, (int)count, (uint32_t*)buf.p, ctx->consts, ctx->stream));
  *tab = (const uint32_t*)buf.p;
  return 0;
}

int small_vec_to_host(mpvss_ctx* ctx, int space, const uint8_t* src, size_t bytes, std::vector<uint8_t>& out) {
  out.resize(bytes);
  if (space == MPVSS_DEVICE) {
    HIPCHK(ctx, hipMemcpyAsync(out.data(), src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    memcpy(out.data(), src, bytes);
  }
  return 0;
}

int copy_out(mpvss_ctx* ctx, int space, void* dst, const void* dev_src, size_t bytes) {
  if (dst == nullptr || dst == dev_src) return 0;
  HIPCHK(ctx, hipMemcpyAsync(dst, dev_src, bytes,
                             space == MPVSS_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             ctx->stream));
  return 0;
}

// out = B1^e1 * B2^e2 over 16-entry window tables (k_modp_dual_exp's arguments); with e1b / outb a second exponent set over the same
// tables.  A SMALL batch -- the sizes of the reference's own tests and examples -- is the latency of one number's chain of 2 555+
// operations: it takes the row layout (16 lanes per number, modp_row_kernels.hip: 3.5 instead of 5.7 us per operation of a wave that has
// its SIMD to itself; ModpGroup::exp of 16 numbers 14.5 -> 7.4 ms, profiles/r06_small_box_latency.txt) up to one wave per SIMD, larger
// ones the quad layout's throughput.
constexpr size_t ROW_MAX_NUMBERS = 4096;
int dual_exp_any(mpvss_ctx* ctx, const uint32_t* t1, size_t s1, const uint32_t* t2, size_t s2, const uint8_t* e1, const uint8_t* e2,
                 size_t e2_stride, int e2_windows, size_t cnt, uint8_t* out, const uint8_t* e1b = nullptr, const uint8_t* e2b = nullptr,
                 uint8_t* outb = nullptr) {
  const size_t sets = e1b ? 2 : 1;
  if (cnt <= ROW_MAX_NUMBERS) {          // (two sets of up to 4096: two waves per SIMD, still shorter than the quad layout's chain twice)
    RET_IF(ensure(ctx, ctx->w->row_m, sets * cnt * MODP_L * 4));
    uint32_t* rm = (uint32_t*)ctx->w->row_m.p;
    TIMED_LAUNCH(ctx, 3, modp_launch_dual_exp_row(t1, s1, t2, s2, e1, e2, e2_stride, e2_windows, e1b, e2b, (int)cnt, rm, ctx->consts, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_from_mont(rm, (int)cnt, out, nullptr, ctx->consts, ctx->stream));
    if (e1b) LAUNCHCHK(ctx, modp_launch_from_mont(rm + cnt * MODP_L, (int)cnt, outb, nullptr, ctx->consts, ctx->stream));
    return 0;
  }
  TIMED_LAUNCH(ctx, 3, modp_launch_dual_exp(t1, s1, t2, s2, e1, e2, e2_stride, e2_windows, (int)cnt, out, ctx->consts, ctx->stream));
  if (e1b) TIMED_LAUNCH(ctx, 3, modp_launch_dual_exp(t1, s1, t2, s2, e1b, e2b, e2_stride, e2_windows, (int)cnt, outb, ctx->consts, ctx->stream));
  return 0;
}

// out = B^e (count numbers); bases/exps device pointers; uses tab1
int exp_dev(mpvss_ctx* ctx, const uint8_t* bases_dev, const uint8_t* exps_dev, size_t count, uint8_t* out_dev) {
  const uint32_t* t1;
  RET_IF(number_tables(ctx, bases_dev, count, ctx->w->tab1, &t1));
  return dual_exp_any(ctx, t1, TABW, t1, TABW, exps_dev, exps_dev, EB, 0, count, out_dev);
}

}  // namespace

// -------------------------------------------------------------------------------------------

extern "C" int mpvss_process_init(void) {
  if (getenv("GPU_MAX_HW_QUEUES")) return 0;
  return setenv("GPU_MAX_HW_QUEUES", "8", 0) == 0 ? 1 : MPVSS_E_INVALID;
}

extern "C" int mpvss_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {
// streams and ordering events of one workspace (created on first use)
int work_init(mpvss_ctx* ctx, mpvss_ctx::Work& w, hipStream_t main_stream) {
  if (w.ready) return 0;
  if (main_stream) {
    w.sa = main_stream;
    HIPCHK(ctx, hipStreamCreateWithFlags(&w.sb, hipStreamNonBlocking));
  } else {
    // a block slot's stream carries the latency-bound chain of its box: highest priority, so that it never queues
    // behind (or shares a hardware queue with) the wide launches, which go to the slot's low-priority second stream
    // (the X path on CUs of its own -- CU-masked streams -- was measured and is 2.2x slower: profiles/r04_cu_partition_ab.txt)
    HIPCHK(ctx, hipStreamCreateWithPriority(&w.sa, hipStreamNonBlocking, ctx->prio_high));
    HIPCHK(ctx, hipStreamCreateWithPriority(&w.sb, hipStreamNonBlocking, ctx->prio_low));
  }
  for (hipEvent_t* e : {&w.ev_fork, &w.ev_gr})
    HIPCHK(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_a2, hipEventDisableTiming));
  HIPCHK(ctx, hipHostMalloc((void**)&w.root, 2 * sizeof(*w.root), hipHostMallocDefault));   // one mailbox per seeding level
  w.root[0].one = 1;
  w.ready = true;
  return 0;
}
void work_destroy(mpvss_ctx::Work& w, bool owns_sa) {
  if (w.sa) (void)hipStreamSynchronize(w.sa);
  if (w.sb) (void)hipStreamSynchronize(w.sb);
  for (DevBuf* b : w.all())
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t e : {w.ev_fork, w.ev_gr})
    if (e) (void)hipEventDestroy(e);
  if (w.ev_a2) (void)hipEventDestroy(w.ev_a2);
  if (w.root) (void)hipHostFree(w.root);
  if (w.sb) (void)hipStreamDestroy(w.sb);
  if (w.sa && owns_sa) (void)hipStreamDestroy(w.sa);
}
}  // namespace

extern "C" int mpvss_ctx_create(int device_id, mpvss_ctx** out) {
  if (!out) return MPVSS_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return MPVSS_E_DEVICE;
  mpvss_ctx* ctx = new mpvss_ctx();
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
    delete ctx;
    return MPVSS_E_DEVICE;
  }
  ctx->own_stream = true;
  if (work_init(ctx, ctx->work0, ctx->stream) != 0 ||
      hipDeviceGetStreamPriorityRange(&ctx->prio_low, &ctx->prio_high) != hipSuccess ||
      modp_consts_upload(&ctx->consts) != 0 || modp_pair_tables_upload(&ctx->pair_tables) != 0) {
    delete ctx;
    return MPVSS_E_DEVICE;
  }
  ctx->stream_b = ctx->work0.sb;
  ctx->pipelined_env = fd_env("MPVSS_PIPELINED", 0) != 0;
  *out = ctx;
  return MPVSS_OK;
}

extern "C" void mpvss_ctx_destroy(mpvss_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& sl : ctx->slot) work_destroy(sl.work, true);
  work_destroy(ctx->work0, ctx->own_stream);
  for (DevBuf* b : {&ctx->comb[0], &ctx->comb[1], &ctx->comb16[0], &ctx->comb16[1], &ctx->qbounds})
    if (b->p) (void)hipFree(b->p);
  for (DevBuf* b : ctx->ecwork.all())
    if (b->p) (void)hipFree(b->p);
  for (auto& sl : ctx->slot)
    for (DevBuf* b : sl.ecw.all())
      if (b->p) (void)hipFree(b->p);
  for (DevBuf& b : ctx->ec_comb)
    if (b.p) (void)hipFree(b.p);
  for (mpvss_ctx::XBatch* xb : ctx->ec_xb) {
    if (xb->s) { (void)hipStreamSynchronize(xb->s); (void)hipStreamDestroy(xb->s); }
    if (xb->done) (void)hipEventDestroy(xb->done);
    if (xb->pin) (void)hipHostFree(xb->pin);
    for (DevBuf* b : xb->all())
      if (b->p) (void)hipFree(b->p);
    delete xb;
  }
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->consts) (void)hipFree(ctx->consts);
  if (ctx->consts_q) (void)hipFree(ctx->consts_q);
  for (auto& e : ctx->scalar_ring) {
    if (e.pin) (void)hipHostFree(e.pin);
    if (e.dev.p) (void)hipFree(e.dev.p);
    if (e.done) (void)hipEventDestroy(e.done);
  }
  if (ctx->scalar_stream) (void)hipStreamDestroy(ctx->scalar_stream);
  if (ctx->pair_tables) (void)hipFree(ctx->pair_tables);
  for (mpvss_ctx::HostStage* h : ctx->stage_pool) {
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
  }
  for (auto& x : ctx->kc)
    if (x.ks) {
      if (x.ks->table.p) (void)hipFree(x.ks->table.p);
      if (x.ks->keys.p) (void)hipFree(x.ks->keys.p);
      delete x.ks;
    }
  for (mpvss_ctx::DealBufs* d : ctx->deal_pool) {
    for (DevBuf* b : {&d->keys, &d->p, &d->w, &d->r, &d->pos, &d->coef, &d->c})
      if (b->p) (void)hipFree(b->p);
    if (d->pin) (void)hipHostFree(d->pin);
    delete d;
  }
  ctx->drop_spare_table();
  for (hipEvent_t e : ctx->main_spans.ev_pool) (void)hipEventDestroy(e);
  for (auto& sl : ctx->slot) {
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.done) (void)hipEventDestroy(sl.done);
    for (hipEvent_t e : sl.spans.ev_pool) (void)hipEventDestroy(e);
  }
  delete ctx;
}

extern "C" const char* mpvss_last_error(const mpvss_ctx* ctx) {
  if (!ctx) return "null context";
  // a copy per calling thread: the context's string may be reassigned by another thread's failing call
  static thread_local std::string copy;
  {
    std::lock_guard<std::mutex> g(ctx->err_mu);
    copy = ctx->err;
  }
  return copy.c_str();
}

extern "C" int mpvss_ctx_set_stream(mpvss_ctx* ctx, void* hip_stream) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;
  ctx->work0.sa = ctx->stream;
  ctx->own_stream = false;
  return MPVSS_OK;
}

extern "C" int mpvss_ctx_synchronize(mpvss_ctx* ctx) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->work0.sb));
  for (auto& sl : ctx->slot)
    if (sl.work.ready) {
      HIPCHK(ctx, hipStreamSynchronize(sl.work.sa));
      if (sl.work.sb) HIPCHK(ctx, hipStreamSynchronize(sl.work.sb));
    }
  return MPVSS_OK;
}

extern "C" double mpvss_last_kernel_ms(const mpvss_ctx* ctx, int kernel_id) {
  if (!ctx || kernel_id < 0 || kernel_id > 3) return -1;
  return ctx->kernel_ms[kernel_id];
}

extern "C" int mpvss_last_kernel_launches(const mpvss_ctx* ctx, int kernel_id) {
  if (!ctx || kernel_id < 0 || kernel_id > 3) return -1;
  return ctx->kernel_launches[kernel_id];
}

extern "C" int mpvss_modp_fd_stats(mpvss_ctx* ctx, unsigned long long* blocks, unsigned long long* fallbacks) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (blocks) *blocks = ctx->fd_blocks;
  if (fallbacks) *fallbacks = ctx->fd_fallbacks;
  return MPVSS_OK;
}

extern "C" void mpvss_sha256(const uint8_t* data, size_t len, uint8_t out32[32]) { mpvss::sha256(data, len, out32); }

extern "C" void mpvss_modp_hash_to_scalar(const uint8_t* data, size_t len, uint8_t out256[256]) {
  memset(out256, 0, EB);
  mpvss::sha256(data, len, out256 + EB - 32);  // 2^256 < (q-1)/2: the reduction is the identity
}

// ---- Group::mul ------------------------------------------------------------------------------
extern "C" int mpvss_modp_batch_mul(mpvss_ctx* ctx, int space, const uint8_t* a, const uint8_t* b, size_t n,
                                    uint8_t* out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!a || !b || !out || n > 0x7fffffff) return fail(ctx, MPVSS_E_INVALID, "batch_mul: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  const void *da, *db;
  RET_IF(stage_in(ctx, space, a, n * EB, ctx->w->in_a, &da));
  RET_IF(stage_in(ctx, space, b, n * EB, ctx->w->in_b, &db));
  uint8_t* dout = out;
  if (space == MPVSS_HOST) {
    RET_IF(ensure(ctx, ctx->w->out1, n * EB));
    dout = (uint8_t*)ctx->w->out1.p;
  }
  LAUNCHCHK(ctx, modp_launch_mul((const uint8_t*)da, (const uint8_t*)db, dout, (int)n, ctx->consts, ctx->stream));
  if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, out, dout, n * EB));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return MPVSS_OK;
}

// ---- Group::exp ------------------------------------------------------------------------------
extern "C" int mpvss_modp_batch_exp(mpvss_ctx* ctx, int space, const uint8_t* bases, const uint8_t* exps, size_t n,
                                    uint8_t* out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!bases || !exps || !out) return fail(ctx, MPVSS_E_INVALID, "batch_exp: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *db, *de;
    RET_IF(stage_in(ctx, space, bases + off * EB, cnt * EB, ctx->w->in_a, &db));
    RET_IF(stage_in(ctx, space, exps + off * EB, cnt * EB, ctx->w->in_b, &de));
    uint8_t* dout = out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
      dout = (uint8_t*)ctx->w->out1.p;
    }
    RET_IF(exp_dev(ctx, (const uint8_t*)db, (const uint8_t*)de, cnt, dout));
    if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, out + off * EB, dout, cnt * EB));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

namespace {
int pair_mask();     // which kernels take the pair layout (defined with the DLEQ launchers below)
}
extern "C" int mpvss_modp_batch_exp_fixed_base(mpvss_ctx* ctx, int space, const uint8_t* base_host,
                                               const uint8_t* exps, size_t n, uint8_t* out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!base_host || !exps || !out) return fail(ctx, MPVSS_E_INVALID, "batch_exp_fixed_base: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  const uint32_t* tg = nullptr;
  const uint32_t* cg = nullptr;
  if (generator_id(base_host) >= 0)
    RET_IF(comb_table(ctx, generator_id(base_host), &cg, n));
  else
    RET_IF(shared_table(ctx, base_host, &tg));
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void* de;
    RET_IF(stage_in(ctx, space, exps + off * EB, cnt * EB, ctx->w->in_b, &de));
    uint8_t* dout = out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
      dout = (uint8_t*)ctx->w->out1.p;
    }
    if (cg && (pair_mask() & 1) && cnt >= 64 && comb_bits_of(ctx, cg) == 16)      // batch keygen, commitments C_j = g^a_j: 128 products on the pair layout
      TIMED_LAUNCH(ctx, 1, modp_launch_comb16_twin_exp_pair(cg, (const uint8_t*)de, nullptr, (int)cnt, dout, nullptr, ctx->consts,
                                                            ctx->pair_tables, ctx->stream));
    else if (cg)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cg, cg, 0, (const uint8_t*)de, (const uint8_t*)de, EB, 0, (int)cnt,
                                                     dout, comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
    else
      RET_IF(dual_exp_any(ctx, tg, 0, tg, 0, (const uint8_t*)de, (const uint8_t*)de, EB, 0, cnt, dout));
    if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, out + off * EB, dout, cnt * EB));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- commitment multi-exp -----------------------------------------------------------------------
namespace {
// commitments (space) -> Montgomery limbs in ctx->w->cm
int stage_commitments(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t) {
  const void* dc;
  RET_IF(stage_in(ctx, space, commitments, t * EB, ctx->w->cbuf, &dc));
  ctx->w->cm_bytes_dev = (const uint8_t*)dc;
  RET_IF(ensure(ctx, ctx->w->cm, t * MODP_L * 4));
  LAUNCHCHK(ctx, modp_launch_to_mont((const uint8_t*)dc, (uint32_t*)ctx->w->cm.p, (int)t, ctx->consts, ctx->stream));
  return 0;
}

int stage_positions(mpvss_ctx* ctx, int space, const int64_t* positions, size_t n, const int64_t** dpos) {
  if (space == MPVSS_HOST) {
    RET_IF(check_positions_host(ctx, positions, n));
  } else {
    std::vector<int64_t> tmp(n);
    HIPCHK(ctx, hipMemcpyAsync(tmp.data(), positions, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    RET_IF(check_positions_host(ctx, tmp.data(), n));
  }
  const void* d;
  RET_IF(stage_in(ctx, space, positions, n * 8, ctx->w->pos, &d));
  *dpos = (const int64_t*)d;
  return 0;
}

// ---- X_i for a run of shares: Horner's rule, or forward differences when the positions are consecutive -------
// Forward differences need t seed values of X and their inverses per chain (the Horner kernel, then Montgomery's
// simultaneous inversion) and then cost ONE Montgomery product per share and coefficient (modp_kernels.hip).
// Whether the path applies (consecutive positions, no X that is 0 mod q) is decided ON THE DEVICE through a flag that
// gates the kernels, so the pipelined callers never synchronise.
// MPVSS_FD=0 disables the path (every X by the Horner kernel); MPVSS_FD_MIN_SHARES lowers the batch size it is used
// from, so that tests can drive it with small inputs.

// stream callback: out = in^-1 mod q (canonical big-endian), ok = 0 when in is 0 mod q.  No HIP calls in here.
// q as 256 big-endian bytes, assembled from its limbs
void modq_modulus_bytes(uint8_t* qbe) {
  for (size_t byte = 0; byte < EB; ++byte) {
    unsigned v = 0;
    for (int bit = 0; bit < 8; ++bit) {
      const size_t b = byte * 8 + bit;
      v |= ((MODP_N_LIMBS[b / MODP_W] >> (b % MODP_W)) & 1u) << bit;
    }
    qbe[EB - 1 - byte] = (uint8_t)v;
  }
}

void invert_root_on_host(void* p) {
  static const hostq::Field* field = [] {
    uint8_t qbe[EB];
    modq_modulus_bytes(qbe);
    return new hostq::Field(qbe);
  }();
  auto* job = (mpvss_ctx::Work::RootJob*)p;
  job->ok = hostq::batch_invert(*field, job->in_be, 1, job->out_be) ? 1 : 0;
}

// does the forward-difference path apply to this run of shares?  (host-side part of the decision)
bool fd_applies(size_t t, const int64_t* hpos, size_t cnt) {
  static const int fd_on = fd_env("MPVSS_FD", 1);
  static const size_t min_shares = (size_t)fd_env("MPVSS_FD_MIN_SHARES", 4096);
  constexpr size_t max_t = 1024;
  bool fd = fd_on && t >= 16 && t <= max_t && cnt >= 16 * t && cnt >= min_shares;
  if (fd && hpos) {                       // host positions: decide here; device positions are checked by a kernel
    for (size_t i = 0; i < cnt && fd; ++i) fd = hpos[i] == hpos[0] + (int64_t)i;
    fd = fd && hpos[0] >= 0 && hpos[0] < ((int64_t)1 << 61);
  }
  return fd;
}

// boxes > 1: a GROUP of same-shaped boxes in one set of launches -- box b has its commitments at w.cm + b * t rows, its `cnt`
// positions at dpos + b * box_positions (0: the boxes share one array) and its X at dX + b * cnt rows; one flag for the group.
int pair_mask();     // which kernels take the pair layout (below)
// x_alone: nothing of the caller runs beside this X path (the stand-alone mpvss_modp_commit_eval): its Horner seeds then take the
// row-layout kernel (modp_row_kernels.hip: 16 lanes per seed, 1024 waves instead of 256 -- the seed launch 52 -> 28 ms, the whole
// X path 68.5 -> 47.7 ms at (65536, 256), profiles/r06_commit_eval_alone.txt).  NOT inside a verifier's block: there the box is
// bound by the sum of its work, the wider layout costs 1.3x the issue slots per seed, and the call gets slower (148 against 122 ms,
// profiles/r06_lone_box_schedule_ab.txt).
// direct: Horner for every share whatever the positions are (a one-box call of up to ROW_MAX_NUMBERS shares, below).
int eval_x(mpvss_ctx* ctx, size_t t, const int64_t* dpos, const int64_t* hpos, size_t cnt, uint8_t* dX, size_t boxes = 1,
           size_t box_positions = 0, bool x_alone = false, bool direct = false) {
  const int B = (int)boxes;
  // The stepping kernels on the pair layout (MPVSS_PAIR bit 5) from MPVSS_FD_PAIR_MIN_T commitments: stages of 32 levels, 122
  // instead of 191 issue slots per product, but half as many waves with longer steps.  Measured (profiles/r03_fd_pair_ab.txt):
  // one GPU's slice of C5 (131072, 1024), where the stepping is a third of the work, 0.60 -> 0.69 M share verifications/s;
  // the headline shape (t = 256) within noise, if anything slower (1.08 against 1.11 M): its X path is latency, not issue.
  static const size_t pair_min_t = (size_t)fd_env("MPVSS_FD_PAIR_MIN_T", 512);
  const bool step_pair = (pair_mask() & 32) != 0 && t >= pair_min_t;
  auto launch_step = [&](const uint32_t* sf, const uint32_t* sb, size_t bx_st, int chains, int tt, int w0_, int clen, int cnt_, uint32_t* xm_,
                         size_t bx_xm_, uint32_t* hand_, size_t bx_hand_, int fault) -> int {
    // MPVSS_FD_TILE (with the pair-layout stepping): the stepping as wide launches over the anti-diagonals of the (stage, block of
    // MPVSS_FD_TILE_STEPS steps) grid instead of a pipeline of persistent stages -- no wave waits for another one.  0: never,
    // 1: when other blocks are in flight (a lone call keeps the pipeline: its latency is shorter), 2: always.
    static const int tile_mode = fd_env("MPVSS_FD_TILE", 0), tile_steps = fd_env("MPVSS_FD_TILE_STEPS", 64);
    if (step_pair && fault == 0 && (tile_mode >= 2 || (tile_mode == 1 && ctx->busy_with_others())))
      return modp_launch_fd_step_pair_tiled_boxes(const_cast<uint32_t*>(sf), const_cast<uint32_t*>(sb), bx_st, chains, tt, w0_, clen, cnt_, xm_,
                                                  bx_xm_, hand_, bx_hand_, B, (const int*)ctx->w->fd_flag.p, tile_steps, ctx->consts,
                                                  ctx->pair_tables, ctx->stream);
    if (step_pair)
      return modp_launch_fd_step_pair_boxes(sf, sb, bx_st, chains, tt, w0_, clen, cnt_, xm_, bx_xm_, hand_, bx_hand_, B, (int*)ctx->w->fd_flag.p,
                                            fault, ctx->consts, ctx->pair_tables, ctx->stream);
    return modp_launch_fd_step_boxes(sf, sb, bx_st, chains, tt, w0_, clen, cnt_, xm_, bx_xm_, hand_, bx_hand_, B, (int*)ctx->w->fd_flag.p, fault,
                                     ctx->consts, ctx->stream);
  };
  const bool fd = !direct && fd_applies(t, hpos, cnt);
  if (!fd && B == 1 && cnt <= ROW_MAX_NUMBERS) {
    // a small box is the latency of one share's Horner chain ((t - 1) x about 26 operations): the row layout's kernel, 3.5 instead of
    // 5.7 us per operation (Montgomery limbs out, converted by k_modp_from_mont)
    RET_IF(ensure(ctx, ctx->w->fd_xm, cnt * MODP_L * 4));
    uint32_t* xm_small = (uint32_t*)ctx->w->fd_xm.p;
    TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval_row_boxes((const uint32_t*)ctx->w->cm.p, (int)t, dpos, 0, (int)cnt, 1, xm_small, cnt, nullptr, 0,
                                                           ctx->consts, ctx->stream, 0));
    LAUNCHCHK(ctx, modp_launch_from_mont(xm_small, (int)cnt, dX, nullptr, ctx->consts, ctx->stream));
    return 0;
  }
  if (!fd) {
    if (B > 1)
      TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval_boxes((const uint32_t*)ctx->w->cm.p, (int)t, dpos, box_positions, (int)cnt, B,
                                                         nullptr, dX, cnt, nullptr, 0, ctx->consts, ctx->stream));
    else
    TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval((const uint32_t*)ctx->w->cm.p, (int)t, dpos, (int)cnt, nullptr, dX,
                                                 ctx->consts, ctx->stream));
    return 0;
  }
  // number of chains S.  Chain c owns the positions c, c+S, c+2S, .. (index j inside the chain); its seeds are the t
  // indices w0 .. w0+t-1 in the MIDDLE of the chain, from which one pipeline steps forward and one backward: the
  // seeds of all chains together are the S*t consecutive positions from S*w0 (Horner, and outputs at the same time).
  // The seed launch is latency-bound (2048 seeds are an eighth of a wave per SIMD), so fewer seeds would not finish
  // sooner; more chains shorten the stepping but cost Horner work.
  // measured optima on MI355X: chains of about 8192 members (n=65536: 8 chains for t = 16..256), fewer when the
  // seeds are dear (t = 512: 4), never longer than 16384 members (n=131072, t=1024: 8)
  int S = (int)std::max<size_t>(std::max<size_t>(std::min<size_t>(2048 / t, cnt / 8192), cnt / 16384), 4);
  // a call that has the GPU to itself: more, shorter chains -- the extra seeds are one wide launch on an idle chip, the
  // stepping (the serial part) shrinks in proportion
  const int S_pipelined = S;
  constexpr int lone_chains = 16;     // measured: 8 -> 109 ms per box, 16 -> 96.5, 32 -> 104, 64 -> 115
  if (lone_chains > S && !ctx->busy_with_others() && t <= 256) S = lone_chains;
  const int s_max = (int)(cnt / (4 * t));      // cnt >= 16 t, so at least 4
  if (S > s_max) S = s_max;
  if (S < 1) S = 1;
  const int chain_len = (int)((cnt + S - 1) / S);
  const int w0 = (chain_len - (int)t) / 2;
  const size_t seed0 = (size_t)S * w0;    // index of the first seed position
  const int m0 = (int)(S * t);
  // Two-level seeding (MPVSS_FD_L1, default on): Horner's rule -- about 7 000 products per seed at (65536, 256) -- only
  // for the first t of the S*t seed positions; a stride-1 forward-difference chain built from those t values steps
  // through the remaining (S-1)*t seed positions at t products each.  Same values (uniqueness of the group element);
  // 180 products per share fewer at (65536, 256), for about 15 ms more latency of the box's chain.
  // tests only: one pipeline stage gives up and the flag falls -- 1 / 2: a stage of the stepping pipelines, 3: of the
  // stride-1 seeding chain, 4: of the table pipeline
  static const int inject_fault = fd_env("MPVSS_FD_TEST_FAULT", 0);
  // MPVSS_FD_L1: 0 never, 1 (default) when other blocks are in flight, 2 always.  The second level saves 180 products per
  // share and costs a serial chain (about 15 ms of a lone box's latency): a call that has the GPU to itself keeps the
  // wide Horner launch for all S*t seeds.
  static const int two_level_env = fd_env("MPVSS_FD_L1", 1);
  const bool two_level = S > 1 && (two_level_env >= 2 || (two_level_env == 1 && ctx->busy_with_others()));
  // product tree of the simultaneous inversion: level l turns ms[l] numbers into ms[l+1] group totals
  constexpr int G = 16;
  auto tree_sizes = [&](int m) {
    std::vector<int> ms{m};
    while (ms.back() > 1) ms.push_back((ms.back() + G - 1) / G);
    return ms;
  };

  mpvss_ctx::Work& w = *ctx->w;
  w.fd_used = true;
  RET_IF(ensure(ctx, w.fd_flag, 64));
  RET_IF(ensure(ctx, w.fd_root, 4 * EB));
  RET_IF(ensure(ctx, w.fd_xm, boxes * cnt * MODP_L * 4));
  // The configuration (chains, seeding levels) depends on whether other blocks are in flight, and a slot sees both
  // over its life: the buffers are sized for the largest of them at once -- growing one later means hipFree, which
  // waits for the device (58-87 ms inside the enqueue of a timed box, measured with MPVSS_TRACE_ENQUEUE).
  struct FdBytes { size_t xinv = 0, pre = 0, tot = 0, state = 0, hand_t = 0, hand_s = 0; } need_b;
  {
    const int s_cap = (int)(cnt / (4 * t)) < 1 ? 1 : (int)(cnt / (4 * t));
    int cands[3] = {S, std::min(S_pipelined, s_cap), (t <= 256 && lone_chains > 0) ? std::min(lone_chains, s_cap) : S};
    for (int c = 0; c < 3; ++c) {
      const int Sc = cands[c] < 1 ? 1 : cands[c];
      const int m0c = (int)(Sc * t), clc = (int)((cnt + Sc - 1) / Sc);
      const std::vector<int> msc = tree_sizes(m0c * B);
      size_t prec = 0, totc = 0;
      for (size_t l = 0; l + 1 < msc.size(); ++l) prec += (size_t)msc[l];
      for (size_t l = 1; l < msc.size(); ++l) totc += (size_t)msc[l];
      const size_t l1t = Sc > 1 ? modp_fd_table_hand_words(1, (int)t) * 4 : 0;
      const size_t l1s = Sc > 1 ? modp_fd_step_hand_words(1, (int)t, m0c) * 4 : 0;
      need_b.xinv = std::max(need_b.xinv, boxes * m0c * MODP_L * 4);
      need_b.pre = std::max(need_b.pre, prec * MODP_L * 4);
      need_b.tot = std::max(need_b.tot, totc * MODP_L * 4);
      need_b.state = std::max(need_b.state, boxes * 2 * m0c * MODP_L * 4);
      need_b.hand_t = std::max(need_b.hand_t, boxes * (modp_fd_table_hand_words(Sc, (int)t) * 4 + l1t));
      need_b.hand_s = std::max(need_b.hand_s, boxes * (modp_fd_step_hand_words(Sc, (int)t, clc) * 4 + l1s));
    }
  }
  RET_IF(ensure(ctx, w.fd_xinv, need_b.xinv));
  RET_IF(ensure(ctx, w.fd_pre, need_b.pre));
  RET_IF(ensure(ctx, w.fd_tot, need_b.tot));
  RET_IF(ensure(ctx, w.fd_totinv, need_b.tot));
  RET_IF(ensure(ctx, w.fd_state, need_b.state));
  if (B > 1) RET_IF(ensure(ctx, w.fd_gather, need_b.xinv));
  if (!w.root) return fail(ctx, MPVSS_E_DEVICE, "eval_x: workspace not initialised");
  int* flag = (int*)w.fd_flag.p;
  int* dok = flag + 1;
  uint8_t* root_be = (uint8_t*)w.fd_root.p;
  uint8_t* rootinv_be = root_be + EB;
  uint32_t* xm = (uint32_t*)w.fd_xm.p;
  uint32_t* xseed = xm + seed0 * MODP_L;
  uint32_t* state_fwd = (uint32_t*)w.fd_state.p;
  uint32_t* state_bwd = state_fwd + (size_t)m0 * MODP_L;
  // inverses of the m Montgomery-form numbers at `in` into `inv_out`: Montgomery's trick on the device (3 products per
  // number), the single inversion of the root on the host, in stream order (no host synchronisation).  A root that
  // is 0 mod q (some X is 0: a commitment was) clears the flag.  `job`: pinned mailbox of the host callback.
  auto invert_batch = [&](const uint32_t* in, int m, uint32_t* inv_out, mpvss_ctx::Work::RootJob* job) -> int {
    const std::vector<int> ms = tree_sizes(m);
    const int nlev = (int)ms.size() - 1;
    std::vector<size_t> pre_off(nlev + 1, 0), tot_off(nlev + 1, 0);
    size_t po = 0, to = 0;
    for (int l = 0; l < nlev; ++l) { pre_off[l] = po; po += (size_t)ms[l]; }
    for (int l = 1; l <= nlev; ++l) { tot_off[l] = to; to += (size_t)ms[l]; }
    auto level_in = [&](int l) { return l == 0 ? in : (const uint32_t*)w.fd_tot.p + tot_off[l] * MODP_L; };
    auto level_out = [&](int l) { return (uint32_t*)w.fd_tot.p + tot_off[l] * MODP_L; };
    auto level_inv = [&](int l) { return l == 0 ? inv_out : (uint32_t*)w.fd_totinv.p + tot_off[l] * MODP_L; };
    for (int l = 0; l < nlev; ++l)
      LAUNCHCHK(ctx, modp_launch_binv_up(level_in(l), ms[l], G, (uint32_t*)w.fd_pre.p + pre_off[l] * MODP_L, level_out(l + 1),
                                         flag, ctx->consts, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_from_mont(level_in(nlev), 1, root_be, flag, ctx->consts, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(job->in_be, root_be, EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipLaunchHostFunc(ctx->stream, invert_root_on_host, job));
    HIPCHK(ctx, hipMemcpyAsync(rootinv_be, job->out_be, EB, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dok, &job->ok, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_fd_apply_ok(dok, flag, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_to_mont(rootinv_be, level_inv(nlev), 1, ctx->consts, ctx->stream));
    for (int l = nlev - 1; l >= 0; --l)
      LAUNCHCHK(ctx, modp_launch_binv_down(level_in(l), (const uint32_t*)w.fd_pre.p + pre_off[l] * MODP_L, level_inv(l + 1),
                                           ms[l], G, level_inv(l), flag, ctx->consts, ctx->stream));
    return 0;
  };
  // handoff buffers of the pipelined kernels (zeroed per call): [tables L2][stepping L2][tables L1][stepping L1]
  const size_t hand_t = modp_fd_table_hand_words(S, (int)t) * 4, hand_s = modp_fd_step_hand_words(S, (int)t, chain_len) * 4;
  const size_t hand_t1 = two_level ? modp_fd_table_hand_words(1, (int)t) * 4 : 0;
  const size_t hand_s1 = two_level ? modp_fd_step_hand_words(1, (int)t, m0) * 4 : 0;
  RET_IF(ensure(ctx, w.fd_hand_t, std::max(need_b.hand_t, boxes * (hand_t + hand_t1))));
  RET_IF(ensure(ctx, w.fd_hand_s, std::max(need_b.hand_s, boxes * (hand_s + hand_s1))));
  // strides from one box of the group to the next, in 32-bit words
  const size_t bx_xm = cnt * MODP_L, bx_state = (size_t)2 * m0 * MODP_L, bx_hand_t = (hand_t + hand_t1) / 4, bx_hand_s = (hand_s + hand_s1) / 4;
  // the seeds of the group's boxes side by side (one simultaneous inversion for all of them); a lone box's are already
  auto seeds_together = [&](int per_box) -> const uint32_t* {
    if (B == 1) return xseed;
    if (modp_launch_gather_rows(xseed, bx_xm, (size_t)per_box * MODP_L, B, (uint32_t*)w.fd_gather.p, ctx->stream) != 0) return nullptr;
    return (const uint32_t*)w.fd_gather.p;
  };
  w.root[0].one = 1;     // pinned: the copy below is asynchronous and reads it in stream order
  HIPCHK(ctx, hipMemcpyAsync(flag, &w.root[0].one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  RET_IF(span_begin(ctx, 0));
  if (!hpos) LAUNCHCHK(ctx, modp_launch_fd_check_positions_boxes(dpos, box_positions, (int)cnt, box_positions ? B : 1, flag, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(w.fd_hand_t.p, 0, boxes * (hand_t + hand_t1), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(w.fd_hand_s.p, 0, boxes * (hand_s + hand_s1), ctx->stream));
  // seeds: X at the S*t positions from seed0, kept in Montgomery form
  if (two_level) {
    LAUNCHCHK(ctx, modp_launch_commit_eval_boxes((const uint32_t*)w.cm.p, (int)t, dpos + seed0, box_positions, (int)t, B, xseed, nullptr,
                                                 cnt, flag, 1, ctx->consts, ctx->stream));
    const uint32_t* seeds1 = seeds_together((int)t);
    if (!seeds1) return fail(ctx, MPVSS_E_DEVICE, "eval_x: gather launch");
    RET_IF(invert_batch(seeds1, B * (int)t, (uint32_t*)w.fd_xinv.p, &w.root[1]));
    // one chain of stride 1 over the seed window: tables from its first t values, then m0 - 1 steps forward
    LAUNCHCHK(ctx, modp_launch_fd_table_boxes(xseed, bx_xm, (const uint32_t*)w.fd_xinv.p, t * MODP_L, 1, (int)t, state_fwd, state_bwd,
                                              bx_state, (uint32_t*)((uint8_t*)w.fd_hand_t.p + hand_t), bx_hand_t, B, flag, 0,
                                              ctx->consts, ctx->stream));
    LAUNCHCHK(ctx, launch_step(state_fwd, state_bwd, bx_state, 1, (int)t, 0, m0, m0, xseed, bx_xm,
                               (uint32_t*)((uint8_t*)w.fd_hand_s.p + hand_s), bx_hand_s, inject_fault == 3 ? 1 : 0));
  } else if (x_alone && !ctx->busy_with_others()) {
    LAUNCHCHK(ctx, modp_launch_commit_eval_row_boxes((const uint32_t*)w.cm.p, (int)t, dpos + seed0, box_positions, m0, B, xseed, cnt, flag, 1,
                                                     ctx->consts, ctx->stream, 1));
  } else {
    LAUNCHCHK(ctx, modp_launch_commit_eval_boxes((const uint32_t*)w.cm.p, (int)t, dpos + seed0, box_positions, m0, B, xseed, nullptr,
                                                 cnt, flag, 1, ctx->consts, ctx->stream));
  }
  const uint32_t* seeds2 = seeds_together(m0);
  if (!seeds2) return fail(ctx, MPVSS_E_DEVICE, "eval_x: gather launch");
  RET_IF(invert_batch(seeds2, B * m0, (uint32_t*)w.fd_xinv.p, &w.root[0]));
  // difference tables, stepping, conversion -- all gated on flag == 1.  Both kernels are pipelines of single-wave
  // stages that hand numbers down through zeroed buffers (see modp_kernels.hip).
  LAUNCHCHK(ctx, modp_launch_fd_table_boxes(xseed, bx_xm, (const uint32_t*)w.fd_xinv.p, (size_t)m0 * MODP_L, S, (int)t, state_fwd,
                                            state_bwd, bx_state, (uint32_t*)w.fd_hand_t.p, bx_hand_t, B, flag, inject_fault, ctx->consts,
                                            ctx->stream));
  LAUNCHCHK(ctx, launch_step(state_fwd, state_bwd, bx_state, S, (int)t, w0, chain_len, (int)cnt, xm, bx_xm, (uint32_t*)w.fd_hand_s.p,
                             bx_hand_s, inject_fault));
  LAUNCHCHK(ctx, modp_launch_from_mont(xm, B * (int)cnt, dX, flag, ctx->consts, ctx->stream));
  // fallback: plain Horner when the flag was cleared
  LAUNCHCHK(ctx, modp_launch_commit_eval_boxes((const uint32_t*)w.cm.p, (int)t, dpos, box_positions, (int)cnt, B, nullptr, dX, cnt, flag,
                                               0, ctx->consts, ctx->stream));
  RET_IF(span_end(ctx));
  return 0;
}

}  // namespace

extern "C" int mpvss_modp_commit_eval(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                      const int64_t* positions, size_t n, uint8_t* x_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!commitments || !positions || !x_out || t == 0 || t > 0x7fffffff || n > 0x7fffffff)
    return fail(ctx, MPVSS_E_INVALID, "commit_eval: bad argument (t must be >= 1)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  RET_IF(stage_commitments(ctx, space, commitments, t));
  const int64_t* dpos;
  RET_IF(stage_positions(ctx, space, positions, n, &dpos));
  uint8_t* dout = x_out;
  if (space == MPVSS_HOST) {
    RET_IF(ensure(ctx, ctx->w->xbe, n * EB));
    dout = (uint8_t*)ctx->w->xbe.p;
  }
  RET_IF(eval_x(ctx, t, dpos, space == MPVSS_HOST ? positions : nullptr, n, dout, 1, 0, true));
  if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, x_out, dout, n * EB));
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- DLEQ verifier commitments -------------------------------------------------------------------
namespace {
// a = B1^r * B2^c with the 64-entry table of B1.  Two kernels compute it: the pair-layout one, whose Montgomery reduction
// runs on the matrix cores (modp_pair_kernels.hip; default), and the VALU-only one (modp_kernels.hip; MPVSS_PAIR bit 0 clear).
// Same tables, exponents and results; measured in the headline pipeline 1.08-1.12 against 0.92-0.98 M share verifications/s.
// c_sched: sliding-window schedule of ONE shared challenge (then c_dev is unused), else fixed 4-bit windows of c_dev
// (stride c_stride; null: B1^r alone).
// MPVSS_PAIR: which kernels of the verifier's block path take the pair layout (bit 0: a2 = y^r Y^c, bit 1: the window tables,
// bit 2: g^r through the wide comb, bit 3: a1 = g^r X^c, bit 4: the bucket phase of the dealer's / participant's twin
// exponentiation -- 58.6 against 87.8 ms per 65536 shares alone on the chip, dealer 0.85 -> 0.92 M shares/s,
// profiles/r03_dealer_ab.txt; bit 5: the forward-difference stepping kernels of the X path when t >= MPVSS_FD_PAIR_MIN_T,
// eval_x).  Default 49: a2, the twin exponentiation and the stepping of large thresholds.  With the a2 kernel's registers
// allocated for two waves per SIMD (modp_pair_kernels.hip) the tables, g^r and a1 run as fast in either layout (MPVSS_PAIR 17 /
// 21 / 25 / 29: 1.08-1.10 M share verifications/s, profiles/r03_pair_occupancy_ab.txt) and stay VALU-only; at one wave per
// SIMD every further pair kernel cost throughput (1.063 M for a2 alone, 1.044 M with the tables, 1.022 M with all four, 0.942 M
// with none: profiles/r03_pair_ab.txt).  MPVSS_PAIR=0 is the VALU-only engine (no MFMA anywhere).
int pair_mask() {
  static const int m = fd_env("MPVSS_PAIR", 49) & 63;
  return m;
}
int launch_table_odd(mpvss_ctx* ctx, const uint8_t* base_dev, size_t cnt, uint32_t* tab) {
  if (pair_mask() & 2) return modp_launch_build_table_pair(base_dev, (int)cnt, tab, 16, 1, ctx->consts, ctx->pair_tables, ctx->stream);
  return modp_launch_build_table_odd(base_dev, (int)cnt, tab, ctx->consts, ctx->stream);
}
int launch_table64(mpvss_ctx* ctx, const uint8_t* base_dev, size_t cnt, uint32_t* tab) {
  if (pair_mask() & 2) return modp_launch_build_table_pair(base_dev, (int)cnt, tab, 64, 0, ctx->consts, ctx->pair_tables, ctx->stream);
  return modp_launch_build_table64(base_dev, (int)cnt, tab, ctx->consts, ctx->stream);
}

int launch_dual_exp_w6(mpvss_ctx* ctx, const uint32_t* t1, const uint32_t* t2, const uint8_t* r_dev, const uint8_t* c_dev,
                       size_t c_stride, const uint16_t* c_sched, size_t cnt, uint8_t* out_dev) {
  if (pair_mask() & 1)
    return modp_launch_dual_exp_w6_pair(t1, t2, r_dev, c_dev, c_stride, c_sched, (int)cnt, out_dev, ctx->consts, ctx->pair_tables,
                                        ctx->stream);
  if (c_sched) return modp_launch_dual_exp_w6_sched(t1, t2, r_dev, c_sched, (int)cnt, out_dev, ctx->consts, ctx->stream);
  return modp_launch_dual_exp_w6(t1, t2, r_dev, c_dev, c_stride, (int)cnt, out_dev, ctx->consts, ctx->stream);
}

// one base, two exponents, two results: the bucket phase in the pair layout when bit 4 of MPVSS_PAIR is set
int launch_twin_exp(mpvss_ctx* ctx, const uint8_t* base, const uint8_t* e1, const uint8_t* e2, size_t cnt, uint32_t* buckets,
                    uint32_t* occupancy, uint8_t* out1, uint8_t* out2) {
  if (pair_mask() & 16)
    return modp_launch_twin_exp_pair(base, e1, e2, (int)cnt, buckets, occupancy, out1, out2, ctx->consts, ctx->pair_tables, ctx->stream);
  return modp_launch_twin_exp(base, e1, e2, (int)cnt, buckets, occupancy, out1, out2, ctx->consts, ctx->stream);
}

// a = B1^r * B2^c for `cnt` shares.  tab_b1: shared table (stride 0) or nullptr -> per-number tables
// from b1_dev.  c: device pointer, stride c_stride (0 shared).
int dleq_side(mpvss_ctx* ctx, const uint32_t* shared_b1, const uint8_t* b1_dev, const uint8_t* b2_dev,
              const uint8_t* r_dev, const uint8_t* c_dev, size_t c_stride, int c_windows, size_t cnt,
              uint8_t* out_dev, const uint32_t* comb_b1 = nullptr, DevBuf* t2buf = nullptr) {
  const uint32_t *t1, *t2;
  size_t s1 = TABW;
  RET_IF(number_tables(ctx, b2_dev, cnt, t2buf ? *t2buf : ctx->w->tab2, &t2));
  if (comb_b1) {   // B1 is a generator with a comb table: no squarings for B1^r
    if ((pair_mask() & 1) && c_windows == 64 && cnt >= 64 && comb_bits_of(ctx, comb_b1) == 16) {
      // (verify_share's a1 = G^r pk^c: 45 K instead of 75 K issue slots per share on the pair layout)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb16_dual_exp_pair(comb_b1, t2, r_dev, c_dev, c_stride, (int)cnt, out_dev, ctx->consts,
                                                            ctx->pair_tables, ctx->stream));
      return 0;
    }
    TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(comb_b1, t2, TABW, r_dev, c_dev, c_stride, c_windows, (int)cnt,
                                                   out_dev, comb_bits_of(ctx, comb_b1), ctx->consts, ctx->stream));
    return 0;
  }
  if (!shared_b1 && c_windows == 64 && (cnt > ROW_MAX_NUMBERS || (cnt >= 1024 && ctx->crowded()))) {
    // per-share base with a full-width exponent and 256-bit second exponent(s): 6-bit windows for B1^r
    // (up to ROW_MAX_NUMBERS shares the row layout's shorter chain wins -- 1024 shares: 23.6 -> 11.9 ms per verify call -- unless the
    //  context is crowded)
    RET_IF(ensure(ctx, ctx->w->tab1, cnt * 4 * TABW * 4));
    TIMED_LAUNCH(ctx, 2, launch_table64(ctx, b1_dev, cnt, (uint32_t*)ctx->w->tab1.p));
    TIMED_LAUNCH(ctx, 3, launch_dual_exp_w6(ctx, (const uint32_t*)ctx->w->tab1.p, t2, r_dev, c_dev, c_stride, nullptr, cnt, out_dev));
    return 0;
  }
  if (shared_b1) {
    t1 = shared_b1;
    s1 = 0;
  } else {
    RET_IF(number_tables(ctx, b1_dev, cnt, ctx->w->tab1, &t1));
  }
  return dual_exp_any(ctx, t1, s1, t2, TABW, r_dev, c_dev, c_stride, c_windows, cnt, out_dev);
}
}  // namespace

extern "C" int mpvss_modp_dleq_commitments(mpvss_ctx* ctx, int space, const uint8_t* g1_host, const uint8_t* h1,
                                           const uint8_t* g2, const uint8_t* h2, const uint8_t* r, const uint8_t* c,
                                           int c_per_share, size_t n, uint8_t* a1_out, uint8_t* a2_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!g1_host || !h1 || !g2 || !h2 || !r || !c || !a1_out || !a2_out)
    return fail(ctx, MPVSS_E_INVALID, "dleq_commitments: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  const uint32_t* tg = nullptr;
  const uint32_t* cg = nullptr;
  if (generator_id(g1_host) >= 0)
    RET_IF(comb_table(ctx, generator_id(g1_host), &cg, n));
  else
    RET_IF(shared_table(ctx, g1_host, &tg));
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *dh1, *dg2, *dh2, *dr, *dc;
    RET_IF(stage_in(ctx, space, h1 + off * EB, cnt * EB, ctx->w->in_a, &dh1));
    RET_IF(stage_in(ctx, space, g2 + off * EB, cnt * EB, ctx->w->in_b, &dg2));
    RET_IF(stage_in(ctx, space, h2 + off * EB, cnt * EB, ctx->w->in_c, &dh2));
    RET_IF(stage_in(ctx, space, r + off * EB, cnt * EB, ctx->w->in_d, &dr));
    if (c_per_share)
      RET_IF(stage_in(ctx, space, c + off * EB, cnt * EB, ctx->w->in_e, &dc));
    else
      RET_IF(stage_in(ctx, MPVSS_HOST, c, EB, ctx->w->in_e, &dc));
    uint8_t *d1 = a1_out + off * EB, *d2 = a2_out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
      RET_IF(ensure(ctx, ctx->w->out2, cnt * EB));
      d1 = (uint8_t*)ctx->w->out1.p;
      d2 = (uint8_t*)ctx->w->out2.p;
    }
    const size_t cs = c_per_share ? EB : 0;
    int cw = 64;   // a 256-bit challenge only touches the low 64 windows
    if (c_per_share) {
      std::vector<uint8_t> hc;
      RET_IF(small_vec_to_host(ctx, space, c + off * EB, cnt * EB, hc));
      for (size_t i = 0; i < cnt && cw == 64; ++i)
        if (!fits_256_bits(hc.data() + i * EB)) cw = 512;
    } else if (!fits_256_bits(c)) {
      cw = 512;
    }
    RET_IF(dleq_side(ctx, tg, nullptr, (const uint8_t*)dh1, (const uint8_t*)dr, (const uint8_t*)dc, cs, cw, cnt, d1, cg));
    RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)dg2, (const uint8_t*)dh2, (const uint8_t*)dr, (const uint8_t*)dc, cs,
                     cw, cnt, d2));
    if (space == MPVSS_HOST) {
      RET_IF(copy_out(ctx, space, a1_out + off * EB, d1, cnt * EB));
      RET_IF(copy_out(ctx, space, a2_out + off * EB, d2, cnt * EB));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- verify_distribution_shares --------------------------------------------------------------------
// Split in three so that a box can be sharded over several engines (one per GPU):
//   compute : GPU work of one contiguous block of shares; X, Y, a1, a2 land in pinned host staging
//   absorb  : waits for the GPU and extends the ordered transcript hash with the block
//   verdict : finishes the hash and compares with the challenge
namespace {

static_assert(sizeof(mpvss::Sha256) <= MPVSS_TRANSCRIPT_STATE_BYTES, "transcript state size");

// q | q - 1 as 2 x 256 big-endian bytes in HBM (once per context)
int wellformed_bounds(mpvss_ctx* ctx, const uint8_t** out) {
  if (!ctx->qbounds.p) {
    uint8_t hb[2 * EB];
    modq_modulus_bytes(hb);
    memcpy(hb + EB, hb, EB);
    hb[2 * EB - 1] -= 1;                       // q is odd: q - 1 only changes the last byte
    DevBuf b;
    hipError_t e = hipMalloc(&b.p, sizeof(hb));
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipMalloc(bounds)", e);
    e = hipMemcpy(b.p, hb, sizeof(hb), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(b.p); return fail(ctx, MPVSS_E_DEVICE, "hipMemcpy(bounds)", e); }
    b.cap = sizeof(hb);
    ctx->qbounds = b;
  }
  *out = (const uint8_t*)ctx->qbounds.p;
  return 0;
}

int verify_block_compute_locked(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                const uint8_t* responses, size_t n, const uint8_t* challenge_host,
                                const mpvss_keyset* ks = nullptr, size_t key_offset = 0, uint8_t* wf_dev_out = nullptr,
                                const uint8_t* prestaged = nullptr) {
  // prestaged (space == MPVSS_HOST only): PINNED memory that already holds pubkeys | shares | responses (n x 256 each) | commitments
  // (t x 256) and stays valid until the block is absorbed: the inputs are not copied again
  if (!challenge_host) return fail(ctx, MPVSS_E_INVALID, "verify: null challenge");
  if (n > 0 && (!commitments || !positions || (!pubkeys && !ks) || !shares || !responses || t == 0 || t > 0x7fffffff ||
                n > 0x7fffffff))
    return fail(ctx, MPVSS_E_INVALID, "verify: bad argument (t must be >= 1, n < 2^31)");
  if (ks && (key_offset > ks->n || n > ks->n - key_offset))
    return fail(ctx, MPVSS_E_INVALID, "verify: shares outside the registered key set");
  int key_space = space;
  if (ks) {          // the keys themselves come from the key set's device copy
    pubkeys = (const uint8_t*)ks->keys.p + key_offset * EB;
    key_space = MPVSS_DEVICE;
  }
  mpvss_ctx::BlockSlot& sl = ctx->head_slot();
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "verify: every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
  const auto t_enq0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done)      // blocking wait: threads that absorb blocks sleep until the GPU is done instead of spinning
    HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
  sl.n = n;
  sl.kind = 0;
  sl.nbox = 1;
  sl.check_positions = false;
  sl.fd_used = false;
  sl.fd_chunks = 0;
  sl.enqueue_ms = 0;
  if (n == 0) {
    sl.busy = true;
    ctx->commit_head(sl);
    return MPVSS_OK;
  }
  if (space == MPVSS_HOST) RET_IF(check_positions_host(ctx, positions, n));
  // this block runs in its slot's own workspace and stream pair, so that boxes overlap on the GPU
  RET_IF(work_init(ctx, sl.work, nullptr));
  struct Restore {       // also: an early (error) return leaves nothing of this block running on the slot's streams
    mpvss_ctx* c;
    hipStream_t a, b;
    mpvss_ctx::BlockSlot* sl;
    ~Restore() {
      if (!sl->busy) {
        if (sl->work.sa) (void)hipStreamSynchronize(sl->work.sa);
        if (sl->work.sb) (void)hipStreamSynchronize(sl->work.sb);
      }
      c->sp = &c->main_spans; c->w = &c->work0; c->stream = a; c->stream_b = b;
    }
  } restore{ctx, ctx->stream, ctx->stream_b, &sl};
  ctx->w = &sl.work;
  sl.work.fd_used = false;
  ctx->stream = sl.work.sa;
  ctx->stream_b = sl.work.sb;
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  // Pinned staging of the slot: the outputs X, Y, a1, a2, the positions, one forward-difference flag per chunk and --
  // for callers that hand over host memory -- a copy of the inputs, so that every transfer is asynchronous and
  // nothing of the caller's is referenced after this call returns.
  constexpr size_t FLAGS = 64;                 // chunks per block whose flags are kept (the rest count as held)
  const size_t out_bytes = n * EB * 4 + n * 8 + FLAGS * 4;
  const size_t need = out_bytes + ((space == MPVSS_HOST && !prestaged) ? 3 * n * EB + t * EB : 0);
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  uint8_t* hX = (uint8_t*)sl.pin;
  uint8_t* hY = hX + n * EB;
  uint8_t* h1 = hY + n * EB;
  uint8_t* h2 = h1 + n * EB;
  int64_t* hpos = (int64_t*)(h2 + n * EB);
  int* hflags = (int*)((uint8_t*)sl.pin + n * EB * 4 + n * 8);
  if (space == MPVSS_HOST && prestaged) {
    memcpy(hpos, positions, n * 8);
    if (!ks) pubkeys = prestaged;
    shares = prestaged + n * EB;
    responses = prestaged + 2 * n * EB;
    commitments = prestaged + 3 * n * EB;
  } else if (space == MPVSS_HOST) {
    uint8_t* in = (uint8_t*)sl.pin + out_bytes;
    // three arrays of n x 256 bytes (48 MB at the headline shape) into pinned memory: this thread holds the context lock, so the
    // copies run side by side on helper threads (4-5 ms -> 1.5 ms of lock-held time per box).  (The host_buffers figure of the bench
    // did not move with it -- 1.05 beside 1.11 M from HBM, as before: the 5 % are not this copy.)
    if (n * EB >= ((size_t)4 << 20)) {
      hsc::parallel_indices(3, [&](unsigned k) {      // (serial on this thread if no helper thread can be had)
        if (k == 0) memcpy(in + n * EB, shares, n * EB);
        else if (k == 1) memcpy(in + 2 * n * EB, responses, n * EB);
        else if (!ks) memcpy(in, pubkeys, n * EB);
      });
    } else {
      if (!ks) memcpy(in, pubkeys, n * EB);
      memcpy(in + n * EB, shares, n * EB);
      memcpy(in + 2 * n * EB, responses, n * EB);
    }
    memcpy(in + 3 * n * EB, commitments, t * EB);
    memcpy(hpos, positions, n * 8);
    if (!ks) pubkeys = in;
    shares = in + n * EB;
    responses = in + 2 * n * EB;
    commitments = in + 3 * n * EB;
  }
  memcpy(sl.work.root->challenge, challenge_host, EB);
  // MPVSS_TRACE_ENQUEUE=1: a block whose enqueueing took more than 3 ms reports where the time went (stderr)
  static const int trace_enq = fd_env("MPVSS_TRACE_ENQUEUE", 0);
  double marks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto mark = [&](int i) {
    if (trace_enq) marks[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  };
  mark(0);
  RET_IF(stage_commitments(ctx, space, commitments, t));
  mark(1);
  const uint32_t* cg;
  RET_IF(comb_table(ctx, 0, &cg, n));
  RET_IF(ensure(ctx, ctx->w->in_e, EB));
  HIPCHK(ctx, hipMemcpyAsync(ctx->w->in_e.p, sl.work.root->challenge, EB, hipMemcpyHostToDevice, ctx->stream));
  const void* dchal = ctx->w->in_e.p;
  const int c_windows = fits_256_bits(challenge_host) ? 64 : 512;
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const int64_t* dpos;
    if (space == MPVSS_HOST) {
      const void* d;
      RET_IF(stage_in(ctx, space, hpos + off, cnt * 8, ctx->w->pos, &d));      // from the pinned copy
      dpos = (const int64_t*)d;
    } else {
      // device-resident positions are validated when the block is absorbed (no host sync here)
      dpos = positions + off;
      HIPCHK(ctx, hipMemcpyAsync(hpos + off, dpos, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
      sl.check_positions = true;
    }
    const void *dy, *dY, *dr;
    RET_IF(stage_in(ctx, key_space, pubkeys + off * EB, cnt * EB, ctx->w->in_a, &dy));
    RET_IF(stage_in(ctx, space, shares + off * EB, cnt * EB, ctx->w->in_b, &dY));
    RET_IF(stage_in(ctx, space, responses + off * EB, cnt * EB, ctx->w->in_c, &dr));
    if (wf_dev_out) {    // per-share well-formedness bytes for the caller's all-gather (never part of the box verdict)
      const uint8_t* dq;
      RET_IF(wellformed_bounds(ctx, &dq));
      LAUNCHCHK(ctx, verdict_launch_modp_wellformed((const uint8_t*)dy, (const uint8_t*)dY, (const uint8_t*)dr, dq, (int)cnt,
                                                    wf_dev_out + off, ctx->stream));
    }
    RET_IF(ensure(ctx, ctx->w->xbe, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out2, cnt * EB));
    uint8_t* dX = (uint8_t*)ctx->w->xbe.p;
    uint8_t* da1 = (uint8_t*)ctx->w->out1.p;
    uint8_t* da2 = (uint8_t*)ctx->w->out2.p;
    sl.work.fd_used = false;
    const int64_t* hp = space == MPVSS_HOST ? hpos + off : nullptr;
    struct Swap {
      mpvss_ctx* c; hipStream_t a;
      Swap(mpvss_ctx* c_, hipStream_t s) : c(c_), a(c_->stream) { c->stream = s; }
      ~Swap() { c->stream = a; }
    };
    // A one-box call of up to ROW_MAX_NUMBERS shares (BASELINE config C2 and everything the reference's tests and examples use) is the
    // latency of its chains, not work: Horner for every share ((t - 1) x 26 operations on the row layout) is shorter than the
    // forward-difference pipeline's seeds + inversion + tables + stepping, and a2 takes the row layout too -- C2: 35.5 -> 17 ms per call.
    // (Groups of small boxes -- mpvss_modp_verify_many -- keep the forward differences: there the work counts.  MPVSS_FD_MIN_SHARES
    // set: the tests' way to reach the forward-difference path at small sizes, honoured.)
    static const bool fd_min_default = getenv("MPVSS_FD_MIN_SHARES") == nullptr;
    const bool small_direct = fd_min_default && n <= ROW_MAX_NUMBERS && !ctx->crowded();
    if (ctx->w->sb && !small_direct && fd_applies(t, hp, cnt)) {
      // The forward-difference X path is a chain of latency-bound launches that occupy few wave slots (seeds,
      // inversion tree, difference tables, stepping) and runs on the block slot's high-priority stream.  a2 = y^r Y^c
      // and g^r do not depend on X: they run beside it on the slot's low-priority stream.
      RET_IF(ensure(ctx, ctx->w->tab1, cnt * TABW * 4 * 4));     // no reallocation while two streams are live
      RET_IF(ensure(ctx, ctx->w->tab2, cnt * TABW * 4));
      RET_IF(ensure(ctx, ctx->w->tab3, cnt * TABW * 4));
      RET_IF(ensure(ctx, ctx->w->gr_m, cnt * MODP_L * 4));
      const bool use_keys = ks && c_windows == 64;
      HIPCHK(ctx, hipEventRecord(ctx->w->ev_fork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->w->sb, ctx->w->ev_fork, 0));
      // Boxes in flight share the chip equally: eight boxes enqueued together finish together (a convoy).  What that
      // costs is the host's turn-around at the end of a convoy, and the cure that works is to enqueue the next box as
      // soon as a box's GPU work is done instead of after its transcript is hashed (run_box_pipeline: 0.90-0.93 -> 0.97 M
      // share verifications/s at K = 20).  What was measured and LOST, and is no longer in the source (DESIGN.md section 10
      // names the files): ordering the wide launches between boxes so that they finish oldest first; a2 in share ranges
      // with the hash following range by range; the inputs' and early outputs' copies ahead of the last kernel.
      // one challenge for every share of the box: a sliding-window schedule made on the host replaces the 64 fixed windows
      // of X^c and Y^c (about 51 products each) and their tables hold the odd powers only (8 instead of 14 products)
      const uint16_t* dsched = nullptr;
      if (c_windows == 64) {        // (with registered keys too: a1's X^c follows it; the key-table kernel keeps Y's full table)
        uint16_t* hs = sl.work.root[0].csched;
        sliding_schedule(sl.work.root[0].challenge, hs);
        if (hs[0] > 0) {
          RET_IF(ensure(ctx, ctx->w->csched, sizeof(sl.work.root[0].csched)));
          dsched = (const uint16_t*)ctx->w->csched.p;
        }
      }
      // registered keys: without the chain of 2046 squarings beside them, g^r and a1 = g^r X^c pay for their layout -- both on the
      // pair layout then (measured 1.65-1.72 -> 1.80-1.86 M share verifications/s, profiles/r05_keyset_ab.txt)
      // (a box that has the chip to itself too: g^r and a1 are the END of its critical path -- 123.8 -> 121 ms per lone call; in the
      //  pipeline the layouts measure the same, profiles/r06_pair_gr_ab.txt)
      const bool short_tail = (use_keys || !ctx->busy_with_others()) && (pair_mask() & 1);
      const bool pair_gr = (pair_mask() & 4) || short_tail;
      const bool pair_a1 = (pair_mask() & 8) || short_tail;
      {
        Swap sw(ctx, ctx->w->sb);      // the box's own low-priority stream: the boxes in flight share the chip
        // the schedule travels on THIS stream, ahead of a2; the a1 launch on the other stream waits for ev_gr, recorded below
        if (dsched)
          HIPCHK(ctx, hipMemcpyAsync(ctx->w->csched.p, sl.work.root[0].csched, (1 + 2 * (size_t)sl.work.root[0].csched[0]) * 2,
                                     hipMemcpyHostToDevice, ctx->stream));
        uint32_t* t2p = (uint32_t*)ctx->w->tab2.p;
        if (dsched && !use_keys) TIMED_LAUNCH(ctx, 2, launch_table_odd(ctx, (const uint8_t*)dY, cnt, t2p));
        else TIMED_LAUNCH(ctx, 2, modp_launch_build_table((const uint8_t*)dY, (int)cnt, t2p, ctx->consts, ctx->stream));
        if (use_keys) {
          // registered keys: y^r from the per-key tables (256 products, no squarings of its own) beside Y^c; on the pair
          // layout with the a2 kernel's pair bit (85 / 122 instead of 153 / 191 issue slots per operation)
          const uint32_t* kt = (const uint32_t*)ks->table.p + (key_offset + off) * modp_keyset_words_per_key();
          if ((pair_mask() & 1) && cnt >= 64)
            TIMED_LAUNCH(ctx, 3, modp_launch_keyset_dual_exp_pair(kt, t2p, (const uint8_t*)dr, (const uint8_t*)dchal, (int)cnt, da2, ctx->consts,
                                                                  ctx->pair_tables, ctx->stream));
          else
            TIMED_LAUNCH(ctx, 3, modp_launch_keyset_dual_exp(kt, t2p, (const uint8_t*)dr, (const uint8_t*)dchal, (int)cnt, da2,
                                                             ctx->consts, ctx->stream));
        } else if (c_windows == 64) {
          // 6-bit windows for y^r (64-entry tables, 18 KB per share): 341 products instead of 511
          uint32_t* t1p = (uint32_t*)ctx->w->tab1.p;
          TIMED_LAUNCH(ctx, 2, launch_table64(ctx, (const uint8_t*)dy, cnt, t1p));
          TIMED_LAUNCH(ctx, 3, launch_dual_exp_w6(ctx, t1p, t2p, (const uint8_t*)dr, (const uint8_t*)dchal, 0, dsched, cnt, da2));
        } else {
          uint32_t* t1p = (uint32_t*)ctx->w->tab1.p;
          TIMED_LAUNCH(ctx, 2, modp_launch_build_table((const uint8_t*)dy, (int)cnt, t1p, ctx->consts, ctx->stream));
          TIMED_LAUNCH(ctx, 3, modp_launch_dual_exp(t1p, TABW, t2p, TABW, (const uint8_t*)dr, (const uint8_t*)dchal, 0, c_windows,
                                                    (int)cnt, da2, ctx->consts, ctx->stream));
        }
        HIPCHK(ctx, hipEventRecord(ctx->w->ev_a2, ctx->stream));
        // g^r_i needs only the responses: it runs behind a2 instead of after the stepping phase
        if (pair_gr && comb_bits_of(ctx, cg) == 16)
          TIMED_LAUNCH(ctx, 1, modp_launch_comb16_exp_pair(cg, (const uint8_t*)dr, (int)cnt, (uint32_t*)ctx->w->gr_m.p, ctx->consts,
                                                           ctx->pair_tables, ctx->stream));
        else
          TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp_split(cg, cg, 0, (const uint8_t*)dr, (const uint8_t*)dchal, 0, 0,
                                                               (int)cnt, nullptr, 1, (uint32_t*)ctx->w->gr_m.p, comb_bits_of(ctx, cg), ctx->consts,
                                                               ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->w->ev_gr, ctx->stream));
      }
      mark(2);
      // (registered keys: a2 is a quarter of its usual work, so a box that has the chip to itself is its X path's latency -- there the
      // row-layout seeds pay: 122.8 -> 103.5 ms per lone call, profiles/r06_lone_key_cache.txt; without key tables they lose, eval_x)
      RET_IF(eval_x(ctx, t, dpos, hp, cnt, dX, 1, 0, use_keys));
      mark(3);
      // a1 = g^r * X^c: once X is known only X^c and one product remain
      {
        const uint32_t* tx;
        if (dsched) {
          RET_IF(ensure(ctx, ctx->w->tab3, cnt * TABW * 4));
          TIMED_LAUNCH(ctx, 2, launch_table_odd(ctx, dX, cnt, (uint32_t*)ctx->w->tab3.p));
          tx = (const uint32_t*)ctx->w->tab3.p;
        } else {
          RET_IF(number_tables(ctx, dX, cnt, ctx->w->tab3, &tx));
        }
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_gr, 0));
        if (dsched && pair_a1)
          TIMED_LAUNCH(ctx, 1, modp_launch_sched_exp_mul_pair(tx, TABW, dsched, (const uint32_t*)ctx->w->gr_m.p, (int)cnt, da1, ctx->consts,
                                                              ctx->pair_tables, ctx->stream));
        else if (dsched)
          TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp_sched(cg, tx, TABW, dsched, (int)cnt, da1, (uint32_t*)ctx->w->gr_m.p,
                                                               comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
        else
          TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp_split(cg, tx, TABW, (const uint8_t*)dr, (const uint8_t*)dchal, 0,
                                                               c_windows, (int)cnt, da1, 2, (uint32_t*)ctx->w->gr_m.p,
                                                               comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
      }
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_a2, 0));
    } else if (ctx->w->sb) {
      // A small box, or positions that are not consecutive: a2_i = y_i^r_i * Y_i^c (dleq.rs:79-81) needs nothing of X -- it runs on
      // the slot's second stream beside X_i (participant.rs:423-434) and a1_i = g^r_i * X_i^c (dleq.rs:75-77; X's table in tab3: a2
      // holds tab1 and tab2).  A box of 16 shares: 20.7 -> 7.8 ms per call (profiles/r06_small_box_latency.txt).
      HIPCHK(ctx, hipEventRecord(ctx->w->ev_fork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->w->sb, ctx->w->ev_fork, 0));
      {
        Swap sw(ctx, ctx->w->sb);
        RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)dy, (const uint8_t*)dY, (const uint8_t*)dr, (const uint8_t*)dchal,
                         0, c_windows, cnt, da2));
        HIPCHK(ctx, hipEventRecord(ctx->w->ev_a2, ctx->stream));
      }
      RET_IF(eval_x(ctx, t, dpos, hp, cnt, dX, 1, 0, false, small_direct));
      RET_IF(dleq_side(ctx, nullptr, nullptr, dX, (const uint8_t*)dr, (const uint8_t*)dchal, 0, c_windows, cnt, da1, cg, &ctx->w->tab3));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_a2, 0));
    } else {
      // X_i                                                  participant.rs:423-434
      RET_IF(eval_x(ctx, t, dpos, hp, cnt, dX));
      // a1_i = g^r_i * X_i^c, a2_i = y_i^r_i * Y_i^c           dleq.rs:66-84
      RET_IF(dleq_side(ctx, nullptr, nullptr, dX, (const uint8_t*)dr, (const uint8_t*)dchal, 0, c_windows, cnt, da1, cg));
      RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)dy, (const uint8_t*)dY, (const uint8_t*)dr, (const uint8_t*)dchal,
                       0, c_windows, cnt, da2));
    }
    HIPCHK(ctx, hipMemcpyAsync(hX + off * EB, dX, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hY + off * EB, dY, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h1 + off * EB, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h2 + off * EB, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    if (sl.work.fd_used) {     // the device's decision for this chunk (1 = forward differences held, 0 = fell back)
      sl.fd_used = true;
      if (sl.fd_chunks < FLAGS)
        HIPCHK(ctx, hipMemcpyAsync(hflags + sl.fd_chunks, sl.work.fd_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      ++sl.fd_chunks;
    }
    if (off + MAX_CHUNK < n) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // device buffers are reused
  }
  mark(4);
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  sl.busy = true;      // only a fully enqueued block occupies the slot (an error above leaves it free)
  sl.enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  if (trace_enq && sl.enqueue_ms > 3.0)
    fprintf(stderr, "[mpvss] slow enqueue %.1f ms: staging %.1f | commitments %.1f | tables+a2+g^r %.1f | X path %.1f | a1+copies %.1f | "
            "event %.1f\n", sl.enqueue_ms, marks[0], marks[1] - marks[0], marks[2] - marks[1], marks[3] - marks[2],
            marks[4] - marks[3], sl.enqueue_ms - marks[4]);
  ctx->commit_head(sl);
  return MPVSS_OK;
}

// ---- a GROUP of small boxes as one block ------------------------------------------------------------------------------
// A 4096-share box is 128 waves of the a2 kernel and a chain of latency-bound launches for X; the device runs eight
// launches at a time (its hardware queues; sixteen or more make everything slower) and each of those waves works for the
// same 27 ms whatever the box size: boxes of BASELINE config C2's size, one block each, reach 0.68-0.71 M share
// verifications/s with the chip mostly empty, and fewer with more of them in flight (profiles/r03_c2_ab.txt).
// mpvss_modp_verify_many therefore enqueues runs of consecutive boxes of ONE shape (same n and t; their own commitments,
// keys, positions, shares, responses and challenges) as ONE block: every launch of the block path covers all of them.
//   X path   : forward differences with the box as the second grid dimension (eval_x: box b's chains read its own
//              commitments; one simultaneous inversion for the seeds of all boxes; one flag for the group)
//   c        : one challenge per box, spread to one per share (fixed 4-bit windows instead of the lone box's sliding
//              schedule: about 38 products per share more)
//   absorb   : one transcript per box over its rows of the block's staging
// Same bytes as one block per box (tests/test_gpu_configs.py::test_c2_boxes_share_one_a2_launch).
constexpr size_t GROUP_MAX_BOX = 16384;      // boxes up to this size travel in groups
size_t group_shares_max() {
  constexpr size_t v = 32768;      // shares per group block; measured on C2: 16384 and 32768 -> 1.06-1.07 M, 65536 -> 0.97-1.00 M
  return v < MAX_CHUNK ? v : MAX_CHUNK;
}
bool box_groupable(const mpvss_modp_box& bx, size_t n, size_t t, int space) {
  constexpr size_t max_n = GROUP_MAX_BOX;
  if (!(bx.n == n && bx.t == t && n <= max_n && 2 * n <= group_shares_max() && !bx.keyset && bx.commitments &&
        bx.positions && bx.pubkeys && bx.shares && bx.responses && bx.challenge_host && fits_256_bits(bx.challenge_host)))
    return false;
  // (round 6: boxes too small for the forward differences -- n < 4096, the reference's own sizes -- travel in groups as well: X by
  //  Horner with the box as the second grid dimension, eval_x; 120 boxes of (1024, 32): 0.21 -> 0.85 M share verifications/s, 400 boxes
  //  of (5, 3): 765 -> 26 000 boxes/s, profiles/r06_many_small_boxes.txt)
  if (space == MPVSS_HOST)           // a box with a negative position is rejected on its own, with its own message
    for (size_t i = 0; i < n; ++i)
      if (bx.positions[i] < 0) return false;
  return true;
}

int verify_group_compute_locked(mpvss_ctx* ctx, int space, const mpvss_modp_box* boxes, size_t B) {
  const size_t n = boxes[0].n, t = boxes[0].t, N = B * n;
  mpvss_ctx::BlockSlot& sl = ctx->head_slot();
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "verify: every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
  const auto t_enq0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
  sl.n = N;
  sl.nbox = (unsigned)B;
  sl.kind = 0;
  sl.check_positions = false;
  sl.fd_used = false;
  sl.fd_chunks = 0;
  sl.enqueue_ms = 0;
  RET_IF(work_init(ctx, sl.work, nullptr));
  struct Restore {       // an early (error) return leaves nothing of this block running on the slot's streams
    mpvss_ctx* c;
    hipStream_t a, b;
    mpvss_ctx::BlockSlot* sl;
    ~Restore() {
      if (!sl->busy) {
        if (sl->work.sa) (void)hipStreamSynchronize(sl->work.sa);
        if (sl->work.sb) (void)hipStreamSynchronize(sl->work.sb);
        sl->nbox = 1;
      }
      c->sp = &c->main_spans; c->w = &c->work0; c->stream = a; c->stream_b = b;
    }
  } restore{ctx, ctx->stream, ctx->stream_b, &sl};
  mpvss_ctx::Work& w = sl.work;
  ctx->w = &w;
  w.fd_used = false;
  ctx->stream = w.sa;
  ctx->stream_b = w.sb;
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  // pinned staging as for a lone block (X | Y | a1 | a2 | positions | flags | host inputs), then the boxes' challenges
  constexpr size_t FLAGS = 64;
  const size_t out_bytes = N * EB * 4 + N * 8 + FLAGS * 4;
  const size_t in_bytes = space == MPVSS_HOST ? 3 * N * EB + B * t * EB : 0;
  const size_t need = out_bytes + in_bytes + B * EB;
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  uint8_t* hX = (uint8_t*)sl.pin;
  uint8_t* hY = hX + N * EB;
  uint8_t* h1 = hY + N * EB;
  uint8_t* h2 = h1 + N * EB;
  int64_t* hpos = (int64_t*)(h2 + N * EB);
  int* hflags = (int*)((uint8_t*)sl.pin + N * EB * 4 + N * 8);
  uint8_t* hin = (uint8_t*)sl.pin + out_bytes;
  uint8_t* hch = hin + in_bytes;
  RET_IF(ensure(ctx, w.in_a, N * EB));
  RET_IF(ensure(ctx, w.in_b, N * EB));
  RET_IF(ensure(ctx, w.in_c, N * EB));
  RET_IF(ensure(ctx, w.in_d, B * EB));
  RET_IF(ensure(ctx, w.in_e, N * EB));
  RET_IF(ensure(ctx, w.cbuf, B * t * EB));
  RET_IF(ensure(ctx, w.cm, B * t * MODP_L * 4));
  RET_IF(ensure(ctx, w.pos, N * 8));
  RET_IF(ensure(ctx, w.xbe, N * EB));
  RET_IF(ensure(ctx, w.out1, N * EB));
  RET_IF(ensure(ctx, w.out2, N * EB));
  RET_IF(ensure(ctx, w.tab1, N * TABW * 4 * 4));
  RET_IF(ensure(ctx, w.tab2, N * TABW * 4));
  RET_IF(ensure(ctx, w.tab3, N * TABW * 4));
  RET_IF(ensure(ctx, w.gr_m, N * MODP_L * 4));
  uint8_t *dy = (uint8_t*)w.in_a.p, *dY = (uint8_t*)w.in_b.p, *dr = (uint8_t*)w.in_c.p, *dcm = (uint8_t*)w.cbuf.p;
  int64_t* dpos = (int64_t*)w.pos.p;
  for (size_t b = 0; b < B; ++b) {
    const mpvss_modp_box& bx = boxes[b];
    memcpy(hch + b * EB, bx.challenge_host, EB);
    if (space == MPVSS_HOST) {
      memcpy(hin + b * n * EB, bx.pubkeys, n * EB);
      memcpy(hin + (N + b * n) * EB, bx.shares, n * EB);
      memcpy(hin + (2 * N + b * n) * EB, bx.responses, n * EB);
      memcpy(hin + 3 * N * EB + b * t * EB, bx.commitments, t * EB);
      memcpy(hpos + b * n, bx.positions, n * 8);
    } else {
      HIPCHK(ctx, hipMemcpyAsync(dy + b * n * EB, bx.pubkeys, n * EB, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(dY + b * n * EB, bx.shares, n * EB, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(dr + b * n * EB, bx.responses, n * EB, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(dcm + b * t * EB, bx.commitments, t * EB, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(dpos + b * n, bx.positions, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  if (space == MPVSS_HOST) {
    HIPCHK(ctx, hipMemcpyAsync(dy, hin, N * EB, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dY, hin + N * EB, N * EB, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dr, hin + 2 * N * EB, N * EB, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dcm, hin + 3 * N * EB, B * t * EB, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dpos, hpos, N * 8, hipMemcpyHostToDevice, ctx->stream));
  } else {
    // device-resident positions are validated when the block is absorbed (no host sync here)
    HIPCHK(ctx, hipMemcpyAsync(hpos, dpos, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    sl.check_positions = true;
  }
  HIPCHK(ctx, hipMemcpyAsync(w.in_d.p, hch, B * EB, hipMemcpyHostToDevice, ctx->stream));
  const uint8_t* dchal = (const uint8_t*)w.in_e.p;        // one challenge per share
  LAUNCHCHK(ctx, modp_launch_spread_rows((const uint8_t*)w.in_d.p, (int)n, (int)N, (uint8_t*)w.in_e.p, ctx->stream));
  w.cm_bytes_dev = dcm;
  LAUNCHCHK(ctx, modp_launch_to_mont(dcm, (uint32_t*)w.cm.p, (int)(B * t), ctx->consts, ctx->stream));
  const uint32_t* cg;
  RET_IF(comb_table(ctx, 0, &cg, N));
  uint8_t *dX = (uint8_t*)w.xbe.p, *da1 = (uint8_t*)w.out1.p, *da2 = (uint8_t*)w.out2.p;
  struct Swap {
    mpvss_ctx* c; hipStream_t a;
    Swap(mpvss_ctx* c_, hipStream_t s) : c(c_), a(c_->stream) { c->stream = s; }
    ~Swap() { c->stream = a; }
  };
  HIPCHK(ctx, hipEventRecord(w.ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(w.sb, w.ev_fork, 0));
  {
    Swap sw(ctx, w.sb);      // a2 = y^r Y^c and g^r beside the X path, as in a lone block
    uint32_t *t1p = (uint32_t*)w.tab1.p, *t2p = (uint32_t*)w.tab2.p;
    TIMED_LAUNCH(ctx, 2, modp_launch_build_table(dY, (int)N, t2p, ctx->consts, ctx->stream));
    // a small group with (almost) nothing else in flight -- a short run of tiny boxes -- is the latency of one chain: the row layout
    if (N <= ROW_MAX_NUMBERS && mpvss_ctx::NSLOT - ctx->free_top < 2) {
      TIMED_LAUNCH(ctx, 2, modp_launch_build_table(dy, (int)N, t1p, ctx->consts, ctx->stream));
      RET_IF(dual_exp_any(ctx, t1p, TABW, t2p, TABW, dr, dchal, EB, 64, N, da2));
    } else {
      TIMED_LAUNCH(ctx, 2, launch_table64(ctx, dy, N, t1p));
      TIMED_LAUNCH(ctx, 3, launch_dual_exp_w6(ctx, t1p, t2p, dr, dchal, EB, nullptr, N, da2));
    }
    HIPCHK(ctx, hipEventRecord(w.ev_a2, ctx->stream));
    if ((pair_mask() & 4) && comb_bits_of(ctx, cg) == 16)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb16_exp_pair(cg, dr, (int)N, (uint32_t*)w.gr_m.p, ctx->consts, ctx->pair_tables, ctx->stream));
    else
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp_split(cg, cg, 0, dr, dchal, 0, 0, (int)N, nullptr, 1, (uint32_t*)w.gr_m.p,
                                                           comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
    HIPCHK(ctx, hipEventRecord(w.ev_gr, ctx->stream));
  }
  RET_IF(eval_x(ctx, t, dpos, nullptr, n, dX, B, n));
  {
    const uint32_t* tx;
    RET_IF(number_tables(ctx, dX, N, w.tab3, &tx));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, w.ev_gr, 0));
    TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp_split(cg, tx, TABW, dr, dchal, EB, 64, (int)N, da1, 2, (uint32_t*)w.gr_m.p,
                                                         comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
  }
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, w.ev_a2, 0));
  HIPCHK(ctx, hipMemcpyAsync(hX, dX, N * EB, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(hY, dY, N * EB, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(h1, da1, N * EB, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(h2, da2, N * EB, hipMemcpyDeviceToHost, ctx->stream));
  if (w.fd_used) {
    sl.fd_used = true;
    HIPCHK(ctx, hipMemcpyAsync(hflags, w.fd_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    sl.fd_chunks = 1;
  }
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  sl.busy = true;
  sl.enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  ctx->commit_head(sl);
  return MPVSS_OK;
}

// Room for `count` more blocks in the ring (context lock held; released while waiting).  Blocks whose consumer is at work -- another
// thread's one-call entry point, a pipeline, a claimed block -- free their slots by themselves: wait for them.  When only blocks of
// the explicit block API fill the ring nobody but the caller can absorb them: an error, as it always was.
int wait_for_room(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, unsigned count = 1) {
  auto room = [&] {
    if (ctx->free_top < count) return false;
    for (unsigned k = 0; k < count; ++k)
      if (ctx->ring[(ctx->head + k) % mpvss_ctx::NSLOT] >= 0) return false;
    return true;
  };
  while (!room()) {
    if (!ctx->consumers_at_work())
      return fail(ctx, MPVSS_E_INVALID, "every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
    ctx->slot_cv.wait(lk);
  }
  return 0;
}

// Which block an absorb call works on (context lock held).  ticket == null: the oldest block that waits for a FIFO consumer
// (the explicit block API); ticket + !by_position: the block mpvss_block_claim handed out under this number; ticket +
// by_position: the caller's OWN block of this number (mpvss_ctx::own_last: the one-call entry points and the library's box
// pipeline).  `kind` / `kind2`: the slot kinds the calling entry point absorbs.  Returns null with the error recorded.
mpvss_ctx::BlockSlot* select_block(mpvss_ctx* ctx, const unsigned long long* ticket, bool by_position, int kind, int kind2, const char* who) {
  char msg[160];
  auto bad = [&](const char* what) -> mpvss_ctx::BlockSlot* {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    (void)fail(ctx, MPVSS_E_INVALID, msg);
    return nullptr;
  };
  mpvss_ctx::BlockSlot* sl;
  if (!ticket) {
    sl = &ctx->fifo_front();
    if (!sl->busy) return bad("no block in flight");
    if (sl->kind != kind && sl->kind != kind2) return bad("the oldest block in flight belongs to another entry point");
    ++ctx->tail;
  } else {
    sl = &ctx->ring_slot((unsigned)*ticket);
    if (by_position) {
      if (!sl->busy || !sl->owned || sl->absorbing) return bad("no block of this caller at this position");
    } else {
      if (!sl->busy || !sl->claimed) return bad("no block was claimed with this ticket");
    }
    if (sl->kind != kind && sl->kind != kind2) return bad("the block belongs to another entry point");
    sl->claimed = false;
  }
  return sl;
}

// Called with `lk` (the context lock) held.  The lock is RELEASED while this thread waits for the block's GPU work
// and hashes it, so that other host threads can enqueue blocks or absorb the next ones meanwhile (every box has its
// own transcript; a 65536-share box is 35 ms of SHA-256).
int verify_block_absorb_locked(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, uint8_t* state, uint8_t* x_out,
                               uint8_t* a1_out, uint8_t* a2_out, uint8_t* y_out = nullptr,
                               const unsigned long long* ticket = nullptr, bool by_position = false,
                               unsigned states = 1, char* box_bad = nullptr) {
  // states > 1: `state` holds that many transcript states, one per box of a GROUP block (its nbox); box_bad[b] = 1 marks a
  // box with a negative position (device-resident positions are looked at here)
  if (!state) return fail(ctx, MPVSS_E_INVALID, "absorb: null transcript state");
  {
    // only the library's own pipeline makes groups and knows how many states to bring; nothing is consumed otherwise
    mpvss_ctx::BlockSlot& peek = ticket ? ctx->ring_slot((unsigned)*ticket) : ctx->fifo_front();
    if (peek.busy && peek.kind == 0 && ((peek.nbox ? peek.nbox : 1u) != states || (states > 1 && (x_out || a1_out || a2_out || y_out))))
      return fail(ctx, MPVSS_E_INVALID, "absorb: the block is a group of boxes (one transcript state per box)");
  }
  mpvss_ctx::BlockSlot* slp = select_block(ctx, ticket, by_position, 0, 0, "absorb");
  if (!slp) return MPVSS_E_INVALID;
  mpvss_ctx::BlockSlot& sl = *slp;
  const size_t n = sl.n;
  const size_t nbox = sl.nbox ? sl.nbox : 1;
  if (n == 0) {
    ctx->release(sl);
    sl.absorbing = false;
    if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
    return MPVSS_OK;
  }
  sl.absorbing = true;
  {
    const hipError_t e_dev = hipSetDevice(ctx->device);
    if (e_dev != hipSuccess) {               // give the slot back: the block is lost, the ring is not
      ctx->release(sl);
      sl.absorbing = false;
      if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
      return fail(ctx, MPVSS_E_DEVICE, "absorb: hipSetDevice", e_dev);
    }
  }
  lk.unlock();
  const auto t_w0 = std::chrono::steady_clock::now();
  const uint8_t* hX = (const uint8_t*)sl.pin;
  const uint8_t* hY = hX + n * EB;
  const uint8_t* h1 = hY + n * EB;
  const uint8_t* h2 = h1 + n * EB;
  bool positions_ok = true;
  const hipError_t e = hipEventSynchronize(sl.done);
  const auto t_w1 = std::chrono::steady_clock::now();
  if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
  if (e == hipSuccess) {
    if (sl.check_positions && nbox == 1) {
      const int64_t* pos = (const int64_t*)(h2 + n * EB);
      for (size_t i = 0; i < n && positions_ok; ++i) positions_ok = pos[i] >= 0;
    }
    if (nbox > 1) {
      const size_t per = n / nbox;
      const int64_t* pos = (const int64_t*)(h2 + n * EB);
      for (size_t b = 0; b < nbox; ++b) {
        bool ok = true;
        if (sl.check_positions)
          for (size_t i = b * per; i < (b + 1) * per && ok; ++i) ok = pos[i] >= 0;
        if (box_bad) box_bad[b] = ok ? 0 : 1;
        if (!ok) continue;
        mpvss::Sha256 tr;
        memcpy(&tr, state + b * MPVSS_TRANSCRIPT_STATE_BYTES, sizeof(tr));
        frame_shares(tr, hX, hY, h1, h2, b * per, (b + 1) * per);
        memcpy(state + b * MPVSS_TRANSCRIPT_STATE_BYTES, &tr, sizeof(tr));
      }
    } else if (positions_ok) {
      mpvss::Sha256 tr;
      memcpy(&tr, state, sizeof(tr));
      // the caller's copies of the arrays (the one-call entry points: 64 MB per 65536 shares) beside the hash, not behind it
      const bool copies = (x_out || y_out || a1_out || a2_out) && n * EB >= ((size_t)1 << 20);
      hsc::parallel_indices(copies ? 2 : 1, [&](unsigned k) {
        if (k == 0) {
          frame_shares(tr, hX, hY, h1, h2, 0, n);     // dleq.rs:87-99, share order = array order
          if (copies) return;
        }
        if (x_out) memcpy(x_out, hX, n * EB);
        if (y_out) memcpy(y_out, hY, n * EB);
        if (a1_out) memcpy(a1_out, h1, n * EB);
        if (a2_out) memcpy(a2_out, h2, n * EB);
      });
      memcpy(state, &tr, sizeof(tr));
    }
  }
  const auto t_h1 = std::chrono::steady_clock::now();
  lk.lock();
  ctx->release(sl);
  sl.absorbing = false;
  if (e == hipSuccess && sl.fd_used) {
    const int* hflags = (const int*)(h2 + n * EB + n * 8);
    bool held = true;
    for (unsigned k = 0; k < sl.fd_chunks && k < 64; ++k) held = held && hflags[k] == 1;
    ++ctx->fd_blocks;
    if (!held) ++ctx->fd_fallbacks;
  }
  if (e != hipSuccess) return fail(ctx, MPVSS_E_DEVICE, "absorb: hipEventSynchronize", e);
  if (!positions_ok) return fail(ctx, MPVSS_E_INVALID, "negative position (the reference panics: negative exponent)");
  RET_IF(spans_sum(ctx, sl.spans, ctx->kernel_ms));
  {
    mpvss_ctx::PipeStats& ps = ctx->pstats;
    ps.enqueue_ms += sl.enqueue_ms;
    ps.wait_ms += std::chrono::duration<double, std::milli>(t_w1 - t_w0).count();
    ps.hash_ms += std::chrono::duration<double, std::milli>(t_h1 - t_w1).count();
    for (int k = 0; k < 4; ++k) {
      ps.kernel_ms[k] += ctx->kernel_ms[k];
      ps.kernel_launches[k] += (unsigned long long)ctx->kernel_launches[k];
    }
    ps.blocks += (unsigned long long)nbox;      // a group block counts as its boxes
  }
  return MPVSS_OK;
}

}  // namespace

extern "C" void mpvss_transcript_init(uint8_t* state) {
  memset(state, 0, MPVSS_TRANSCRIPT_STATE_BYTES);
  mpvss::Sha256 tr;
  memcpy(state, &tr, sizeof(tr));
}

extern "C" int mpvss_modp_transcript_absorb(uint8_t* state, const uint8_t* elements, size_t count) {
  if (!state || (count && !elements)) return MPVSS_E_INVALID;
  mpvss::Sha256 tr;
  memcpy(&tr, state, sizeof(tr));
  for (size_t i = 0; i < count; ++i) frame_update(tr, elements + i * EB);
  memcpy(state, &tr, sizeof(tr));
  return MPVSS_OK;
}

extern "C" int mpvss_modp_transcript_verdict(const uint8_t* state, const uint8_t* challenge_host, int* verdict,
                                             uint8_t* digest32_out) {
  if (!state || !challenge_host || !verdict) return MPVSS_E_INVALID;
  mpvss::Sha256 tr;
  memcpy(&tr, state, sizeof(tr));
  uint8_t digest[32];
  tr.final(digest);
  if (digest32_out) memcpy(digest32_out, digest, 32);
  *verdict = challenge_matches(digest, challenge_host) ? 1 : 0;   // participant.rs:451-454
  return MPVSS_OK;
}

extern "C" int mpvss_modp_verify_block_compute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                               const int64_t* positions, const uint8_t* pubkeys,
                                               const uint8_t* shares, const uint8_t* responses, size_t n,
                                               const uint8_t* challenge_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_block_compute_locked(ctx, space, commitments, t, positions, pubkeys, shares, responses, n,
                                     challenge_host);
}

extern "C" int mpvss_modp_verify_block_compute_flags(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                                     const int64_t* positions, const uint8_t* pubkeys,
                                                     const uint8_t* shares, const uint8_t* responses, size_t n,
                                                     const uint8_t* challenge_host, uint8_t* wellformed_dev_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_block_compute_locked(ctx, space, commitments, t, positions, pubkeys, shares, responses, n,
                                     challenge_host, nullptr, 0, wellformed_dev_out);
}

// ---- registered public keys -------------------------------------------------------------------------
namespace {
int keyset_create_locked(mpvss_ctx* ctx, int space, const uint8_t* pubkeys, size_t n, mpvss_keyset** out);
}
extern "C" int mpvss_modp_keyset_create(mpvss_ctx* ctx, int space, const uint8_t* pubkeys, size_t n, mpvss_keyset** out) {
  if (!ctx || !out) return MPVSS_E_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return keyset_create_locked(ctx, space, pubkeys, n, out);
}
namespace {
int keyset_create_locked(mpvss_ctx* ctx, int space, const uint8_t* pubkeys, size_t n, mpvss_keyset** out) {
  *out = nullptr;
  if (!pubkeys || n == 0 || n > 0x7fffffff / 8) return fail(ctx, MPVSS_E_INVALID, "keyset_create: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  mpvss_keyset* ks = new mpvss_keyset();
  ks->n = n;
  auto cleanup = [&](int rc) {
    if (ks->table.p) (void)hipFree(ks->table.p);
    if (ks->keys.p) (void)hipFree(ks->keys.p);
    delete ks;
    return rc;
  };
  static const int trace_ks = fd_env("MPVSS_TRACE_KEYSET", 0);      // stderr: allocation against table build
  const auto t_k0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&ks->keys.p, n * EB);
  const size_t table_bytes = n * modp_keyset_words_per_key() * 4;
  if (e == hipSuccess) {
    if (ctx->spare_table && ctx->spare_table_cap >= table_bytes) {      // the buffer a cache dropped a set from
      ks->table.p = ctx->spare_table;
      ks->table.cap = ctx->spare_table_cap;
      ctx->spare_table = nullptr;
      ctx->spare_table_cap = 0;
    } else {
      ctx->drop_spare_table();                                           // (too small for this set: its memory is better free)
      e = hipMalloc(&ks->table.p, table_bytes);
      if (e == hipSuccess) ks->table.cap = table_bytes;
    }
  }
  if (e != hipSuccess) return cleanup(fail(ctx, MPVSS_E_NOMEM, "keyset_create: hipMalloc", e));
  const auto t_k1 = std::chrono::steady_clock::now();
  e = hipMemcpyAsync(ks->keys.p, pubkeys, n * EB, space == MPVSS_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                     ctx->stream);
  if (e != hipSuccess) return cleanup(fail(ctx, MPVSS_E_DEVICE, "keyset_create: copy", e));
  if (modp_launch_keyset_build((const uint8_t*)ks->keys.p, (int)n, (uint32_t*)ks->table.p, ctx->consts, ctx->stream) != 0)
    return cleanup(fail(ctx, MPVSS_E_DEVICE, "keyset_create: launch"));
  e = hipStreamSynchronize(ctx->stream);     // the tables are read from every stream afterwards
  if (e != hipSuccess) return cleanup(fail(ctx, MPVSS_E_DEVICE, "keyset_create: build", e));
  if (trace_ks)
    fprintf(stderr, "[mpvss] keyset_create n=%zu: hipMalloc %.1f ms, copy + table build %.1f ms\n", n,
            std::chrono::duration<double, std::milli>(t_k1 - t_k0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_k1).count());
  *out = ks;
  return MPVSS_OK;
}

// (no block may still read the tables: the caller's business)  keep: the set belonged to one of the context's key caches and a cache
// is still on -- its table buffer becomes the spare (the larger of the two stays)
void keyset_free_locked(mpvss_ctx* ctx, mpvss_keyset* ks, bool keep = false) {
  (void)hipSetDevice(ctx->device);
  keep = keep && (ctx->key_cache_min_boxes > 0 || ctx->kc_max_sets.load() > 0);
  if (ks->table.p) {
    if (keep && ks->table.cap > ctx->spare_table_cap) {
      ctx->drop_spare_table();
      ctx->spare_table = ks->table.p;
      ctx->spare_table_cap = ks->table.cap;
    } else {
      (void)hipFree(ks->table.p);
    }
  }
  if (ks->keys.p) (void)hipFree(ks->keys.p);
  delete ks;
}

// The cross-call key cache (context lock held): the key set to verify a box of `n` shares against the HOST key array with this SHA-256
// against, or null for the plain path.  A returned set has its `users` count raised: key_cache_release() when the box is absorbed.
mpvss_keyset* key_cache_acquire(mpvss_ctx* ctx, const uint8_t digest[32], const uint8_t* pubkeys_host, size_t n) {
  const int max_sets = ctx->kc_max_sets.load();
  if (max_sets <= 0) return nullptr;
  mpvss_ctx::KeyCacheEntry* e = nullptr;
  for (auto& x : ctx->kc)
    if (x.n == n && memcmp(x.digest, digest, 32) == 0) { e = &x; break; }
  if (!e) {
    if (ctx->kc.size() >= 64) {                    // sightings without tables are cheap, but not unbounded: forget the oldest such record
      size_t victim = ctx->kc.size();
      for (size_t i = 0; i < ctx->kc.size(); ++i)
        if (!ctx->kc[i].ks && (victim == ctx->kc.size() || ctx->kc[i].last_use < ctx->kc[victim].last_use)) victim = i;
      if (victim == ctx->kc.size()) return nullptr;
      ctx->kc.erase(ctx->kc.begin() + (long)victim);
    }
    ctx->kc.emplace_back();
    e = &ctx->kc.back();
    memcpy(e->digest, digest, 32);
    e->n = n;
  }
  e->last_use = ++ctx->kc_clock;
  ++e->sightings;
  if (!e->ks) {
    if ((int)e->sightings < ctx->kc_min_sightings) return nullptr;
    // room: at most max_sets sets with tables -- the least recently used one that no block reads goes
    int with_tables = 0;
    for (auto& x : ctx->kc) with_tables += x.ks != nullptr;
    while (with_tables >= max_sets) {
      mpvss_ctx::KeyCacheEntry* lru = nullptr;
      for (auto& x : ctx->kc)
        if (x.ks && x.users == 0 && (!lru || x.last_use < lru->last_use)) lru = &x;
      if (!lru) return nullptr;                     // every set is in use: this box goes the plain way
      keyset_free_locked(ctx, lru->ks, true);
      lru->ks = nullptr;
      lru->sightings = 0;
      --with_tables;
    }
    // HBM: the tables may take what is free now minus a reserve for the block slots' workspaces (32 GB: a dozen headline boxes)
    size_t free_b = 0, total_b = 0;
    const size_t table_b = n * ((size_t)modp_keyset_words_per_key() * 4 + EB);
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    free_b += ctx->spare_table_cap;                 // (what a dropped set left behind is there to be used)
    if (free_b < table_b || free_b - table_b < ((size_t)32 << 30)) return nullptr;
    mpvss_keyset* ks = nullptr;
    if (keyset_create_locked(ctx, MPVSS_HOST, pubkeys_host, n, &ks) != MPVSS_OK) {      // costs speed, not the call
      (void)hipGetLastError();
      std::lock_guard<std::mutex> g(ctx->err_mu);
      ctx->err.clear();
      return nullptr;
    }
    e->ks = ks;
  }
  ++e->users;
  return e->ks;
}
void key_cache_release(mpvss_ctx* ctx, const mpvss_keyset* ks) {
  for (auto& x : ctx->kc)
    if (x.ks == ks && x.users > 0) { --x.users; return; }
}
// The identity of a host key array in the cache: SHA-256 over the SHA-256 of eight slices of its bytes (hashed side by side on helper
// threads: 16.8 MB in 1-2 ms).  key_slice_digest is one leaf, for callers that run the leaves beside other work.
constexpr unsigned KEY_SLICES = 8;
void key_slice_digest(const uint8_t* pubkeys, size_t n, unsigned j, uint8_t leaf[32]) {
  const size_t bytes = n * EB, slice = (bytes + KEY_SLICES - 1) / KEY_SLICES;
  const size_t lo = std::min(bytes, (size_t)j * slice), hi = std::min(bytes, lo + slice);
  mpvss::sha256(pubkeys + lo, hi - lo, leaf);
}
void key_array_digest(const uint8_t* pubkeys, size_t n, uint8_t out[32]) {
  uint8_t leaves[KEY_SLICES][32];
  hsc::parallel_indices(KEY_SLICES, [&](unsigned j) { key_slice_digest(pubkeys, n, j, leaves[j]); });
  mpvss::sha256(&leaves[0][0], sizeof(leaves), out);
}
struct CacheUse {          // gives a cached key set back when the call that acquired it returns (the context lock is held again then)
  mpvss_ctx* c; const mpvss_keyset* ks;
  ~CacheUse() { if (ks) key_cache_release(c, ks); }
};
// The dealer's side of the cross-call cache (context lock held on entry and on return, released while the keys are hashed): the key
// set for a box dealt to the n keys of this HOST array, or null -- the cache is off, the box is small, or the keys are new.
const mpvss_keyset* dealer_key_cache(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, const uint8_t* pubkeys_host, size_t n) {
  if (ctx->kc_max_sets.load() <= 0 || n <= GROUP_MAX_BOX || !(pair_mask() & 1)) return nullptr;
  uint8_t kd[32];
  lk.unlock();
  key_array_digest(pubkeys_host, n, kd);
  lk.lock();
  return key_cache_acquire(ctx, kd, pubkeys_host, n);
}
}  // namespace

extern "C" int mpvss_ctx_set_key_cache_lru(mpvss_ctx* ctx, int max_sets, int min_sightings) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (max_sets < 0 || max_sets > 8 || min_sightings < 1) return fail(ctx, MPVSS_E_INVALID, "set_key_cache_lru: 0 <= max_sets <= 8, min_sightings >= 1");
  const int prev = ctx->kc_max_sets.load();
  ctx->kc_max_sets.store(max_sets);
  ctx->kc_min_sightings = min_sightings;
  if (max_sets == 0) {                               // off: drop what no block reads (sets in use go when the context does)
    for (auto& x : ctx->kc)
      if (x.ks && x.users == 0) { keyset_free_locked(ctx, x.ks); x.ks = nullptr; }
    ctx->kc.erase(std::remove_if(ctx->kc.begin(), ctx->kc.end(), [](const mpvss_ctx::KeyCacheEntry& x) { return x.ks == nullptr; }), ctx->kc.end());
    if (ctx->key_cache_min_boxes == 0) ctx->drop_spare_table();      // both caches off: the spare buffer goes too
  }
  return prev;
}

extern "C" int mpvss_ctx_set_key_cache(mpvss_ctx* ctx, int min_boxes) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (min_boxes < 0 || min_boxes == 1) return fail(ctx, MPVSS_E_INVALID, "set_key_cache: min_boxes must be 0 (off) or >= 2");
  const int prev = ctx->key_cache_min_boxes;
  ctx->key_cache_min_boxes = min_boxes;
  if (min_boxes == 0 && ctx->kc_max_sets.load() == 0) {
    (void)hipSetDevice(ctx->device);
    ctx->drop_spare_table();                                           // both caches off: the spare buffer goes too
  }
  return prev;
}

extern "C" void mpvss_modp_keyset_destroy(mpvss_ctx* ctx, mpvss_keyset* ks) {
  if (!ks) return;
  if (ctx) {
    (void)mpvss_ctx_synchronize(ctx);        // no launch may still read the tables
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (ks->table.p) (void)hipFree(ks->table.p);
    if (ks->keys.p) (void)hipFree(ks->keys.p);
  }
  delete ks;
}

extern "C" size_t mpvss_modp_keyset_bytes(const mpvss_keyset* ks) {
  return ks ? ks->n * (modp_keyset_words_per_key() * 4 + EB) : 0;
}

extern "C" int mpvss_modp_verify_block_compute_keyset(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                                      const int64_t* positions, const mpvss_keyset* keyset,
                                                      size_t key_offset, const uint8_t* shares,
                                                      const uint8_t* responses, size_t n, const uint8_t* challenge_host) {
  if (!ctx || !keyset) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_block_compute_locked(ctx, space, commitments, t, positions, nullptr, shares, responses, n, challenge_host,
                                     keyset, key_offset);
}

extern "C" int mpvss_modp_verify_block_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host,
                                              uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  return verify_block_absorb_locked(ctx, lk, state, x_out_host, a1_out_host, a2_out_host);
}

// Claim / absorb in two steps: several host threads can then absorb blocks concurrently AND know which block each of
// them holds -- needed when the transcript state of a block comes from somewhere else (the previous rank of a sharded
// verification) and has to be fetched between the two steps.
extern "C" int mpvss_block_claim(mpvss_ctx* ctx, unsigned long long* ticket_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ticket_out) return fail(ctx, MPVSS_E_INVALID, "claim: null ticket");
  mpvss_ctx::BlockSlot& sl = ctx->fifo_front();
  if (!sl.busy) return fail(ctx, MPVSS_E_INVALID, "claim: no block in flight");
  if (sl.kind != 0 && sl.kind != 2) return fail(ctx, MPVSS_E_INVALID, "claim: the oldest block in flight is not a distribution block");
  sl.absorbing = true;
  sl.claimed = true;
  *ticket_out = sl.pos;
  ++ctx->tail;
  return MPVSS_OK;
}

extern "C" int mpvss_modp_verify_block_absorb_claimed(mpvss_ctx* ctx, unsigned long long ticket, uint8_t* state,
                                                      uint8_t* x_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  return verify_block_absorb_locked(ctx, lk, state, x_out_host, a1_out_host, a2_out_host, nullptr, &ticket);
}

extern "C" int mpvss_modp_verify_distribution(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                              const int64_t* positions, const uint8_t* pubkeys,
                                              const uint8_t* shares, const uint8_t* responses, size_t n,
                                              const uint8_t* challenge_host, int* verdict, uint8_t* digest32_out,
                                              uint8_t* x_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  if (!verdict || !challenge_host) return fail(ctx, MPVSS_E_INVALID, "verify_distribution: bad argument");
  *verdict = 0;
  uint8_t state[MPVSS_TRANSCRIPT_STATE_BYTES];
  mpvss_transcript_init(state);
  // Host callers: the inputs go into a pinned buffer of the call's own with the lock RELEASED (other callers enqueue and absorb
  // meanwhile), and -- with the cross-call key cache on (mpvss_ctx_set_key_cache_lru) -- the key array is identified beside those
  // copies by a SHA-256 tree hash of its bytes (eight slices: 16.8 MB in 1-2 ms), looked up, and, seen often enough, verified
  // against per-key tables built once.
  const bool whole = n > 0 && n <= MAX_CHUNK && commitments && positions && pubkeys && shares && responses && t > 0 && t <= 0x7fffffff;
  const bool prestage = space == MPVSS_HOST && whole && n * EB >= ((size_t)4 << 20);
  const bool want_cache = ctx->kc_max_sets.load() > 0 && space == MPVSS_HOST && whole && n > GROUP_MAX_BOX && fits_256_bits(challenge_host);
  mpvss_ctx::HostStage* hs = nullptr;
  if (prestage) {
    const size_t bytes = 3 * n * EB + t * EB;
    for (auto* x : ctx->stage_pool)
      if (!x->in_use && x->cap >= bytes) { hs = x; break; }
    if (!hs) {
      for (auto* x : ctx->stage_pool)
        if (!x->in_use) { hs = x; break; }
      if (!hs) {
        hs = new (std::nothrow) mpvss_ctx::HostStage();
        if (!hs) return fail(ctx, MPVSS_E_NOMEM, "verify_distribution: staging");
        ctx->stage_pool.push_back(hs);
      }
      HIPCHK(ctx, hipSetDevice(ctx->device));
      if (hs->pin) (void)hipHostFree(hs->pin);       // (not in use: nothing reads it)
      hs->pin = nullptr;
      hs->cap = 0;
      const hipError_t e = hipHostMalloc(&hs->pin, bytes, hipHostMallocDefault);
      if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(call staging)", e);
      hs->cap = bytes;
    }
    hs->in_use = true;
  }
  struct StageUse {
    mpvss_ctx::HostStage* h;
    ~StageUse() { if (h) h->in_use = false; }             // (the context lock is held again whenever this call returns)
  } stage_use{hs};
  const mpvss_keyset* cached = nullptr;
  if (prestage || want_cache) {
    lk.unlock();
    uint8_t kd[32], leaves[KEY_SLICES][32];
    uint8_t* in = hs ? (uint8_t*)hs->pin : nullptr;
    const size_t bytes = n * EB;
    hsc::parallel_indices((prestage ? 3u : 0u) + (want_cache ? KEY_SLICES : 0u), [&](unsigned k) {
      if (prestage && k < 3) {
        if (k == 0) memcpy(in, pubkeys, bytes);
        else if (k == 1) memcpy(in + bytes, shares, bytes);
        else { memcpy(in + 2 * bytes, responses, bytes); memcpy(in + 3 * bytes, commitments, t * EB); }
        return;
      }
      key_slice_digest(pubkeys, n, k - (prestage ? 3u : 0u), leaves[k - (prestage ? 3u : 0u)]);
    });
    if (want_cache) mpvss::sha256(&leaves[0][0], sizeof(leaves), kd);
    lk.lock();
    if (want_cache) cached = key_cache_acquire(ctx, kd, pubkeys, n);
  }
  CacheUse cache_use{ctx, cached};
  // Any number of host threads may be in here on one context (the crate goes parallel over dealers the same way,
  // participant.rs:490-500): each enqueues its box under the lock, owns the block by its number and absorbs exactly that one
  // with the lock released -- T callers keep T boxes in flight, which is what the library's own pipeline does for verify_many.
  RET_IF(wait_for_room(ctx, lk));
  RET_IF(verify_block_compute_locked(ctx, space, commitments, t, positions, cached ? nullptr : pubkeys, shares, responses, n,
                                     challenge_host, cached, 0, nullptr, hs ? (const uint8_t*)hs->pin : nullptr));
  const unsigned long long pos = ctx->own_last(1);
  RET_IF(verify_block_absorb_locked(ctx, lk, state, x_out_host, a1_out_host, a2_out_host, nullptr, &pos, true));
  return mpvss_modp_transcript_verdict(state, challenge_host, verdict, digest32_out);
}

// ---- many boxes, pipelined inside the library -------------------------------------------------------------
// The calling thread enqueues the GPU work of up to `depth` boxes ahead; `hash_threads` library threads wait for the
// boxes in enqueue order and hash them (every box has its own transcript, so the hashes of consecutive boxes run side
// by side).  Nothing of this depends on the caller's scheduler.  The pipeline owns its blocks by number (mpvss_ctx::own_last),
// so other host threads may use the one-call entry points or the block API on the same context meanwhile; a second pipeline
// call on the context waits for the first (pipe_mu: the batched X paths' workspaces belong to one run at a time).
namespace {

int ec_verify_block_absorb_locked(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, uint8_t* state, uint8_t* x_out,
                                  uint8_t* a1_out, uint8_t* a2_out, uint8_t* y_out, const unsigned long long* ticket = nullptr,
                                  bool by_position = false);     // capi_ec.inc

// issue(b, &nbox): enqueue box b as ONE block (called with the context lock held); *nbox > 1: the block is a group of the boxes
// b .. b + nbox - 1 (one transcript each).  finish(idx, state): verdict of box idx from its state.
// begin(idx, state) (optional): the state box idx's transcript starts from -- the initial one by default; a chained run (several
// engines, one box: mpvss_modp_verify_many_chained) takes it from the engine before, false = that engine failed;
// give_up(idx) (optional): box idx will not reach finish() (malformed here, or failed upstream): a chained run passes that on
struct PipeInitState { bool operator()(size_t, uint8_t* state) const { mpvss_transcript_init(state); return true; } };
struct PipeNoGiveUp { void operator()(size_t) const {} };
template <class Issue, class Finish, class Begin = PipeInitState, class GiveUp = PipeNoGiveUp>
int run_box_pipeline(mpvss_ctx* ctx, size_t count, int depth, int hash_threads, Issue issue, Finish finish, Begin begin = Begin(),
                     GiveUp give_up = GiveUp(), int max_threads = 16) {
  if (hash_threads < 1) hash_threads = 1;
  if (hash_threads > max_threads) hash_threads = max_threads;
  if (depth < 1) depth = 1;
  if (depth > (int)mpvss_ctx::NSLOT - 8) depth = (int)mpvss_ctx::NSLOT - 8;
  std::lock_guard<std::mutex> one_pipeline(ctx->pipe_mu);      // (always taken before the context lock)
  struct Hint {
    mpvss_ctx* c; bool on;
    Hint(mpvss_ctx* c_, bool on_) : c(c_), on(on_) {
      if (!on) return;
      std::lock_guard<std::mutex> lk(c->mu);
      ++c->pipelines_running;
    }
    ~Hint() {
      if (!on) return;
      std::lock_guard<std::mutex> lk(c->mu);
      --c->pipelines_running;
    }
  } hint(ctx, count > 2 && depth > 1);
  struct Ent { size_t box; unsigned pos; bool bad; bool ec; unsigned nbox; };   // one entry per enqueued block (pos: its number in the context's ring); nbox > 1: a group block of boxes box .. box+nbox-1
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    size_t issued = 0, claimed = 0, absorbed = 0;   // blocks enqueued / handed to workers / absorbed (their slots are free again)
    std::vector<Ent> order;                         // order[k]: the k-th block enqueued by this run
    bool stop = false;                              // no more boxes will be issued
    int rc = MPVSS_OK;
  } sh;
  sh.order.reserve(count + 8);
  // `depth` bounds the blocks whose GPU work is pending; a block whose GPU work is done but whose transcript is still being
  // hashed keeps its slot (the ring has NSLOT of them) without holding back the enqueueing of the next one (measured: issuing
  // on GPU completion instead of after the hash, 0.90-0.93 -> 0.97 M share verifications/s at K = 20)
  std::atomic<unsigned long long> gpu_done{0};      // this run's blocks whose GPU work has completed (seen by an absorbing thread)

  auto worker = [&]() {
    for (;;) {
      Ent ent;
      {
        std::unique_lock<std::mutex> l(sh.m);
        sh.cv.wait(l, [&] { return sh.claimed < sh.issued || sh.stop; });
        if (sh.claimed >= sh.issued) return;       // stop and nothing left
        ent = sh.order[sh.claimed++];
      }
      const unsigned long long pos = ent.pos;
      int rc = MPVSS_OK;
      if (ent.nbox > 1) {          // a group of boxes in one block: one transcript each
        std::vector<uint8_t> states((size_t)ent.nbox * MPVSS_TRANSCRIPT_STATE_BYTES);
        std::vector<char> bad(ent.nbox, 0);
        for (unsigned i = 0; i < ent.nbox; ++i) mpvss_transcript_init(states.data() + (size_t)i * MPVSS_TRANSCRIPT_STATE_BYTES);
        int prc;
        {
          std::unique_lock<std::mutex> lk(ctx->mu);
          prc = verify_block_absorb_locked(ctx, lk, states.data(), nullptr, nullptr, nullptr, nullptr, &pos, true, ent.nbox, bad.data());
        }
        {
          std::lock_guard<std::mutex> l(sh.m);
          ++sh.absorbed;
        }
        sh.cv.notify_all();
        if (prc != MPVSS_OK) rc = prc;
        for (unsigned i = 0; i < ent.nbox && rc == MPVSS_OK; ++i) {
          if (bad[i]) {
            std::lock_guard<std::mutex> lk(ctx->mu);
            (void)fail(ctx, MPVSS_E_INVALID, "negative position (the reference panics: negative exponent)");
            continue;
          }
          if (ent.box + i < count) rc = finish(ent.box + i, states.data() + (size_t)i * MPVSS_TRANSCRIPT_STATE_BYTES);
        }
        if (rc != MPVSS_OK) {
          std::lock_guard<std::mutex> l(sh.m);
          if (sh.rc == MPVSS_OK) sh.rc = rc;
        }
        sh.cv.notify_all();
        continue;
      }
      uint8_t state[MPVSS_TRANSCRIPT_STATE_BYTES];
      bool malformed = ent.bad;
      if (!begin(ent.box, state)) {              // (may wait: a chained run receives the state from the engine before)
        mpvss_transcript_init(state);
        malformed = true;
      }
      int prc;
      {
        std::unique_lock<std::mutex> lk(ctx->mu);
        prc = ent.ec ? ec_verify_block_absorb_locked(ctx, lk, state, nullptr, nullptr, nullptr, nullptr, &pos, true)
                     : verify_block_absorb_locked(ctx, lk, state, nullptr, nullptr, nullptr, nullptr, &pos, true);
      }
      {
        std::lock_guard<std::mutex> l(sh.m);
        ++sh.absorbed;
      }
      sh.cv.notify_all();
      // A box the engine rejects as malformed (an invalid curve encoding, a scalar that is not reduced, a negative
      // position in device memory) is THAT box's business: its verdict stays 0 -- the reference answers `false` to
      // structural problems, participant.rs:415-420 -- and the other boxes of the run are verified as usual
      // (mpvss_last_error still names the reason).  Only device and allocation errors end the run.
      if (prc == MPVSS_E_INVALID) malformed = true;
      else if (prc != MPVSS_OK) rc = prc;
      if (rc == MPVSS_OK && !malformed) rc = finish(ent.box, state);
      else give_up(ent.box);
      if (rc != MPVSS_OK) {
        std::lock_guard<std::mutex> l(sh.m);
        if (sh.rc == MPVSS_OK) sh.rc = rc;
      }
      sh.cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  try {      // (a thread that cannot be created must not unwind through the C ABI: fewer workers do; none is an error of the call)
    pool.reserve((size_t)hash_threads);
    for (int i = 0; i < hash_threads; ++i) pool.emplace_back(worker);
  } catch (...) {
  }
  if (pool.empty()) return fail(ctx, MPVSS_E_NOMEM, "box pipeline: no worker thread could be started");

  unsigned nbox = 1;
  for (size_t b = 0; b < count; b += nbox) {
    {
      std::unique_lock<std::mutex> l(sh.m);
      auto may_issue = [&] {
        if (sh.rc != MPVSS_OK) return true;
        const size_t pending_gpu = sh.issued - (size_t)gpu_done.load();
        return pending_gpu < (size_t)depth && sh.issued - sh.absorbed + 8 < (size_t)mpvss_ctx::NSLOT;
      };
      while (!may_issue()) sh.cv.wait_for(l, std::chrono::microseconds(200));
      if (sh.rc != MPVSS_OK) break;
    }
    int rc;
    nbox = 1;                                // boxes this call enqueued (> 1: one group block)
    unsigned head0, head1;
    bool ec = false;
    {
      std::unique_lock<std::mutex> lk(ctx->mu);
      rc = wait_for_room(ctx, lk);           // (other callers' blocks share the ring)
      head0 = ctx->head;
      if (rc == MPVSS_OK) rc = issue(b, &nbox);
      head1 = ctx->head;
      if (head1 != head0) {                  // what was enqueued is this run's, also when the box was rejected half-way
        ctx->own_last(head1 - head0, &gpu_done);
        ec = ctx->ring_slot(head0).kind == 2;
      }
    }
    if (nbox < 1) nbox = 1;
    {
      std::lock_guard<std::mutex> l(sh.m);
      const bool bad = rc == MPVSS_E_INVALID;     // malformed box: verdict 0, the run goes on (what was enqueued is still absorbed)
      if (rc == MPVSS_OK || bad) {
        for (unsigned p = head0; p != head1; ++p) sh.order.push_back(Ent{b, p, bad, ec, nbox});
        sh.issued += head1 - head0;
        rc = MPVSS_OK;
      } else if (sh.rc == MPVSS_OK) {
        sh.rc = rc;
      }
    }
    if (rc == MPVSS_OK && head1 == head0) give_up(b);      // rejected before anything was enqueued: no worker will see the box
    sh.cv.notify_all();
    if (rc != MPVSS_OK) break;
  }
  {
    std::lock_guard<std::mutex> l(sh.m);
    sh.stop = true;
  }
  sh.cv.notify_all();
  for (auto& th : pool) th.join();     // the workers drain every issued block, so no slot of this run stays busy
  return sh.rc;
}

}  // namespace

extern "C" int mpvss_modp_verify_many(mpvss_ctx* ctx, int space, const mpvss_modp_box* boxes, size_t count, int depth,
                                      int hash_threads, int* verdicts, uint8_t* digests32) {
  if (!ctx) return MPVSS_E_INVALID;
  if (count == 0) return MPVSS_OK;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!boxes || !verdicts) return fail(ctx, MPVSS_E_INVALID, "verify_many: bad argument");
  }
  for (size_t i = 0; i < count; ++i) verdicts[i] = 0;
  if (digests32) memset(digests32, 0, 32 * count);       // a malformed box has no transcript: verdict 0, digest zero
  // The key cache (mpvss_ctx_set_key_cache): a public-key array that enough large boxes of THIS call present -- same pointer, same n:
  // inside one call that is the same array -- gets its tables built once, now; those boxes then take the registered-key path
  // (a2 in 613 instead of 2 620 products), the tables are freed when the call returns.  Boxes that bring a key set of their own, small
  // boxes (they travel in groups) and boxes whose challenge does not fit 256 bits (decided per block) are left as they are.
  // Whatever goes wrong with the tables (no room in HBM, any other failure of the build) costs speed, not the call: those boxes
  // are verified the plain way.
  std::vector<const mpvss_keyset*> auto_ks(count, nullptr);
  struct AutoKeys {
    mpvss_ctx* c;
    std::vector<mpvss_keyset*> made;
    ~AutoKeys() {          // (every block of the call is absorbed: nothing reads the tables) -- the buffer stays for the next call's
      if (made.empty()) return;
      (void)mpvss_ctx_synchronize(c);
      std::lock_guard<std::mutex> lk(c->mu);
      for (mpvss_keyset* k : made) keyset_free_locked(c, k, true);
    }
  } auto_keys{ctx, {}};
  const int cache_min = [&] { std::lock_guard<std::mutex> lk(ctx->mu); return ctx->key_cache_min_boxes; }();
  if (cache_min >= 2) {
    std::vector<char> seen(count, 0);
    for (size_t b = 0; b < count; ++b) {
      const mpvss_modp_box& bx = boxes[b];
      if (seen[b] || bx.keyset || !bx.pubkeys || bx.n <= GROUP_MAX_BOX || bx.n > MAX_CHUNK) continue;
      std::vector<size_t> same;
      for (size_t k = b; k < count; ++k)
        if (!boxes[k].keyset && boxes[k].pubkeys == bx.pubkeys && boxes[k].n == bx.n) { same.push_back(k); seen[k] = 1; }
      if ((int)same.size() < cache_min) continue;
      // leave the block slots' workspaces their room: the tables may take what is free now minus a reserve of 3 GB per box in flight
      size_t free_b = 0, total_b = 0;
      const size_t table_b = bx.n * ((size_t)modp_keyset_words_per_key() * 4 + EB);
      const size_t reserve_b = ((size_t)3 << 30) * (size_t)std::max(1, std::min(depth, (int)mpvss_ctx::NSLOT));
      if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); continue; }
      free_b += [&] { std::lock_guard<std::mutex> lk(ctx->mu); return ctx->spare_table_cap; }();      // what the last call's tables left behind
      if (free_b < table_b || free_b - table_b < reserve_b) continue;
      mpvss_keyset* ks = nullptr;
      if (mpvss_modp_keyset_create(ctx, space, bx.pubkeys, bx.n, &ks) != MPVSS_OK) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> g(ctx->err_mu);
        ctx->err.clear();               // tolerated: nothing of it may show up as this call's error
        continue;
      }
      auto_keys.made.push_back(ks);
      for (size_t k : same) auto_ks[k] = ks;
    }
  }
  return run_box_pipeline(
      ctx, count, depth, hash_threads,
      [&](size_t b, unsigned* nbox) {
        const mpvss_modp_box& bx = boxes[b];
        // a run of consecutive boxes of this box's shape (small ones): ONE block for all of them
        if (bx.n > 0 && 2 * bx.n <= group_shares_max() && box_groupable(bx, bx.n, bx.t, space)) {
          const size_t cap = group_shares_max() / bx.n;
          size_t B = 1;
          while (B < cap && b + B < count && box_groupable(boxes[b + B], bx.n, bx.t, space)) ++B;
          if (B >= 2) {
            const int rc = verify_group_compute_locked(ctx, space, boxes + b, B);
            if (rc != MPVSS_OK) return rc;
            *nbox = (unsigned)B;
            return (int)MPVSS_OK;
          }
        }
        const mpvss_keyset* use_ks = bx.keyset ? bx.keyset : auto_ks[b];
        return verify_block_compute_locked(ctx, space, bx.commitments, bx.t, bx.positions, bx.pubkeys, bx.shares, bx.responses, bx.n,
                                           bx.challenge_host, use_ks, bx.keyset ? bx.key_offset : 0);
      },
      [&](size_t idx, const uint8_t* state) {
        return mpvss_modp_transcript_verdict(state, boxes[idx].challenge_host, &verdicts[idx],
                                             digests32 ? digests32 + 32 * idx : nullptr);
      });
}

// One box over SEVERAL engines (one per GPU: participants sharded): this engine holds one contiguous block of every
// box; the transcript of a box is one ordered hash, so its 128-byte running state travels engine to engine.  The same pipeline
// as mpvss_modp_verify_many -- the calling thread enqueues, library threads absorb -- with two callbacks around every box's
// absorb: state_in(user, box, state, 1) fills in the state this engine's block starts from (it may wait for the engine before;
// null: the initial state, i.e. the first engine of the chain; a non-zero return means an earlier engine failed: the box gets
// verdict 0 here and the failure is passed on), state_out(user, box, state, ok) hands the state on once the block is absorbed
// (ok = 0: this or an earlier engine failed on the box; null: the last engine).  Callbacks run on library threads, several
// boxes at a time, each box exactly once.  verdicts / digests32 are this engine's view of the final state: meaningful on the
// last engine of the chain.  wellformed_dev_out (optional): per box a device buffer of n bytes that receives the shares'
// well-formedness flags (mpvss_modp_verify_block_compute_flags) -- what the engines of a box all-gather.
extern "C" int mpvss_modp_verify_many_chained(mpvss_ctx* ctx, int space, const mpvss_modp_box* boxes, size_t count, int depth,
                                              int hash_threads, uint8_t* const* wellformed_dev_out, mpvss_chain_cb state_in,
                                              mpvss_chain_cb state_out, void* user, int* verdicts, uint8_t* digests32) {
  if (!ctx) return MPVSS_E_INVALID;
  if (count == 0) return MPVSS_OK;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!boxes || !verdicts) return fail(ctx, MPVSS_E_INVALID, "verify_many_chained: bad argument");
  }
  for (size_t i = 0; i < count; ++i) verdicts[i] = 0;
  if (digests32) memset(digests32, 0, 32 * count);
  return run_box_pipeline(
      ctx, count, depth, hash_threads,
      [&](size_t b, unsigned* nbox) {
        const mpvss_modp_box& bx = boxes[b];
        *nbox = 1;
        return verify_block_compute_locked(ctx, space, bx.commitments, bx.t, bx.positions, bx.pubkeys, bx.shares, bx.responses,
                                           bx.n, bx.challenge_host, bx.keyset, bx.key_offset,
                                           wellformed_dev_out ? wellformed_dev_out[b] : nullptr);
      },
      [&](size_t idx, const uint8_t* state) {
        if (state_out) {
          uint8_t copy[MPVSS_TRANSCRIPT_STATE_BYTES];
          memcpy(copy, state, sizeof(copy));
          (void)state_out(user, idx, copy, 1);
        }
        return mpvss_modp_transcript_verdict(state, boxes[idx].challenge_host, &verdicts[idx], digests32 ? digests32 + 32 * idx : nullptr);
      },
      [&](size_t idx, uint8_t* state) {
        if (!state_in) { mpvss_transcript_init(state); return true; }
        return state_in(user, idx, state, 1) == 0;
      },
      [&](size_t idx) {
        if (state_out) {
          uint8_t zero[MPVSS_TRANSCRIPT_STATE_BYTES];
          memset(zero, 0, sizeof(zero));
          (void)state_out(user, idx, zero, 0);
        }
      },
      32);
}

extern "C" int mpvss_blocks_in_flight(mpvss_ctx* ctx, int* in_flight_out, int* gpu_pending_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int busy = 0, pending = 0;
  for (int i = 0; i < mpvss_ctx::NSLOT; ++i) {
    const mpvss_ctx::BlockSlot& sl = ctx->slot[i];
    if (!sl.busy) continue;
    ++busy;
    if (sl.n != 0 && sl.done && hipEventQuery(sl.done) == hipErrorNotReady) ++pending;
  }
  if (in_flight_out) *in_flight_out = busy;
  if (gpu_pending_out) *gpu_pending_out = pending;
  return MPVSS_OK;
}

extern "C" int mpvss_pipeline_stats_get(mpvss_ctx* ctx, mpvss_pipeline_stats* out, int reset) {
  if (!ctx || !out) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  out->enqueue_ms = ctx->pstats.enqueue_ms;
  out->wait_ms = ctx->pstats.wait_ms;
  out->hash_ms = ctx->pstats.hash_ms;
  for (int k = 0; k < 4; ++k) {
    out->kernel_ms[k] = ctx->pstats.kernel_ms[k];
    out->kernel_launches[k] = ctx->pstats.kernel_launches[k];
  }
  out->blocks = ctx->pstats.blocks;
  if (reset) ctx->pstats = mpvss_ctx::PipeStats();
  return MPVSS_OK;
}

extern "C" int mpvss_sha256_uses_shani(void) { return mpvss::sha256_uses_shani() ? 1 : 0; }

// What this device sustains of the kernels' basic instruction right now (bench.py holds its issue-slot accounting against it):
// 4 waves per SIMD on every CU issue the instruction back to back for about target_ms.
extern "C" int mpvss_issue_probe(mpvss_ctx* ctx, int kind, double target_ms, double* insts_per_s_out, double* shader_clock_ghz_out,
                                 double* ms_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (kind < 0 || kind > 5 || !insts_per_s_out || target_ms <= 0 || target_ms > 2000)
    return fail(ctx, MPVSS_E_INVALID, "issue_probe: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  const int blocks = prop.multiProcessorCount * 4;           // 256-thread blocks: one wave per SIMD each
  const size_t waves = (size_t)blocks * 4;
  DevBuf out, st;
  RET_IF(ensure(ctx, out, (size_t)blocks * 256 * 4));
  int rc = ensure(ctx, st, waves * 16);
  if (rc != 0) { (void)hipFree(out.p); return rc; }
  struct Free { DevBuf &a, &b; ~Free() { (void)hipFree(a.p); (void)hipFree(b.p); } } fr{out, st};
  hipEvent_t e0, e1;
  HIPCHK(ctx, hipEventCreate(&e0));
  HIPCHK(ctx, hipEventCreate(&e1));
  struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } ev{e0, e1};
  float ms = 0;
  int iters = 2000;
  for (int pass = 0; pass < 2; ++pass) {                      // a short calibration launch, then the measurement
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    LAUNCHCHK(ctx, issue_probe_launch((uint32_t*)out.p, (unsigned long long*)st.p, blocks, iters, kind, ctx->stream));
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) {
      const double scale = target_ms / (ms > 0.01f ? ms : 0.01f);
      iters = (int)std::min(50000000.0, std::max(2000.0, iters * scale));
    }
  }
  std::vector<unsigned long long> hs(waves * 2);
  HIPCHK(ctx, hipMemcpy(hs.data(), st.p, hs.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (size_t w = 0; w < waves; ++w) { cyc += (double)hs[2 * w]; wall += (double)hs[2 * w + 1]; }
  *insts_per_s_out = (double)waves * iters * 64.0 / (ms * 1e-3);
  if (shader_clock_ghz_out) *shader_clock_ghz_out = wall > 0 ? cyc / wall * 0.1 : 0.0;
  if (ms_out) *ms_out = ms;
  return MPVSS_OK;
}

// ---- verify_share, batched ----------------------------------------------------------------------------
// n independent share-box proofs (participant.rs:361-386 -> dleq.rs:275-302).  Everything runs on the device: the two
// double exponentiations, then K7 (verdict_kernels.hip): one lane per share hashes the four framed elements, applies
// hash_to_scalar and compares with c_i -- a verdict byte per share in HBM.  Batches go through the same block slots
// as the boxes of verify_distribution_shares (own workspace and stream pair each), so several are in flight at once.
namespace {

int verify_shares_compute_locked(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s, const uint8_t* y,
                                 const uint8_t* c, const uint8_t* r, size_t n, uint8_t* verdicts_dev_out) {
  if (n > 0 && (!pk || !s || !y || !c || !r || n > 0x7fffffff)) return fail(ctx, MPVSS_E_INVALID, "verify_shares: bad argument");
  mpvss_ctx::BlockSlot& sl = ctx->head_slot();
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "verify_shares: every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
  const auto t_enq0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
  sl.n = n;
  sl.kind = 1;
  sl.check_positions = false;
  sl.fd_used = false;
  sl.fd_chunks = 0;
  sl.enqueue_ms = 0;
  if (n == 0) {
    sl.busy = true;
    ctx->commit_head(sl);
    return MPVSS_OK;
  }
  RET_IF(work_init(ctx, sl.work, nullptr));
  struct Restore {       // also: an early (error) return leaves nothing of this block running on the slot's streams
    mpvss_ctx* c;
    hipStream_t a, b;
    mpvss_ctx::BlockSlot* sl;
    ~Restore() {
      if (!sl->busy) {
        if (sl->work.sa) (void)hipStreamSynchronize(sl->work.sa);
        if (sl->work.sb) (void)hipStreamSynchronize(sl->work.sb);
      }
      c->sp = &c->main_spans; c->w = &c->work0; c->stream = a; c->stream_b = b;
    }
  } restore{ctx, ctx->stream, ctx->stream_b, &sl};
  ctx->w = &sl.work;
  ctx->stream = sl.work.sa;
  ctx->stream_b = sl.work.sb;
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  // pinned staging: the verdict bytes and, for host callers, a copy of the five input arrays
  const size_t need = n + (space == MPVSS_HOST ? 5 * n * EB : 0);
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  uint8_t* hv = (uint8_t*)sl.pin;
  if (space == MPVSS_HOST) {
    uint8_t* in = hv + n;
    const uint8_t* src[5] = {pk, s, y, c, r};
    for (int k = 0; k < 5; ++k) memcpy(in + (size_t)k * n * EB, src[k], n * EB);
    pk = in; s = in + n * EB; y = in + 2 * n * EB; c = in + 3 * n * EB; r = in + 4 * n * EB;
  }
  const uint32_t* cG;
  RET_IF(comb_table(ctx, 1, &cG, n));
  RET_IF(ensure(ctx, ctx->w->verd, n));
  uint8_t* dv = (uint8_t*)ctx->w->verd.p;
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *dpk, *ds, *dy, *dc, *dr;
    RET_IF(stage_in(ctx, space, pk + off * EB, cnt * EB, ctx->w->in_a, &dpk));
    RET_IF(stage_in(ctx, space, s + off * EB, cnt * EB, ctx->w->in_b, &ds));
    RET_IF(stage_in(ctx, space, y + off * EB, cnt * EB, ctx->w->in_c, &dy));
    RET_IF(stage_in(ctx, space, r + off * EB, cnt * EB, ctx->w->in_d, &dr));
    RET_IF(stage_in(ctx, space, c + off * EB, cnt * EB, ctx->w->in_e, &dc));
    RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out2, cnt * EB));
    uint8_t* da1 = (uint8_t*)ctx->w->out1.p;
    uint8_t* da2 = (uint8_t*)ctx->w->out2.p;
    // a1 = G^r * pk^c ; a2 = S^r * Y^c                       dleq.rs:66-84 via participant.rs:376-385
    // Only the low 256 bits of c_i enter the exponentiations: a challenge >= 2^256 can never equal the 256-bit hash,
    // and K7 gives such a share the verdict 0 whatever a1, a2 are.
    if (ctx->w->sb && cnt <= ROW_MAX_NUMBERS) {
      // a small batch is the latency of its longest chain: a2 (2 620 operations) on the slot's second stream beside a1 (830)
      struct Swap {
        mpvss_ctx* c; hipStream_t a;
        Swap(mpvss_ctx* c_, hipStream_t s) : c(c_), a(c_->stream) { c->stream = s; }
        ~Swap() { c->stream = a; }
      };
      HIPCHK(ctx, hipEventRecord(ctx->w->ev_fork, ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->w->sb, ctx->w->ev_fork, 0));
      {
        Swap sw(ctx, ctx->w->sb);
        RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)ds, (const uint8_t*)dy, (const uint8_t*)dr, (const uint8_t*)dc, EB, 64, cnt, da2));
        HIPCHK(ctx, hipEventRecord(ctx->w->ev_a2, ctx->stream));
      }
      RET_IF(dleq_side(ctx, nullptr, nullptr, (const uint8_t*)dpk, (const uint8_t*)dr, (const uint8_t*)dc, EB, 64, cnt, da1, cG, &ctx->w->tab3));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_a2, 0));
    } else {
      RET_IF(dleq_side(ctx, nullptr, nullptr, (const uint8_t*)dpk, (const uint8_t*)dr, (const uint8_t*)dc, EB, 64, cnt, da1, cG));
      RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)ds, (const uint8_t*)dy, (const uint8_t*)dr, (const uint8_t*)dc, EB, 64,
                       cnt, da2));
    }
    TIMED_LAUNCH(ctx, 0, verdict_launch_modp((const uint8_t*)dpk, (const uint8_t*)dy, da1, da2, (const uint8_t*)dc, (int)cnt,
                                             dv + off, ctx->stream));
    if (off + MAX_CHUNK < n) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // device buffers are reused
  }
  HIPCHK(ctx, hipMemcpyAsync(hv, dv, n, hipMemcpyDeviceToHost, ctx->stream));
  if (verdicts_dev_out) HIPCHK(ctx, hipMemcpyAsync(verdicts_dev_out, dv, n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  sl.busy = true;
  sl.enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  ctx->commit_head(sl);
  return MPVSS_OK;
}

int verify_shares_absorb_locked(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, uint8_t* verdicts_host,
                                const unsigned long long* own_pos = nullptr) {
  mpvss_ctx::BlockSlot* slp = select_block(ctx, own_pos, own_pos != nullptr, 1, 1, "verify_shares_absorb");
  if (!slp) return MPVSS_E_INVALID;
  mpvss_ctx::BlockSlot& sl = *slp;
  const size_t n = sl.n;
  if (n == 0) {
    ctx->release(sl);
    if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
    return MPVSS_OK;
  }
  sl.absorbing = true;
  {
    const hipError_t e_dev = hipSetDevice(ctx->device);
    if (e_dev != hipSuccess) {               // give the slot back: the block is lost, the ring is not
      ctx->release(sl);
      sl.absorbing = false;
      if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
      return fail(ctx, MPVSS_E_DEVICE, "absorb: hipSetDevice", e_dev);
    }
  }
  lk.unlock();
  const auto t_w0 = std::chrono::steady_clock::now();
  const hipError_t e = hipEventSynchronize(sl.done);
  const auto t_w1 = std::chrono::steady_clock::now();
  if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
  if (e == hipSuccess && verdicts_host) memcpy(verdicts_host, sl.pin, n);
  lk.lock();
  ctx->release(sl);
  sl.absorbing = false;
  if (e != hipSuccess) return fail(ctx, MPVSS_E_DEVICE, "verify_shares_absorb: hipEventSynchronize", e);
  RET_IF(spans_sum(ctx, sl.spans, ctx->kernel_ms));
  ctx->pstats.enqueue_ms += sl.enqueue_ms;
  ctx->pstats.wait_ms += std::chrono::duration<double, std::milli>(t_w1 - t_w0).count();
  for (int k = 0; k < 4; ++k) {
    ctx->pstats.kernel_ms[k] += ctx->kernel_ms[k];
    ctx->pstats.kernel_launches[k] += (unsigned long long)ctx->kernel_launches[k];
  }
  ++ctx->pstats.blocks;
  return MPVSS_OK;
}

}  // namespace

extern "C" int mpvss_modp_verify_shares_compute(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s,
                                                const uint8_t* y, const uint8_t* c, const uint8_t* r, size_t n,
                                                uint8_t* verdicts_dev_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_shares_compute_locked(ctx, space, pk, s, y, c, r, n, verdicts_dev_out);
}

extern "C" int mpvss_modp_verify_shares_absorb(mpvss_ctx* ctx, uint8_t* verdicts_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  return verify_shares_absorb_locked(ctx, lk, verdicts_host);
}

extern "C" int mpvss_modp_verify_shares(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s,
                                        const uint8_t* y, const uint8_t* c, const uint8_t* r, size_t n,
                                        uint8_t* verdicts_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!pk || !s || !y || !c || !r || !verdicts_host) return fail(ctx, MPVSS_E_INVALID, "verify_shares: bad argument");
  RET_IF(wait_for_room(ctx, lk));
  RET_IF(verify_shares_compute_locked(ctx, space, pk, s, y, c, r, n, nullptr));
  const unsigned long long pos = ctx->own_last(1);
  return verify_shares_absorb_locked(ctx, lk, verdicts_host, &pos);
}

// ---- distribute_secret, group part ---------------------------------------------------------------------
// Dealer side of src/participant.rs:160-286 in the same compute / absorb form as the verifier's blocks (same slots,
// same pinned staging layout X | Y | a1 | a2, same ordered hash in absorb):
//   stream B (low priority):  64-entry tables of y_i, Y_i = y_i^p_i (:219), a2_i = y_i^w_i (dleq.rs:213-216)
//   stream A (high priority): X_i -- from the commitments exactly as the verifier does (:207-215), or, when the caller
//                             passes NO commitments, as g^p_i through the fixed-base comb (the same element whenever
//                             the commitments are g^a_j of the polynomial behind p_i: 127 products instead of a
//                             forward-difference chain);  a1_i = g^w_i (dleq.rs:207-211) through the comb.
namespace {

// the dealer's polynomial for a block whose P(i) is computed on the device, ahead of the group work and in the same stream
struct DealPoly {
  const uint8_t* coeffs_host;     // t x 256 bytes (null when limbs_dev is given)
  size_t t;
  const int64_t* positions_dev;   // n positions (validated when the block is absorbed)
  uint8_t* p_dev_out;             // n x 256 bytes: P(i) mod (q-1), kept by the caller for the responses
  // a box dealt in several blocks prepares its polynomial once: t x 72 limbs of a'_j in HBM and the two parities (modq_poly_limbs)
  const uint32_t* limbs_dev = nullptr;
  int par_even = 0, par_odd = 0;
};
void modq_poly_limbs(const uint8_t* coeffs_host, size_t t, uint32_t* limbs, int* par_even, int* par_odd);   // capi_scalar.inc
int modq_consts(mpvss_ctx* ctx);

int distribute_block_compute_locked(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t, const int64_t* positions,
                                    const uint8_t* pubkeys, const uint8_t* p_values, const uint8_t* witnesses, size_t n,
                                    uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out, uint8_t* a2_dev_out,
                                    const DealPoly* poly = nullptr, const mpvss_keyset* ks = nullptr, size_t key_offset = 0,
                                    hipEvent_t run_after = nullptr) {
  // run_after: this block's GPU work starts when that event has fired (the `done` event of the block before it in a box that a lone
  // dealer cuts into blocks: they run one after the other, each filling the chip, and block k is hashed while block k + 1 computes)
  // ks: the participants' keys are REGISTERED (a key set of the caller's, or the context's cross-call cache): shares
  // key_offset .. key_offset + n of the set; the keys come from its device copy and Y = y^p, a2 = y^w from its tables
  if (ks && (key_offset > ks->n || n > ks->n - key_offset))
    return fail(ctx, MPVSS_E_INVALID, "distribute: shares outside the registered key set");
  int key_space = space;
  if (ks) {
    pubkeys = (const uint8_t*)ks->keys.p + key_offset * EB;
    key_space = MPVSS_DEVICE;
  }
  if (poly) {
    if (space != MPVSS_DEVICE || commitments || (!poly->coeffs_host && !poly->limbs_dev) || !poly->positions_dev || !poly->p_dev_out ||
        poly->t == 0 || poly->t > 0x7fffffff || n > MAX_CHUNK)
      return fail(ctx, MPVSS_E_INVALID, "deal: bad argument (device buffers, t >= 1, one chunk of shares)");
    p_values = poly->p_dev_out;
  }
  if (n > 0 && (!pubkeys || !p_values || !witnesses || n > 0x7fffffff || (commitments && (!positions || t == 0 || t > 0x7fffffff))))
    return fail(ctx, MPVSS_E_INVALID, "distribute: bad argument");
  // (threshold > n is the whole box's business -- mpvss_modp_distribute checks it; a block of a box may be smaller)
  mpvss_ctx::BlockSlot& sl = ctx->head_slot();
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "distribute: every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
  const auto t_enq0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
  sl.n = n;
  sl.kind = 0;                 // absorbed like a verifier's block: same staging layout, same hash
  sl.nbox = 1;
  sl.check_positions = false;
  sl.fd_used = false;
  sl.fd_chunks = 0;
  sl.enqueue_ms = 0;
  if (n == 0) {
    sl.busy = true;
    ctx->commit_head(sl);
    return MPVSS_OK;
  }
  if (commitments && space == MPVSS_HOST) RET_IF(check_positions_host(ctx, positions, n));
  RET_IF(work_init(ctx, sl.work, nullptr));
  struct Restore {       // also: an early (error) return leaves nothing of this block running on the slot's streams
    mpvss_ctx* c;
    hipStream_t a, b;
    mpvss_ctx::BlockSlot* sl;
    ~Restore() {
      if (!sl->busy) {
        if (sl->work.sa) (void)hipStreamSynchronize(sl->work.sa);
        if (sl->work.sb) (void)hipStreamSynchronize(sl->work.sb);
      }
      c->sp = &c->main_spans; c->w = &c->work0; c->stream = a; c->stream_b = b;
    }
  } restore{ctx, ctx->stream, ctx->stream_b, &sl};
  ctx->w = &sl.work;
  sl.work.fd_used = false;
  ctx->stream = sl.work.sa;
  ctx->stream_b = sl.work.sb;
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  constexpr size_t FLAGS = 64;
  const size_t out_bytes = n * EB * 4 + n * 8 + FLAGS * 4;
  const size_t need = out_bytes + (space == MPVSS_HOST ? 3 * n * EB + (commitments ? t * EB : 0) : 0) +
                      (poly && !poly->limbs_dev ? poly->t * MODP_L * 4 : 0);
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  uint8_t* hX = (uint8_t*)sl.pin;
  uint8_t* hY = hX + n * EB;
  uint8_t* h1 = hY + n * EB;
  uint8_t* h2 = h1 + n * EB;
  int64_t* hpos = (int64_t*)(h2 + n * EB);
  int* hflags = (int*)((uint8_t*)sl.pin + n * EB * 4 + n * 8);
  if (run_after) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, run_after, 0));      // (the second stream forks from this one below)
  if (poly) {
    // P(i) mod (q-1) first, on this block's own stream: the group work below reads it in stream order (k_modq_poly_eval)
    RET_IF(modq_consts(ctx));
    if (poly->limbs_dev) {
      LAUNCHCHK(ctx, modq_launch_poly_eval(poly->limbs_dev, (int)poly->t, poly->positions_dev, (int)n, poly->par_even, poly->par_odd,
                                           poly->p_dev_out, ctx->consts_q, ctx->stream));
    } else {
      uint32_t* hl = (uint32_t*)((uint8_t*)sl.pin + out_bytes);
      const size_t lb = poly->t * MODP_L * 4;
      int par_even, par_odd;
      modq_poly_limbs(poly->coeffs_host, poly->t, hl, &par_even, &par_odd);
      RET_IF(ensure(ctx, ctx->w->cm, lb));
      HIPCHK(ctx, hipMemcpyAsync(ctx->w->cm.p, hl, lb, hipMemcpyHostToDevice, ctx->stream));
      LAUNCHCHK(ctx, modq_launch_poly_eval((const uint32_t*)ctx->w->cm.p, (int)poly->t, poly->positions_dev, (int)n, par_even, par_odd,
                                           poly->p_dev_out, ctx->consts_q, ctx->stream));
      // the staged coefficients are the dealer's secret: zero the device copy and, in stream order, the pinned one with it
      HIPCHK(ctx, hipMemsetAsync(ctx->w->cm.p, 0, lb, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(hl, ctx->w->cm.p, lb, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(hpos, poly->positions_dev, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    sl.check_positions = true;          // a negative position fails the block when it is absorbed
  }
  if (space == MPVSS_HOST) {
    uint8_t* in = (uint8_t*)sl.pin + out_bytes;
    if (!ks) { memcpy(in, pubkeys, n * EB); pubkeys = in; }
    memcpy(in + n * EB, p_values, n * EB); p_values = in + n * EB;
    memcpy(in + 2 * n * EB, witnesses, n * EB); witnesses = in + 2 * n * EB;
    if (commitments) { memcpy(in + 3 * n * EB, commitments, t * EB); commitments = in + 3 * n * EB; memcpy(hpos, positions, n * 8); }
  }
  if (commitments) RET_IF(stage_commitments(ctx, space, commitments, t));
  const uint32_t* cg;
  RET_IF(comb_table(ctx, 0, &cg, n));
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *dy, *dp, *dw;
    RET_IF(stage_in(ctx, key_space, pubkeys + off * EB, cnt * EB, ctx->w->in_a, &dy));
    RET_IF(stage_in(ctx, space, p_values + off * EB, cnt * EB, ctx->w->in_b, &dp));
    RET_IF(stage_in(ctx, space, witnesses + off * EB, cnt * EB, ctx->w->in_c, &dw));
    RET_IF(ensure(ctx, ctx->w->xbe, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->in_d, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out2, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->tab1, cnt * 4 * TABW * 4));
    uint8_t *dX = (uint8_t*)ctx->w->xbe.p, *dY = (uint8_t*)ctx->w->in_d.p, *da1 = (uint8_t*)ctx->w->out1.p, *da2 = (uint8_t*)ctx->w->out2.p;
    struct Swap {
      mpvss_ctx* c; hipStream_t a;
      Swap(mpvss_ctx* c_, hipStream_t s) : c(c_), a(c_->stream) { c->stream = s; }
      ~Swap() { c->stream = a; }
    };
    HIPCHK(ctx, hipEventRecord(ctx->w->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->w->sb, ctx->w->ev_fork, 0));
    {
      Swap sw(ctx, ctx->w->sb);
      // y-tables once, two exponent sets: Y = y^p (participant.rs:219), a2 = y^w (dleq.rs:213-216)
      uint32_t* ty = (uint32_t*)ctx->w->tab1.p;
      if (ks && (pair_mask() & 1) && cnt >= 64) {
        // registered keys: both powers from the per-key tables, 2 x (252 squarings + 296 products) instead of 2 045 + 820
        const uint32_t* kt = (const uint32_t*)ks->table.p + (key_offset + off) * modp_keyset_words_per_key();
        TIMED_LAUNCH(ctx, 3, modp_launch_keyset_twin_exp_pair(kt, (const uint8_t*)dp, (const uint8_t*)dw, (int)cnt, dY, da2, ctx->consts,
                                                              ctx->pair_tables, ctx->stream));
      } else if (cnt > ROW_MAX_NUMBERS) {
        // same base, two exponents: right-to-left buckets share the 2 045 squarings (tab1 holds buckets + occupancy)
        // (smaller blocks: the row-layout launch below, the latency of one chain instead of the bucket kernel's 27 ms)
        const size_t bw = modp_twin_exp_bucket_words();
        RET_IF(ensure(ctx, ctx->w->tab1, cnt * (bw + MODP_TWIN_EXTRA_WORDS) * 4 + MODP_TWIN_SLACK_BYTES));
        uint32_t* bk = (uint32_t*)ctx->w->tab1.p;
        TIMED_LAUNCH(ctx, 3, launch_twin_exp(ctx, (const uint8_t*)dy, (const uint8_t*)dp, (const uint8_t*)dw, cnt, bk, bk + cnt * bw, dY, da2));
      } else {
        TIMED_LAUNCH(ctx, 2, modp_launch_build_table((const uint8_t*)dy, (int)cnt, ty, ctx->consts, ctx->stream));
        // (a small box: both powers in ONE row-layout launch, side by side instead of one after the other)
        RET_IF(dual_exp_any(ctx, ty, TABW, ty, TABW, (const uint8_t*)dp, (const uint8_t*)dp, EB, 0, cnt, dY, (const uint8_t*)dw, (const uint8_t*)dw, da2));
      }
      HIPCHK(ctx, hipEventRecord(ctx->w->ev_a2, ctx->stream));
    }
    // X = g^P(i) and a1 = g^w through the wide comb in ONE pair-layout launch (122 instead of 191 issue slots per product): 12 dealers
    // to registered keys 1.75 -> 2.00 M shares dealt/s; beside the bucket kernels 1.076 -> 1.110 M (profiles/r06_dealer_comb_ab.txt)
    const bool pair_combs = (pair_mask() & 1) && cnt >= 64 && !commitments && comb_bits_of(ctx, cg) == 16;
    if (commitments) {
      const int64_t* dpos;
      const void* d;
      if (space == MPVSS_HOST) {
        RET_IF(stage_in(ctx, space, hpos + off, cnt * 8, ctx->w->pos, &d));
        dpos = (const int64_t*)d;
      } else {
        dpos = positions + off;
        HIPCHK(ctx, hipMemcpyAsync(hpos + off, dpos, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
        sl.check_positions = true;
      }
      sl.work.fd_used = false;
      RET_IF(eval_x(ctx, t, dpos, space == MPVSS_HOST ? hpos + off : nullptr, cnt, dX));          // :207-215
      if (sl.work.fd_used) {
        sl.fd_used = true;
        if (sl.fd_chunks < FLAGS)
          HIPCHK(ctx, hipMemcpyAsync(hflags + sl.fd_chunks, sl.work.fd_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        ++sl.fd_chunks;
      }
    } else if (!pair_combs) {
      // X_i = g^P(i): the dealer knows the polynomial
      TIMED_LAUNCH(ctx, 0, modp_launch_comb_dual_exp(cg, cg, 0, (const uint8_t*)dp, (const uint8_t*)dp, EB, 0, (int)cnt, dX,
                                                     comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
    }
    // a1 = g^w (dleq.rs:207-211)
    if (pair_combs)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb16_twin_exp_pair(cg, (const uint8_t*)dp, (const uint8_t*)dw, (int)cnt, dX, da1, ctx->consts,
                                                            ctx->pair_tables, ctx->stream));
    else
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cg, cg, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)cnt, da1,
                                                     comb_bits_of(ctx, cg), ctx->consts, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_a2, 0));
    HIPCHK(ctx, hipMemcpyAsync(hX + off * EB, dX, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hY + off * EB, dY, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h1 + off * EB, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h2 + off * EB, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    if (space == MPVSS_DEVICE) {      // results stay in HBM for the caller as well
      if (x_dev_out) HIPCHK(ctx, hipMemcpyAsync(x_dev_out + off * EB, dX, cnt * EB, hipMemcpyDeviceToDevice, ctx->stream));
      if (y_dev_out) HIPCHK(ctx, hipMemcpyAsync(y_dev_out + off * EB, dY, cnt * EB, hipMemcpyDeviceToDevice, ctx->stream));
      if (a1_dev_out) HIPCHK(ctx, hipMemcpyAsync(a1_dev_out + off * EB, da1, cnt * EB, hipMemcpyDeviceToDevice, ctx->stream));
      if (a2_dev_out) HIPCHK(ctx, hipMemcpyAsync(a2_dev_out + off * EB, da2, cnt * EB, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (off + MAX_CHUNK < n) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // device buffers are reused
      HIPCHK(ctx, hipStreamSynchronize(ctx->w->sb));
    }
  }
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  sl.busy = true;
  sl.enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  ctx->commit_head(sl);
  return MPVSS_OK;
}

}  // namespace

extern "C" int mpvss_modp_distribute_compute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                             const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                                             const uint8_t* witnesses, size_t n, uint8_t* x_dev_out, uint8_t* y_dev_out,
                                             uint8_t* a1_dev_out, uint8_t* a2_dev_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return distribute_block_compute_locked(ctx, space, commitments, t, positions, pubkeys, p_values, witnesses, n, x_dev_out,
                                         y_dev_out, a1_dev_out, a2_dev_out);
}

extern "C" int mpvss_modp_deal_compute(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                                       const uint8_t* pubkeys_dev, const uint8_t* witnesses_dev, size_t n, uint8_t* p_dev_out,
                                       uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out, uint8_t* a2_dev_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return fail(ctx, MPVSS_E_INVALID, "deal_compute: no shares");
  const DealPoly poly{coeffs_host, t, positions_dev, p_dev_out};
  return distribute_block_compute_locked(ctx, MPVSS_DEVICE, nullptr, 0, nullptr, pubkeys_dev, p_dev_out, witnesses_dev, n, x_dev_out,
                                         y_dev_out, a1_dev_out, a2_dev_out, &poly);
}

extern "C" int mpvss_modp_deal_compute_keyset(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                                              const mpvss_keyset* keyset, size_t key_offset, const uint8_t* witnesses_dev, size_t n,
                                              uint8_t* p_dev_out, uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out,
                                              uint8_t* a2_dev_out) {
  if (!ctx || !keyset) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return fail(ctx, MPVSS_E_INVALID, "deal_compute_keyset: no shares");
  const DealPoly poly{coeffs_host, t, positions_dev, p_dev_out};
  return distribute_block_compute_locked(ctx, MPVSS_DEVICE, nullptr, 0, nullptr, nullptr, p_dev_out, witnesses_dev, n, x_dev_out,
                                         y_dev_out, a1_dev_out, a2_dev_out, &poly, keyset, key_offset);
}

extern "C" int mpvss_modp_distribute_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host, uint8_t* y_out_host,
                                            uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  return verify_block_absorb_locked(ctx, lk, state, x_out_host, a1_out_host, a2_out_host, y_out_host);
}

extern "C" int mpvss_modp_distribute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                     const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                                     const uint8_t* witnesses, size_t n, uint8_t* x_out, uint8_t* y_out,
                                     uint8_t* a1_out, uint8_t* a2_out, uint8_t* digest32_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  if (n > 0 && (!commitments || !positions || !pubkeys || !p_values || !witnesses || !x_out || !y_out || !a1_out ||
                !a2_out || t == 0 || t > 0x7fffffff))
    return fail(ctx, MPVSS_E_INVALID, "distribute: bad argument");
  if (t > n) return fail(ctx, MPVSS_E_INVALID, "distribute: threshold > number of public keys (participant.rs:166)");
  uint8_t state[MPVSS_TRANSCRIPT_STATE_BYTES];
  mpvss_transcript_init(state);
  const bool dev = space == MPVSS_DEVICE;
  // (cross-call key cache, host callers: dealers to the same long-lived participants find the keys' tables built)
  const mpvss_keyset* cached = (!dev && n > 0 && n <= MAX_CHUNK) ? dealer_key_cache(ctx, lk, pubkeys, n) : nullptr;
  CacheUse cache_use{ctx, cached};
  RET_IF(wait_for_room(ctx, lk));
  RET_IF(distribute_block_compute_locked(ctx, space, commitments, t, positions, pubkeys, p_values, witnesses, n, dev ? x_out : nullptr,
                                         dev ? y_out : nullptr, dev ? a1_out : nullptr, dev ? a2_out : nullptr, nullptr, cached, 0));
  const unsigned long long pos = ctx->own_last(1);
  RET_IF(verify_block_absorb_locked(ctx, lk, state, dev ? nullptr : x_out, dev ? nullptr : a1_out, dev ? nullptr : a2_out,
                                    dev ? nullptr : y_out, &pos, true));
  if (digest32_out) {
    mpvss::Sha256 tr;
    memcpy(&tr, state, sizeof(tr));
    tr.final(digest32_out);                       // participant.rs:251: the dealer's transcript digest
  }
  return MPVSS_OK;
}

// ---- extract_secret_share, batched ---------------------------------------------------
// n participants decrypt their share and prove it at once, src/participant.rs:294-353:
//   S_i = Y_i^(1/x_i)            (:310-314; the inverse mod q-1 is host work, util.rs:33-41, and an input here)
//   a1_i = G^w_i, a2_i = S_i^w_i (dleq.rs:207-216)
//   c_i = hash_to_scalar(SHA256(framed(pk_i) framed(Y_i) framed(a1_i) framed(a2_i)))   (:329-343)
// The response r_i = w_i - x_i*c_i (dleq.rs:42-50) is scalar-field work and stays with the host.
extern "C" int mpvss_modp_extract_shares(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* y,
                                         const uint8_t* xinv, const uint8_t* w, size_t n, uint8_t* s_out,
                                         uint8_t* c_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!pk || !y || !xinv || !w || !s_out || !c_out_host) return fail(ctx, MPVSS_E_INVALID, "extract_shares: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  const uint32_t* cG;
  RET_IF(comb_table(ctx, 1, &cG, n));
  std::vector<uint8_t> hpk, hy;
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *dy, *dxi, *dw;
    RET_IF(stage_in(ctx, space, y + off * EB, cnt * EB, ctx->w->in_a, &dy));
    RET_IF(stage_in(ctx, space, xinv + off * EB, cnt * EB, ctx->w->in_b, &dxi));
    RET_IF(stage_in(ctx, space, w + off * EB, cnt * EB, ctx->w->in_c, &dw));
    uint8_t* dS = s_out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->w->xbe, cnt * EB));
      dS = (uint8_t*)ctx->w->xbe.p;
    }
    RET_IF(ensure(ctx, ctx->w->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->w->out2, cnt * EB));
    uint8_t* da1 = (uint8_t*)ctx->w->out1.p;
    uint8_t* da2 = (uint8_t*)ctx->w->out2.p;
    // a2 = S^w = Y^(w/x): the same base as S = Y^(1/x), so both come out of ONE chain of squarings (the bucket kernels of
    // the dealer) once the host has the second exponent  e2 = w * (1/x) mod (q-1)  -- exact for every Y that is a unit
    // mod q; a Y that is 0 mod q (not a group element) would make S = 0 and the reduced exponent matter, so such a
    // batch takes the two dependent exponentiations.  Host buffers.  Large batches: the dealer's bucket kernels; up to ROW_MAX_NUMBERS
    // shares (the reference calls extract_secret_share for ONE participant at a time): both powers side by side in one row-layout
    // launch over Y's window table -- the latency of one chain instead of two.
    bool shared = space == MPVSS_HOST;
    if (shared) {
      uint8_t qb[EB];
      modq_modulus_bytes(qb);
      static const uint8_t zero[EB] = {0};
      const uint8_t* hy = y + off * EB;
      for (size_t i = 0; i < cnt && shared; ++i)
        shared = memcmp(hy + i * EB, zero, EB) != 0 && memcmp(hy + i * EB, qb, EB) != 0;
    }
    bool a1_done = false;
    if (shared) {
      // e2 = w / x mod (q-1) on the device (the scalar ring's product kernel): no host threads, whatever the host has of them
      RET_IF(modq_consts(ctx));
      RET_IF(ensure(ctx, ctx->w->in_d, cnt * EB));
      uint8_t* de2 = (uint8_t*)ctx->w->in_d.p;
      LAUNCHCHK(ctx, modq_launch_mul((const uint8_t*)dw, (const uint8_t*)dxi, (int)cnt, de2, ctx->consts_q, ctx->stream));
      if (cnt <= ROW_MAX_NUMBERS && ctx->stream_b) {
        // (a1 = G^w, 511 comb products, on the second stream beside the two powers)
        HIPCHK(ctx, hipEventRecord(ctx->w->ev_fork, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_b, ctx->w->ev_fork, 0));
        {
          hipStream_t main_stream = ctx->stream;
          ctx->stream = ctx->stream_b;
          const int rc1 = [&]() -> int {
            TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cG, cG, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)cnt, da1,
                                                           comb_bits_of(ctx, cG), ctx->consts, ctx->stream));
            HIPCHK(ctx, hipEventRecord(ctx->w->ev_a2, ctx->stream));
            return 0;
          }();
          ctx->stream = main_stream;
          RET_IF(rc1);
        }
        const uint32_t* ty;
        RET_IF(number_tables(ctx, (const uint8_t*)dy, cnt, ctx->w->tab1, &ty));
        RET_IF(dual_exp_any(ctx, ty, TABW, ty, TABW, (const uint8_t*)dxi, (const uint8_t*)dxi, EB, 0, cnt, dS, de2, de2, da2));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->w->ev_a2, 0));
        a1_done = true;
      } else if (cnt <= ROW_MAX_NUMBERS) {
        const uint32_t* ty;
        RET_IF(number_tables(ctx, (const uint8_t*)dy, cnt, ctx->w->tab1, &ty));
        RET_IF(dual_exp_any(ctx, ty, TABW, ty, TABW, (const uint8_t*)dxi, (const uint8_t*)dxi, EB, 0, cnt, dS, de2, de2, da2));
      } else {
        const size_t bw = modp_twin_exp_bucket_words();
        RET_IF(ensure(ctx, ctx->w->tab1, cnt * (bw + MODP_TWIN_EXTRA_WORDS) * 4 + MODP_TWIN_SLACK_BYTES));
        uint32_t* bk = (uint32_t*)ctx->w->tab1.p;
        TIMED_LAUNCH(ctx, 3, launch_twin_exp(ctx, (const uint8_t*)dy, (const uint8_t*)dxi, (const uint8_t*)de2, cnt, bk, bk + cnt * bw, dS, da2));
      }
      if (a1_done) {
      } else if ((pair_mask() & 1) && cnt >= 64 && comb_bits_of(ctx, cG) == 16)                                            // a1 = G^w
        TIMED_LAUNCH(ctx, 1, modp_launch_comb16_twin_exp_pair(cG, (const uint8_t*)dw, nullptr, (int)cnt, da1, nullptr, ctx->consts,
                                                              ctx->pair_tables, ctx->stream));
      else
        TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cG, cG, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)cnt,
                                                       da1, comb_bits_of(ctx, cG), ctx->consts, ctx->stream));
    } else {
      RET_IF(exp_dev(ctx, (const uint8_t*)dy, (const uint8_t*)dxi, cnt, dS));                       // S = Y^(1/x)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cG, cG, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)cnt,
                                                     da1, comb_bits_of(ctx, cG), ctx->consts, ctx->stream));              // a1 = G^w
      RET_IF(exp_dev(ctx, dS, (const uint8_t*)dw, cnt, da2));                                        // a2 = S^w
    }
    RET_IF(ensure_pinned(ctx, cnt * EB * 3));
    uint8_t* hS = (uint8_t*)ctx->pin;
    uint8_t* h1 = hS + cnt * EB;
    uint8_t* h2 = h1 + cnt * EB;
    HIPCHK(ctx, hipMemcpyAsync(hS, dS, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h1, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h2, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    RET_IF(small_vec_to_host(ctx, space, pk + off * EB, cnt * EB, hpk));
    RET_IF(small_vec_to_host(ctx, space, y + off * EB, cnt * EB, hy));
    for (size_t i = 0; i < cnt; ++i) {
      mpvss::Sha256 h;
      frame_update(h, hpk.data() + i * EB);
      frame_update(h, hy.data() + i * EB);
      frame_update(h, h1 + i * EB);
      frame_update(h, h2 + i * EB);
      uint8_t digest[32];
      h.final(digest);
      mpvss_modp_hash_to_scalar(digest, 32, c_out_host + (off + i) * EB);
    }
    if (space == MPVSS_HOST) memcpy(s_out + off * EB, hS, cnt * EB);
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- extract_secret_share in block form -------------------------------------------------------------------------
// compute ONLY enqueues: e2 = w / x mod (q-1) by the scalar ring's product kernel, S and a2 = S^w from one chain of squarings (the
// dealer's bucket kernels), a1 = G^w through the comb, the challenge c_i by K7 on the device, D2H of S and c into the slot's
// pinned staging.  absorb: waits for the oldest batch and hands out S (n x 256) and c (n x 256, big-endian scalars).
// Batches share the block slots and the FIFO order with the other block calls (slot kind 4).  Host buffers; every Y must be
// a unit mod q (the shared chain is exact for units only: a batch with Y = 0 mod q returns MPVSS_E_UNSUPPORTED and belongs to
// mpvss_modp_extract_shares).
namespace {

int extract_shares_compute_locked(mpvss_ctx* ctx, const uint8_t* pk, const uint8_t* y, const uint8_t* xinv, const uint8_t* w,
                                  size_t n) {
  if (n == 0 || !pk || !y || !xinv || !w || n > 0x7fffffff) return fail(ctx, MPVSS_E_INVALID, "extract_shares_compute: bad argument");
  mpvss_ctx::BlockSlot& sl = ctx->head_slot();
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "extract_shares_compute: every block slot (MPVSS_BLOCK_SLOTS) is in flight, absorb one first");
  if (n > MAX_CHUNK) return fail(ctx, MPVSS_E_UNSUPPORTED, "extract_shares_compute: batch larger than one chunk");
  {
    uint8_t qb[EB];
    modq_modulus_bytes(qb);
    static const uint8_t zero[EB] = {0};
    for (size_t i = 0; i < n; ++i)
      if (memcmp(y + i * EB, zero, EB) == 0 || memcmp(y + i * EB, qb, EB) == 0)
        return fail(ctx, MPVSS_E_UNSUPPORTED, "extract_shares_compute: an encrypted share is 0 mod q (use mpvss_modp_extract_shares)");
  }
  const auto t_enq0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
  sl.n = n;
  sl.kind = 4;
  sl.check_positions = false;
  sl.fd_used = false;
  sl.fd_chunks = 0;
  sl.enqueue_ms = 0;
  RET_IF(work_init(ctx, sl.work, nullptr));
  struct Restore {       // also: an early (error) return leaves nothing of this block running on the slot's streams
    mpvss_ctx* c;
    hipStream_t a, b;
    mpvss_ctx::BlockSlot* sl;
    ~Restore() {
      if (!sl->busy) {
        if (sl->work.sa) (void)hipStreamSynchronize(sl->work.sa);
        if (sl->work.sb) (void)hipStreamSynchronize(sl->work.sb);
      }
      c->sp = &c->main_spans; c->w = &c->work0; c->stream = a; c->stream_b = b;
    }
  } restore{ctx, ctx->stream, ctx->stream_b, &sl};
  ctx->w = &sl.work;
  ctx->stream = sl.work.sa;
  ctx->stream_b = sl.work.sb;
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  // pinned staging: outputs S [n][256], c [n][32]; inputs pk, y, xinv, w [n][256] each
  const size_t need = n * EB + n * 32 + 4 * n * EB;
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  uint8_t* hS = (uint8_t*)sl.pin;
  uint8_t* hc = hS + n * EB;
  uint8_t* in = hc + n * 32;
  uint8_t *hpk = in, *hy = in + n * EB, *hxi = in + 2 * n * EB, *hw = in + 3 * n * EB;
  const unsigned par = n * EB >= ((size_t)4 << 20) ? 4 : 1;      // (side by side from 4 MB each: the lock is held)
  hsc::parallel_indices(par, [&](unsigned k) {
    for (unsigned j = k; j < 4; j += par) {
      if (j == 0) memcpy(hpk, pk, n * EB);
      else if (j == 1) memcpy(hy, y, n * EB);
      else if (j == 2) memcpy(hxi, xinv, n * EB);
      else memcpy(hw, w, n * EB);
    }
  });
  const uint32_t* cG;
  RET_IF(comb_table(ctx, 1, &cG, n));
  RET_IF(modq_consts(ctx));
  const void *dpk, *dy, *dxi, *dw;
  RET_IF(stage_in(ctx, MPVSS_HOST, hpk, n * EB, ctx->w->in_a, &dpk));
  RET_IF(stage_in(ctx, MPVSS_HOST, hy, n * EB, ctx->w->in_b, &dy));
  RET_IF(stage_in(ctx, MPVSS_HOST, hxi, n * EB, ctx->w->in_c, &dxi));
  RET_IF(stage_in(ctx, MPVSS_HOST, hw, n * EB, ctx->w->in_d, &dw));
  // e2 = w * (1/x) mod (q-1): the scalar ring's product kernel, in stream order (round 6: this was 16 host threads under the lock)
  RET_IF(ensure(ctx, ctx->w->in_e, n * EB));
  const uint8_t* de2 = (const uint8_t*)ctx->w->in_e.p;
  LAUNCHCHK(ctx, modq_launch_mul((const uint8_t*)dw, (const uint8_t*)dxi, (int)n, (uint8_t*)ctx->w->in_e.p, ctx->consts_q, ctx->stream));
  RET_IF(ensure(ctx, ctx->w->xbe, n * EB));
  RET_IF(ensure(ctx, ctx->w->out1, n * EB));
  RET_IF(ensure(ctx, ctx->w->out2, n * EB));
  RET_IF(ensure(ctx, ctx->w->verd, n * 32));
  uint8_t *dS = (uint8_t*)ctx->w->xbe.p, *da1 = (uint8_t*)ctx->w->out1.p, *da2 = (uint8_t*)ctx->w->out2.p, *dc = (uint8_t*)ctx->w->verd.p;
  const size_t bw = modp_twin_exp_bucket_words();
  RET_IF(ensure(ctx, ctx->w->tab1, n * (bw + MODP_TWIN_EXTRA_WORDS) * 4 + MODP_TWIN_SLACK_BYTES));
  uint32_t* bk = (uint32_t*)ctx->w->tab1.p;
  // S = Y^(1/x) and a2 = S^w = Y^(w/x) from one chain of squarings (participant.rs:310-314, dleq.rs:213-216)
  TIMED_LAUNCH(ctx, 3, launch_twin_exp(ctx, (const uint8_t*)dy, (const uint8_t*)dxi, (const uint8_t*)de2, n, bk, bk + n * bw, dS, da2));
  if ((pair_mask() & 1) && n >= 64 && comb_bits_of(ctx, cG) == 16)                                                           // a1 = G^w
    TIMED_LAUNCH(ctx, 1, modp_launch_comb16_twin_exp_pair(cG, (const uint8_t*)dw, nullptr, (int)n, da1, nullptr, ctx->consts,
                                                          ctx->pair_tables, ctx->stream));
  else
    TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cG, cG, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)n, da1,
                                                   comb_bits_of(ctx, cG), ctx->consts, ctx->stream));
  // c_i = hash_to_scalar(SHA256(framed(pk_i) framed(Y_i) framed(a1_i) framed(a2_i)))   (participant.rs:329-343), K7
  TIMED_LAUNCH(ctx, 0, verdict_launch_modp_challenge((const uint8_t*)dpk, (const uint8_t*)dy, da1, da2, (int)n, dc, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(hS, dS, n * EB, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(hc, dc, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  sl.busy = true;
  sl.enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
  ctx->commit_head(sl);
  return MPVSS_OK;
}

int extract_shares_absorb_locked(mpvss_ctx* ctx, std::unique_lock<std::mutex>& lk, uint8_t* s_out_host, uint8_t* c_out_host) {
  mpvss_ctx::BlockSlot* slp = select_block(ctx, nullptr, false, 4, 4, "extract_shares_absorb");
  if (!slp) return MPVSS_E_INVALID;
  mpvss_ctx::BlockSlot& sl = *slp;
  const size_t n = sl.n;
  sl.absorbing = true;
  {
    const hipError_t e_dev = hipSetDevice(ctx->device);
    if (e_dev != hipSuccess) {
      ctx->release(sl);
      sl.absorbing = false;
      if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
      return fail(ctx, MPVSS_E_DEVICE, "absorb: hipSetDevice", e_dev);
    }
  }
  lk.unlock();
  const hipError_t e = hipEventSynchronize(sl.done);
  if (sl.gpu_done_ctr) sl.gpu_done_ctr->fetch_add(1);
  if (e == hipSuccess) {
    const uint8_t* hS = (const uint8_t*)sl.pin;
    const uint8_t* hc = hS + n * EB;
    if (s_out_host) memcpy(s_out_host, hS, n * EB);
    if (c_out_host)
      for (size_t i = 0; i < n; ++i) {             // 256-byte big-endian scalars, as mpvss_modp_hash_to_scalar writes them
        memset(c_out_host + i * EB, 0, EB - 32);
        memcpy(c_out_host + i * EB + EB - 32, hc + i * 32, 32);
      }
  }
  lk.lock();
  ctx->release(sl);
  sl.absorbing = false;
  if (e != hipSuccess) return fail(ctx, MPVSS_E_DEVICE, "extract_shares_absorb: hipEventSynchronize", e);
  RET_IF(spans_sum(ctx, sl.spans, ctx->kernel_ms));
  return MPVSS_OK;
}

}  // namespace

extern "C" int mpvss_modp_extract_shares_compute(mpvss_ctx* ctx, const uint8_t* pk, const uint8_t* y, const uint8_t* xinv,
                                                 const uint8_t* w, size_t n) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return extract_shares_compute_locked(ctx, pk, y, xinv, w, n);
}

extern "C" int mpvss_modp_extract_shares_absorb(mpvss_ctx* ctx, uint8_t* s_out_host, uint8_t* c_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  return extract_shares_absorb_locked(ctx, lk, s_out_host, c_out_host);
}

// Elliptic-curve groups (secp256k1, ristretto255): same translation unit, separate file
#include "capi_ec.inc"
// scalar-field entry points and reconstruct
#include "capi_scalar.inc"
// flat wire format of a box
#include "capi_wire.inc"
