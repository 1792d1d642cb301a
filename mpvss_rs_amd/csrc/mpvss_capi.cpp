// C-ABI of the engine (include/mpvss_hip.h): context, workspace, kernel orchestration and the
// host-side Fiat-Shamir transcript.  No CPU fallback exists: without a HIP device every compute
// entry point fails with MPVSS_E_DEVICE.
#include "../../include/mpvss_hip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "modp_kernels.h"
#include "sha256.h"

namespace {

constexpr size_t EB = MPVSS_MODP_BYTES;          // element / scalar bytes
constexpr size_t TABW = MODP_TABLE_WORDS;        // words per 16-entry window table
constexpr size_t MAX_CHUNK = 1u << 18;           // shares per pass (bounds the table workspace: 2 x 1.2 GiB)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct mpvss_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* consts = nullptr;
  std::string err;
  std::mutex mu;
  // grow-only device workspace
  DevBuf in_a, in_b, in_c, in_d, in_e, pos, cm, xbe, out1, out2, tab1, tab2, tabg, cbuf;
  DevBuf comb[2];            // fixed-base comb tables of g = 4 (index 0) and G = 2 (index 1), built on first use
  bool comb_ready[2] = {false, false};
  // pinned host staging
  void* pin = nullptr;
  size_t pin_cap = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double kernel_ms[3] = {-1, -1, -1};
  struct Span { int id; hipEvent_t a, b; };
  struct SpanSet {
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
  };
  SpanSet main_spans;
  SpanSet* sp = &main_spans;   // where TIMED_LAUNCH records
  // Two verify blocks may be in flight (compute of block k+1 is enqueued before block k is absorbed):
  // each has its own pinned staging, timing events and completion event.
  struct BlockSlot {
    void* pin = nullptr;
    size_t cap = 0;
    size_t n = 0;
    bool busy = false;
    bool check_positions = false;
    hipEvent_t done = nullptr;
    SpanSet spans;
    double kernel_ms[3] = {0, 0, 0};
  };
  BlockSlot slot[2];
  unsigned head = 0, tail = 0;   // next slot to fill / to absorb
};

namespace {

int fail(mpvss_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  if (ctx) {
    char buf[512];
    if (e != hipSuccess)
      snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(buf, sizeof(buf), "%s", what);
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
    HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipMalloc(workspace)", e);
  b.cap = bytes;
  return 0;
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
int spans_sum(mpvss_ctx* ctx, mpvss_ctx::SpanSet& ss, double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = 0;
  for (auto& s : ss.spans) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, s.a, s.b));
    out[s.id] += ms;
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

// the shared 16-entry table of one base (host bytes) into ctx->tabg
int shared_table(mpvss_ctx* ctx, const uint8_t* base_host, const uint32_t** tab) {
  RET_IF(ensure(ctx, ctx->tabg, TABW * 4 + EB));
  uint8_t* dbase = (uint8_t*)ctx->tabg.p + TABW * 4;
  HIPCHK(ctx, hipMemcpyAsync(dbase, base_host, EB, hipMemcpyHostToDevice, ctx->stream));
  TIMED_LAUNCH(ctx, 2, modp_launch_build_table(dbase, 1, (uint32_t*)ctx->tabg.p, ctx->consts, ctx->stream));
  *tab = (const uint32_t*)ctx->tabg.p;
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

// fixed-base comb table of generator `gid` (modp.rs:64-66), built once per context (about 25 ms)
int comb_table(mpvss_ctx* ctx, int gid, const uint32_t** comb) {
  if (!ctx->comb_ready[gid]) {
    RET_IF(ensure(ctx, ctx->comb[gid], (size_t)MODP_COMB_WORDS * 4 + EB));
    uint8_t* dbase = (uint8_t*)ctx->comb[gid].p + (size_t)MODP_COMB_WORDS * 4;
    HIPCHK(ctx, hipMemcpyAsync(dbase, g_bytes(gid), EB, hipMemcpyHostToDevice, ctx->stream));
    LAUNCHCHK(ctx, modp_launch_comb_build(dbase, (uint32_t*)ctx->comb[gid].p, ctx->consts, ctx->stream));
    ctx->comb_ready[gid] = true;
  }
  *comb = (const uint32_t*)ctx->comb[gid].p;
  return 0;
}

// per-number tables of `count` bases (device bytes) into buf
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

// out = B^e (count numbers); bases/exps device pointers; uses tab1
int exp_dev(mpvss_ctx* ctx, const uint8_t* bases_dev, const uint8_t* exps_dev, size_t count, uint8_t* out_dev) {
  const uint32_t* t1;
  RET_IF(number_tables(ctx, bases_dev, count, ctx->tab1, &t1));
  TIMED_LAUNCH(ctx, 1, modp_launch_dual_exp(t1, TABW, t1, TABW, exps_dev, exps_dev, EB, 0, (int)count, out_dev,
                                            ctx->consts, ctx->stream));
  return 0;
}

}  // namespace

// -------------------------------------------------------------------------------------------

extern "C" int mpvss_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

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
  if (modp_consts_upload(&ctx->consts) != 0) {
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MPVSS_E_DEVICE;
  }
  *out = ctx;
  return MPVSS_OK;
}

extern "C" void mpvss_ctx_destroy(mpvss_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (DevBuf* b : {&ctx->in_a, &ctx->in_b, &ctx->in_c, &ctx->in_d, &ctx->in_e, &ctx->pos, &ctx->cm, &ctx->xbe,
                    &ctx->out1, &ctx->out2, &ctx->tab1, &ctx->tab2, &ctx->tabg, &ctx->cbuf, &ctx->comb[0], &ctx->comb[1]})
    if (b->p) (void)hipFree(b->p);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->consts) (void)hipFree(ctx->consts);
  for (hipEvent_t e : ctx->main_spans.ev_pool) (void)hipEventDestroy(e);
  for (auto& sl : ctx->slot) {
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.done) (void)hipEventDestroy(sl.done);
    for (hipEvent_t e : sl.spans.ev_pool) (void)hipEventDestroy(e);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* mpvss_last_error(const mpvss_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mpvss_ctx_set_stream(mpvss_ctx* ctx, void* hip_stream) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return MPVSS_OK;
}

extern "C" int mpvss_ctx_synchronize(mpvss_ctx* ctx) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return MPVSS_OK;
}

extern "C" double mpvss_last_kernel_ms(const mpvss_ctx* ctx, int kernel_id) {
  if (!ctx || kernel_id < 0 || kernel_id > 2) return -1;
  return ctx->kernel_ms[kernel_id];
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
  RET_IF(stage_in(ctx, space, a, n * EB, ctx->in_a, &da));
  RET_IF(stage_in(ctx, space, b, n * EB, ctx->in_b, &db));
  uint8_t* dout = out;
  if (space == MPVSS_HOST) {
    RET_IF(ensure(ctx, ctx->out1, n * EB));
    dout = (uint8_t*)ctx->out1.p;
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
    RET_IF(stage_in(ctx, space, bases + off * EB, cnt * EB, ctx->in_a, &db));
    RET_IF(stage_in(ctx, space, exps + off * EB, cnt * EB, ctx->in_b, &de));
    uint8_t* dout = out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->out1, cnt * EB));
      dout = (uint8_t*)ctx->out1.p;
    }
    RET_IF(exp_dev(ctx, (const uint8_t*)db, (const uint8_t*)de, cnt, dout));
    if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, out + off * EB, dout, cnt * EB));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
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
    RET_IF(comb_table(ctx, generator_id(base_host), &cg));
  else
    RET_IF(shared_table(ctx, base_host, &tg));
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void* de;
    RET_IF(stage_in(ctx, space, exps + off * EB, cnt * EB, ctx->in_b, &de));
    uint8_t* dout = out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->out1, cnt * EB));
      dout = (uint8_t*)ctx->out1.p;
    }
    if (cg)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cg, cg, 0, (const uint8_t*)de, (const uint8_t*)de, EB, 0, (int)cnt,
                                                     dout, ctx->consts, ctx->stream));
    else
      TIMED_LAUNCH(ctx, 1, modp_launch_dual_exp(tg, 0, tg, 0, (const uint8_t*)de, (const uint8_t*)de, EB, 0, (int)cnt,
                                                dout, ctx->consts, ctx->stream));
    if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, out + off * EB, dout, cnt * EB));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- commitment multi-exp -----------------------------------------------------------------------
namespace {
// commitments (space) -> Montgomery limbs in ctx->cm
int stage_commitments(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t) {
  const void* dc;
  RET_IF(stage_in(ctx, space, commitments, t * EB, ctx->cbuf, &dc));
  RET_IF(ensure(ctx, ctx->cm, t * MODP_L * 4));
  LAUNCHCHK(ctx, modp_launch_to_mont((const uint8_t*)dc, (uint32_t*)ctx->cm.p, (int)t, ctx->consts, ctx->stream));
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
  RET_IF(stage_in(ctx, space, positions, n * 8, ctx->pos, &d));
  *dpos = (const int64_t*)d;
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
    RET_IF(ensure(ctx, ctx->xbe, n * EB));
    dout = (uint8_t*)ctx->xbe.p;
  }
  TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval((const uint32_t*)ctx->cm.p, (int)t, dpos, (int)n, nullptr, dout,
                                               ctx->consts, ctx->stream));
  if (space == MPVSS_HOST) RET_IF(copy_out(ctx, space, x_out, dout, n * EB));
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- DLEQ verifier commitments -------------------------------------------------------------------
namespace {
// a = B1^r * B2^c for `cnt` shares.  tab_b1: shared table (stride 0) or nullptr -> per-number tables
// from b1_dev.  c: device pointer, stride c_stride (0 shared).
int dleq_side(mpvss_ctx* ctx, const uint32_t* shared_b1, const uint8_t* b1_dev, const uint8_t* b2_dev,
              const uint8_t* r_dev, const uint8_t* c_dev, size_t c_stride, int c_windows, size_t cnt,
              uint8_t* out_dev, const uint32_t* comb_b1 = nullptr) {
  const uint32_t *t1, *t2;
  size_t s1 = TABW;
  RET_IF(number_tables(ctx, b2_dev, cnt, ctx->tab2, &t2));
  if (comb_b1) {   // B1 is a generator with a comb table: no squarings for B1^r
    TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(comb_b1, t2, TABW, r_dev, c_dev, c_stride, c_windows, (int)cnt,
                                                   out_dev, ctx->consts, ctx->stream));
    return 0;
  }
  if (shared_b1) {
    t1 = shared_b1;
    s1 = 0;
  } else {
    RET_IF(number_tables(ctx, b1_dev, cnt, ctx->tab1, &t1));
  }
  TIMED_LAUNCH(ctx, 1, modp_launch_dual_exp(t1, s1, t2, TABW, r_dev, c_dev, c_stride, c_windows, (int)cnt, out_dev,
                                            ctx->consts, ctx->stream));
  return 0;
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
    RET_IF(comb_table(ctx, generator_id(g1_host), &cg));
  else
    RET_IF(shared_table(ctx, g1_host, &tg));
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const void *dh1, *dg2, *dh2, *dr, *dc;
    RET_IF(stage_in(ctx, space, h1 + off * EB, cnt * EB, ctx->in_a, &dh1));
    RET_IF(stage_in(ctx, space, g2 + off * EB, cnt * EB, ctx->in_b, &dg2));
    RET_IF(stage_in(ctx, space, h2 + off * EB, cnt * EB, ctx->in_c, &dh2));
    RET_IF(stage_in(ctx, space, r + off * EB, cnt * EB, ctx->in_d, &dr));
    if (c_per_share)
      RET_IF(stage_in(ctx, space, c + off * EB, cnt * EB, ctx->in_e, &dc));
    else
      RET_IF(stage_in(ctx, MPVSS_HOST, c, EB, ctx->in_e, &dc));
    uint8_t *d1 = a1_out + off * EB, *d2 = a2_out + off * EB;
    if (space == MPVSS_HOST) {
      RET_IF(ensure(ctx, ctx->out1, cnt * EB));
      RET_IF(ensure(ctx, ctx->out2, cnt * EB));
      d1 = (uint8_t*)ctx->out1.p;
      d2 = (uint8_t*)ctx->out2.p;
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

int verify_block_compute_locked(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                const uint8_t* responses, size_t n, const uint8_t* challenge_host) {
  if (!challenge_host) return fail(ctx, MPVSS_E_INVALID, "verify: null challenge");
  if (n > 0 && (!commitments || !positions || !pubkeys || !shares || !responses || t == 0 || t > 0x7fffffff))
    return fail(ctx, MPVSS_E_INVALID, "verify: bad argument (t must be >= 1)");
  mpvss_ctx::BlockSlot& sl = ctx->slot[ctx->head & 1];
  if (sl.busy) return fail(ctx, MPVSS_E_INVALID, "verify: two blocks already in flight, absorb one first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  sl.n = n;
  sl.check_positions = false;
  sl.busy = true;
  ++ctx->head;
  if (n == 0) return MPVSS_OK;
  struct Restore {
    mpvss_ctx* c;
    ~Restore() { c->sp = &c->main_spans; }
  } restore{ctx};
  ctx->sp = &sl.spans;
  spans_reset(ctx);
  const size_t need = n * EB * 4 + n * 8;
  if (need > sl.cap) {
    if (sl.pin) HIPCHK(ctx, hipHostFree(sl.pin));
    sl.pin = nullptr;
    sl.cap = 0;
    hipError_t e = hipHostMalloc(&sl.pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, MPVSS_E_NOMEM, "hipHostMalloc(block staging)", e);
    sl.cap = need;
  }
  RET_IF(stage_commitments(ctx, space, commitments, t));
  const uint32_t* cg;
  RET_IF(comb_table(ctx, 0, &cg));
  const void* dchal;
  RET_IF(stage_in(ctx, MPVSS_HOST, challenge_host, EB, ctx->in_e, &dchal));
  const int c_windows = fits_256_bits(challenge_host) ? 64 : 512;
  uint8_t* hX = (uint8_t*)sl.pin;
  uint8_t* hY = hX + n * EB;
  uint8_t* h1 = hY + n * EB;
  uint8_t* h2 = h1 + n * EB;
  int64_t* hpos = (int64_t*)(h2 + n * EB);
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    const int64_t* dpos;
    if (space == MPVSS_HOST) {
      RET_IF(stage_positions(ctx, space, positions + off, cnt, &dpos));
    } else {
      // device-resident positions are validated when the block is absorbed (no host sync here)
      dpos = positions + off;
      HIPCHK(ctx, hipMemcpyAsync(hpos + off, dpos, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
      sl.check_positions = true;
    }
    const void *dy, *dY, *dr;
    RET_IF(stage_in(ctx, space, pubkeys + off * EB, cnt * EB, ctx->in_a, &dy));
    RET_IF(stage_in(ctx, space, shares + off * EB, cnt * EB, ctx->in_b, &dY));
    RET_IF(stage_in(ctx, space, responses + off * EB, cnt * EB, ctx->in_c, &dr));
    RET_IF(ensure(ctx, ctx->xbe, cnt * EB));
    RET_IF(ensure(ctx, ctx->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->out2, cnt * EB));
    uint8_t* dX = (uint8_t*)ctx->xbe.p;
    uint8_t* da1 = (uint8_t*)ctx->out1.p;
    uint8_t* da2 = (uint8_t*)ctx->out2.p;
    // X_i                                                  participant.rs:423-434
    TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval((const uint32_t*)ctx->cm.p, (int)t, dpos, (int)cnt, nullptr, dX,
                                                 ctx->consts, ctx->stream));
    // a1_i = g^r_i * X_i^c, a2_i = y_i^r_i * Y_i^c           dleq.rs:66-84
    RET_IF(dleq_side(ctx, nullptr, nullptr, dX, (const uint8_t*)dr, (const uint8_t*)dchal, 0, c_windows, cnt, da1, cg));
    RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)dy, (const uint8_t*)dY, (const uint8_t*)dr, (const uint8_t*)dchal,
                     0, c_windows, cnt, da2));
    HIPCHK(ctx, hipMemcpyAsync(hX + off * EB, dX, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hY + off * EB, dY, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h1 + off * EB, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h2 + off * EB, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    if (off + MAX_CHUNK < n) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // device buffers are reused
  }
  HIPCHK(ctx, hipEventRecord(sl.done, ctx->stream));
  return MPVSS_OK;
}

int verify_block_absorb_locked(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out, uint8_t* a1_out, uint8_t* a2_out) {
  if (!state) return fail(ctx, MPVSS_E_INVALID, "absorb: null transcript state");
  mpvss_ctx::BlockSlot& sl = ctx->slot[ctx->tail & 1];
  if (!sl.busy) return fail(ctx, MPVSS_E_INVALID, "absorb: no block in flight");
  const size_t n = sl.n;
  sl.busy = false;
  ++ctx->tail;
  if (n == 0) return MPVSS_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipEventSynchronize(sl.done));
  RET_IF(spans_sum(ctx, sl.spans, ctx->kernel_ms));
  const uint8_t* hX = (const uint8_t*)sl.pin;
  const uint8_t* hY = hX + n * EB;
  const uint8_t* h1 = hY + n * EB;
  const uint8_t* h2 = h1 + n * EB;
  if (sl.check_positions) RET_IF(check_positions_host(ctx, (const int64_t*)(h2 + n * EB), n));
  mpvss::Sha256 tr;
  memcpy(&tr, state, sizeof(tr));
  for (size_t i = 0; i < n; ++i) {                     // dleq.rs:87-99, share order = array order
    frame_update(tr, hX + i * EB);
    frame_update(tr, hY + i * EB);
    frame_update(tr, h1 + i * EB);
    frame_update(tr, h2 + i * EB);
  }
  memcpy(state, &tr, sizeof(tr));
  if (x_out) memcpy(x_out, hX, n * EB);
  if (a1_out) memcpy(a1_out, h1, n * EB);
  if (a2_out) memcpy(a2_out, h2, n * EB);
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

extern "C" int mpvss_modp_verify_block_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host,
                                              uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_block_absorb_locked(ctx, state, x_out_host, a1_out_host, a2_out_host);
}

extern "C" int mpvss_modp_verify_distribution(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                              const int64_t* positions, const uint8_t* pubkeys,
                                              const uint8_t* shares, const uint8_t* responses, size_t n,
                                              const uint8_t* challenge_host, int* verdict, uint8_t* digest32_out,
                                              uint8_t* x_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!verdict || !challenge_host) return fail(ctx, MPVSS_E_INVALID, "verify_distribution: bad argument");
  *verdict = 0;
  uint8_t state[MPVSS_TRANSCRIPT_STATE_BYTES];
  mpvss_transcript_init(state);
  RET_IF(verify_block_compute_locked(ctx, space, commitments, t, positions, pubkeys, shares, responses, n,
                                     challenge_host));
  RET_IF(verify_block_absorb_locked(ctx, state, x_out_host, a1_out_host, a2_out_host));
  return mpvss_modp_transcript_verdict(state, challenge_host, verdict, digest32_out);
}

// ---- verify_share, batched ----------------------------------------------------------------------------
extern "C" int mpvss_modp_verify_shares(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s,
                                        const uint8_t* y, const uint8_t* c, const uint8_t* r, size_t n,
                                        uint8_t* verdicts_host) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return MPVSS_OK;
  if (!pk || !s || !y || !c || !r || !verdicts_host) return fail(ctx, MPVSS_E_INVALID, "verify_shares: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  const uint32_t* cG;
  RET_IF(comb_table(ctx, 1, &cG));
  std::vector<uint8_t> hc, hpk, hy;
  for (size_t off = 0; off < n; off += MAX_CHUNK) {
    const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
    RET_IF(small_vec_to_host(ctx, space, c + off * EB, cnt * EB, hc));
    bool small = true;
    for (size_t i = 0; i < cnt && small; ++i) small = fits_256_bits(hc.data() + i * EB);
    const int c_windows = small ? 64 : 512;
    const void *dpk, *ds, *dy, *dc, *dr;
    RET_IF(stage_in(ctx, space, pk + off * EB, cnt * EB, ctx->in_a, &dpk));
    RET_IF(stage_in(ctx, space, s + off * EB, cnt * EB, ctx->in_b, &ds));
    RET_IF(stage_in(ctx, space, y + off * EB, cnt * EB, ctx->in_c, &dy));
    RET_IF(stage_in(ctx, space, r + off * EB, cnt * EB, ctx->in_d, &dr));
    RET_IF(stage_in(ctx, space, c + off * EB, cnt * EB, ctx->in_e, &dc));
    RET_IF(ensure(ctx, ctx->out1, cnt * EB));
    RET_IF(ensure(ctx, ctx->out2, cnt * EB));
    uint8_t* da1 = (uint8_t*)ctx->out1.p;
    uint8_t* da2 = (uint8_t*)ctx->out2.p;
    // a1 = G^r * pk^c ; a2 = S^r * Y^c                       dleq.rs:66-84 via participant.rs:376-385
    RET_IF(dleq_side(ctx, nullptr, nullptr, (const uint8_t*)dpk, (const uint8_t*)dr, (const uint8_t*)dc, EB,
                     c_windows, cnt, da1, cG));
    RET_IF(dleq_side(ctx, nullptr, (const uint8_t*)ds, (const uint8_t*)dy, (const uint8_t*)dr, (const uint8_t*)dc, EB,
                     c_windows, cnt, da2));
    RET_IF(ensure_pinned(ctx, cnt * EB * 2));
    uint8_t* h1 = (uint8_t*)ctx->pin;
    uint8_t* h2 = h1 + cnt * EB;
    HIPCHK(ctx, hipMemcpyAsync(h1, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h2, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    RET_IF(small_vec_to_host(ctx, space, pk + off * EB, cnt * EB, hpk));
    RET_IF(small_vec_to_host(ctx, space, y + off * EB, cnt * EB, hy));
    for (size_t i = 0; i < cnt; ++i) {
      mpvss::Sha256 h;                                      // dleq.rs:289 fresh hasher per proof
      frame_update(h, hpk.data() + i * EB);
      frame_update(h, hy.data() + i * EB);
      frame_update(h, h1 + i * EB);
      frame_update(h, h2 + i * EB);
      uint8_t digest[32];
      h.final(digest);
      verdicts_host[off + i] = challenge_matches(digest, hc.data() + i * EB) ? 1 : 0;
    }
  }
  RET_IF(spans_collect(ctx));
  return MPVSS_OK;
}

// ---- distribute_secret, group part ---------------------------------------------------------------------
extern "C" int mpvss_modp_distribute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                     const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                                     const uint8_t* witnesses, size_t n, uint8_t* x_out, uint8_t* y_out,
                                     uint8_t* a1_out, uint8_t* a2_out, uint8_t* digest32_out) {
  if (!ctx) return MPVSS_E_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n > 0 && (!commitments || !positions || !pubkeys || !p_values || !witnesses || !x_out || !y_out || !a1_out ||
                !a2_out || t == 0 || t > 0x7fffffff))
    return fail(ctx, MPVSS_E_INVALID, "distribute: bad argument");
  if (t > n) return fail(ctx, MPVSS_E_INVALID, "distribute: threshold > number of public keys (participant.rs:166)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  spans_reset(ctx);
  mpvss::Sha256 transcript;
  if (n > 0) {
    RET_IF(stage_commitments(ctx, space, commitments, t));
    const uint32_t* cg;
    RET_IF(comb_table(ctx, 0, &cg));
    for (size_t off = 0; off < n; off += MAX_CHUNK) {
      const size_t cnt = (n - off < MAX_CHUNK) ? n - off : MAX_CHUNK;
      const int64_t* dpos;
      RET_IF(stage_positions(ctx, space, positions + off, cnt, &dpos));
      const void *dy, *dp, *dw;
      RET_IF(stage_in(ctx, space, pubkeys + off * EB, cnt * EB, ctx->in_a, &dy));
      RET_IF(stage_in(ctx, space, p_values + off * EB, cnt * EB, ctx->in_b, &dp));
      RET_IF(stage_in(ctx, space, witnesses + off * EB, cnt * EB, ctx->in_c, &dw));
      uint8_t *dX = x_out + off * EB, *dY = y_out + off * EB, *da1 = a1_out + off * EB, *da2 = a2_out + off * EB;
      if (space == MPVSS_HOST) {
        RET_IF(ensure(ctx, ctx->xbe, cnt * EB));
        RET_IF(ensure(ctx, ctx->out1, cnt * EB));
        RET_IF(ensure(ctx, ctx->out2, cnt * EB));
        RET_IF(ensure(ctx, ctx->in_d, cnt * EB));
        dX = (uint8_t*)ctx->xbe.p;
        dY = (uint8_t*)ctx->in_d.p;
        da1 = (uint8_t*)ctx->out1.p;
        da2 = (uint8_t*)ctx->out2.p;
      }
      TIMED_LAUNCH(ctx, 0, modp_launch_commit_eval((const uint32_t*)ctx->cm.p, (int)t, dpos, (int)cnt, nullptr, dX,
                                                   ctx->consts, ctx->stream));
      // y-tables once, two exponent sets: Y = y^p (participant.rs:219), a2 = y^w (dleq.rs:214-216)
      const uint32_t* ty;
      RET_IF(number_tables(ctx, (const uint8_t*)dy, cnt, ctx->tab1, &ty));
      TIMED_LAUNCH(ctx, 1, modp_launch_dual_exp(ty, TABW, ty, TABW, (const uint8_t*)dp, (const uint8_t*)dp, EB, 0,
                                                (int)cnt, dY, ctx->consts, ctx->stream));
      TIMED_LAUNCH(ctx, 1, modp_launch_dual_exp(ty, TABW, ty, TABW, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0,
                                                (int)cnt, da2, ctx->consts, ctx->stream));
      // a1 = g^w (dleq.rs:207-211)
      TIMED_LAUNCH(ctx, 1, modp_launch_comb_dual_exp(cg, cg, 0, (const uint8_t*)dw, (const uint8_t*)dw, EB, 0, (int)cnt,
                                                     da1, ctx->consts, ctx->stream));
      RET_IF(ensure_pinned(ctx, cnt * EB * 4));
      uint8_t* hX = (uint8_t*)ctx->pin;
      uint8_t* hY = hX + cnt * EB;
      uint8_t* h1 = hY + cnt * EB;
      uint8_t* h2 = h1 + cnt * EB;
      HIPCHK(ctx, hipMemcpyAsync(hX, dX, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(hY, dY, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(h1, da1, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(h2, da2, cnt * EB, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      for (size_t i = 0; i < cnt; ++i) {                     // participant.rs:238-245
        frame_update(transcript, hX + i * EB);
        frame_update(transcript, hY + i * EB);
        frame_update(transcript, h1 + i * EB);
        frame_update(transcript, h2 + i * EB);
      }
      if (space == MPVSS_HOST) {
        memcpy(x_out + off * EB, hX, cnt * EB);
        memcpy(y_out + off * EB, hY, cnt * EB);
        memcpy(a1_out + off * EB, h1, cnt * EB);
        memcpy(a2_out + off * EB, h2, cnt * EB);
      }
    }
    RET_IF(spans_collect(ctx));
  }
  if (digest32_out) transcript.final(digest32_out);
  return MPVSS_OK;
}
