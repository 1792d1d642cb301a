// The reference's call shape from COMPILED callers: T std::threads, each calling the ONE-box entry point
// mpvss_modp_verify_distribution (what Participant::verify_distribution_shares binds, src/participant.rs:399-455) on ONE engine
// context over K different dealers' boxes against the same participants -- the way the crate itself goes parallel
// (rayon, src/participant.rs:490-500), here over dealers.  The C ABI directly (flat arrays, as rust/src/batch.rs::flatten makes them):
// no maps, no Python.  bench.py's `drop_in` leg measures the same thing through ctypes threads; this program is the check that those
// behave like compiled callers (profiles/r06_drop_in_threads_compiled.txt).
//   build: make -C mpvss_rs_amd/csrc examples        run: ./examples/drop_in_threads [n] [t] [K boxes] [T threads] [passes] [key cache 0|1]
// key cache 1: the same once more with mpvss_ctx_set_key_cache_lru(ctx, 1, 2) -- the participants' key tables built once, taken by the
// verifiers' calls AND by T concurrent dealers (mpvss_modp_deal, every dealer its own polynomial; boxes checked against the first deal).
// Exit code 0 only if every box verified with the digest its dealer produced, from every thread, in every pass.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../include/mpvss_hip.h"

namespace {
typedef std::vector<uint8_t> Bytes;
constexpr size_t EB = MPVSS_MODP_BYTES;

void check(mpvss_ctx* ctx, int rc, const char* what) {
  if (rc != MPVSS_OK) {
    fprintf(stderr, "%s: rc %d: %s\n", what, rc, mpvss_last_error(ctx));
    exit(2);
  }
}

Bytes random_scalars(std::mt19937_64& gen, size_t count) {       // below 2^2040: below q - 1 (src/groups/modp.rs:47-58)
  Bytes out(count * EB, 0);
  for (size_t i = 0; i < count; ++i)
    for (size_t b = 1; b < EB; b += 8) {
      const uint64_t v = gen();
      memcpy(&out[i * EB + b], &v, b + 8 <= EB ? 8 : EB - b);
    }
  return out;
}

struct Box {
  Bytes commitments, shares, responses, challenge;
  uint8_t digest[32];
};
}  // namespace

int main(int argc, char** argv) {
  mpvss_process_init();      // before the first HIP call: 8 hardware queues for the block pipeline
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 65536, t = argc > 2 ? strtoull(argv[2], nullptr, 0) : 256;
  const size_t K = argc > 3 ? strtoull(argv[3], nullptr, 0) : 20;
  const unsigned T = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 0) : 12;
  const int passes = argc > 5 ? atoi(argv[5]) : 2;
  const int key_cache = argc > 6 ? atoi(argv[6]) : 0;
  mpvss_ctx* ctx = nullptr;
  if (mpvss_ctx_create(0, &ctx) != MPVSS_OK) {
    fprintf(stderr, "no HIP device: the engine has no CPU fallback\n");
    return 2;
  }
  std::mt19937_64 gen(20261004);
  std::vector<int64_t> positions(n);
  for (size_t i = 0; i < n; ++i) positions[i] = (int64_t)i + 1;
  uint8_t two[EB] = {0}, four[EB] = {0};
  two[EB - 1] = 2;
  four[EB - 1] = 4;
  // long-lived participants: y_i = G^x_i (src/groups/modp.rs:176-178)
  Bytes pubkeys(n * EB);
  {
    const Bytes x = random_scalars(gen, n);
    check(ctx, mpvss_modp_batch_exp_fixed_base(ctx, MPVSS_HOST, two, x.data(), n, pubkeys.data()), "keygen");
  }
  // K dealers: commitments C_j = g^a_j, then the whole box in one call (src/participant.rs:160-286)
  std::vector<Box> boxes(K);
  std::vector<Bytes> all_coeffs(K), all_witnesses(K);
  for (size_t b = 0; b < K; ++b) {
    const Bytes coeffs = random_scalars(gen, t), witnesses = random_scalars(gen, n);
    all_coeffs[b] = coeffs;
    all_witnesses[b] = witnesses;
    Box& bx = boxes[b];
    bx.commitments.resize(t * EB);
    bx.shares.resize(n * EB);
    bx.responses.resize(n * EB);
    bx.challenge.resize(EB);
    check(ctx, mpvss_modp_batch_exp_fixed_base(ctx, MPVSS_HOST, four, coeffs.data(), t, bx.commitments.data()), "commitments");
    check(ctx, mpvss_modp_deal(ctx, coeffs.data(), t, positions.data(), pubkeys.data(), witnesses.data(), n, nullptr, bx.shares.data(), nullptr,
                               nullptr, bx.digest, bx.challenge.data(), bx.responses.data()), "deal");
  }
  auto verify = [&](const Box& bx) -> bool {
    int verdict = 0;
    uint8_t digest[32];
    check(ctx, mpvss_modp_verify_distribution(ctx, MPVSS_HOST, bx.commitments.data(), t, positions.data(), pubkeys.data(), bx.shares.data(),
                                              bx.responses.data(), n, bx.challenge.data(), &verdict, digest, nullptr, nullptr, nullptr),
          "verify_distribution");
    return verdict == 1 && memcmp(digest, bx.digest, 32) == 0;
  };
  auto run = [&](unsigned threads, size_t count) -> double {       // `count` verifications, box k % K; returns seconds; exits on a wrong result
    std::vector<std::thread> pool;
    std::vector<int> bad(threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 0; k < threads; ++k)
      pool.emplace_back([&, k] {
        for (size_t i = k; i < count; i += threads)
          if (!verify(boxes[i % K])) bad[k] = 1;
      });
    for (auto& th : pool) th.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int b : bad)
      if (b) {
        fprintf(stderr, "a box did not verify with its dealer's digest\n");
        exit(1);
      }
    return s;
  };
  run(T + 2, T + 2);                         // every slot the callers will use gets its workspace and pinned staging (set-up, untimed)
  const double lone = run(1, 3) / 3;
  double best = 1e30;
  for (int p = 0; p < passes; ++p) best = std::min(best, run(T, 2 * K) / (double)(2 * K));
  // a tampered box is rejected by a concurrent caller as well
  Box bad = boxes[0];
  bad.responses[(n / 2) * EB + 100] ^= 1;
  int verdict = 1;
  check(ctx, mpvss_modp_verify_distribution(ctx, MPVSS_HOST, bad.commitments.data(), t, positions.data(), pubkeys.data(), bad.shares.data(),
                                            bad.responses.data(), n, bad.challenge.data(), &verdict, nullptr, nullptr, nullptr, nullptr), "verify (tampered)");
  if (verdict != 0) {
    fprintf(stderr, "a tampered box verified\n");
    return 1;
  }
  printf("n=%zu t=%zu, %zu boxes: %u compiled callers, one box per call: %.1f ms per box = %.3f M share verifications/s; one caller %.1f ms = %.3f M/s\n",
         n, t, K, T, best * 1e3, (double)n / best / 1e6, lone * 1e3, (double)n / lone / 1e6);
  // T concurrent dealers, each dealing box k % K again from its polynomial and witnesses: the same shares, digest and challenge
  auto deal_run = [&](unsigned threads, size_t count) -> double {
    std::vector<std::thread> pool;
    std::vector<int> bad_deal(threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 0; k < threads; ++k)
      pool.emplace_back([&, k] {
        Bytes y(n * EB), r(n * EB), ch(EB);
        uint8_t dg[32];
        for (size_t i = k; i < count; i += threads) {
          const size_t b = i % K;
          check(ctx, mpvss_modp_deal(ctx, all_coeffs[b].data(), t, positions.data(), pubkeys.data(), all_witnesses[b].data(), n, nullptr, y.data(),
                                     nullptr, nullptr, dg, ch.data(), r.data()), "deal (threads)");
          if (memcmp(dg, boxes[b].digest, 32) != 0 || y != boxes[b].shares || r != boxes[b].responses || ch != boxes[b].challenge) bad_deal[k] = 1;
        }
      });
    for (auto& th : pool) th.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int b : bad_deal)
      if (b) {
        fprintf(stderr, "a concurrent deal returned another box\n");
        exit(1);
      }
    return s;
  };
  deal_run(T, T);
  const double deal_s = deal_run(T, 2 * T) / (double)(2 * T);
  printf("%u compiled dealers, one mpvss_modp_deal call each: %.1f ms per box = %.3f M shares dealt/s\n", T, deal_s * 1e3, (double)n / deal_s / 1e6);
  if (key_cache) {
    if (mpvss_ctx_set_key_cache_lru(ctx, 1, 2) < 0) check(ctx, MPVSS_E_INVALID, "set_key_cache_lru");
    run(2, 4);                               // second sighting of the key array: its tables are built here
    run(T + 2, T + 2);
    double kbest = 1e30;
    for (int p = 0; p < passes; ++p) kbest = std::min(kbest, run(T, 2 * K) / (double)(2 * K));
    const double klone = run(1, 3) / 3;
    deal_run(T, T);
    const double kdeal = deal_run(T, 2 * T) / (double)(2 * T);
    const double kdeal_lone = deal_run(1, 3) / 3;
    printf("with the cross-call key cache: %u verifying callers %.1f ms per box = %.3f M share verifications/s (one caller %.1f ms); "
           "%u dealers %.1f ms per box = %.3f M shares dealt/s (one dealer %.1f ms)\n",
           T, kbest * 1e3, (double)n / kbest / 1e6, klone * 1e3, T, kdeal * 1e3, (double)n / kdeal / 1e6, kdeal_lone * 1e3);
    (void)mpvss_ctx_set_key_cache_lru(ctx, 0, 1);
  }
  printf("ok\n");
  mpvss_ctx_destroy(ctx);
  return 0;
}
