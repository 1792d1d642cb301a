// Streaming SHA-256 for the host-side Fiat-Shamir transcript (reference: sha2 0.10 `Sha256`,
// call sites src/dleq.rs:58-61,123 and src/participant.rs:405,451).  The transcript of
// verify_distribution_shares is ONE running hash over all shares in order, so it cannot be
// parallelised across shares; it runs on the host, using the SHA-NI instructions when the
// CPU has them.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mpvss {

struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t total;   // bytes absorbed
  size_t fill;      // bytes waiting in buf

  Sha256() { reset(); }
  void reset();
  void update(const void* data, size_t len);
  void final(uint8_t out[32]);
};

void sha256(const void* data, size_t len, uint8_t out[32]);
bool sha256_uses_shani();

}  // namespace mpvss
