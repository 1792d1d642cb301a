// Shared body of the two curve-group examples: the full PVSS protocol for n = 3, t = 3, as the reference's
// examples/mpvss_all_secp256k1.rs:10-92 and examples/mpvss_all_ristretto255.rs run it, against the C++ mirror of the
// crate's Participant API (mpvss_rs_amd/host/mpvss_host_ec.hpp) whose group arithmetic runs on the MI355X engine.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "../mpvss_rs_amd/host/mpvss_host_ec.hpp"

template <class Traits>
int run_mpvss_all(const char* secret_message_c, int argc, char** argv) {
  using namespace mpvss_host;
  typedef EcParticipant<Traits> P;
  mpvss_process_init();   // before the first HIP call: 8 hardware queues for the block pipeline
  Rng rng(argc > 1 ? strtoull(argv[1], nullptr, 0) : std::random_device{}());
  auto group = EcGroup<Traits>::create();
  const std::string secret_message = secret_message_c;
  P dealer = P::with_arc(group);
  dealer.initialize(rng);
  P p1 = P::with_arc(group), p2 = P::with_arc(group), p3 = P::with_arc(group);
  p1.initialize(rng);
  p2.initialize(rng);
  p3.initialize(rng);
  std::vector<Bytes> publickeys = {p1.publickey, p2.publickey, p3.publickey};

  EcDistributionSharesBox box = dealer.distribute_secret(string_to_secret(secret_message), publickeys, 3, rng);

  if (!p1.verify_distribution_shares(box) || !p2.verify_distribution_shares(box) || !p3.verify_distribution_shares(box)) {
    fprintf(stderr, "verify_distribution_shares failed\n");
    return 1;
  }
  const BigUint w = group->generate_private_key(rng);
  EcShareBox s1 = *p1.extract_secret_share(box, p1.privatekey, w);
  EcShareBox s2 = *p2.extract_secret_share(box, p2.privatekey, w);
  EcShareBox s3 = *p3.extract_secret_share(box, p3.privatekey, w);
  if (!p1.verify_share(s2, box, p2.publickey) || !p2.verify_share(s3, box, p3.publickey) ||
      !p3.verify_share(s1, box, s1.publickey)) {
    fprintf(stderr, "verify_share failed\n");
    return 1;
  }
  // a tampered share box must be rejected (the reference's tests never show a rejection)
  EcShareBox bad = s2;
  bad.response = group->scalar_sub(bad.response, BigUint(1));
  if (p1.verify_share(bad, box, p2.publickey)) {
    fprintf(stderr, "tampered share box accepted\n");
    return 1;
  }
  std::vector<EcShareBox> share_boxs = {s1, s2, s3};
  const std::string r1 = string_from_secret(*dealer.reconstruct(share_boxs, box));
  const std::string r2 = string_from_secret(*dealer.reconstruct(share_boxs, box));
  const std::string r3 = string_from_secret(*dealer.reconstruct(share_boxs, box));
  if (r1 != secret_message || r2 != secret_message || r3 != secret_message) {
    fprintf(stderr, "reconstruction mismatch\n");
    return 1;
  }
  printf("secret message: %s\n", secret_message.c_str());
  printf("r1 str: %s\n", r1.c_str());
  printf("r2 str: %s\n", r2.c_str());
  printf("r3 str: %s\n", r3.c_str());
  return 0;
}
