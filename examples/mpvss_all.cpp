// Counterpart of the reference's examples/mpvss_all.rs (examples/mpvss_all.rs:10-95): the full PVSS
// protocol for n = 3, t = 3 over the 2048-bit MODP group, written against the C++ mirror of the crate's
// Participant API (mpvss_rs_amd/host/mpvss_host.hpp) whose group arithmetic runs on the MI355X engine.
//   build: make -C mpvss_rs_amd/csrc examples      run: ./examples/mpvss_all [seed]
#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "../mpvss_rs_amd/host/mpvss_host.hpp"

using namespace mpvss_host;

int main(int argc, char** argv) {
  mpvss_process_init();   // before the first HIP call: 8 hardware queues for the block pipeline
  Rng rng(argc > 1 ? strtoull(argv[1], nullptr, 0) : std::random_device{}());
  auto group = ModpGroup::create();
  const std::string secret_message = "Hello MPVSS Example.";
  Participant dealer = Participant::with_arc(group);
  dealer.initialize(rng);
  Participant p1 = Participant::with_arc(group), p2 = Participant::with_arc(group), p3 = Participant::with_arc(group);
  p1.initialize(rng);
  p2.initialize(rng);
  p3.initialize(rng);
  std::vector<BigUint> publickeys = {p1.publickey, p2.publickey, p3.publickey};

  DistributionSharesBox box = dealer.distribute_secret(string_to_secret(secret_message), publickeys, 3, rng);

  if (!p1.verify_distribution_shares(box) || !p2.verify_distribution_shares(box) || !p3.verify_distribution_shares(box)) {
    fprintf(stderr, "verify_distribution_shares failed\n");
    return 1;
  }
  const BigUint w = rng.below(group->modulus());
  ShareBox s1 = *p1.extract_secret_share(box, p1.privatekey, w);
  ShareBox s2 = *p2.extract_secret_share(box, p2.privatekey, w);
  ShareBox s3 = *p3.extract_secret_share(box, p3.privatekey, w);
  if (!p1.verify_share(s2, box, p2.publickey) || !p2.verify_share(s3, box, p3.publickey) ||
      !p3.verify_share(s1, box, s1.publickey)) {
    fprintf(stderr, "verify_share failed\n");
    return 1;
  }
  std::vector<ShareBox> share_boxs = {s1, s2, s3};
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
