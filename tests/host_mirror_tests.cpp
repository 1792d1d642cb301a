// The reference's own MODP tests, restated against the C++ mirror of its API running on the GPU engine:
//   tests/mpvss_tests.rs:10-87, src/participant.rs:593-743, src/dleq.rs:406-441, src/mpvss.rs:151-287.
// Plus the negative cases the reference never tests.  Exit code 0 = all passed.  Run by tests/test_gpu_host_mirror.py.
#include <cstdio>
#include <cstdlib>

#include "../mpvss_rs_amd/host/mpvss_host.hpp"

using namespace mpvss_host;

static int failures = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) { fprintf(stderr, "FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

static void test_end_to_end_modp(std::shared_ptr<ModpGroup> group) {       // participant.rs:593-698
  Rng rng(1);
  Participant dealer = Participant::with_arc(group);
  dealer.initialize(rng);
  Participant p[3] = {Participant::with_arc(group), Participant::with_arc(group), Participant::with_arc(group)};
  std::vector<BigUint> pks;
  for (auto& x : p) { x.initialize(rng); pks.push_back(x.publickey); }
  const BigUint secret = string_to_secret("Hello MPVSS End-to-End Test!");
  DistributionSharesBox box = dealer.distribute_secret(secret, pks, 3, rng);
  for (auto& x : p) CHECK(x.verify_distribution_shares(box));
  CHECK(box.publickeys.size() == 3 && box.commitments.size() == 3 && box.shares.size() == 3);
  CHECK(!box.U.is_zero());
  const BigUint w = rng.below(group->modulus());
  std::vector<ShareBox> sb;
  for (auto& x : p) sb.push_back(*x.extract_secret_share(box, x.privatekey, w));
  for (int i = 0; i < 3; ++i) { CHECK(sb[i].publickey == p[i].publickey); CHECK(!sb[i].share.is_zero()); }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (i != j) CHECK(p[i].verify_share(sb[j], box, p[j].publickey));
  CHECK(*dealer.reconstruct(sb, box) == secret);
  // negative cases
  DistributionSharesBox bad = box;
  bad.responses.begin()->second = bad.responses.begin()->second ^ BigUint(1);
  CHECK(!p[0].verify_distribution_shares(bad));
  bad = box;
  bad.shares.erase(bad.shares.begin());
  CHECK(!p[0].verify_distribution_shares(bad));                            // participant.rs:415-420
  ShareBox sbad = sb[1];
  sbad.response = sbad.response ^ BigUint(2);
  CHECK(!p[0].verify_share(sbad, box, p[1].publickey));
  CHECK(!p[0].verify_share(sb[1], box, BigUint(12345)));                   // unknown public key -> false (:370-373)
  std::vector<ShareBox> two = {sb[0], sb[1]};
  CHECK(!dealer.reconstruct(two, box).has_value());                        // fewer than t shares (:467-469)
}

static void test_threshold_subset_positions_1_and_3(std::shared_ptr<ModpGroup> group) {   // participant.rs:703-743
  Rng rng(2);
  Participant dealer = Participant::with_arc(group);
  dealer.initialize(rng);
  Participant p1 = Participant::with_arc(group), p2 = Participant::with_arc(group), p3 = Participant::with_arc(group);
  p1.initialize(rng); p2.initialize(rng); p3.initialize(rng);
  const BigUint secret(123456);
  DistributionSharesBox box = dealer.distribute_secret(secret, {p1.publickey, p2.publickey, p3.publickey}, 2, rng);
  const BigUint w = rng.below(group->modulus());
  ShareBox s1 = *p1.extract_secret_share(box, p1.privatekey, w);
  ShareBox s3 = *p3.extract_secret_share(box, p3.privatekey, w);
  CHECK(*dealer.reconstruct({s1, s3}, box) == secret);
}

static void test_dleq_verify(std::shared_ptr<ModpGroup> group) {            // dleq.rs:406-441
  Rng rng(3);
  const BigUint q = group->modulus();
  const BigUint alpha = rng.below(q), w = rng.below(q);
  const BigUint g1(8443), g2(1299721);
  DLEQ dleq(group);
  dleq.init(g1, group->exp(g1, alpha), g2, group->exp(g2, alpha), alpha, w);
  uint8_t st[MPVSS_TRANSCRIPT_STATE_BYTES];
  mpvss_transcript_init(st);
  Bytes four;
  append(four, be256(dleq.h1)); append(four, be256(dleq.h2)); append(four, be256(dleq.get_a1())); append(four, be256(dleq.get_a2()));
  mpvss_modp_transcript_absorb(st, four.data(), 4);
  uint8_t digest[32]; int v = 0; Bytes zero(256, 0);
  mpvss_modp_transcript_verdict(st, zero.data(), &v, digest);
  dleq.c = group->hash_to_scalar(Bytes(digest, digest + 32));
  dleq.r = dleq.get_r();
  CHECK(dleq.verify());
  dleq.r = *dleq.r ^ BigUint(1);
  CHECK(!dleq.verify());
  DLEQ empty(group);
  CHECK(!empty.verify());                                                   // c / r missing -> false (dleq.rs:280-287)
}

static void test_group_basics(std::shared_ptr<ModpGroup> group) {          // modp.rs:243-268, dleq.rs:380-403
  CHECK(group->exp(group->generator(), BigUint(1)) == group->generator());
  CHECK(group->exp(group->generator(), BigUint()) == group->identity());
  CHECK(group->mul(BigUint(5), BigUint(3)) == BigUint(15));
  const Bytes data = {'t', 'e', 's', 't', ' ', 'd', 'a', 't', 'a'};
  CHECK(group->hash_to_scalar(data) < group->subgroup_order());
  DLEQ d(group);
  d.init(BigUint(8443), BigUint(531216), BigUint(1299721), BigUint(14767239), BigUint(163027), BigUint(81647));
  d.c = BigUint(127997);
  const BigUint expect = (BigUint(81647) + group->order()) - (BigUint(163027) * BigUint(127997)) % group->order();
  CHECK(*d.get_r() == expect % group->order());
  CHECK(d.get_a1() == BigUint::modpow(BigUint(8443), BigUint(81647), group->modulus()));
  bool threw = false;
  try {
    Rng r(9);
    Participant dealer = Participant::with_arc(group);
    dealer.distribute_secret(BigUint(1), {BigUint(4)}, 2, r);              // threshold > n panics (participant.rs:166)
  } catch (const std::logic_error&) { threw = true; }
  CHECK(threw);
}

static void test_wire_format_round_trip(std::shared_ptr<ModpGroup> group) {
  Rng rng(77);
  Participant dealer = Participant::with_arc(group);
  dealer.initialize(rng);
  std::vector<Participant> ps;
  std::vector<BigUint> keys;
  for (int i = 0; i < 4; ++i) {
    ps.push_back(Participant::with_arc(group));
    ps.back().initialize(rng);
    keys.push_back(ps.back().publickey);
  }
  const BigUint secret = BigUint::from_bytes_be(reinterpret_cast<const uint8_t*>("wire"), 4);
  DistributionSharesBox box = dealer.distribute_secret(secret, keys, 3, rng);
  Bytes flat;
  CHECK(serialize_box(box, *group, flat));
  CHECK(flat.size() == 40 + 3 * 256 + 4 * 8 + 3 * 4 * 256 + 256 + box.U.to_bytes_be().size());
  DistributionSharesBox back;
  CHECK(parse_box(flat, *group, back));
  CHECK(back.commitments == box.commitments && back.publickeys == box.publickeys && back.positions == box.positions);
  CHECK(back.shares == box.shares && back.responses == box.responses && back.challenge == box.challenge && back.U == box.U);
  CHECK(ps[0].verify_distribution_shares(back));                 // the parsed box verifies on the GPU
  Bytes again;
  CHECK(serialize_box(back, *group, again) && again == flat);     // canonical: one encoding per box
  flat[flat.size() - 1] ^= 1;                                     // U changes, the proofs still hold
  CHECK(parse_box(flat, *group, back) && ps[0].verify_distribution_shares(back) && !(back.U == box.U));
  flat.pop_back();
  CHECK(!parse_box(flat, *group, back));                          // truncated
  DistributionSharesBox missing = box;
  missing.responses.erase(missing.responses.begin());
  CHECK(!serialize_box(missing, *group, again));
}

int main() {
  mpvss_process_init();   // before the first HIP call: 8 hardware queues for the block pipeline
  auto group = ModpGroup::create();
  test_wire_format_round_trip(group);
  test_group_basics(group);
  test_dleq_verify(group);
  test_end_to_end_modp(group);
  test_threshold_subset_positions_1_and_3(group);
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("host mirror tests: all passed\n");
  return 0;
}
