// Host side of the drop-in for the reference's two curve groups: C++ mirror of `Secp256k1Group`
// (src/groups/secp256k1.rs:38-189), `Ristretto255Group` (src/groups/ristretto255.rs:45-253) and the two hand-specialised
// `impl Participant<...>` blocks (src/participant.rs:1085-1558, 1564-2003), with the same method names, argument
// meaning and error behaviour, on top of the C ABI (include/mpvss_hip.h).  Counterpart of mpvss_host.hpp (MODP).
//
// An element is its canonical encoding (33-byte SEC1 compressed / 32-byte ristretto255): the host never touches curve
// arithmetic -- every group operation goes to the GPU engine, the three hot loops through the batched entry points.
// Scalars are BigUint values below the group order; the byte order of the boundary (big-endian for secp256k1,
// little-endian for ristretto255: secp256k1.rs:154-156, ristretto255.rs:222-225) is applied when they cross it.
#pragma once
#include <algorithm>

#include "mpvss_host.hpp"

namespace mpvss_host {

struct Secp256k1Traits {
  static constexpr int GROUP = MPVSS_GROUP_SECP256K1;
  static constexpr size_t ENC = 33;
  static constexpr bool SCALAR_BE = true;
  static const char* order_hex() { return "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141"; }   // secp256k1.rs:47-51
  static const char* generator_hex() { return "0279be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"; }
};
struct Ristretto255Traits {
  static constexpr int GROUP = MPVSS_GROUP_RISTRETTO255;
  static constexpr size_t ENC = 32;
  static constexpr bool SCALAR_BE = false;
  static const char* order_hex() { return "1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3ed"; }   // ristretto255.rs:55-59
  static const char* generator_hex() { return "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"; }
};

inline Bytes hex_bytes(const char* h) {
  Bytes out;
  for (size_t i = 0; h[i] && h[i + 1]; i += 2) out.push_back((uint8_t)strtoul(std::string(h + i, 2).c_str(), nullptr, 16));
  return out;
}

template <class T>
class EcGroup {
 public:
  typedef BigUint Scalar;
  typedef Bytes Element;
  static constexpr size_t ENC = T::ENC;

  static std::shared_ptr<EcGroup> create(int device = 0) { return std::shared_ptr<EcGroup>(new EcGroup(device)); }

  const Scalar& order() const { return n_; }                                       // order_as_bigint, secp256k1.rs:186-188
  const Scalar& subgroup_order() const { return n_; }
  Element generator() const { return gen_; }                                       // secp256k1.rs:78-80, ristretto255.rs:148-150
  Element subgroup_generator() const { return gen_; }                              // :82-85 / :152-155
  Element identity() const { return Bytes(ENC, 0); }                               // :87-89 / :157-159 (33 / 32 zero bytes)

  Bytes scalar_bytes(const Scalar& s) const {                                      // scalar_to_bytes: 32 bytes
    Bytes b = s.to_fixed_be(32);
    if (!T::SCALAR_BE) std::reverse(b.begin(), b.end());
    return b;
  }
  Scalar scalar_from(const uint8_t* b) const {
    Bytes t(b, b + 32);
    if (!T::SCALAR_BE) std::reverse(t.begin(), t.end());
    return BigUint::from_bytes_be(t);
  }
  Element exp(const Element& base, const Scalar& scalar) const {                    // secp256k1.rs:91-100, ristretto255.rs:161-170
    Bytes k = scalar_bytes(scalar % n_), out(ENC);
    eng_.check(mpvss_ec_batch_exp(eng_.ctx(), T::GROUP, MPVSS_HOST, base.data(), k.data(), 1, out.data()), "exp");
    return out;
  }
  Element mul(const Element& a, const Element& b) const {                           // :102-107 / :172-177
    Bytes out(ENC);
    eng_.check(mpvss_ec_batch_mul(eng_.ctx(), T::GROUP, MPVSS_HOST, a.data(), b.data(), 1, out.data()), "mul");
    return out;
  }
  std::optional<Scalar> scalar_inverse(const Scalar& x) const {                     // :109-112 / :179-187
    BigUint r;
    if (x.is_zero() || !BigUint::mod_inverse(x, n_, r)) return std::nullopt;
    return r;
  }
  std::optional<Element> element_inverse(const Element& x) const {                  // point negation, :114-119 / :189-194
    return exp(x, n_ - BigUint(1));
  }
  Scalar hash_to_scalar(const Bytes& data) const {                                  // :121-131 / :196-205
    uint8_t out[32];
    mpvss_ec_hash_to_scalar(T::GROUP, data.data(), data.size(), out);
    return scalar_from(out);
  }
  Bytes element_to_bytes(const Element& e) const { return e; }                      // :133-136 / :207-210
  std::optional<Element> bytes_to_element(const Bytes& b) const {                   // :138-152 / :212-220: validated by decoding
    if (b.size() != ENC) return std::nullopt;
    Bytes out(ENC);
    if (mpvss_ec_batch_mul(eng_.ctx(), T::GROUP, MPVSS_HOST, b.data(), identity().data(), 1, out.data()) != MPVSS_OK) return std::nullopt;
    return b;
  }
  Bytes scalar_to_bytes(const Scalar& s) const { return scalar_bytes(s); }          // :154-156 / :222-225
  Scalar generate_private_key(Rng& rng) const {                                     // :158-166 / :227-237: 32 random bytes, reduced
    return rng.below(BigUint(1).shl(256)) % n_;
  }
  Element generate_public_key(const Scalar& priv) const {                           // :168-171 / :239-242, fixed-base comb
    Bytes k = scalar_bytes(priv), out(ENC);
    eng_.check(mpvss_ec_batch_exp_generator(eng_.ctx(), T::GROUP, MPVSS_HOST, k.data(), 1, out.data()), "generate_public_key");
    return out;
  }
  Scalar scalar_mul(const Scalar& a, const Scalar& b) const {                       // :173-176 / :244-247
    Bytes x = scalar_bytes(a), y = scalar_bytes(b), out(32);
    mpvss_ec_scalar_mul(T::GROUP, x.data(), y.data(), out.data());
    return scalar_from(out.data());
  }
  Scalar scalar_sub(const Scalar& a, const Scalar& b) const {                       // :178-181 / :249-252
    Bytes x = scalar_bytes(a), y = scalar_bytes(b), out(32);
    mpvss_ec_scalar_sub(T::GROUP, x.data(), y.data(), out.data());
    return scalar_from(out.data());
  }
  // BigInt coefficient -> scalar: secp256k1 right-aligns the big-endian bytes into 32 (participant.rs:1134-1143, the
  // coefficients are below the order); ristretto255's bigint_to_scalar reduces mod l (ristretto255.rs:78-105)
  Scalar scalar_from_bigint(const BigUint& v) const { return v % n_; }
  // int_BE(SHA256(bytes(e))) mod order: the mask XORed onto the secret (participant.rs:1246-1260, 1696-1703)
  BigUint secret_mask(const Element& e) const {
    uint8_t h[32];
    mpvss_sha256(e.data(), e.size(), h);
    return BigUint::from_bytes_be(h, 32) % n_;
  }
  const HipEngine& engine() const { return eng_; }

 private:
  explicit EcGroup(int device) : eng_(device), n_(BigUint::from_hex(T::order_hex())), gen_(hex_bytes(T::generator_hex())) {}
  HipEngine eng_;
  BigUint n_;
  Bytes gen_;
};

// src/sharebox.rs:21-134 with elements as encodings
struct EcShareBox {
  Bytes publickey, share;
  BigUint challenge, response;
};
struct EcDistributionSharesBox {
  std::vector<Bytes> commitments;
  std::map<Bytes, int64_t> positions;
  std::map<Bytes, Bytes> shares;
  std::vector<Bytes> publickeys;
  BigUint challenge;
  std::map<Bytes, BigUint> responses;
  BigUint U;
};

template <class T>
class EcParticipant {
 public:
  typedef EcGroup<T> Group;
  std::shared_ptr<Group> group;
  BigUint privatekey;
  Bytes publickey;

  static EcParticipant with_arc(std::shared_ptr<Group> g) { EcParticipant p; p.group = std::move(g); return p; }
  void initialize(Rng& rng) {                                                       // participant.rs:139-146
    privatekey = group->generate_private_key(rng);
    publickey = group->generate_public_key(privatekey);
  }

  // participant.rs:1094-1274 (secp256k1), 1573-1717 (ristretto255).  Throws when threshold > publickeys.len().
  EcDistributionSharesBox distribute_secret(const BigUint& secret, const std::vector<Bytes>& publickeys, uint32_t threshold,
                                            Rng& rng) const {
    if (threshold > publickeys.size()) throw std::logic_error("assertion failed: threshold <= publickeys.len()");
    const HipEngine& eng = group->engine();
    const size_t n = publickeys.size(), L = Group::ENC;
    Polynomial polynomial;
    polynomial.init((int)threshold - 1, group->order(), rng);                      // coefficients below the order
    Bytes coeffs, cm(threshold * L);
    for (auto& a : polynomial.coefficients) append(coeffs, group->scalar_bytes(group->scalar_from_bigint(a)));
    eng.check(mpvss_ec_batch_exp_generator(eng.ctx(), T::GROUP, MPVSS_HOST, coeffs.data(), threshold, cm.data()),
              "distribute_secret: commitments");                                   // C_j = a_j G
    std::vector<int64_t> pos(n);
    Bytes pk, ws, r(n * 32);
    std::vector<BigUint> wits(n);
    for (size_t i = 0; i < n; ++i) {
      pos[i] = (int64_t)i + 1;
      wits[i] = group->generate_private_key(rng);
      append(pk, publickeys[i]);
      append(ws, group->scalar_bytes(wits[i]));
    }
    // P(i) mod order (polynomial.rs:50-58 + `% order`), X_i, Y_i, a1_i, a2_i, the transcript digest, the challenge
    // c = hash_to_scalar(digest) and the responses r_i = w_i - P(i) c in ONE call, the scalar side on the device too
    Bytes Y(n * L), cb(32);
    uint8_t digest[32];
    eng.check(mpvss_ec_deal(eng.ctx(), T::GROUP, coeffs.data(), threshold, pos.data(), pk.data(), ws.data(), n, nullptr, Y.data(), nullptr,
                            nullptr, digest, cb.data(), r.data()), "distribute_secret");
    const BigUint challenge = group->hash_to_scalar(Bytes(digest, digest + 32));
    if (!(group->scalar_bytes(challenge) == cb)) throw std::logic_error("distribute_secret: challenge");
    EcDistributionSharesBox box;
    for (uint32_t j = 0; j < threshold; ++j) box.commitments.emplace_back(cm.begin() + j * L, cm.begin() + (j + 1) * L);
    for (size_t i = 0; i < n; ++i) {
      const Bytes& key = publickeys[i];
      box.positions[key] = pos[i];
      box.shares[key] = Bytes(Y.begin() + i * L, Y.begin() + (i + 1) * L);
      box.responses[key] = group->scalar_from(r.data() + i * 32);
    }
    box.publickeys = publickeys;
    box.challenge = challenge;
    // U = secret XOR (int(SHA256(bytes(s G))) mod order), s = P(0)
    const BigUint s = group->scalar_from_bigint(polynomial.get_value(BigUint()));
    box.U = secret ^ group->secret_mask(group->generate_public_key(s));
    return box;
  }

  // participant.rs:1282-1338, 1725-1781
  std::optional<EcShareBox> extract_secret_share(const EcDistributionSharesBox& box, const BigUint& private_key,
                                                 const BigUint& w) const {
    const Bytes public_key = group->generate_public_key(private_key);
    const auto it = box.shares.find(public_key);
    if (it == box.shares.end()) return std::nullopt;
    const auto inv = group->scalar_inverse(private_key);
    if (!inv) return std::nullopt;
    const HipEngine& eng = group->engine();
    const Bytes xi = group->scalar_bytes(*inv), wb = group->scalar_bytes(w % group->order());
    Bytes S(Group::ENC), c(32), r(32);
    eng.check(mpvss_ec_extract_shares(eng.ctx(), T::GROUP, MPVSS_HOST, public_key.data(), it->second.data(), xi.data(), wb.data(), 1,
                                      S.data(), c.data()), "extract_secret_share");
    const Bytes xb = group->scalar_bytes(private_key);
    mpvss_ec_dleq_responses(T::GROUP, wb.data(), xb.data(), c.data(), 1, 1, r.data(), 1);        // r = w - x c, dleq.rs:42-50
    return EcShareBox{public_key, S, group->scalar_from(c.data()), group->scalar_from(r.data())};
  }

  // participant.rs:1346-1371, 1789-1814
  bool verify_share(const EcShareBox& sb, const EcDistributionSharesBox& box, const Bytes& publickey_) const {
    const auto it = box.shares.find(publickey_);
    if (it == box.shares.end()) return false;
    const HipEngine& eng = group->engine();
    const Bytes c = group->scalar_bytes(sb.challenge), r = group->scalar_bytes(sb.response);
    uint8_t verdict = 0;
    if (mpvss_ec_verify_shares(eng.ctx(), T::GROUP, MPVSS_HOST, publickey_.data(), sb.share.data(), it->second.data(), c.data(),
                               r.data(), 1, &verdict) != MPVSS_OK)
      return false;                                        // an encoding the reference's types could not hold
    return verdict == 1;
  }

  // participant.rs:1384-1442, 1827-1885
  bool verify_distribution_shares(const EcDistributionSharesBox& box) const {
    Bytes cm, pk, sh, rs;
    std::vector<int64_t> pos;
    for (auto& c : box.commitments) append(cm, c);
    for (auto& y : box.publickeys) {
      const auto p = box.positions.find(y);
      const auto r = box.responses.find(y);
      const auto s = box.shares.find(y);
      if (p == box.positions.end() || r == box.responses.end() || s == box.shares.end()) return false;
      pos.push_back(p->second);
      append(pk, y); append(sh, s->second); append(rs, group->scalar_bytes(r->second));
    }
    int verdict = 0;
    const Bytes ch = group->scalar_bytes(box.challenge);
    const HipEngine& eng = group->engine();
    if (mpvss_ec_verify_distribution(eng.ctx(), T::GROUP, MPVSS_HOST, cm.data(), box.commitments.size(), pos.data(), pk.data(),
                                     sh.data(), rs.data(), pos.size(), ch.data(), &verdict, nullptr, nullptr, nullptr,
                                     nullptr) != MPVSS_OK)
      return false;
    return verdict == 1;
  }

  // participant.rs:1452-1557, 1895-2002
  std::optional<BigUint> reconstruct(const std::vector<EcShareBox>& share_boxes, const EcDistributionSharesBox& box) const {
    if (share_boxes.size() < box.commitments.size()) return std::nullopt;
    std::map<int64_t, Bytes> shares;
    for (auto& sb : share_boxes) {
      const auto p = box.positions.find(sb.publickey);
      if (p == box.positions.end()) return std::nullopt;
      shares[p->second] = sb.share;
    }
    std::vector<int64_t> pos;
    Bytes S;
    for (auto& kv : shares) { pos.push_back(kv.first); append(S, kv.second); }
    Bytes gs(Group::ENC);
    uint8_t mask[32];
    const HipEngine& eng = group->engine();
    if (mpvss_ec_reconstruct(eng.ctx(), T::GROUP, MPVSS_HOST, pos.data(), S.data(), pos.size(), gs.data(), mask) != MPVSS_OK)
      return std::nullopt;
    return BigUint::from_bytes_be(mask, 32) ^ box.U;
  }
};

typedef EcGroup<Secp256k1Traits> Secp256k1Group;
typedef EcGroup<Ristretto255Traits> Ristretto255Group;
typedef EcParticipant<Secp256k1Traits> Secp256k1Participant;
typedef EcParticipant<Ristretto255Traits> Ristretto255Participant;

}  // namespace mpvss_host
