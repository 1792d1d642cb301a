// Host side of the drop-in: a C++ mirror of the reference crate's operator interface for the MODP
// group -- `Group` trait (src/group.rs:24-124), `ModpGroup` (src/groups/modp.rs), `DLEQ` (src/dleq.rs),
// `Polynomial` (src/polynomial.rs), `ShareBox` / `DistributionSharesBox` (src/sharebox.rs) and
// `Participant<ModpGroup>` (src/participant.rs:63-562) -- with the same method names, argument meaning
// and error behaviour, sitting on top of the C ABI (include/mpvss_hip.h).  It plays the role the Rust
// host code plays in the reference (this image has no Rust toolchain; INTEGRATION.md shows the Rust
// binding).  Every group exponentiation goes to the GPU engine; the three loops of the hot path
// (distribute_secret, verify_distribution_shares, verify_share) use the batched entry points.
#pragma once
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mpvss_hip.h"
#include "biguint.hpp"

namespace mpvss_host {

typedef std::vector<uint8_t> Bytes;

// ---- engine handle (the Arc<G> of the reference holds one of these) --------------------------------
class HipEngine {
 public:
  explicit HipEngine(int device = 0) {
    if (mpvss_ctx_create(device, &ctx_) != MPVSS_OK) throw std::runtime_error("mpvss_ctx_create failed: no HIP device (the engine has no CPU fallback)");
  }
  ~HipEngine() { mpvss_ctx_destroy(ctx_); }
  HipEngine(const HipEngine&) = delete;
  mpvss_ctx* ctx() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc != MPVSS_OK) throw std::runtime_error(std::string(what) + ": " + mpvss_last_error(ctx_));
  }
 private:
  mpvss_ctx* ctx_ = nullptr;
};

// deterministic stand-in for rand::thread_rng (tests need reproducibility)
class Rng {
 public:
  explicit Rng(uint64_t seed = std::random_device{}()) : gen_(seed) {}
  BigUint below(const BigUint& bound) {          // RandBigInt::gen_biguint_below
    const size_t nb = bound.bits();
    while (true) {
      BigUint r;
      r.d.assign((nb + 31) / 32, 0);
      for (auto& w : r.d) w = (uint32_t)gen_();
      if (nb % 32) r.d.back() &= (1u << (nb % 32)) - 1;
      while (!r.d.empty() && r.d.back() == 0) r.d.pop_back();
      if (r < bound) return r;
    }
  }
 private:
  std::mt19937_64 gen_;
};

inline Bytes be256(const BigUint& v) { return v.to_fixed_be(MPVSS_MODP_BYTES); }
inline void append(Bytes& dst, const Bytes& src) { dst.insert(dst.end(), src.begin(), src.end()); }

// ---- ModpGroup: src/groups/modp.rs:29-197 ---------------------------------------------------------------
class ModpGroup {
 public:
  typedef BigUint Scalar;
  typedef BigUint Element;

  static std::shared_ptr<ModpGroup> create(int device = 0) { return std::shared_ptr<ModpGroup>(new ModpGroup(device)); }

  const Scalar& order() const { return q_minus_1_; }                 // modp.rs:101-103
  const Scalar& subgroup_order() const { return g_; }                // modp.rs:105-107
  Element generator() const { return G_; }                           // modp.rs:109-111
  Element subgroup_generator() const { return g_gen_; }              // modp.rs:113-116
  Element identity() const { return BigUint(1); }                    // modp.rs:118-120
  const BigUint& modulus() const { return q_; }                      // modp.rs:194-196

  Element exp(const Element& base, const Scalar& scalar) const {     // modp.rs:122-128
    Bytes b = be256(base), e = be256(scalar), out(MPVSS_MODP_BYTES);
    eng_.check(mpvss_modp_batch_exp(eng_.ctx(), MPVSS_HOST, b.data(), e.data(), 1, out.data()), "exp");
    return BigUint::from_bytes_be(out);
  }
  Element mul(const Element& a, const Element& b) const {            // modp.rs:130-132
    Bytes x = be256(a), y = be256(b), out(MPVSS_MODP_BYTES);
    eng_.check(mpvss_modp_batch_mul(eng_.ctx(), MPVSS_HOST, x.data(), y.data(), 1, out.data()), "mul");
    return BigUint::from_bytes_be(out);
  }
  std::optional<Scalar> scalar_inverse(const Scalar& x) const {      // modp.rs:134-136
    BigUint r;
    if (!BigUint::mod_inverse(x, q_minus_1_, r)) return std::nullopt;
    return r;
  }
  std::optional<Element> element_inverse(const Element& x) const {   // modp.rs:138-140
    BigUint r;
    if (!BigUint::mod_inverse(x, q_, r)) return std::nullopt;
    return r;
  }
  Scalar hash_to_scalar(const Bytes& data) const {                   // modp.rs:142-148
    uint8_t out[MPVSS_MODP_BYTES];
    mpvss_modp_hash_to_scalar(data.data(), data.size(), out);
    return BigUint::from_bytes_be(out, sizeof(out)) % g_;
  }
  Bytes element_to_bytes(const Element& e) const { return e.to_bytes_be(); }                 // modp.rs:150-152
  std::optional<Element> bytes_to_element(const Bytes& b) const { return BigUint::from_bytes_be(b); }   // :154-156
  Bytes scalar_to_bytes(const Scalar& s) const { return s.to_bytes_be(); }                   // modp.rs:158-160
  Scalar generate_private_key(Rng& rng) const {                      // modp.rs:162-174
    while (true) {
      BigUint k = rng.below(q_);
      if (BigUint::gcd(k, q_minus_1_) == BigUint(1)) return k;
    }
  }
  Element generate_public_key(const Scalar& priv) const { return exp(G_, priv); }            // modp.rs:176-178
  Scalar scalar_mul(const Scalar& a, const Scalar& b) const { return (a * b) % q_minus_1_; } // modp.rs:180-182
  Scalar scalar_sub(const Scalar& a, const Scalar& b) const {                                // modp.rs:184-192
    if (a >= b) return (a - b) % q_minus_1_;
    return (a + q_minus_1_) - b;   // a - b + order (operands are reduced on every reference call path)
  }
  const HipEngine& engine() const { return eng_; }

 private:
  explicit ModpGroup(int device) : eng_(device) {
    q_ = BigUint::from_hex(                                          // modp.rs:47-58
        "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404dd"
        "ef9519b3cd3a431b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7ed"
        "ee386bfb5a899fa5ae9f24117c4b1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f"
        "83655d23dca3ad961c62f356208552bb9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3b"
        "e39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6955817183995497cea956ae515d2261898fa0510"
        "15728e5a8aacaa68ffffffffffffffff");
    q_minus_1_ = q_ - BigUint(1);
    g_ = q_minus_1_.shr(1);
    G_ = BigUint(2);
    g_gen_ = BigUint(4);
  }
  HipEngine eng_;
  BigUint q_, g_, G_, g_gen_, q_minus_1_;
};

// ---- Polynomial: src/polynomial.rs:18-59 ------------------------------------------------------------------
struct Polynomial {
  std::vector<BigUint> coefficients;
  void init_coefficients(const std::vector<BigUint>& c) { coefficients = c; }
  void init(int degree, const BigUint& q, Rng& rng) {                // polynomial.rs:34-47
    coefficients.clear();
    for (int i = 0; i <= degree; ++i) coefficients.push_back(rng.below(q));
  }
  BigUint get_value(const BigUint& x) const {                        // polynomial.rs:50-58 (over the integers)
    BigUint result = coefficients[0], xp(1);
    for (size_t i = 1; i < coefficients.size(); ++i) {
      xp = xp * x;
      result = result + coefficients[i] * xp;
    }
    return result;
  }
};

// ---- share boxes: src/sharebox.rs:21-134 -------------------------------------------------------------------
struct ShareBox {
  BigUint publickey, share, challenge, response;
};
struct DistributionSharesBox {
  std::vector<BigUint> commitments;
  std::map<Bytes, int64_t> positions;
  std::map<Bytes, BigUint> shares;
  std::vector<BigUint> publickeys;
  BigUint challenge;
  std::map<Bytes, BigUint> responses;
  BigUint U;
};

// ---- flat wire format (the reference has none; sharebox.rs:21-27,74-86 are HashMap-keyed in memory) ------
// The positions-ordered structure-of-arrays the C ABI consumes, so a box can be streamed to the GPUs (or sharded by
// byte ranges across ranks) without rebuilding maps:
//   "MPVSSBX1" | u32 group (0 = MODP-2048) | u32 element bytes | u64 n | u64 t | u64 U bytes
//   commitments [t][256] | positions i64-LE [n] | publickeys [n][256] | shares [n][256] | responses [n][256]
//   challenge [256] | U (big-endian, U bytes)
// Rows are in `publickeys` order (the order the reference iterates in, participant.rs:408-448).
inline void put_u64(Bytes& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
inline uint64_t get_u64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

// false when an entry of the maps is missing (the box would not verify either, participant.rs:415-420)
inline bool serialize_box(const DistributionSharesBox& box, const ModpGroup& group, Bytes& out) {
  out.clear();
  const Bytes ub = box.U.to_bytes_be();
  const char magic[8] = {'M', 'P', 'V', 'S', 'S', 'B', 'X', '1'};
  out.insert(out.end(), magic, magic + 8);
  for (uint32_t v : {0u, (uint32_t)MPVSS_MODP_BYTES}) for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(v >> (8 * i)));
  put_u64(out, box.publickeys.size());
  put_u64(out, box.commitments.size());
  put_u64(out, ub.size());
  for (auto& c : box.commitments) append(out, be256(c));
  Bytes sh, rs;
  for (auto& pk : box.publickeys) {
    const Bytes key = group.element_to_bytes(pk);
    const auto p = box.positions.find(key);
    const auto s = box.shares.find(key);
    const auto r = box.responses.find(key);
    if (p == box.positions.end() || s == box.shares.end() || r == box.responses.end()) return false;
    put_u64(out, (uint64_t)p->second);
    append(sh, be256(s->second));
    append(rs, be256(r->second));
  }
  for (auto& pk : box.publickeys) append(out, be256(pk));
  append(out, sh);
  append(out, rs);
  append(out, be256(box.challenge));
  append(out, ub);
  return true;
}

inline bool parse_box(const Bytes& in, const ModpGroup& group, DistributionSharesBox& box) {
  const size_t EBn = MPVSS_MODP_BYTES;
  if (in.size() < 40 || memcmp(in.data(), "MPVSSBX1", 8) != 0) return false;
  if (in[8] != 0 || in[9] != 0 || in[10] != 0 || in[11] != 0) return false;               // group: MODP-2048
  if ((uint32_t)(in[12] | in[13] << 8 | in[14] << 16 | (uint32_t)in[15] << 24) != EBn) return false;
  const uint64_t n = get_u64(&in[16]), t = get_u64(&in[24]), ulen = get_u64(&in[32]);
  if (n > (1ull << 32) || t > (1ull << 32) || ulen > (1ull << 20)) return false;
  const size_t need = 40 + t * EBn + n * 8 + 3 * n * EBn + EBn + ulen;
  if (in.size() != need) return false;
  const uint8_t* p = in.data() + 40;
  box = DistributionSharesBox();
  for (uint64_t j = 0; j < t; ++j, p += EBn) box.commitments.push_back(BigUint::from_bytes_be(p, EBn));
  const uint8_t* pos = p;
  const uint8_t* pk = pos + n * 8;
  const uint8_t* sh = pk + n * EBn;
  const uint8_t* rs = sh + n * EBn;
  for (uint64_t i = 0; i < n; ++i) {
    BigUint key = BigUint::from_bytes_be(pk + i * EBn, EBn);
    const Bytes kb = group.element_to_bytes(key);
    box.positions[kb] = (int64_t)get_u64(pos + i * 8);
    box.shares[kb] = BigUint::from_bytes_be(sh + i * EBn, EBn);
    box.responses[kb] = BigUint::from_bytes_be(rs + i * EBn, EBn);
    box.publickeys.push_back(std::move(key));
  }
  box.challenge = BigUint::from_bytes_be(rs + n * EBn, EBn);
  box.U = BigUint::from_bytes_be(rs + n * EBn + EBn, ulen);
  return true;
}

// ---- DLEQ: src/dleq.rs:153-335 ----------------------------------------------------------------------------
struct DLEQ {
  BigUint g1, h1, g2, h2, w, alpha;
  std::optional<BigUint> c, r;
  std::shared_ptr<ModpGroup> group;
  explicit DLEQ(std::shared_ptr<ModpGroup> g) : group(std::move(g)) {}
  void init(const BigUint& g1_, const BigUint& h1_, const BigUint& g2_, const BigUint& h2_, const BigUint& alpha_,
            const BigUint& w_) { g1 = g1_; h1 = h1_; g2 = g2_; h2 = h2_; alpha = alpha_; w = w_; }
  BigUint get_a1() const { return group->exp(g1, w); }               // dleq.rs:207-211
  BigUint get_a2() const { return group->exp(g2, w); }               // dleq.rs:213-216
  std::optional<BigUint> get_r() const {                             // dleq.rs:221-228
    if (!c) return std::nullopt;
    return group->scalar_sub(w, group->scalar_mul(alpha, *c));
  }
  bool verify() const {                                              // dleq.rs:275-302
    if (!c || !r) return false;
    Bytes pk = be256(h1), s = be256(g2), y = be256(h2), cc = be256(*c), rr = be256(*r);
    uint8_t verdict = 0;
    // g1 must be the main generator on this path (participant.rs:361-386); other bases use the generic call
    if (g1 == group->generator()) {
      group->engine().check(mpvss_modp_verify_shares(group->engine().ctx(), MPVSS_HOST, pk.data(), s.data(), y.data(),
                                                     cc.data(), rr.data(), 1, &verdict), "DLEQ::verify");
      return verdict == 1;
    }
    Bytes g1b = be256(g1), a1(MPVSS_MODP_BYTES), a2(MPVSS_MODP_BYTES);
    group->engine().check(mpvss_modp_dleq_commitments(group->engine().ctx(), MPVSS_HOST, g1b.data(), pk.data(), s.data(),
                                                      y.data(), rr.data(), cc.data(), 0, 1, a1.data(), a2.data()),
                          "DLEQ::verify");
    uint8_t st[MPVSS_TRANSCRIPT_STATE_BYTES];
    mpvss_transcript_init(st);
    Bytes four;
    append(four, pk); append(four, y); append(four, a1); append(four, a2);
    mpvss_modp_transcript_absorb(st, four.data(), 4);
    int v = 0;
    mpvss_modp_transcript_verdict(st, cc.data(), &v, nullptr);
    return v == 1;
  }
};

inline BigUint string_to_secret(const std::string& m) {             // lib.rs:49-53
  return BigUint::from_bytes_be((const uint8_t*)m.data(), m.size());
}
inline std::string string_from_secret(const BigUint& s) {           // lib.rs:55-57
  Bytes b = s.to_bytes_be();
  return std::string(b.begin(), b.end());
}

// ---- Participant<ModpGroup>: src/participant.rs:63-562 -------------------------------------------------------
class Participant {
 public:
  std::shared_ptr<ModpGroup> group;
  BigUint privatekey, publickey;

  static Participant with_arc(std::shared_ptr<ModpGroup> g) { Participant p; p.group = std::move(g); return p; }   // :98
  void initialize(Rng& rng) {                                        // participant.rs:139-146
    privatekey = group->generate_private_key(rng);
    publickey = group->generate_public_key(privatekey);
  }

  // participant.rs:160-286.  Panics (throws) when threshold > publickeys.len() (:166).
  DistributionSharesBox distribute_secret(const BigUint& secret, const std::vector<BigUint>& publickeys, uint32_t threshold,
                                          Rng& rng) const {
    if (threshold > publickeys.size()) throw std::logic_error("assertion failed: threshold <= publickeys.len()");
    const BigUint& order = group->order();
    Polynomial polynomial;
    polynomial.init((int)threshold - 1, order, rng);
    const size_t n = publickeys.size();
    const HipEngine& eng = group->engine();
    // commitments C_j = g^a_j                                         :189-193
    Bytes coeffs, cm(threshold * MPVSS_MODP_BYTES);
    for (auto& a : polynomial.coefficients) append(coeffs, be256(a));
    Bytes gbytes = be256(group->subgroup_generator());
    eng.check(mpvss_modp_batch_exp_fixed_base(eng.ctx(), MPVSS_HOST, gbytes.data(), coeffs.data(), threshold, cm.data()),
              "distribute_secret: commitments");
    // per participant: position and witness; P(i) mod order (:200-202), X_i, Y_i, a1_i, a2_i (:207-249), the transcript
    // digest and the challenge (:251-252) and the responses r_i = w_i - P(i) c (:255-264) are ONE engine call
    DistributionSharesBox box;
    std::vector<int64_t> pos(n);
    Bytes pk, ws;
    for (size_t i = 0; i < n; ++i) {
      pos[i] = (int64_t)i + 1;
      append(pk, be256(publickeys[i]));
      append(ws, be256(group->generate_private_key(rng)));
    }
    Bytes Y(n * MPVSS_MODP_BYTES), R(Y.size()), cbytes(MPVSS_MODP_BYTES);
    uint8_t digest[32];
    eng.check(mpvss_modp_deal(eng.ctx(), coeffs.data(), threshold, pos.data(), pk.data(), ws.data(), n, nullptr, Y.data(), nullptr,
                              nullptr, digest, cbytes.data(), R.data()), "distribute_secret");
    const BigUint challenge = BigUint::from_bytes_be(cbytes.data(), MPVSS_MODP_BYTES);
    if (!(challenge == group->hash_to_scalar(Bytes(digest, digest + 32)))) throw std::logic_error("distribute_secret: challenge");
    for (uint32_t j = 0; j < threshold; ++j)
      box.commitments.push_back(BigUint::from_bytes_be(cm.data() + j * MPVSS_MODP_BYTES, MPVSS_MODP_BYTES));
    for (size_t i = 0; i < n; ++i) {
      const Bytes key = group->element_to_bytes(publickeys[i]);
      box.positions[key] = pos[i];
      box.shares[key] = BigUint::from_bytes_be(Y.data() + i * MPVSS_MODP_BYTES, MPVSS_MODP_BYTES);
      box.responses[key] = BigUint::from_bytes_be(R.data() + i * MPVSS_MODP_BYTES, MPVSS_MODP_BYTES);
    }
    box.publickeys = publickeys;
    box.challenge = challenge;
    // U = secret XOR (SHA256(bytes(G^s)) mod q)                        :267-272
    const BigUint s = polynomial.get_value(BigUint()) % order;
    const BigUint g_s = group->exp(group->generator(), s);
    const Bytes gb = group->element_to_bytes(g_s);
    uint8_t h[32];
    mpvss_sha256(gb.data(), gb.size(), h);
    box.U = secret ^ (BigUint::from_bytes_be(h, 32) % group->modulus());
    return box;
  }

  // participant.rs:294-353
  std::optional<ShareBox> extract_secret_share(const DistributionSharesBox& box, const BigUint& private_key,
                                               const BigUint& w) const {
    const BigUint public_key = group->generate_public_key(private_key);
    const auto it = box.shares.find(group->element_to_bytes(public_key));
    if (it == box.shares.end()) return std::nullopt;
    const BigUint& enc = it->second;
    BigUint inv;
    if (!BigUint::mod_inverse(private_key, group->order(), inv)) return std::nullopt;
    const BigUint share = group->exp(enc, inv);
    DLEQ dleq(group);
    dleq.init(group->generator(), public_key, share, enc, private_key, w);
    const BigUint a1 = dleq.get_a1(), a2 = dleq.get_a2();
    uint8_t st[MPVSS_TRANSCRIPT_STATE_BYTES];
    mpvss_transcript_init(st);
    Bytes four;
    append(four, be256(public_key)); append(four, be256(enc)); append(four, be256(a1)); append(four, be256(a2));
    mpvss_modp_transcript_absorb(st, four.data(), 4);                // dleq.rs:87-99
    uint8_t digest[32];
    int unused = 0;
    Bytes zero(MPVSS_MODP_BYTES, 0);
    mpvss_modp_transcript_verdict(st, zero.data(), &unused, digest);
    dleq.c = group->hash_to_scalar(Bytes(digest, digest + 32));
    const auto r = dleq.get_r();
    if (!r) return std::nullopt;
    return ShareBox{public_key, share, *dleq.c, *r};
  }

  // participant.rs:361-386
  bool verify_share(const ShareBox& sb, const DistributionSharesBox& box, const BigUint& publickey_) const {
    const auto it = box.shares.find(group->element_to_bytes(publickey_));
    if (it == box.shares.end()) return false;
    DLEQ dleq(group);
    dleq.g1 = group->generator();
    dleq.h1 = publickey_;
    dleq.g2 = sb.share;
    dleq.h2 = it->second;
    dleq.c = sb.challenge;
    dleq.r = sb.response;
    return dleq.verify();
  }

  // participant.rs:399-455 (= PVSS::verify_distribution_shares, mpvss.rs:90-144)
  bool verify_distribution_shares(const DistributionSharesBox& box) const {
    Bytes cm, pk, sh, rs;
    std::vector<int64_t> pos;
    for (auto& c : box.commitments) append(cm, be256(c));
    for (auto& y : box.publickeys) {
      const Bytes key = group->element_to_bytes(y);
      const auto p = box.positions.find(key);
      const auto r = box.responses.find(key);
      const auto s = box.shares.find(key);
      if (p == box.positions.end() || r == box.responses.end() || s == box.shares.end()) return false;   // :415-420
      pos.push_back(p->second);
      append(pk, be256(y)); append(sh, be256(s->second)); append(rs, be256(r->second));
    }
    int verdict = 0;
    const Bytes ch = be256(box.challenge);
    const HipEngine& eng = group->engine();
    eng.check(mpvss_modp_verify_distribution(eng.ctx(), MPVSS_HOST, cm.data(), box.commitments.size(), pos.data(), pk.data(),
                                             sh.data(), rs.data(), pos.size(), ch.data(), &verdict, nullptr, nullptr, nullptr,
                                             nullptr), "verify_distribution_shares");
    return verdict == 1;
  }

  // participant.rs:462-561
  std::optional<BigUint> reconstruct(const std::vector<ShareBox>& share_boxes, const DistributionSharesBox& box) const {
    if (share_boxes.size() < box.commitments.size()) return std::nullopt;
    std::map<int64_t, BigUint> shares;
    for (auto& sb : share_boxes) {
      const auto p = box.positions.find(group->element_to_bytes(sb.publickey));
      if (p == box.positions.end()) return std::nullopt;
      shares[p->second] = sb.share;
    }
    const BigUint& sub = group->subgroup_order();
    Bytes bases, exps;
    std::vector<bool> negative;
    for (auto& kv : shares) {
      // util.rs:47-64 lagrange_coefficient as |num| / |den| with a sign
      BigUint num(1), den(1);
      bool neg = false;
      for (auto& other : shares) {
        if (other.first == kv.first) continue;
        num = num * BigUint((uint64_t)other.first);
        const int64_t diff = other.first - kv.first;
        if (diff < 0) neg = !neg;
        den = den * BigUint((uint64_t)(diff < 0 ? -diff : diff));
      }
      const BigUint g = BigUint::gcd(num, den);
      num = num / g;
      den = den / g;
      BigUint den_inv;
      if (!BigUint::mod_inverse(den, sub, den_inv)) return std::nullopt;
      append(bases, be256(kv.second));
      append(exps, be256((num * den_inv) % sub));
      negative.push_back(neg);
    }
    const size_t m = shares.size();
    Bytes factors(m * MPVSS_MODP_BYTES);
    const HipEngine& eng = group->engine();
    eng.check(mpvss_modp_batch_exp(eng.ctx(), MPVSS_HOST, bases.data(), exps.data(), m, factors.data()), "reconstruct");
    BigUint secret = group->identity();
    for (size_t i = 0; i < m; ++i) {
      BigUint f = BigUint::from_bytes_be(factors.data() + i * MPVSS_MODP_BYTES, MPVSS_MODP_BYTES);
      if (negative[i]) {
        const auto inv = group->element_inverse(f);
        if (!inv) return std::nullopt;
        f = *inv;
      }
      secret = (secret * f) % group->modulus();
    }
    const Bytes gb = group->element_to_bytes(secret);
    uint8_t h[32];
    mpvss_sha256(gb.data(), gb.size(), h);
    return (BigUint::from_bytes_be(h, 32) % group->modulus()) ^ box.U;   // :512-517
  }
};

}  // namespace mpvss_host
