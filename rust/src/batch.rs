//! The three loops of the hot path as single batched calls, written against the reference's own box types
//! (`mpvss_rs::sharebox::{DistributionSharesBox, ShareBox}`) for the MODP group; the curve groups follow the same
//! shape with `mpvss_ec_*` and 33 / 32-byte encodings (helpers at the bottom).
//!
//!   verify_distribution_shares   src/participant.rs:399-455  (= src/mpvss.rs:90-144)  -> mpvss_modp_verify_distribution
//!   many boxes                   one call per dealer in the reference               -> mpvss_modp_verify_many
//!   verify_share (n boxes)       src/participant.rs:361-386 -> src/dleq.rs:275-302   -> mpvss_modp_verify_shares
//!   distribute_secret            src/participant.rs:160-286                          -> mpvss_modp_distribute (+ scalar side)
//!   extract_secret_share (n)     src/participant.rs:294-353                          -> mpvss_modp_extract_shares
//!   reconstruct                  src/participant.rs:462-561                          -> mpvss_modp_reconstruct
//!
//! `crate::participant::Participant<G>` wraps these in the reference's method names.  Every function here may be called from
//! several threads at once on one engine (the library keeps one box per caller in flight: rayon over dealers, as
//! participant.rs:490-500 goes parallel over shares).
//!
//! Never compiled in this repository's environment (no Rust toolchain).
use mpvss_rs::group::Group;
use mpvss_rs::polynomial::Polynomial;
use mpvss_rs::sharebox::{DistributionSharesBox, ShareBox};
use num_bigint::{BigInt, BigUint, Sign};
use num_traits::Zero;

use crate::engine::EngineError;
use crate::ffi;
use crate::groups::{be256, HipModpGroup, HipRistretto255Group, HipSecp256k1Group};

/// Flat, positions-ordered view of a box: exactly the arrays the C ABI takes.
pub struct FlatBox {
    pub commitments: Vec<u8>,
    pub positions: Vec<i64>,
    pub pubkeys: Vec<u8>,
    pub shares: Vec<u8>,
    pub responses: Vec<u8>,
    pub challenge: [u8; 256],
}

/// `None` when an entry of the maps is missing for a listed public key: the reference returns `false` then
/// (participant.rs:415-420).
pub fn flatten(group: &HipModpGroup, bx: &DistributionSharesBox<HipModpGroup>) -> Option<FlatBox> {
    let n = bx.publickeys.len();
    let mut f = FlatBox {
        commitments: Vec::with_capacity(bx.commitments.len() * 256),
        positions: Vec::with_capacity(n),
        pubkeys: Vec::with_capacity(n * 256),
        shares: Vec::with_capacity(n * 256),
        responses: Vec::with_capacity(n * 256),
        challenge: be256(&bx.challenge),
    };
    for c in &bx.commitments {
        f.commitments.extend_from_slice(&be256(c));
    }
    for pk in &bx.publickeys {
        let key = group.element_to_bytes(pk);
        f.positions.push(*bx.positions.get(&key)?);
        f.shares.extend_from_slice(&be256(bx.shares.get(&key)?));
        f.responses.extend_from_slice(&be256(bx.responses.get(&key)?));
        f.pubkeys.extend_from_slice(&be256(pk));
    }
    Some(f)
}

/// Drop-in body of `Participant<ModpGroup>::verify_distribution_shares` / `PVSS::verify_distribution_shares`.
pub fn verify_distribution_shares(group: &HipModpGroup, bx: &DistributionSharesBox<HipModpGroup>) -> bool {
    let Some(f) = flatten(group, bx) else { return false };
    let mut verdict = 0i32;
    let rc = unsafe {
        ffi::mpvss_modp_verify_distribution(
            group.engine.raw(), ffi::MPVSS_HOST, f.commitments.as_ptr(), bx.commitments.len(), f.positions.as_ptr(),
            f.pubkeys.as_ptr(), f.shares.as_ptr(), f.responses.as_ptr(), f.positions.len(), f.challenge.as_ptr(),
            &mut verdict, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(),
        )
    };
    group.engine.expect(rc, "verify_distribution_shares");     // the reference panics on a negative position too
    verdict == 1
}

/// Every dealer's box at once: the library pipelines them (GPU work of up to `depth` boxes ahead, `hash_threads`
/// transcript hashers).  Boxes with a missing map entry get `false` without touching the GPU.
pub fn verify_many(group: &HipModpGroup, boxes: &[&DistributionSharesBox<HipModpGroup>], depth: i32, hash_threads: i32)
    -> Result<Vec<bool>, EngineError> {
    let flats: Vec<Option<FlatBox>> = boxes.iter().map(|b| flatten(group, b)).collect();
    let mut descs = Vec::new();
    let mut index = Vec::new();
    // Boxes of different dealers for the same participants carry the same public keys: hand the engine ONE array for them (the
    // first box's), so that its key cache (`Engine::set_key_cache`) recognises them -- it goes by pointer and n, never by content.
    let mut distinct: Vec<&Vec<u8>> = Vec::new();
    for (i, (f, b)) in flats.iter().zip(boxes).enumerate() {
        if let Some(f) = f {
            index.push(i);
            let keys: &Vec<u8> = match distinct.iter().find(|k| **k == &f.pubkeys) {
                Some(k) => *k,
                None => { distinct.push(&f.pubkeys); &f.pubkeys }
            };
            descs.push(ffi::mpvss_modp_box {
                commitments: f.commitments.as_ptr(), t: b.commitments.len(), positions: f.positions.as_ptr(),
                pubkeys: keys.as_ptr(), shares: f.shares.as_ptr(), responses: f.responses.as_ptr(), n: f.positions.len(),
                challenge_host: f.challenge.as_ptr(), keyset: std::ptr::null(), key_offset: 0,
            });
        }
    }
    let mut verdicts = vec![0i32; descs.len()];
    let rc = unsafe {
        ffi::mpvss_modp_verify_many(group.engine.raw(), ffi::MPVSS_HOST, descs.as_ptr(), descs.len(), depth, hash_threads,
                                    verdicts.as_mut_ptr(), std::ptr::null_mut())
    };
    group.engine.check(rc)?;
    let mut out = vec![false; boxes.len()];
    for (k, i) in index.into_iter().enumerate() {
        out[i] = verdicts[k] == 1;
    }
    Ok(out)
}

/// n calls of `Participant::verify_share` (participant.rs:361-386) against one distribution box: one verdict each.
/// `publickeys[i]` is the `publickey` argument of the i-th call: the key the encrypted share is looked up under and h1 of the proof.
pub fn verify_shares(group: &HipModpGroup, share_boxes: &[ShareBox<HipModpGroup>], publickeys: &[BigInt],
                     bx: &DistributionSharesBox<HipModpGroup>) -> Vec<bool> {
    assert_eq!(share_boxes.len(), publickeys.len());
    let mut live = Vec::new();
    let (mut pk, mut s, mut y, mut c, mut r) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for (i, (sb, key)) in share_boxes.iter().zip(publickeys).enumerate() {
        if let Some(enc) = bx.shares.get(&group.element_to_bytes(key)) {     // missing share -> false (:368-372)
            live.push(i);
            pk.extend_from_slice(&be256(key));
            s.extend_from_slice(&be256(&sb.share));
            y.extend_from_slice(&be256(enc));
            c.extend_from_slice(&be256(&sb.challenge));
            r.extend_from_slice(&be256(&sb.response));
        }
    }
    let mut verdicts = vec![0u8; live.len()];
    let rc = unsafe {
        ffi::mpvss_modp_verify_shares(group.engine.raw(), ffi::MPVSS_HOST, pk.as_ptr(), s.as_ptr(), y.as_ptr(), c.as_ptr(), r.as_ptr(),
                                      live.len(), verdicts.as_mut_ptr())
    };
    group.engine.expect(rc, "verify_shares");
    let mut out = vec![false; share_boxes.len()];
    for (k, i) in live.into_iter().enumerate() {
        out[i] = verdicts[k] == 1;
    }
    out
}

/// Drop-in body of `Participant<ModpGroup>::distribute_secret` (participant.rs:160-286).
pub fn distribute_secret(group: &HipModpGroup, secret: &BigInt, publickeys: &[BigInt], threshold: u32) -> DistributionSharesBox<HipModpGroup> {
    assert!(threshold as usize <= publickeys.len());                                   // participant.rs:166
    let n = publickeys.len();
    let t = threshold as usize;
    let order = group.order().clone();
    let mut polynomial = Polynomial::new();
    polynomial.init((threshold - 1) as i32, &order);                                   // participant.rs:175
    let coeffs: Vec<u8> = polynomial.coefficients.iter().flat_map(|a| be256(a)).collect();
    let g = be256(&group.subgroup_generator());
    let mut cm = vec![0u8; t * 256];
    let rc = unsafe { ffi::mpvss_modp_batch_exp_fixed_base(group.engine.raw(), ffi::MPVSS_HOST, g.as_ptr(), coeffs.as_ptr(), t, cm.as_mut_ptr()) };
    group.engine.expect(rc, "distribute_secret: commitments");                         // C_j = g^a_j, :189-193
    let positions: Vec<i64> = (1..=n as i64).collect();                                // :186,198,247
    let witnesses: Vec<BigInt> = (0..n).map(|_| group.generate_private_key()).collect();    // :223
    let pk: Vec<u8> = publickeys.iter().flat_map(|y| be256(y)).collect();
    let ws: Vec<u8> = witnesses.iter().flat_map(|w| be256(w)).collect();
    // P(i) % order (:200-202), X_i, Y_i, a1_i, a2_i (:207-249), the transcript digest and the challenge (:251-252) and the
    // responses (:255-264) in one call: the 2048-bit scalar arithmetic runs on the device too (any n: the library cuts a box
    // of more than 262144 participants into blocks itself and keeps ONE transcript over them)
    let mut y = vec![0u8; n * 256];
    let mut r = vec![0u8; n * 256];
    let mut digest = [0u8; 32];
    let mut c256 = [0u8; 256];
    let rc = unsafe {
        ffi::mpvss_modp_deal(group.engine.raw(), coeffs.as_ptr(), t, positions.as_ptr(), pk.as_ptr(), ws.as_ptr(), n, std::ptr::null_mut(),
                             y.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), digest.as_mut_ptr(), c256.as_mut_ptr(),
                             r.as_mut_ptr())
    };
    group.engine.expect(rc, "distribute_secret");
    let challenge = BigInt::from_bytes_be(Sign::Plus, &c256);
    debug_assert_eq!(challenge, group.hash_to_scalar(&digest));
    let mut bx = DistributionSharesBox::new();
    let big = |b: &[u8]| BigInt::from_bytes_be(Sign::Plus, b);
    let mut pos_map = std::collections::HashMap::new();
    let mut share_map = std::collections::HashMap::new();
    let mut resp_map = std::collections::HashMap::new();
    for (i, pkey) in publickeys.iter().enumerate() {
        let key = group.element_to_bytes(pkey);
        pos_map.insert(key.clone(), positions[i]);
        share_map.insert(key.clone(), big(&y[i * 256..(i + 1) * 256]));
        resp_map.insert(key, big(&r[i * 256..(i + 1) * 256]));
    }
    // U = secret XOR (SHA256(bytes(G^s)) mod q)                                       :267-272
    let s = polynomial.get_value(&BigInt::zero()) % &order;
    let g_s = group.generate_public_key(&s);
    let mut h = [0u8; 32];
    let gb = group.element_to_bytes(&g_s);
    unsafe { ffi::mpvss_sha256(gb.as_ptr(), gb.len(), h.as_mut_ptr()) };
    let mask = BigUint::from_bytes_be(&h) % group.modulus().unwrap().to_biguint().unwrap();
    let u = secret.to_biguint().unwrap() ^ mask;
    let commitments: Vec<BigInt> = (0..t).map(|j| big(&cm[j * 256..(j + 1) * 256])).collect();
    bx.init(&commitments, pos_map, share_map, publickeys, &challenge, resp_map, &BigInt::from_biguint(Sign::Plus, u));
    bx
}

/// n participants decrypt and prove at once (participant.rs:294-353): `witnesses[i]` is the `w` argument of the i-th call.
pub fn extract_secret_shares(group: &HipModpGroup, bx: &DistributionSharesBox<HipModpGroup>, private_keys: &[BigInt], witnesses: &[BigInt])
    -> Vec<Option<ShareBox<HipModpGroup>>> {
    assert_eq!(private_keys.len(), witnesses.len());
    let mut idx = Vec::new();
    let (mut pk, mut y, mut xinv, mut w) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
    let mut pubs = Vec::new();
    for (i, x) in private_keys.iter().enumerate() {
        let public_key = group.generate_public_key(x);
        let (Some(enc), Some(inv)) = (bx.shares.get(&group.element_to_bytes(&public_key)), group.scalar_inverse(x)) else { continue };
        idx.push(i);
        pk.extend_from_slice(&be256(&public_key));
        y.extend_from_slice(&be256(enc));
        xinv.extend_from_slice(&be256(&inv));
        w.extend_from_slice(&be256(&witnesses[i]));
        pubs.push(public_key);
    }
    let m = idx.len();
    let (mut s, mut c, mut r) = (vec![0u8; m * 256], vec![0u8; m * 256], vec![0u8; m * 256]);
    let rc = unsafe { ffi::mpvss_modp_extract_shares(group.engine.raw(), ffi::MPVSS_HOST, pk.as_ptr(), y.as_ptr(), xinv.as_ptr(), w.as_ptr(), m, s.as_mut_ptr(), c.as_mut_ptr()) };
    group.engine.expect(rc, "extract_secret_shares");
    let xs: Vec<u8> = idx.iter().flat_map(|&i| be256(&private_keys[i])).collect();
    unsafe { ffi::mpvss_modp_dleq_responses(w.as_ptr(), xs.as_ptr(), c.as_ptr(), 1, m, r.as_mut_ptr(), 0) };     // r = w - x c, dleq.rs:42-50
    let big = |b: &[u8]| BigInt::from_bytes_be(Sign::Plus, b);
    let mut out: Vec<Option<ShareBox<HipModpGroup>>> = (0..private_keys.len()).map(|_| None).collect();
    for (k, &i) in idx.iter().enumerate() {
        let mut sb = ShareBox::new();
        sb.init(pubs[k].clone(), big(&s[k * 256..(k + 1) * 256]), big(&c[k * 256..(k + 1) * 256]), big(&r[k * 256..(k + 1) * 256]));
        out[i] = Some(sb);
    }
    out
}

/// Drop-in body of `Participant<ModpGroup>::reconstruct` (participant.rs:462-519).
pub fn reconstruct(group: &HipModpGroup, share_boxes: &[ShareBox<HipModpGroup>], bx: &DistributionSharesBox<HipModpGroup>) -> Option<BigInt> {
    if share_boxes.len() < bx.commitments.len() {
        return None;
    }
    let mut shares = std::collections::BTreeMap::new();
    for sb in share_boxes {
        let position = bx.positions.get(&group.element_to_bytes(&sb.publickey))?;
        shares.insert(*position, sb.share.clone());
    }
    let positions: Vec<i64> = shares.keys().copied().collect();
    let s: Vec<u8> = shares.values().flat_map(|v| be256(v)).collect();
    let (mut gs, mut mask) = ([0u8; 256], [0u8; 32]);
    let rc = unsafe { ffi::mpvss_modp_reconstruct(group.engine.raw(), ffi::MPVSS_HOST, positions.as_ptr(), s.as_ptr(), positions.len(), gs.as_mut_ptr(), mask.as_mut_ptr()) };
    if rc != ffi::MPVSS_OK {
        return None;        // a share without an inverse: the reference returns None as well (:551-553)
    }
    let secret = BigUint::from_bytes_be(&mask) ^ bx.U.to_biguint().unwrap();
    Some(BigInt::from_biguint(Sign::Plus, secret))
}

// ---- curve groups: the same flat layout with 33 / 32-byte elements and 32-byte scalars --------------------------------
/// What the generic curve bodies below need from a HIP-backed curve group beyond `Group`: its id at the C boundary, the
/// boundary encodings (33-byte SEC1 / 32-byte ristretto255 elements, 32-byte scalars in the group's byte order) and its order.
pub trait CurveCodec: Group {
    const GROUP_ID: i32;
    const ENC: usize;
    fn engine(&self) -> &crate::Engine;
    fn enc(e: &Self::Element) -> Vec<u8>;
    fn dec(b: &[u8]) -> Option<Self::Element>;
    fn sc(s: &Self::Scalar) -> [u8; 32];
    fn sc_from(b: &[u8; 32]) -> Self::Scalar;
    /// the group order as the reference's `order_as_bigint()` gives it
    fn order_bigint(&self) -> &BigInt;
    /// a BigInt below the order as a scalar, as participant.rs:1134-1143 / ristretto255.rs:78-106 convert it
    fn scalar_of(v: &BigInt) -> Self::Scalar;
}

impl CurveCodec for HipSecp256k1Group {
    const GROUP_ID: i32 = ffi::MPVSS_GROUP_SECP256K1;
    const ENC: usize = 33;
    fn engine(&self) -> &crate::Engine { &self.engine }
    fn enc(e: &Self::Element) -> Vec<u8> { crate::groups::secp::point_bytes(e).to_vec() }
    fn dec(b: &[u8]) -> Option<Self::Element> { crate::groups::secp::point_from(b.try_into().ok()?) }
    fn sc(s: &Self::Scalar) -> [u8; 32] { crate::groups::secp::scalar_bytes(s) }
    fn sc_from(b: &[u8; 32]) -> Self::Scalar { crate::groups::secp::scalar_from(b) }
    fn order_bigint(&self) -> &BigInt { self.order_as_bigint() }
    fn scalar_of(v: &BigInt) -> Self::Scalar { crate::groups::secp::scalar_from_bigint(v) }
}

impl CurveCodec for HipRistretto255Group {
    const GROUP_ID: i32 = ffi::MPVSS_GROUP_RISTRETTO255;
    const ENC: usize = 32;
    fn engine(&self) -> &crate::Engine { &self.engine }
    fn enc(e: &Self::Element) -> Vec<u8> { crate::groups::rist::point_bytes(e).to_vec() }
    fn dec(b: &[u8]) -> Option<Self::Element> { crate::groups::rist::point_from(b.try_into().ok()?) }
    fn sc(s: &Self::Scalar) -> [u8; 32] { s.to_bytes() }
    fn sc_from(b: &[u8; 32]) -> Self::Scalar { curve25519_dalek::scalar::Scalar::from_bytes_mod_order(*b) }
    fn order_bigint(&self) -> &BigInt { self.order_as_bigint() }
    fn scalar_of(v: &BigInt) -> Self::Scalar { HipRistretto255Group::bigint_to_scalar(v) }
}

/// SHA256(bytes(G^s)) as a big-endian integer reduced mod the group order: what the curve groups XOR onto the secret
/// (participant.rs:1229-1246 / 1694-1701, and back in :1495-1508 / 1939-1946)
fn ec_mask<G: CurveCodec>(group: &G, g_s: &G::Element) -> BigUint {
    let gb = group.element_to_bytes(g_s);
    let mut h = [0u8; 32];
    unsafe { ffi::mpvss_sha256(gb.as_ptr(), gb.len(), h.as_mut_ptr()) };
    BigUint::from_bytes_be(&h) % group.order_bigint().to_biguint().unwrap()
}

/// Drop-in body of `Participant<Secp256k1Group / Ristretto255Group>::verify_distribution_shares` (participant.rs:1384-1444, 1827-1887)
pub fn ec_verify_distribution_shares<G: CurveCodec>(group: &G, bx: &DistributionSharesBox<G>) -> bool {
    let (mut cm, mut pos, mut pk, mut sh, mut rs) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for c in &bx.commitments {
        cm.extend(G::enc(c));
    }
    for y in &bx.publickeys {
        let key = group.element_to_bytes(y);
        let (Some(p), Some(r), Some(s)) = (bx.positions.get(&key), bx.responses.get(&key), bx.shares.get(&key)) else { return false };
        pos.push(*p);
        pk.extend(G::enc(y));
        sh.extend(G::enc(s));
        rs.extend_from_slice(&G::sc(r));
    }
    let ch = G::sc(&bx.challenge);
    let mut verdict = 0i32;
    let rc = unsafe {
        ffi::mpvss_ec_verify_distribution(group.engine().raw(), G::GROUP_ID, ffi::MPVSS_HOST, cm.as_ptr(), bx.commitments.len(), pos.as_ptr(),
                                          pk.as_ptr(), sh.as_ptr(), rs.as_ptr(), pos.len(), ch.as_ptr(), &mut verdict, std::ptr::null_mut(),
                                          std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())
    };
    rc == ffi::MPVSS_OK && verdict == 1        // an encoding the engine rejects is `false`, like a failed check in the reference
}

/// Drop-in body of `distribute_secret` for the curve groups (participant.rs:1094-1274, 1573-1717): polynomial and witnesses drawn as
/// the reference draws them, then ONE `mpvss_ec_deal` call -- P(i) mod order, X_i, Y_i, a1_i, a2_i, the transcript digest, the
/// challenge and the responses on the device.
pub fn ec_distribute_secret<G: CurveCodec>(group: &G, secret: &BigInt, publickeys: &[G::Element], threshold: u32) -> DistributionSharesBox<G> {
    assert!(threshold as usize <= publickeys.len());                                   // participant.rs:1100
    let (n, t) = (publickeys.len(), threshold as usize);
    let order = group.order_bigint().clone();
    let mut polynomial = Polynomial::new();
    polynomial.init((threshold - 1) as i32, &order);                                   // :1107-1113
    let coeff_scalars: Vec<G::Scalar> = polynomial.coefficients.iter().map(|a| G::scalar_of(a)).collect();
    let coeffs: Vec<u8> = coeff_scalars.iter().flat_map(|a| G::sc(a)).collect();
    let eng = group.engine();
    let mut cm = vec![0u8; t * G::ENC];
    let rc = unsafe { ffi::mpvss_ec_batch_exp_generator(eng.raw(), G::GROUP_ID, ffi::MPVSS_HOST, coeffs.as_ptr(), t, cm.as_mut_ptr()) };
    eng.expect(rc, "distribute_secret: commitments");                                  // C_j = a_j G, :1130-1152
    let positions: Vec<i64> = (1..=n as i64).collect();                                // :1127, 1155-1157, 1212
    let witnesses: Vec<u8> = (0..n).flat_map(|_| G::sc(&group.generate_private_key())).collect();   // :1177
    let pk: Vec<u8> = publickeys.iter().flat_map(|y| G::enc(y)).collect();
    let mut y = vec![0u8; n * G::ENC];
    let mut r = vec![0u8; n * 32];
    let (mut digest, mut challenge) = ([0u8; 32], [0u8; 32]);
    let rc = unsafe {
        ffi::mpvss_ec_deal(eng.raw(), G::GROUP_ID, coeffs.as_ptr(), t, positions.as_ptr(), pk.as_ptr(), witnesses.as_ptr(), n,
                           std::ptr::null_mut(), y.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), digest.as_mut_ptr(),
                           challenge.as_mut_ptr(), r.as_mut_ptr())
    };
    eng.expect(rc, "distribute_secret");
    let mut pos_map = std::collections::HashMap::new();
    let mut share_map = std::collections::HashMap::new();
    let mut resp_map = std::collections::HashMap::new();
    for (i, pkey) in publickeys.iter().enumerate() {
        let key = group.element_to_bytes(pkey);
        pos_map.insert(key.clone(), positions[i]);
        share_map.insert(key.clone(), G::dec(&y[i * G::ENC..(i + 1) * G::ENC]).expect("engine returns canonical encodings"));
        resp_map.insert(key, G::sc_from(r[i * 32..(i + 1) * 32].try_into().unwrap()));
    }
    // U = secret XOR (SHA256(bytes(s G)) mod order), s = P(0)                         :1226-1250 / 1690-1702
    let g_s = group.generate_public_key(&coeff_scalars[0]);
    let u = secret.to_biguint().unwrap() ^ ec_mask(group, &g_s);
    let commitments: Vec<G::Element> =
        (0..t).map(|j| G::dec(&cm[j * G::ENC..(j + 1) * G::ENC]).expect("engine returns canonical encodings")).collect();
    let mut bx = DistributionSharesBox::new();
    bx.init(&commitments, pos_map, share_map, publickeys, &G::sc_from(&challenge), resp_map, &BigInt::from_biguint(Sign::Plus, u));
    bx
}

/// n calls of `extract_secret_share` (participant.rs:1282-1338, 1725-1781) at once: `witnesses[i]` is the i-th call's `w`.
pub fn ec_extract_secret_shares<G: CurveCodec>(group: &G, bx: &DistributionSharesBox<G>, private_keys: &[G::Scalar], witnesses: &[G::Scalar])
    -> Vec<Option<ShareBox<G>>> {
    assert_eq!(private_keys.len(), witnesses.len());
    let mut idx = Vec::new();
    let (mut pk, mut y, mut xinv, mut w, mut xs) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    let mut pubs = Vec::new();
    for (i, x) in private_keys.iter().enumerate() {
        let public_key = group.generate_public_key(x);
        let (Some(enc), Some(inv)) = (bx.shares.get(&group.element_to_bytes(&public_key)), group.scalar_inverse(x)) else { continue };
        idx.push(i);
        pk.extend(G::enc(&public_key));
        y.extend(G::enc(enc));
        xinv.extend_from_slice(&G::sc(&inv));
        w.extend_from_slice(&G::sc(&witnesses[i]));
        xs.extend_from_slice(&G::sc(x));
        pubs.push(public_key);
    }
    let m = idx.len();
    let mut out: Vec<Option<ShareBox<G>>> = (0..private_keys.len()).map(|_| None).collect();
    if m == 0 {
        return out;
    }
    let (mut s, mut c, mut r) = (vec![0u8; m * G::ENC], vec![0u8; m * 32], vec![0u8; m * 32]);
    let eng = group.engine();
    let rc = unsafe {
        ffi::mpvss_ec_extract_shares(eng.raw(), G::GROUP_ID, ffi::MPVSS_HOST, pk.as_ptr(), y.as_ptr(), xinv.as_ptr(), w.as_ptr(), m, s.as_mut_ptr(),
                                     c.as_mut_ptr())
    };
    eng.expect(rc, "extract_secret_shares");
    unsafe { ffi::mpvss_ec_dleq_responses(G::GROUP_ID, w.as_ptr(), xs.as_ptr(), c.as_ptr(), 1, m, r.as_mut_ptr(), 0) };     // r = w - x c, dleq.rs:42-50
    for (k, &i) in idx.iter().enumerate() {
        let mut sb = ShareBox::new();
        sb.init(pubs[k].clone(), G::dec(&s[k * G::ENC..(k + 1) * G::ENC]).expect("engine returns canonical encodings"),
                G::sc_from(c[k * 32..(k + 1) * 32].try_into().unwrap()), G::sc_from(r[k * 32..(k + 1) * 32].try_into().unwrap()));
        out[i] = Some(sb);
    }
    out
}

/// n calls of `verify_share` (participant.rs:1346-1371, 1789-1814) against one distribution box: one verdict each.
pub fn ec_verify_shares<G: CurveCodec>(group: &G, share_boxes: &[ShareBox<G>], publickeys: &[G::Element], bx: &DistributionSharesBox<G>) -> Vec<bool> {
    assert_eq!(share_boxes.len(), publickeys.len());
    let mut live = Vec::new();
    let (mut pk, mut s, mut y, mut c, mut r) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for (i, (sb, key)) in share_boxes.iter().zip(publickeys).enumerate() {
        if let Some(enc) = bx.shares.get(&group.element_to_bytes(key)) {     // missing share -> false (:1355-1361)
            live.push(i);
            pk.extend(G::enc(key));
            s.extend(G::enc(&sb.share));
            y.extend(G::enc(enc));
            c.extend_from_slice(&G::sc(&sb.challenge));
            r.extend_from_slice(&G::sc(&sb.response));
        }
    }
    let mut out = vec![false; share_boxes.len()];
    if live.is_empty() {
        return out;
    }
    let mut verdicts = vec![0u8; live.len()];
    let rc = unsafe {
        ffi::mpvss_ec_verify_shares(group.engine().raw(), G::GROUP_ID, ffi::MPVSS_HOST, pk.as_ptr(), s.as_ptr(), y.as_ptr(), c.as_ptr(),
                                    r.as_ptr(), live.len(), verdicts.as_mut_ptr())
    };
    group.engine().expect(rc, "verify_shares");
    for (k, i) in live.into_iter().enumerate() {
        out[i] = verdicts[k] == 1;
    }
    out
}

/// Drop-in body of `reconstruct` for the curve groups (participant.rs:1452-1516, 1895-1953).
pub fn ec_reconstruct<G: CurveCodec>(group: &G, share_boxes: &[ShareBox<G>], bx: &DistributionSharesBox<G>) -> Option<BigInt> {
    if share_boxes.len() < bx.commitments.len() {
        return None;
    }
    let mut shares = std::collections::BTreeMap::new();
    for sb in share_boxes {
        let position = bx.positions.get(&group.element_to_bytes(&sb.publickey))?;
        shares.insert(*position, G::enc(&sb.share));
    }
    let positions: Vec<i64> = shares.keys().copied().collect();
    let s: Vec<u8> = shares.values().flatten().copied().collect();
    let (mut gs, mut mask) = (vec![0u8; G::ENC], [0u8; 32]);
    let rc = unsafe {
        ffi::mpvss_ec_reconstruct(group.engine().raw(), G::GROUP_ID, ffi::MPVSS_HOST, positions.as_ptr(), s.as_ptr(), positions.len(), gs.as_mut_ptr(),
                                  mask.as_mut_ptr())
    };
    if rc != ffi::MPVSS_OK {
        return None;
    }
    let secret = BigUint::from_bytes_be(&mask) ^ bx.U.to_biguint().unwrap();
    Some(BigInt::from_biguint(Sign::Plus, secret))
}
