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
//! Never compiled in this repository's environment (no Rust toolchain).
use mpvss_rs::group::Group;
use mpvss_rs::polynomial::Polynomial;
use mpvss_rs::sharebox::{DistributionSharesBox, ShareBox};
use num_bigint::{BigInt, BigUint, Sign};
use num_traits::Zero;

use crate::engine::EngineError;
use crate::ffi;
use crate::groups::{be256, HipModpGroup};

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
pub fn verify_shares(group: &HipModpGroup, share_boxes: &[ShareBox<HipModpGroup>], bx: &DistributionSharesBox<HipModpGroup>) -> Vec<bool> {
    let mut live = Vec::new();
    let (mut pk, mut s, mut y, mut c, mut r) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for (i, sb) in share_boxes.iter().enumerate() {
        if let Some(enc) = bx.shares.get(&group.element_to_bytes(&sb.publickey)) {     // missing share -> false (:368-372)
            live.push(i);
            pk.extend_from_slice(&be256(&sb.publickey));
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

/// n participants decrypt and prove at once (participant.rs:294-353); witnesses drawn here like the reference's callers do.
pub fn extract_secret_shares(group: &HipModpGroup, bx: &DistributionSharesBox<HipModpGroup>, private_keys: &[BigInt]) -> Vec<Option<ShareBox<HipModpGroup>>> {
    let order = group.order().clone();
    let mut idx = Vec::new();
    let (mut pk, mut y, mut xinv, mut w) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
    let mut wits = Vec::new();
    let mut pubs = Vec::new();
    for (i, x) in private_keys.iter().enumerate() {
        let public_key = group.generate_public_key(x);
        let (Some(enc), Some(inv)) = (bx.shares.get(&group.element_to_bytes(&public_key)), group.scalar_inverse(x)) else { continue };
        let wit = group.generate_private_key();
        idx.push(i);
        pk.extend_from_slice(&be256(&public_key));
        y.extend_from_slice(&be256(enc));
        xinv.extend_from_slice(&be256(&inv));
        w.extend_from_slice(&be256(&wit));
        wits.push(wit);
        pubs.push(public_key);
    }
    let m = idx.len();
    let (mut s, mut c, mut r) = (vec![0u8; m * 256], vec![0u8; m * 256], vec![0u8; m * 256]);
    let rc = unsafe { ffi::mpvss_modp_extract_shares(group.engine.raw(), ffi::MPVSS_HOST, pk.as_ptr(), y.as_ptr(), xinv.as_ptr(), w.as_ptr(), m, s.as_mut_ptr(), c.as_mut_ptr()) };
    group.engine.expect(rc, "extract_secret_shares");
    let xs: Vec<u8> = idx.iter().flat_map(|&i| be256(&private_keys[i])).collect();
    unsafe { ffi::mpvss_modp_dleq_responses(w.as_ptr(), xs.as_ptr(), c.as_ptr(), 1, m, r.as_mut_ptr(), 0) };     // r = w - x c, dleq.rs:42-50
    let _ = order;
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
/// Generic over the two curve groups: `enc` serialises an element, `sc` a scalar, in the boundary's byte order.
pub fn ec_verify_distribution_shares<G: Group>(engine: &crate::Engine, group_id: i32, group: &G, bx: &DistributionSharesBox<G>,
                                               enc: impl Fn(&G::Element) -> Vec<u8>, sc: impl Fn(&G::Scalar) -> Vec<u8>) -> bool {
    let (mut cm, mut pos, mut pk, mut sh, mut rs) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for c in &bx.commitments {
        cm.extend(enc(c));
    }
    for y in &bx.publickeys {
        let key = group.element_to_bytes(y);
        let (Some(p), Some(r), Some(s)) = (bx.positions.get(&key), bx.responses.get(&key), bx.shares.get(&key)) else { return false };
        pos.push(*p);
        pk.extend(enc(y));
        sh.extend(enc(s));
        rs.extend(sc(r));
    }
    let ch = sc(&bx.challenge);
    let mut verdict = 0i32;
    let rc = unsafe {
        ffi::mpvss_ec_verify_distribution(engine.raw(), group_id, ffi::MPVSS_HOST, cm.as_ptr(), bx.commitments.len(), pos.as_ptr(), pk.as_ptr(),
                                          sh.as_ptr(), rs.as_ptr(), pos.len(), ch.as_ptr(), &mut verdict, std::ptr::null_mut(),
                                          std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())
    };
    rc == ffi::MPVSS_OK && verdict == 1
}

/// The dealer's whole box for a curve group in one call (`mpvss_ec_deal`): the body of `distribute_secret`
/// (participant.rs:1094-1274 secp256k1, 1573-1717 ristretto255) between "draw the polynomial and the witnesses" and "put the maps
/// together" -- P(i) mod order (:1155-1157 / :1619-1621), X_i, Y_i, a1_i, a2_i, the transcript digest, the challenge
/// c = hash_to_scalar(digest) (:1200-1210 / :1662-1672) and the responses r_i = w_i - P(i) c -- everything on the device.
/// `coeffs` / `witnesses`: 32-byte scalars in the boundary's byte order, `pubkeys`: encoded elements; returns
/// (Y encodings, digest, challenge bytes, response bytes) or the library's error code.
pub fn ec_deal(engine: &crate::Engine, group_id: i32, enc_len: usize, coeffs: &[u8], pubkeys: &[u8], witnesses: &[u8])
               -> Result<(Vec<u8>, [u8; 32], [u8; 32], Vec<u8>), i32> {
    let (t, n) = (coeffs.len() / 32, witnesses.len() / 32);
    let positions: Vec<i64> = (1..=n as i64).collect();                                 // :1139,1151,1198
    let mut y = vec![0u8; n * enc_len];
    let mut r = vec![0u8; n * 32];
    let (mut digest, mut challenge) = ([0u8; 32], [0u8; 32]);
    let rc = unsafe {
        ffi::mpvss_ec_deal(engine.raw(), group_id, coeffs.as_ptr(), t, positions.as_ptr(), pubkeys.as_ptr(), witnesses.as_ptr(), n,
                           std::ptr::null_mut(), y.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), digest.as_mut_ptr(),
                           challenge.as_mut_ptr(), r.as_mut_ptr())
    };
    if rc != ffi::MPVSS_OK {
        return Err(rc);
    }
    Ok((y, digest, challenge, r))
}
