//! `impl Group for HipRistretto255Group`: src/groups/ristretto255.rs:45-253 with `exp` / `mul` on the GPU.
//! Elements stay `RistrettoPoint`, scalars `curve25519_dalek::Scalar`; they cross the boundary as the canonical
//! 32-byte ristretto255 encoding and 32-byte little-endian scalars.
use std::sync::Arc;

use curve25519_dalek::constants::RISTRETTO_BASEPOINT_POINT;
use curve25519_dalek::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek::scalar::Scalar;
use curve25519_dalek::traits::Identity;
use mpvss_rs::group::Group;
use num_bigint::BigInt;

use crate::engine::Engine;
use crate::ffi;

const G: i32 = ffi::MPVSS_GROUP_RISTRETTO255;

#[derive(Debug, Clone)]
pub struct HipRistretto255Group {
    pub(crate) engine: Engine,
    order_scalar: Scalar,        // placeholder like the reference's (ristretto255.rs:128-146): l does not fit a Scalar
    order_bigint: BigInt,
}

pub(crate) fn point_bytes(p: &RistrettoPoint) -> [u8; 32] { p.compress().to_bytes() }
pub(crate) fn point_from(b: &[u8; 32]) -> Option<RistrettoPoint> { CompressedRistretto(*b).decompress() }

impl HipRistretto255Group {
    /// Same convention as `Ristretto255Group::new()` (ristretto255.rs:51): the process's shared engine (device `MPVSS_DEVICE`, default 0).
    pub fn new() -> Arc<Self> {
        Self::with_engine(Engine::shared())
    }
    /// A group bound to a context of its own on GPU `device_id`.
    pub fn with_device(device_id: i32) -> Arc<Self> {
        Self::with_engine(Engine::new(device_id).expect("MI355X engine"))
    }
    fn with_engine(engine: Engine) -> Arc<Self> {
        let order_bigint = BigInt::parse_bytes(b"1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3ed", 16).unwrap();
        Arc::new(HipRistretto255Group { engine, order_scalar: Scalar::ZERO, order_bigint })
    }
    /// ristretto255.rs:66-68
    pub fn order_as_bigint(&self) -> &BigInt { &self.order_bigint }

    /// ristretto255.rs:78-106: a BigInt already reduced mod l, big-endian, to dalek's little-endian Scalar
    pub fn bigint_to_scalar(bigint: &BigInt) -> Scalar {
        let be = bigint.to_bytes_be().1;
        let mut le = [0u8; 32];
        for (dst, src) in le.iter_mut().zip(be.iter().rev()) {
            *dst = *src;
        }
        Scalar::from_bytes_mod_order(le)
    }

    /// ristretto255.rs:108-125
    pub fn scalar_to_bigint(scalar: &Scalar) -> BigInt {
        BigInt::from_bytes_le(num_bigint::Sign::Plus, &scalar.to_bytes())
    }
}

impl Group for HipRistretto255Group {
    type Scalar = Scalar;
    type Element = RistrettoPoint;

    fn order(&self) -> &Scalar { &self.order_scalar }
    fn subgroup_order(&self) -> &Scalar { &self.order_scalar }
    fn generator(&self) -> RistrettoPoint { RISTRETTO_BASEPOINT_POINT }                                  // ristretto255.rs:148-150
    fn subgroup_generator(&self) -> RistrettoPoint { RISTRETTO_BASEPOINT_POINT }                         // :152-155
    fn identity(&self) -> RistrettoPoint { RistrettoPoint::identity() }                                  // :157-159

    /// ristretto255.rs:161-170
    fn exp(&self, base: &RistrettoPoint, scalar: &Scalar) -> RistrettoPoint {
        let (p, k) = (point_bytes(base), scalar.to_bytes());
        let mut out = [0u8; 32];
        let rc = unsafe { ffi::mpvss_ec_batch_exp(self.engine.raw(), G, ffi::MPVSS_HOST, p.as_ptr(), k.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipRistretto255Group::exp");
        point_from(&out).expect("engine returns canonical encodings")
    }

    /// ristretto255.rs:172-177
    fn mul(&self, a: &RistrettoPoint, b: &RistrettoPoint) -> RistrettoPoint {
        let (x, y) = (point_bytes(a), point_bytes(b));
        let mut out = [0u8; 32];
        let rc = unsafe { ffi::mpvss_ec_batch_mul(self.engine.raw(), G, ffi::MPVSS_HOST, x.as_ptr(), y.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipRistretto255Group::mul");
        point_from(&out).expect("engine returns canonical encodings")
    }

    fn scalar_inverse(&self, x: &Scalar) -> Option<Scalar> {                                             // ristretto255.rs:179-187
        if *x == Scalar::ZERO { None } else { Some(x.invert()) }
    }
    fn element_inverse(&self, x: &RistrettoPoint) -> Option<RistrettoPoint> { Some(-x) }                 // :189-194

    /// ristretto255.rs:196-205: SHA-512, 64 bytes little-endian, mod l
    fn hash_to_scalar(&self, data: &[u8]) -> Scalar {
        let mut out = [0u8; 32];
        unsafe { ffi::mpvss_ec_hash_to_scalar(G, data.as_ptr(), data.len(), out.as_mut_ptr()) };
        Scalar::from_bytes_mod_order(out)
    }

    fn element_to_bytes(&self, elem: &RistrettoPoint) -> Vec<u8> { point_bytes(elem).to_vec() }          // ristretto255.rs:207-210
    fn bytes_to_element(&self, bytes: &[u8]) -> Option<RistrettoPoint> {                                 // ristretto255.rs:212-220
        if bytes.len() != 32 { return None; }
        let mut b = [0u8; 32];
        b.copy_from_slice(bytes);
        point_from(&b)
    }
    fn scalar_to_bytes(&self, scalar: &Scalar) -> Vec<u8> { scalar.to_bytes().to_vec() }                 // ristretto255.rs:222-225

    /// ristretto255.rs:227-237
    fn generate_private_key(&self) -> Scalar {
        let mut bytes = [0u8; 32];
        rand::Rng::fill(&mut rand::thread_rng(), &mut bytes);
        Scalar::from_bytes_mod_order(bytes)
    }

    /// ristretto255.rs:239-242: k B through the fixed-base comb
    fn generate_public_key(&self, private_key: &Scalar) -> RistrettoPoint {
        let k = private_key.to_bytes();
        let mut out = [0u8; 32];
        let rc = unsafe { ffi::mpvss_ec_batch_exp_generator(self.engine.raw(), G, ffi::MPVSS_HOST, k.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipRistretto255Group::generate_public_key");
        point_from(&out).expect("engine returns canonical encodings")
    }

    fn scalar_mul(&self, a: &Scalar, b: &Scalar) -> Scalar { a * b }                                     // ristretto255.rs:244-247
    fn scalar_sub(&self, a: &Scalar, b: &Scalar) -> Scalar { a - b }                                     // ristretto255.rs:249-252
}
