//! `impl Group for HipSecp256k1Group`: src/groups/secp256k1.rs:38-189 with `exp` / `mul` on the GPU.
//! Elements stay `k256::AffinePoint`, scalars `k256::Scalar` (the reference's types); they cross the boundary as
//! 33-byte SEC1 compressed points (33 zero bytes = identity) and 32-byte big-endian scalars.
use std::sync::Arc;

use k256::elliptic_curve::group::GroupEncoding;
use k256::elliptic_curve::ops::Reduce;
use k256::elliptic_curve::PrimeField;
use k256::{AffinePoint, FieldBytes, ProjectivePoint, Scalar, U256};
use mpvss_rs::group::Group;
use num_bigint::{BigInt, Sign};

use crate::engine::Engine;
use crate::ffi;

const G: i32 = ffi::MPVSS_GROUP_SECP256K1;

#[derive(Debug, Clone)]
pub struct HipSecp256k1Group {
    pub(crate) engine: Engine,
    order_placeholder: Scalar,   // `order()` returns a placeholder ONE in the reference as well (secp256k1.rs:65-76)
    order_bigint: BigInt,
}

pub(crate) fn point_bytes(p: &AffinePoint) -> [u8; 33] {
    let mut out = [0u8; 33];
    out.copy_from_slice(ProjectivePoint::from(*p).to_affine().to_bytes().as_slice());   // identity -> 33 zero bytes
    out
}
pub(crate) fn point_from(b: &[u8; 33]) -> Option<AffinePoint> {
    Option::from(AffinePoint::from_bytes(b.into()))
}
pub(crate) fn scalar_bytes(s: &Scalar) -> [u8; 32] {
    s.to_bytes().into()
}
pub(crate) fn scalar_from(b: &[u8; 32]) -> Scalar {
    Option::from(Scalar::from_repr(FieldBytes::clone_from_slice(b))).expect("engine returns reduced scalars")
}

impl HipSecp256k1Group {
    /// Same convention as `Secp256k1Group::new()` (secp256k1.rs:44): the process's shared engine (device `MPVSS_DEVICE`, default 0).
    pub fn new() -> Arc<Self> {
        Self::with_engine(Engine::shared())
    }
    /// A group bound to a context of its own on GPU `device_id`.
    pub fn with_device(device_id: i32) -> Arc<Self> {
        Self::with_engine(Engine::new(device_id).expect("MI355X engine"))
    }
    fn with_engine(engine: Engine) -> Arc<Self> {
        let order_bigint = BigInt::parse_bytes(b"fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141", 16).unwrap();
        Arc::new(HipSecp256k1Group { engine, order_placeholder: Scalar::ONE, order_bigint })
    }
    /// secp256k1.rs:186-188
    pub fn order_as_bigint(&self) -> &BigInt { &self.order_bigint }
}

impl Group for HipSecp256k1Group {
    type Scalar = Scalar;
    type Element = AffinePoint;

    fn order(&self) -> &Scalar { &self.order_placeholder }
    fn subgroup_order(&self) -> &Scalar { &self.order_placeholder }
    fn generator(&self) -> AffinePoint { AffinePoint::GENERATOR }
    fn subgroup_generator(&self) -> AffinePoint { AffinePoint::GENERATOR }
    fn identity(&self) -> AffinePoint { AffinePoint::IDENTITY }

    /// secp256k1.rs:91-100
    fn exp(&self, base: &AffinePoint, scalar: &Scalar) -> AffinePoint {
        let (p, k) = (point_bytes(base), scalar_bytes(scalar));
        let mut out = [0u8; 33];
        let rc = unsafe { ffi::mpvss_ec_batch_exp(self.engine.raw(), G, ffi::MPVSS_HOST, p.as_ptr(), k.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipSecp256k1Group::exp");
        point_from(&out).expect("engine returns canonical encodings")
    }

    /// secp256k1.rs:102-107
    fn mul(&self, a: &AffinePoint, b: &AffinePoint) -> AffinePoint {
        let (x, y) = (point_bytes(a), point_bytes(b));
        let mut out = [0u8; 33];
        let rc = unsafe { ffi::mpvss_ec_batch_mul(self.engine.raw(), G, ffi::MPVSS_HOST, x.as_ptr(), y.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipSecp256k1Group::mul");
        point_from(&out).expect("engine returns canonical encodings")
    }

    fn scalar_inverse(&self, x: &Scalar) -> Option<Scalar> { Option::from(x.invert()) }                 // secp256k1.rs:109-112
    fn element_inverse(&self, x: &AffinePoint) -> Option<AffinePoint> { Some((-ProjectivePoint::from(*x)).to_affine()) }   // :114-119

    /// secp256k1.rs:121-131
    fn hash_to_scalar(&self, data: &[u8]) -> Scalar {
        let mut out = [0u8; 32];
        unsafe { ffi::mpvss_ec_hash_to_scalar(G, data.as_ptr(), data.len(), out.as_mut_ptr()) };
        <Scalar as Reduce<U256>>::reduce_bytes(FieldBytes::from_slice(&out))
    }

    fn element_to_bytes(&self, elem: &AffinePoint) -> Vec<u8> { point_bytes(elem).to_vec() }             // secp256k1.rs:133-136
    fn bytes_to_element(&self, bytes: &[u8]) -> Option<AffinePoint> {                                   // secp256k1.rs:138-152
        if bytes.len() != 33 { return None; }
        let mut b = [0u8; 33];
        b.copy_from_slice(bytes);
        point_from(&b)
    }
    fn scalar_to_bytes(&self, scalar: &Scalar) -> Vec<u8> { scalar_bytes(scalar).to_vec() }              // secp256k1.rs:154-156

    /// secp256k1.rs:158-166
    fn generate_private_key(&self) -> Scalar {
        let mut bytes = [0u8; 32];
        rand::Rng::fill(&mut rand::thread_rng(), &mut bytes);
        <Scalar as Reduce<U256>>::reduce_bytes(FieldBytes::from_slice(&bytes))
    }

    /// secp256k1.rs:168-171: k G through the fixed-base comb
    fn generate_public_key(&self, private_key: &Scalar) -> AffinePoint {
        let k = scalar_bytes(private_key);
        let mut out = [0u8; 33];
        let rc = unsafe { ffi::mpvss_ec_batch_exp_generator(self.engine.raw(), G, ffi::MPVSS_HOST, k.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipSecp256k1Group::generate_public_key");
        point_from(&out).expect("engine returns canonical encodings")
    }

    fn scalar_mul(&self, a: &Scalar, b: &Scalar) -> Scalar { a * b }                                     // secp256k1.rs:173-176
    fn scalar_sub(&self, a: &Scalar, b: &Scalar) -> Scalar { a - b }                                     // secp256k1.rs:178-181
}

/// BigInt (below the order) -> Scalar as participant.rs:1134-1143 does it: right-aligned big-endian bytes
pub(crate) fn scalar_from_bigint(v: &BigInt) -> Scalar {
    let bytes = v.to_bytes_be().1;
    let mut fb = [0u8; 32];
    let take = bytes.len().min(32);
    fb[32 - take..].copy_from_slice(&bytes[..take]);
    scalar_from(&fb)
}
#[allow(dead_code)]
pub(crate) fn bigint_from_scalar(s: &Scalar) -> BigInt { BigInt::from_bytes_be(Sign::Plus, &scalar_bytes(s)) }
