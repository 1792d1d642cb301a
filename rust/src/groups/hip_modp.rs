//! `impl Group for HipModpGroup`: src/groups/modp.rs:29-197 with `exp` / `mul` on the GPU.
use std::sync::Arc;

use mpvss_rs::group::Group;
use num_bigint::{BigInt, RandBigInt, Sign};
use num_integer::Integer;
use num_traits::{One, Zero};

use super::be256;
use crate::engine::Engine;
use crate::ffi;

#[derive(Debug, Clone)]
pub struct HipModpGroup {
    pub(crate) engine: Engine,
    q: BigInt,         // RFC 3526 group 14 (modp.rs:47-58)
    g: BigInt,         // (q - 1) / 2
    order: BigInt,     // q - 1
    gen_main: BigInt,  // 2
    gen_sub: BigInt,   // 4
}

impl HipModpGroup {
    /// Same construction convention as `ModpGroup::new()` (modp.rs:44-69): an `Arc` to share between participants.  The engine
    /// is the process's shared one for device `MPVSS_DEVICE` (default 0): every group made this way talks to the same context,
    /// so participants built from separate `new()` calls (examples/mpvss_all.rs:11-16) still share one pipeline.
    pub fn new() -> Arc<Self> {
        Self::with_engine(Engine::shared())
    }

    /// A group bound to a context of its own on GPU `device_id`.
    pub fn with_device(device_id: i32) -> Arc<Self> {
        Self::with_engine(Engine::new(device_id).expect("MI355X engine"))
    }

    fn with_engine(engine: Engine) -> Arc<Self> {
        let q = BigInt::parse_bytes(
            b"ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404dd\
              ef9519b3cd3a431b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7ed\
              ee386bfb5a899fa5ae9f24117c4b1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f\
              83655d23dca3ad961c62f356208552bb9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3b\
              e39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6955817183995497cea956ae515d2261898fa0510\
              15728e5a8aacaa68ffffffffffffffff",
            16,
        )
        .unwrap();
        let order = &q - BigInt::one();
        let g = &order / BigInt::from(2);
        Arc::new(HipModpGroup { engine, q, g, order, gen_main: BigInt::from(2), gen_sub: BigInt::from(4) })
    }

    fn from_be256(b: &[u8; 256]) -> BigInt {
        BigInt::from_bytes_be(Sign::Plus, b)
    }

    /// modp.rs:87-89 (inherent, like the reference's: `group.modulus().to_biguint()` in examples/mpvss_all.rs:42)
    pub fn modulus(&self) -> &BigInt {
        &self.q
    }

    /// modp.rs:92-94
    pub fn subgroup_order_value(&self) -> &BigInt {
        &self.g
    }
}

/// x with a x = 1 (mod m), or None when gcd(a, m) != 1 -- the contract of the reference's crate-private `Util::mod_inverse`
/// (util.rs:33-41; `mod util` is private in src/lib.rs:27, so a crate outside the reference carries its own).  Iterative extended
/// Euclid on (m, a mod m): the same residue in [0, m) as the reference's recursive form returns.
pub(crate) fn mod_inverse(a: &BigInt, m: &BigInt) -> Option<BigInt> {
    let (mut r0, mut r1) = (m.clone(), a.mod_floor(m));
    let (mut t0, mut t1) = (BigInt::zero(), BigInt::one());
    while !r1.is_zero() {
        let (q, r2) = r0.div_mod_floor(&r1);
        let t2 = &t0 - &q * &t1;
        r0 = r1;
        r1 = r2;
        t0 = t1;
        t1 = t2;
    }
    if r0 != BigInt::one() {
        return None;
    }
    Some(t0.mod_floor(m))
}

impl Group for HipModpGroup {
    type Scalar = BigInt;
    type Element = BigInt;

    fn order(&self) -> &BigInt { &self.order }
    fn subgroup_order(&self) -> &BigInt { &self.g }
    fn generator(&self) -> BigInt { self.gen_main.clone() }
    fn subgroup_generator(&self) -> BigInt { self.gen_sub.clone() }
    fn identity(&self) -> BigInt { BigInt::one() }

    /// modp.rs:122-128 `base.modpow(scalar, q)` -> `mpvss_modp_batch_exp` with n = 1
    fn exp(&self, base: &BigInt, scalar: &BigInt) -> BigInt {
        let (b, e) = (be256(base), be256(scalar));
        let mut out = [0u8; 256];
        let rc = unsafe { ffi::mpvss_modp_batch_exp(self.engine.raw(), ffi::MPVSS_HOST, b.as_ptr(), e.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipModpGroup::exp");
        Self::from_be256(&out)
    }

    /// modp.rs:130-132 `(a * b) % q`
    fn mul(&self, a: &BigInt, b: &BigInt) -> BigInt {
        let (x, y) = (be256(a), be256(b));
        let mut out = [0u8; 256];
        let rc = unsafe { ffi::mpvss_modp_batch_mul(self.engine.raw(), ffi::MPVSS_HOST, x.as_ptr(), y.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipModpGroup::mul");
        Self::from_be256(&out)
    }

    fn scalar_inverse(&self, x: &BigInt) -> Option<BigInt> { mod_inverse(x, &self.order) }   // modp.rs:134-136
    fn element_inverse(&self, x: &BigInt) -> Option<BigInt> { mod_inverse(x, &self.q) }      // modp.rs:138-140

    /// modp.rs:142-148
    fn hash_to_scalar(&self, data: &[u8]) -> BigInt {
        let mut out = [0u8; 256];
        unsafe { ffi::mpvss_modp_hash_to_scalar(data.as_ptr(), data.len(), out.as_mut_ptr()) };
        Self::from_be256(&out).mod_floor(&self.g)
    }

    fn element_to_bytes(&self, elem: &BigInt) -> Vec<u8> { elem.to_bytes_be().1 }                       // modp.rs:150-152
    fn bytes_to_element(&self, bytes: &[u8]) -> Option<BigInt> { Some(BigInt::from_bytes_be(Sign::Plus, bytes)) }   // :154-156
    fn scalar_to_bytes(&self, scalar: &BigInt) -> Vec<u8> { scalar.to_bytes_be().1 }                   // modp.rs:158-160

    /// modp.rs:162-174
    fn generate_private_key(&self) -> BigInt {
        let mut rng = rand::thread_rng();
        loop {
            let k = rng.gen_bigint_range(&BigInt::zero(), &self.q);
            if k.gcd(&self.order) == BigInt::one() {
                return k;
            }
        }
    }

    /// modp.rs:176-178: G^k through the fixed-base comb
    fn generate_public_key(&self, private_key: &BigInt) -> BigInt {
        let (g, e) = (be256(&self.gen_main), be256(private_key));
        let mut out = [0u8; 256];
        let rc = unsafe { ffi::mpvss_modp_batch_exp_fixed_base(self.engine.raw(), ffi::MPVSS_HOST, g.as_ptr(), e.as_ptr(), 1, out.as_mut_ptr()) };
        self.engine.expect(rc, "HipModpGroup::generate_public_key");
        Self::from_be256(&out)
    }

    /// modp.rs:180-182
    fn scalar_mul(&self, a: &BigInt, b: &BigInt) -> BigInt {
        let (x, y) = (be256(a), be256(b));
        let mut out = [0u8; 256];
        unsafe { ffi::mpvss_modp_scalar_mul(x.as_ptr(), y.as_ptr(), out.as_mut_ptr()) };
        Self::from_be256(&out)
    }

    /// modp.rs:184-192
    fn scalar_sub(&self, a: &BigInt, b: &BigInt) -> BigInt {
        let (x, y) = (be256(a), be256(b));
        let mut out = [0u8; 256];
        unsafe { ffi::mpvss_modp_scalar_sub(x.as_ptr(), y.as_ptr(), out.as_mut_ptr()) };
        Self::from_be256(&out)
    }

    fn modulus(&self) -> Option<&BigInt> { Some(&self.q) }                                             // modp.rs:194-196
}
