//! `Participant<G>` with the reference's method surface (src/participant.rs:64-155 generic part, :158-561 MODP, :1085-1558
//! secp256k1, :1564-2003 ristretto255) over the HIP-backed groups.
//!
//! Why a type of this crate and not an extension trait on `mpvss_rs::Participant<HipModpGroup>`: the reference keeps
//! `Participant::group` private (participant.rs:65) and offers no accessor, and its hot methods live in hand-specialised
//! `impl Participant<ModpGroup>` / `<Secp256k1Group>` / `<Ristretto255Group>` blocks (participant.rs:158, 1085, 1564) -- a
//! `Participant<HipModpGroup>` of the reference has `new` / `with_arc` / `initialize` and nothing to reach its group with.  So the
//! drop-in is this type: same fields (`privatekey`, `publickey` public), same constructors, same five methods with the same
//! names, argument order and types (the group type aside).  A program written against the reference changes its `use` lines and
//! the group type (the diff is in INTEGRATION.md section 1); rust/examples/roundtrip.rs runs the whole protocol through this surface.
//! tests/test_capi_host.py holds the method names and arities of the three blocks below against a listing of the reference's
//! (tests/reference_api/reference_api.json, made by tools/gen_reference_api.py in the build container).
//!
//! Bodies are the batched calls of [`crate::batch`]: one library call per method, whatever n.  Every method may be called from
//! many threads at once on participants that share an engine (`Group: Send + Sync`, group.rs:24): the library keeps one box per
//! caller in flight, which is how a verifier of many dealers reaches the pipeline's throughput through this API.
//!
//! Never compiled in this repository's environment (no Rust toolchain).
use std::sync::Arc;

use curve25519_dalek::ristretto::RistrettoPoint;
use curve25519_dalek::scalar::Scalar as RistrettoScalar;
use k256::{AffinePoint, Scalar};
use mpvss_rs::group::Group;
use mpvss_rs::sharebox::{DistributionSharesBox, ShareBox};
use num_bigint::BigInt;

use crate::batch;
use crate::groups::{HipModpGroup, HipRistretto255Group, HipSecp256k1Group};

/// participant.rs:63-68
#[derive(Debug)]
pub struct Participant<G: Group> {
    group: Arc<G>,
    pub privatekey: G::Scalar,
    pub publickey: G::Element,
}

/// participant.rs:72-84 (no `G: Clone` bound)
impl<G: Group> Clone for Participant<G>
where
    G::Scalar: Clone,
    G::Element: Clone,
{
    fn clone(&self) -> Self {
        Participant { group: Arc::clone(&self.group), privatekey: self.privatekey.clone(), publickey: self.publickey.clone() }
    }
}

impl<G: Group> Participant<G> {
    /// participant.rs:98-108
    pub fn with_arc(group: Arc<G>) -> Self
    where
        G::Scalar: Default,
        G::Element: Default,
    {
        Participant { group, privatekey: Default::default(), publickey: Default::default() }
    }

    /// participant.rs:126-136
    pub fn new(group: G) -> Self
    where
        G::Scalar: Default,
        G::Element: Default,
    {
        Participant { group: Arc::new(group), privatekey: Default::default(), publickey: Default::default() }
    }

    /// participant.rs:139-146
    pub fn initialize(&mut self)
    where
        G::Scalar: Default,
        G::Element: Default,
    {
        self.privatekey = self.group.generate_private_key();
        self.publickey = self.group.generate_public_key(&self.privatekey);
    }

    /// The group this participant works in (not in the reference, whose field is private; the batch API of this crate takes it).
    pub fn group(&self) -> &Arc<G> {
        &self.group
    }
}

/// participant.rs:158-561
impl Participant<HipModpGroup> {
    /// participant.rs:160-286
    pub fn distribute_secret(&mut self, secret: &BigInt, publickeys: &[BigInt], threshold: u32) -> DistributionSharesBox<HipModpGroup> {
        batch::distribute_secret(&self.group, secret, publickeys, threshold)
    }

    /// participant.rs:294-353
    pub fn extract_secret_share(&self, shares_box: &DistributionSharesBox<HipModpGroup>, private_key: &BigInt, w: &BigInt)
        -> Option<ShareBox<HipModpGroup>> {
        batch::extract_secret_shares(&self.group, shares_box, std::slice::from_ref(private_key), std::slice::from_ref(w)).pop().flatten()
    }

    /// participant.rs:361-386
    pub fn verify_share(&self, sharebox: &ShareBox<HipModpGroup>, distribution_sharebox: &DistributionSharesBox<HipModpGroup>,
                        publickey: &BigInt) -> bool {
        batch::verify_shares(&self.group, std::slice::from_ref(sharebox), std::slice::from_ref(publickey), distribution_sharebox)[0]
    }

    /// participant.rs:399-455
    pub fn verify_distribution_shares(&self, distribute_sharesbox: &DistributionSharesBox<HipModpGroup>) -> bool {
        batch::verify_distribution_shares(&self.group, distribute_sharesbox)
    }

    /// participant.rs:462-519
    pub fn reconstruct(&self, share_boxes: &[ShareBox<HipModpGroup>], distribute_share_box: &DistributionSharesBox<HipModpGroup>)
        -> Option<BigInt> {
        batch::reconstruct(&self.group, share_boxes, distribute_share_box)
    }
}

/// participant.rs:1085-1558
impl Participant<HipSecp256k1Group> {
    /// participant.rs:1094-1274
    pub fn distribute_secret(&mut self, secret: &BigInt, publickeys: &[AffinePoint], threshold: u32)
        -> DistributionSharesBox<HipSecp256k1Group> {
        batch::ec_distribute_secret(self.group.as_ref(), secret, publickeys, threshold)
    }

    /// participant.rs:1282-1338
    pub fn extract_secret_share(&self, shares_box: &DistributionSharesBox<HipSecp256k1Group>, private_key: &Scalar, w: &Scalar)
        -> Option<ShareBox<HipSecp256k1Group>> {
        batch::ec_extract_secret_shares(self.group.as_ref(), shares_box, std::slice::from_ref(private_key), std::slice::from_ref(w)).pop().flatten()
    }

    /// participant.rs:1346-1371
    pub fn verify_share(&self, sharebox: &ShareBox<HipSecp256k1Group>, distribution_sharebox: &DistributionSharesBox<HipSecp256k1Group>,
                        publickey: &AffinePoint) -> bool {
        batch::ec_verify_shares(self.group.as_ref(), std::slice::from_ref(sharebox), std::slice::from_ref(publickey), distribution_sharebox)[0]
    }

    /// participant.rs:1384-1444
    pub fn verify_distribution_shares(&self, distribute_sharesbox: &DistributionSharesBox<HipSecp256k1Group>) -> bool {
        batch::ec_verify_distribution_shares(self.group.as_ref(), distribute_sharesbox)
    }

    /// participant.rs:1452-1516
    pub fn reconstruct(&self, share_boxes: &[ShareBox<HipSecp256k1Group>], distribute_share_box: &DistributionSharesBox<HipSecp256k1Group>)
        -> Option<BigInt> {
        batch::ec_reconstruct(self.group.as_ref(), share_boxes, distribute_share_box)
    }
}

/// participant.rs:1564-2003
impl Participant<HipRistretto255Group> {
    /// participant.rs:1573-1717
    pub fn distribute_secret(&mut self, secret: &BigInt, publickeys: &[RistrettoPoint], threshold: u32)
        -> DistributionSharesBox<HipRistretto255Group> {
        batch::ec_distribute_secret(self.group.as_ref(), secret, publickeys, threshold)
    }

    /// participant.rs:1725-1781
    pub fn extract_secret_share(&self, shares_box: &DistributionSharesBox<HipRistretto255Group>, private_key: &RistrettoScalar,
                                w: &RistrettoScalar) -> Option<ShareBox<HipRistretto255Group>> {
        batch::ec_extract_secret_shares(self.group.as_ref(), shares_box, std::slice::from_ref(private_key), std::slice::from_ref(w)).pop().flatten()
    }

    /// participant.rs:1789-1814
    pub fn verify_share(&self, sharebox: &ShareBox<HipRistretto255Group>, distribution_sharebox: &DistributionSharesBox<HipRistretto255Group>,
                        publickey: &RistrettoPoint) -> bool {
        batch::ec_verify_shares(self.group.as_ref(), std::slice::from_ref(sharebox), std::slice::from_ref(publickey), distribution_sharebox)[0]
    }

    /// participant.rs:1827-1887
    pub fn verify_distribution_shares(&self, distribute_sharesbox: &DistributionSharesBox<HipRistretto255Group>) -> bool {
        batch::ec_verify_distribution_shares(self.group.as_ref(), distribute_sharesbox)
    }

    /// participant.rs:1895-1953
    pub fn reconstruct(&self, share_boxes: &[ShareBox<HipRistretto255Group>], distribute_share_box: &DistributionSharesBox<HipRistretto255Group>)
        -> Option<BigInt> {
        batch::ec_reconstruct(self.group.as_ref(), share_boxes, distribute_share_box)
    }
}

/// `mpvss_rs::ModpParticipant` / `Secp256k1Participant` / `Ristretto255Participant` (src/lib.rs:36-46)
pub type ModpParticipant = Participant<HipModpGroup>;
pub type Secp256k1Participant = Participant<HipSecp256k1Group>;
pub type Ristretto255Participant = Participant<HipRistretto255Group>;
