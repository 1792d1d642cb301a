//! The whole protocol once per group -- deal, verify the distribution, extract, verify the shares, reconstruct -- through
//! `mpvss_hip::Participant`, i.e. the reference's `Participant<G>` calls (participant.rs:160, 294, 361, 399, 462 and their curve
//! twins) with the group type swapped.  A program written against the reference (its examples/mpvss_all*.rs) needs exactly the
//! edit INTEGRATION.md section 2 shows: `use mpvss_hip::Participant;` for `use mpvss_rs::Participant;` and `HipModpGroup` for
//! `ModpGroup`.  Not a copy of those examples: one macro instantiated three times, five participants, threshold three.
//!
//! Never compiled in this repository's environment (no Rust toolchain).
use mpvss_hip::groups::{HipModpGroup, HipRistretto255Group, HipSecp256k1Group};
use mpvss_hip::{string_from_secret, string_to_secret, Participant};
use mpvss_rs::group::Group;

macro_rules! roundtrip {
    ($group_ty:ty, $label:expr) => {{
        let group = <$group_ty>::new();
        let message = format!("five keys, any three open this ({})", $label);
        let mut dealer = Participant::with_arc(group.clone());
        dealer.initialize();
        let mut holders: Vec<_> = (0..5).map(|_| Participant::with_arc(<$group_ty>::new())).collect();
        holders.iter_mut().for_each(|h| h.initialize());
        let keys: Vec<_> = holders.iter().map(|h| h.publickey.clone()).collect();

        let shares_box = dealer.distribute_secret(&string_to_secret(&message), &keys, 3);
        // public verifiability: anybody checks the dealer, holders or not
        assert!(holders.iter().all(|h| h.verify_distribution_shares(&shares_box)));
        assert!(dealer.verify_distribution_shares(&shares_box));

        // holders 0, 2 and 4 decrypt their shares and prove it; everybody checks the proofs
        let opened: Vec<_> = [0usize, 2, 4]
            .iter()
            .map(|&i| {
                let w = group.generate_private_key();
                let sb = holders[i].extract_secret_share(&shares_box, &holders[i].privatekey, &w).expect("share of a listed key");
                assert!(holders[(i + 1) % 5].verify_share(&sb, &shares_box, &holders[i].publickey));
                sb
            })
            .collect();
        let secret = holders[1].reconstruct(&opened, &shares_box).expect("three of five shares");
        assert_eq!(string_from_secret(&secret), message);
        // two shares are not enough
        assert!(holders[1].reconstruct(&opened[..2], &shares_box).is_none());
        println!("{}: {}", $label, string_from_secret(&secret));
    }};
}

fn main() {
    mpvss_hip::process_init();      // before anything touches HIP: 8 hardware queues for the block pipeline
    roundtrip!(HipModpGroup, "MODP-2048");
    roundtrip!(HipSecp256k1Group, "secp256k1");
    roundtrip!(HipRistretto255Group, "ristretto255");
}
