//! HIP-backed `Group` implementations and batched Participant paths for the `mpvss-rs` crate (MI355X, gfx950).
//!
//! The reference keeps its public surface -- `trait Group` (src/group.rs:24-124), `Participant<G>`, `DLEQ<G>`,
//! `PVSS<G>` -- and this crate plugs in underneath it:
//!
//! * [`groups::HipModpGroup`], [`groups::HipSecp256k1Group`], [`groups::HipRistretto255Group`] implement `Group`
//!   with the same `Scalar` / `Element` types as the reference's groups, so they are drop-in type parameters;
//!   every `exp` / `mul` is one call into `libmpvss_hip.so` (a batch of one).
//! * [`batch`] holds what makes the GPU worthwhile: the three loops of the hot path
//!   (`distribute_secret`, `verify_distribution_shares`, `verify_share`) as single batched calls, plus batched
//!   `extract_secret_share`, `reconstruct` and key generation.
//! * [`participant::Participant`] is the reference's `Participant<G>` surface over those calls -- same constructors, fields and
//!   method names (participant.rs:158, 1085, 1564) -- so that `examples/mpvss_all*.rs` change one `use` line and the group type.
//!
//! STATUS: this crate has never been compiled -- the repository's build image has no cargo / rustc.  The C ABI
//! it binds (include/mpvss_hip.h) is what the repository tests; the C++ mirror under mpvss_rs_amd/host/ is the
//! host side that actually runs there.  `ffi.rs` is checked symbol by symbol against the header.
pub mod batch;
pub mod engine;
pub mod ffi;
pub mod groups;
pub mod participant;

pub use engine::{Engine, EngineError};
pub use groups::{HipModpGroup, HipRistretto255Group, HipSecp256k1Group};
pub use participant::{ModpParticipant, Participant, Ristretto255Participant, Secp256k1Participant};
// the reference's helpers a program written against it imports from the crate root (src/lib.rs:49-58)
pub use mpvss_rs::{string_from_secret, string_to_secret};

/// Call once at the top of `main`, before anything in the process touches HIP: asks the ROCm runtime for the 8 hardware
/// queues the block pipeline is tuned for (sets `GPU_MAX_HW_QUEUES=8` unless the variable is already set; the library
/// never edits the environment by itself).  Returns true when it set the variable.
pub fn process_init() -> bool {
    unsafe { ffi::mpvss_process_init() == 1 }
}
