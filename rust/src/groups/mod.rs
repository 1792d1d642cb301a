//! HIP-backed counterparts of `mpvss_rs::groups::{ModpGroup, Secp256k1Group, Ristretto255Group}`.
mod hip_modp;
mod hip_ristretto255;
mod hip_secp256k1;

/// boundary encodings of the two curve groups (crate::batch::CurveCodec)
pub(crate) mod secp {
    pub(crate) use super::hip_secp256k1::{point_bytes, point_from, scalar_bytes, scalar_from, scalar_from_bigint};
}
pub(crate) mod rist {
    pub(crate) use super::hip_ristretto255::{point_bytes, point_from};
}

pub use hip_modp::HipModpGroup;
pub use hip_ristretto255::HipRistretto255Group;
pub use hip_secp256k1::HipSecp256k1Group;

/// 256-byte big-endian, zero padded: the MODP encoding of the boundary (include/mpvss_hip.h).
pub(crate) fn be256(v: &num_bigint::BigInt) -> [u8; 256] {
    let (_, bytes) = v.to_bytes_be();
    let mut out = [0u8; 256];
    let take = bytes.len().min(256);
    out[256 - take..].copy_from_slice(&bytes[bytes.len() - take..]);
    out
}
