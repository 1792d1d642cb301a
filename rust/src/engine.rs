//! Owner of one `mpvss_ctx` (one engine bound to one GPU).  `Send + Sync`: the C library serialises calls on a
//! context internally, which is what `Group: Send + Sync` (src/group.rs:24) and rayon's use of `exp` in
//! `reconstruct` (src/participant.rs:490-500) need.
use std::ffi::CStr;
use std::fmt;
use std::ptr;
use std::sync::{Arc, OnceLock};

use crate::ffi;

#[derive(Debug, Clone)]
pub struct EngineError {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for EngineError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "mpvss_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for EngineError {}

#[derive(Debug)]
struct Ctx(*mut ffi::mpvss_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::mpvss_ctx_destroy(self.0) }
    }
}

/// Shared handle; cloning is cheap (the groups keep one in their `Arc`).
#[derive(Debug, Clone)]
pub struct Engine {
    ctx: Arc<Ctx>,
}

impl Engine {
    /// Fails when no HIP device is visible: there is no CPU fallback.
    pub fn new(device_id: i32) -> Result<Engine, EngineError> {
        let mut raw: *mut ffi::mpvss_ctx = ptr::null_mut();
        let rc = unsafe { ffi::mpvss_ctx_create(device_id, &mut raw) };
        if rc != ffi::MPVSS_OK || raw.is_null() {
            return Err(EngineError { code: rc, message: "mpvss_ctx_create failed (no MI355X / HIP device?)".into() });
        }
        Ok(Engine { ctx: Arc::new(Ctx(raw)) })
    }

    pub fn device_count() -> i32 {
        unsafe { ffi::mpvss_device_count() }
    }

    /// The process's shared engine: one context on device `MPVSS_DEVICE` (default 0), created on first use.  `HipModpGroup::new()`
    /// and its curve twins hand every caller this one, as `ModpGroup::new()` hands every caller the same group parameters:
    /// participants built from separate `new()` calls then share one block pipeline, and T threads each verifying its own dealer's
    /// box (`verify_distribution_shares`, one box per call) keep T boxes in flight on the GPU.
    pub fn shared() -> Engine {
        static SHARED: OnceLock<Engine> = OnceLock::new();
        SHARED
            .get_or_init(|| {
                let device = std::env::var("MPVSS_DEVICE").ok().and_then(|v| v.parse::<i32>().ok()).unwrap_or(0);
                Engine::new(device).expect("MI355X engine")
            })
            .clone()
    }

    pub(crate) fn raw(&self) -> *mut ffi::mpvss_ctx {
        self.ctx.0
    }

    /// Maps a return code to `Result`, attaching `mpvss_last_error`.
    pub(crate) fn check(&self, rc: i32) -> Result<(), EngineError> {
        if rc == ffi::MPVSS_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(ffi::mpvss_last_error(self.ctx.0)) }.to_string_lossy().into_owned();
        Err(EngineError { code: rc, message: msg })
    }

    /// Blocks of the block API in flight in this context, and how many of them still have GPU work pending.
    pub fn blocks_in_flight(&self) -> Result<(i32, i32), EngineError> {
        let (mut a, mut b) = (0i32, 0i32);
        self.check(unsafe { ffi::mpvss_blocks_in_flight(self.ctx.0, &mut a, &mut b) })?;
        Ok((a, b))
    }

    /// Takes the oldest MODP distribution block in flight; the ticket counts the blocks of this context in enqueue
    /// order.  For callers whose transcript state arrives from elsewhere (the previous rank of a sharded
    /// verification): claim, fetch the state of box `ticket`, then `absorb_claimed`.
    pub fn block_claim(&self) -> Result<u64, EngineError> {
        let mut t: u64 = 0;
        self.check(unsafe { ffi::mpvss_block_claim(self.ctx.0, &mut t) })?;
        Ok(t)
    }

    /// Waits for the claimed block's GPU work and extends `state` (MPVSS_TRANSCRIPT_STATE_BYTES) with its shares.
    pub fn absorb_claimed(&self, ticket: u64, state: &mut [u8]) -> Result<(), EngineError> {
        assert!(state.len() >= ffi::MPVSS_TRANSCRIPT_STATE_BYTES);
        self.check(unsafe {
            ffi::mpvss_modp_verify_block_absorb_claimed(self.ctx.0, ticket, state.as_mut_ptr(), ptr::null_mut(), ptr::null_mut(), ptr::null_mut())
        })
    }

    /// `batch::verify_many` builds per-key tables by itself for a public-key array that at least `min_boxes` large boxes of one
    /// call present (the same `Vec<u8>` of flattened keys: same pointer, same n) and frees them when the call returns -- what a
    /// verifier of many dealers' boxes against the same participants gets without changing its calls
    /// (`Participant::verify_distribution_shares`, src/participant.rs:399-455, with the same `publickeys` every time).  0: off (the
    /// default).  Returns the previous setting.
    pub fn set_key_cache(&self, min_boxes: i32) -> Result<i32, EngineError> {
        let rc = unsafe { ffi::mpvss_ctx_set_key_cache(self.ctx.0, min_boxes) };
        if rc < 0 {
            self.check(rc)?;
        }
        Ok(rc)
    }

    /// Key tables ACROSS calls for one-box callers (`Participant::verify_distribution_shares` against the same participants again
    /// and again): the engine recognises a key array by its SHA-256, builds its per-key tables at the `min_sightings`-th box and keeps
    /// at most `max_sets` sets (19.3 GB per 65536 keys), least recently used first out.  Dealers (`Participant::distribute_secret`) to
    /// the same participants take their Y_i = y_i^P(i) and a2_i = y_i^w_i from the same tables.  0: off.  Returns the previous `max_sets`.
    pub fn set_key_cache_lru(&self, max_sets: i32, min_sightings: i32) -> Result<i32, EngineError> {
        let rc = unsafe { ffi::mpvss_ctx_set_key_cache_lru(self.ctx.0, max_sets, min_sightings) };
        if rc < 0 {
            self.check(rc)?;
        }
        Ok(rc)
    }

    /// For trait methods that cannot return an error (`Group::exp`): a failing engine is a programmer / hardware
    /// error there, as a panic in the reference's arithmetic crates would be.
    pub(crate) fn expect(&self, rc: i32, what: &str) {
        if let Err(e) = self.check(rc) {
            panic!("{what}: {e}");
        }
    }
}
