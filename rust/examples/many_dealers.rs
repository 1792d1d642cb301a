//! One verifier, many dealers: T threads, each calling the reference-shaped `verify_distribution_shares(&box)` -- ONE box per call
//! (participant.rs:399-455) -- on participants that share the process's engine.  The library keeps one box per caller in flight, so
//! the threads together run at the block pipeline's rate (bench.py `drop_in`: 12-16 callers reach what `batch::verify_many`
//! reaches with its own threads); a lone caller gets half of that, its box's latency.  This is how the crate itself goes
//! parallel (rayon over shares, participant.rs:490-500) -- here over dealers, with plain threads to stay free of extra crates.
//!
//! usage: many_dealers [participants] [threshold] [dealers] [threads] [key cache: 0 | 1]
//! key cache 1 (boxes of more than 16384 shares): the engine builds per-key tables for the participants' keys once and both the
//! dealers' `distribute_secret` and the verifiers' calls take them (`Engine::set_key_cache_lru`; 295 KB of HBM per key).
//! Never compiled in this repository's environment (no Rust toolchain).
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::Arc;
use std::time::Instant;

use mpvss_hip::groups::HipModpGroup;
use mpvss_hip::{string_to_secret, Participant};
use mpvss_rs::group::Group;

fn main() {
    mpvss_hip::process_init();
    let arg = |i: usize, d: usize| std::env::args().nth(i).and_then(|v| v.parse().ok()).unwrap_or(d);
    let (n, t, dealers, threads) = (arg(1, 4096), arg(2, 64) as u32, arg(3, 24), arg(4, 12));
    let group = HipModpGroup::new();
    if arg(5, 0) == 1 {
        mpvss_hip::Engine::shared().set_key_cache_lru(1, 1).expect("key cache");
    }
    // long-lived participant keys (batched key generation would be one call: batch_exp_fixed_base)
    let keys: Vec<_> = (0..n).map(|_| group.generate_public_key(&group.generate_private_key())).collect();
    let boxes: Vec<_> = (0..dealers)
        .map(|d| {
            let mut dealer = Participant::with_arc(group.clone());
            dealer.initialize();
            dealer.distribute_secret(&string_to_secret(&format!("dealer {d}")), &keys, t)
        })
        .collect();
    let boxes = Arc::new(boxes);
    let next = Arc::new(AtomicUsize::new(0));
    let started = Instant::now();
    let workers: Vec<_> = (0..threads)
        .map(|_| {
            let (boxes, next) = (Arc::clone(&boxes), Arc::clone(&next));
            let verifier = Participant::with_arc(HipModpGroup::new());      // shares the process's engine
            std::thread::spawn(move || {
                let mut ok = 0usize;
                loop {
                    let i = next.fetch_add(1, Ordering::Relaxed);
                    if i >= boxes.len() {
                        return ok;
                    }
                    ok += verifier.verify_distribution_shares(&boxes[i]) as usize;      // one box per call
                }
            })
        })
        .collect();
    let verified: usize = workers.into_iter().map(|w| w.join().unwrap()).sum();
    let secs = started.elapsed().as_secs_f64();
    assert_eq!(verified, dealers);
    println!("{dealers} boxes of {n} shares by {threads} callers: {:.3} M share verifications/s", (dealers * n) as f64 / secs / 1e6);
}
