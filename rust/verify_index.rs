//! rust/verify_index.rs -- documentation, like ffi.rs (there is no rustc in the build image): the check a RustSolver maintainer runs ONCE per
//! abstraction before loading bucket files written by the reference's own `gen_abstraction` (`src/gen_abstraction/main.rs:372-380`) into
//! `rs_card_abs` / `rs_deal_trainer`.
//!
//! Those files are flat little-endian u32 arrays indexed by `rust_poker::hand_indexer_s::get_index` (`src/solver/card_abstraction.rs:204-209`,
//! `:227-229`).  librustsolver_amd computes the same suit-isomorphic index from the published algorithm; the partition is pinned by the reference's
//! own numbers (1 286 792 flop hands, 12 888 turn clusters) but the ORDER inside it is not -- `rust_poker` is not vendored in the reference, so
//! nothing in this repository could compare the two.  This does, on the integrator's machine, where both exist.
//!
//! Put it next to `ffi.rs` (e.g. `src/solver/gpu/verify_index.rs`), call `verify_hand_index_order()` from `MCCFRTrainer::init` or a `#[test]`.
use super::ffi::*;
use rand::seq::SliceRandom;
use rust_poker::hand_indexer_s;
use std::ffi::CStr;

/// Compares `hand_indexer_s::get_index` with `rs_hand_index` on `n` random hands of every betting round (hole cards + 3 / 4 / 5 board cards,
/// the indexers `card_abstraction.rs:88-90` builds).  Err(message) names the first hand the two disagree on.
pub fn verify_hand_index_order(n: usize) -> Result<(), String> {
    let mut rng = rand::thread_rng();
    for board_cards in 3u8..=5 {
        let cards_per_round = [2u8, board_cards];
        let theirs = hand_indexer_s::init(2, cards_per_round.to_vec());
        let mut ours: *mut rs_hand_indexer = std::ptr::null_mut();
        check(unsafe { rs_hand_indexer_create(2, cards_per_round.as_ptr(), &mut ours) })?;
        let nc = 2 + board_cards as usize;
        let mut cards = vec![0u8; n * nc];
        let mut expect = vec![0u64; n];
        let mut deck: Vec<u8> = (0..52).collect();
        for i in 0..n {
            deck.shuffle(&mut rng);
            cards[i * nc..(i + 1) * nc].copy_from_slice(&deck[..nc]);      // card = 4 * rank + suit on both sides (cfr.rs:592)
            expect[i] = theirs.get_index(&cards[i * nc..(i + 1) * nc]);    // card_abstraction.rs:205
        }
        let (mut first_bad, mut got) = (0usize, 0u64);
        let rc = unsafe { rs_hand_index_verify(ours, 1, cards.as_ptr(), n, expect.as_ptr(), &mut first_bad, &mut got) };
        unsafe { rs_hand_indexer_destroy(ours) };
        // sizes first: a different partition would be a different algorithm altogether
        if rc == RS_ERR_MISMATCH {
            return Err(format!(
                "hand index ORDER differs on round with {} board cards: hand {:?} is {} in rust_poker, {} in librustsolver_amd -- do not load bucket \
                 files written by gen_abstraction; re-index them (rs_hand_unindex -> get_index) or generate them through rs_kmeans_* instead",
                board_cards, &cards[first_bad * nc..(first_bad + 1) * nc], expect[first_bad], got));
        }
        check(rc)?;
    }
    Ok(())
}

fn check(rc: std::os::raw::c_int) -> Result<(), String> {
    if rc == 0 { Ok(()) } else { Err(unsafe { CStr::from_ptr(rs_last_error()) }.to_string_lossy().into_owned()) }
}
