// pin_draws.rs — drop into rustlight/examples/ and `cargo run --release --example=pin_draws`.
// Prints what oracle/pin/make_pin_inputs.py writes to expected_draws.txt, from the real rand 0.8.5 crate:
// 64 next_u64 of SmallRng::seed_from_u64(0), 8 gen::<f32>() bit patterns of a fresh stream, and the first four per-block seeds
// generate_img_blocks would fork for `-r independent:0` (one next_u64 of the master per 16x16 block, x-major).
use rand::rngs::SmallRng;
use rand::{Rng, RngCore, SeedableRng};

fn main() {
    let mut r = SmallRng::seed_from_u64(0);
    for _ in 0..64 {
        println!("u64 {:016x}", r.next_u64());
    }
    let mut f = SmallRng::seed_from_u64(0);
    for _ in 0..8 {
        println!("f32 {:08x}", f.gen::<f32>().to_bits());
    }
    let mut master = SmallRng::seed_from_u64(0);
    for _ in 0..4 {
        let seed = master.next_u64(); // IndependentSampler::clone_box: SmallRng::seed_from_u64(self.rnd.next_u64())
        let mut block = SmallRng::seed_from_u64(seed);
        println!("blk {:016x} {:016x}", seed, block.next_u64());
    }
}
