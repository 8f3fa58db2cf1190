// frames.cpp — rl_render_path_frames: frames in flight behind one C-ABI call (include/rustlight_amd.h).  Host only: K contexts of one scene, one std::thread each, every
// frame an ordinary rl_render_path on its context (contexts share nothing mutable; each has its own stream) — the idle tail of one frame's launch chain is filled by
// another frame's workgroups ("Frames in flight").  What the progressive wrappers of the reference (src/integrators/avg.rs:5-131, equal_time.rs:4-66: N
// independent renders of one scene) can call as it is.
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rustlight_amd.h"
#include "../kernels/wavefront.h"     // rl_set_error

extern "C" int rl_render_path_frames(rl_context* const* ctxs, size_t n_ctx, const rl_path_params* params, const uint64_t* const* block_seeds, size_t n_blocks,
                                     size_t n_frames, float* const* out_rgb, rl_render_stats* stats) {
    if (!ctxs || n_ctx == 0 || !params || (n_frames && (!block_seeds || !out_rgb))) { rl_set_error("rl_render_path_frames: null argument"); return RL_ERR_INVALID_ARGUMENT; }
    for (size_t c = 0; c < n_ctx; c++) {
        if (!ctxs[c]) { rl_set_error("rl_render_path_frames: null context"); return RL_ERR_INVALID_ARGUMENT; }
        for (size_t d = 0; d < c; d++)
            if (ctxs[d] == ctxs[c]) { rl_set_error("rl_render_path_frames: the contexts must be distinct (a context renders one frame at a time)"); return RL_ERR_INVALID_ARGUMENT; }
    }
    for (size_t f = 0; f < n_frames; f++)
        if (!block_seeds[f] || !out_rgb[f]) { rl_set_error("rl_render_path_frames: null frame argument"); return RL_ERR_INVALID_ARGUMENT; }
    const size_t k = std::min(n_ctx, std::max<size_t>(n_frames, 1));
    std::vector<int> rc(k, RL_OK);
    std::vector<std::string> msg(k);
    std::atomic<bool> failed{false};         // one frame failed: the other threads stop before their next frame instead of rendering the rest of a call that already has its answer
    auto work = [&](size_t c) {
        for (size_t f = c; f < n_frames && !failed.load(std::memory_order_relaxed); f += k) {
            rl_render_stats st{};
            const int r = rl_render_path(ctxs[c], params, block_seeds[f], n_blocks, out_rgb[f], 0, nullptr, &st);
            if (stats) stats[f] = st;
            if (r != RL_OK) { rc[c] = r; msg[c] = rl_last_error(); failed.store(true, std::memory_order_relaxed); return; }      // (the error string is per thread: carried over to the caller's below)
        }
    };
    if (k == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t c = 0; c < k; c++) th.emplace_back(work, c);
        for (std::thread& t : th) t.join();
    }
    for (size_t c = 0; c < k; c++)
        if (rc[c] != RL_OK) { rl_set_error(msg[c]); return rc[c]; }
    return RL_OK;
}
