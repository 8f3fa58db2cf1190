// integrator.hpp — C++ mirror of the reference's plugin surface for the `path` hot path, layered
// on the C-ABI (include/rustlight_amd.h).  Names, argument meaning and defaults follow rustlight:
//   trait Sampler / IndependentSampler      src/samplers/mod.rs:3-9, independent.rs:5-34
//   struct Scene                            src/scene.rs:16-30
//   struct BufferCollection ("primal")      src/integrators/mod.rs:48-216
//   trait Integrator::compute               src/integrators/mod.rs:219-233
//   struct IntegratorPathTracing            src/integrators/explicit/path.rs:14-20
//   IntegratorType::compute (BVH build is untimed, then "Elapsed Integrator")  mod.rs:274-338
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/rustlight_amd.h"

namespace rustlight {

struct IndependentSampler {   // owns the concrete SmallRng so the raw u64 stream is reachable
    rl_sampler rnd;
    int variant = 0;
    explicit IndependentSampler(uint64_t seed, int variant_ = 0) : variant(variant_) { rl_sampler_seed(&rnd, seed, variant_); }
    float next() { return rl_sampler_next_f32(&rnd); }
    uint64_t next_u64() { return rl_sampler_next_u64(&rnd); }
};

struct Scene {
    rl_scene* handle = nullptr;
    size_t nb_samples = 1;           // Scene::nb_samples (CLI global -n)
    std::string output_img_path = "out.pfm";
    explicit Scene(rl_scene* h) : handle(h) {}
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { rl_scene_destroy(handle); }
    static Scene* load(const std::string& path, bool use_shading_normals = true) {
        rl_scene* h = nullptr;
        int rc = rl_scene_load(path.c_str(), use_shading_normals ? 1 : 0, &h);   // SceneLoaderManager: .pbrt | .xml
        if (rc != RL_OK) throw std::runtime_error(std::string("error on loading the scene: ") + rl_last_error());
        return new Scene(h);
    }
    void build_emitters(bool build_ats = false) {   // Scene::build_emitters(build_ats) (scene.rs:53-123)
        if (rl_scene_enable_ats(handle, build_ats ? 1 : 0) != RL_OK || rl_scene_build_emitters(handle) != RL_OK)
            throw std::runtime_error(std::string("build_emitters failed: ") + rl_last_error());
    }
};

struct BufferCollection {   // only the "primal" buffer exists on this path
    uint32_t width = 0, height = 0;
    std::vector<float> primal;   // W*H*3, row-major, origin top-left
    void save(const std::string& /*name = "primal"*/, const std::string& path) const {   // Bitmap::save: .pfm | .png | .exr
        if (rl_save_image(path.c_str(), primal.data(), width, height) != RL_OK) throw std::runtime_error("cannot write " + path);
    }
    void scale(float v) { for (float& x : primal) x *= v; }                                   // Bitmap::scale
    void accumulate_bitmap(const BufferCollection& o) { for (size_t i = 0; i < primal.size(); i++) primal[i] += o.primal[i]; }
};

enum class IntegratorPathTracingStrategies { All = RL_STRATEGY_ALL, BSDF = RL_STRATEGY_BSDF, Emitter = RL_STRATEGY_EMITTER };

struct IntegratorPathTracing {
    std::optional<uint32_t> min_depth = 0, max_depth = std::nullopt, rr_depth = 0;
    IntegratorPathTracingStrategies strategy = IntegratorPathTracingStrategies::All;
    bool single_scattering = false;
    // MI355X-specific knobs (not in the reference)
    int device = 0;
    // RL_STREAM_REFERENCE_ORDER = rustlight's own stream assignment (seed-for-seed the reference's image); RL_STREAM_PER_SAMPLE is the
    // throughput decomposition (statistically the same image, ~5x faster on the Cornell box at 1080p x 128 spp since round 4): opt-in
    rl_stream_mode stream_mode = RL_STREAM_REFERENCE_ORDER;
    uint32_t numerics = RL_NUMERICS_EXACT;      // RL_NUMERICS_FAST: opt-in tolerance mode
    uint32_t shard_index = 0, shard_count = 1;
    rl_render_stats last_stats{};

    // several GPUs of one node from one process (the CLI's --gpus N): GPU g renders the blocks b % N == g on its
    // own host thread into its own framebuffer in HBM; ONE ncclReduce over xGMI merges them on the first GPU (rl_multi_*)
    int n_gpus = 1;

    // frames in flight (one GPU): how many independent frames `compute_frames` — and the progressive wrappers through it — keep on the GPU at once, one device
    // context and one host thread each.  A frame's render is a chain of dependent launches that leaves much of the chip idle at its tail (reference-order
    // streams: the chain pass ends with its slowest wave); another context's frame fills it.  The images are those of one frame after the other.
    int frames_in_flight = 1;

    // execution options handed to every device context this integrator creates (rl_context_set_option: test / measurement hooks, none changes an image;
    // e.g. {"spec_draws_per_sample", "40"}); `rustlight-amd --option name=value`
    std::vector<std::pair<std::string, std::string>> options;
    void apply_options(rl_context* c) const {
        for (const auto& o : options)
            if (rl_context_set_option(c, o.first.c_str(), o.second.c_str()) != RL_OK) throw std::runtime_error(std::string("rl_context_set_option: ") + rl_last_error());
    }

    // the device contexts (BVH + uploaded scene) are built once per scene, like `BVHAccel::new` in IntegratorType::compute
    rl_context* ctx = nullptr;
    std::vector<rl_context*> extra_ctx;      // frames in flight: contexts 2 .. frames_in_flight
    rl_multi* multi = nullptr;
    const Scene* ctx_scene = nullptr;
    int ctx_gpus = 0;
    IntegratorPathTracing() = default;
    IntegratorPathTracing(const IntegratorPathTracing&) = delete;
    ~IntegratorPathTracing() { release(); }
    void release() {
        if (ctx) rl_context_destroy(ctx);
        for (rl_context* c : extra_ctx) rl_context_destroy(c);
        extra_ctx.clear();
        if (multi) rl_multi_destroy(multi);
        ctx = nullptr; multi = nullptr;
    }
    rl_path_params params_for(const IndependentSampler& sampler, const Scene& scene) const {
        rl_path_params p;
        rl_path_params_default(&p);
        p.spp = (uint32_t)scene.nb_samples;
        p.has_min_depth = min_depth.has_value(); p.min_depth = min_depth.value_or(0);
        p.has_max_depth = max_depth.has_value(); p.max_depth = max_depth.value_or(0);
        p.has_rr_depth = rr_depth.has_value(); p.rr_depth = rr_depth.value_or(0);
        p.strategy = (int)strategy;
        p.single_scattering = single_scattering;
        p.stream_mode = stream_mode;
        p.numerics = numerics;
        p.seed_variant = sampler.variant;
        p.shard_index = shard_index; p.shard_count = shard_count;
        return p;
    }

    // `n_frames` consecutive compute() calls with up to `frames_in_flight` of them on the GPU at once: the block seeds of every frame are drawn from the master
    // sampler in call order (exactly what that many sequential calls draw), frame j renders on context j % k from host thread j % k; images in frame order.
    std::vector<BufferCollection> compute_frames(IndependentSampler& sampler, Scene& scene, size_t n_frames) {
        std::vector<BufferCollection> out;
        const size_t k = std::min<size_t>((size_t)std::max(1, frames_in_flight), std::max<size_t>(1, n_frames));
        if (k <= 1 || std::max(1, n_gpus) > 1) {
            for (size_t j = 0; j < n_frames; j++) out.push_back(compute(sampler, scene));
            return out;
        }
        if (ctx_scene != &scene || ctx_gpus != 1 || !ctx) {
            release();
            if (rl_context_create(scene.handle, device, &ctx) != RL_OK) throw std::runtime_error(std::string("rl_context_create: ") + rl_last_error());
            apply_options(ctx);
            ctx_scene = &scene; ctx_gpus = 1;
        }
        while (extra_ctx.size() + 1 < k) {
            rl_context* c = nullptr;
            if (rl_context_create(scene.handle, device, &c) != RL_OK) throw std::runtime_error(std::string("rl_context_create: ") + rl_last_error());
            extra_ctx.push_back(c);
            apply_options(c);
        }
        const rl_path_params p = params_for(sampler, scene);
        out.resize(n_frames);
        std::vector<std::vector<uint64_t>> seeds(n_frames);
        for (size_t j = 0; j < n_frames; j++) {
            rl_scene_image_size(scene.handle, &out[j].width, &out[j].height);
            out[j].primal.assign((size_t)3 * out[j].width * out[j].height, 0.0f);
            seeds[j].resize(rl_block_count(out[j].width, out[j].height));
            rl_generate_block_seeds(&sampler.rnd, out[j].width, out[j].height, seeds[j].data(), seeds[j].size());   // generate_img_blocks, in frame order
        }
        std::vector<std::string> errors(k);
        std::vector<rl_render_stats> stats(k);
        std::vector<std::thread> threads;
        for (size_t c = 0; c < k; c++)
            threads.emplace_back([&, c]() {
                rl_context* cx = c == 0 ? ctx : extra_ctx[c - 1];
                for (size_t j = c; j < n_frames; j += k) {
                    if (rl_render_path(cx, &p, seeds[j].data(), seeds[j].size(), out[j].primal.data(), 0, nullptr, &stats[c]) != RL_OK) { errors[c] = rl_last_error(); return; }   // (the error string is per thread)
                }
            });
        for (std::thread& t : threads) t.join();
        for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error("rl_render_path: " + e);
        last_stats = stats[(n_frames - 1) % k];
        return out;
    }

    // IntegratorType::compute + Integrator::compute: builds the BVH (untimed, first call), renders, returns the image
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        const int n = std::max(1, n_gpus);
        if (ctx_scene != &scene || ctx_gpus != n || (!ctx && !multi)) {
            release();
            if (n == 1) {
                int rc = rl_context_create(scene.handle, device, &ctx);
                if (rc != RL_OK) throw std::runtime_error(std::string("rl_context_create: ") + rl_last_error());
                apply_options(ctx);
            } else {
                int n_dev = 0;
                rl_device_count(&n_dev);
                std::vector<int> devs(n);
                for (int g = 0; g < n; g++) devs[g] = n_dev > 0 ? (device + g) % n_dev : device + g;
                int rc = rl_multi_create(scene.handle, devs.data(), n, &multi);
                if (rc != RL_OK) throw std::runtime_error(std::string("rl_multi_create: ") + rl_last_error());
            }
            ctx_scene = &scene; ctx_gpus = n;
        }
        BufferCollection img;
        rl_scene_image_size(scene.handle, &img.width, &img.height);
        img.primal.assign((size_t)3 * img.width * img.height, 0.0f);
        rl_path_params p = params_for(sampler, scene);
        std::vector<uint64_t> seeds(rl_block_count(img.width, img.height));
        rl_generate_block_seeds(&sampler.rnd, img.width, img.height, seeds.data(), seeds.size());   // generate_img_blocks
        if (n == 1) {
            int rc = rl_render_path(ctx, &p, seeds.data(), seeds.size(), img.primal.data(), 0, nullptr, &last_stats);
            if (rc != RL_OK) throw std::runtime_error(std::string("rl_render_path: ") + rl_last_error());
            return img;
        }
        p.shard_index = 0; p.shard_count = 1;      // (rl_multi deals the blocks itself)
        int rc = rl_multi_render_path(multi, &p, seeds.data(), seeds.size(), img.primal.data(), &last_stats);
        if (rc != RL_OK) throw std::runtime_error(std::string("rl_multi_render_path: ") + rl_last_error());
        return img;
    }
};

// struct IntegratorAO (src/integrators/ao.rs:4-7) / struct IntegratorDirect (src/integrators/direct.rs:5-8)
struct IntegratorMC {
    int device = 0;
    rl_stream_mode stream_mode = RL_STREAM_REFERENCE_ORDER;
    rl_render_stats last_stats{};
  protected:
    BufferCollection run(bool direct, rl_mc_params p, IndependentSampler& sampler, Scene& scene) {
        rl_context* ctx = nullptr;
        int rc = rl_context_create(scene.handle, device, &ctx);
        if (rc != RL_OK) throw std::runtime_error(std::string("rl_context_create: ") + rl_last_error());
        BufferCollection img;
        rl_scene_image_size(scene.handle, &img.width, &img.height);
        img.primal.assign((size_t)3 * img.width * img.height, 0.0f);
        p.spp = (uint32_t)scene.nb_samples;
        p.stream_mode = stream_mode;
        p.seed_variant = sampler.variant;
        p.shard_index = 0; p.shard_count = 1;
        std::vector<uint64_t> seeds(rl_block_count(img.width, img.height));
        rl_generate_block_seeds(&sampler.rnd, img.width, img.height, seeds.data(), seeds.size());
        rc = (direct ? rl_render_direct : rl_render_ao)(ctx, &p, seeds.data(), seeds.size(), img.primal.data(), 0, nullptr, &last_stats);
        rl_context_destroy(ctx);
        if (rc != RL_OK) throw std::runtime_error(std::string("render: ") + rl_last_error());
        return img;
    }
};
struct IntegratorAO : IntegratorMC {
    std::optional<float> max_distance = 1.0f;
    bool normal_correction = false;
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        rl_mc_params p{};
        p.has_max_distance = max_distance.has_value(); p.max_distance = max_distance.value_or(0.0f);
        p.normal_correction = normal_correction;
        return run(false, p, sampler, scene);
    }
};
struct IntegratorDirect : IntegratorMC {
    size_t nb_bsdf_samples = 1, nb_light_samples = 1;
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        rl_mc_params p{};
        p.nb_bsdf_samples = (uint32_t)nb_bsdf_samples; p.nb_light_samples = (uint32_t)nb_light_samples;
        return run(true, p, sampler, scene);
    }
};

// IntegratorAverage (src/integrators/avg.rs:5-131) and IntegratorEqualTime (src/integrators/equal_time.rs:4-66):
// host loops around any inner integrator with `compute(IndependentSampler&, Scene&)`.
template <class T, class = void> struct has_frames_in_flight : std::false_type {};
template <class T> struct has_frames_in_flight<T, std::void_t<decltype(std::declval<T&>().frames_in_flight)>> : std::true_type {};
template <class Inner>
struct IntegratorAverage {
    Inner& integrator;
    std::optional<size_t> time_out;   // seconds
    bool dump_all = true;
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        if (!dump_all && !time_out) throw std::runtime_error("Impossible to have infinite approach and not dumping all images");
        const std::string& out = scene.output_img_path;
        size_t dot = out.rfind('.');
        if (dot == std::string::npos) throw std::runtime_error("No file extension provided");
        const std::string base = out.substr(0, dot), ext = out.substr(dot + 1);
        FILE* csv = dump_all ? std::fopen((base + "_time.csv").c_str(), "w") : nullptr;
        if (dump_all && !csv) throw std::runtime_error("cannot write " + base + "_time.csv");
        BufferCollection bitmap;
        size_t iteration = 1;
        double elapsed = 0.0;
        // frames in flight (an inner integrator with frames_in_flight > 1): the passes are rendered a batch at a time and folded in pass order, each charged its
        // share of the batch's time; the passes of the last batch beyond the time-out are dropped (their seeds are drawn)
        std::vector<BufferCollection> pending;
        size_t next_pending = 0;
        double share = 0.0;
        for (;;) {
            auto t0 = std::chrono::steady_clock::now();
            BufferCollection nb;
            if constexpr (has_frames_in_flight<Inner>::value) {
                if (integrator.frames_in_flight > 1) {
                    if (next_pending == pending.size()) {
                        pending = integrator.compute_frames(sampler, scene, (size_t)integrator.frames_in_flight);
                        next_pending = 0;
                        share = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)pending.size();
                    }
                    nb = std::move(pending[next_pending++]);
                    elapsed += share;
                    t0 = std::chrono::steady_clock::now();
                } else nb = integrator.compute(sampler, scene);
            } else nb = integrator.compute(sampler, scene);
            if (iteration == 1) bitmap = nb;
            else { bitmap.scale((float)iteration); bitmap.accumulate_bitmap(nb); bitmap.scale(1.0f / (float)(iteration + 1)); }   // avg.rs:59-61
            elapsed += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dump_all) {
                bitmap.save("primal", base + "_" + std::to_string(iteration) + "." + ext);
                std::fprintf(csv, "%llu.%u,\n", (unsigned long long)elapsed, (unsigned)((elapsed - (double)(unsigned long long)elapsed) * 1000.0));
            }
            if (time_out && (size_t)elapsed >= *time_out) break;
            iteration++;
        }
        if (csv) std::fclose(csv);
        return bitmap;
    }
};
template <class Inner>
struct IntegratorEqualTime {
    Inner& integrator;
    double target_time_ms;
    size_t iterations = 0;
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        BufferCollection bitmap;
        size_t iteration = 1;
        double elapsed_ms = 0.0;
        std::vector<BufferCollection> pending;      // frames in flight: as in IntegratorAverage
        size_t next_pending = 0;
        double share_ms = 0.0;
        for (;;) {
            auto t0 = std::chrono::steady_clock::now();
            BufferCollection nb;
            if constexpr (has_frames_in_flight<Inner>::value) {
                if (integrator.frames_in_flight > 1) {
                    if (next_pending == pending.size()) {
                        pending = integrator.compute_frames(sampler, scene, (size_t)integrator.frames_in_flight);
                        next_pending = 0;
                        share_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)pending.size();
                    }
                    nb = std::move(pending[next_pending++]);
                    elapsed_ms += share_ms;
                    t0 = std::chrono::steady_clock::now();
                } else nb = integrator.compute(sampler, scene);
            } else nb = integrator.compute(sampler, scene);
            if (iteration == 1) bitmap = nb; else bitmap.accumulate_bitmap(nb);
            elapsed_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (elapsed_ms >= target_time_ms) break;
            iteration++;
        }
        bitmap.scale(1.0f / (float)iteration);
        iterations = iteration;
        return bitmap;
    }
};

}  // namespace rustlight
