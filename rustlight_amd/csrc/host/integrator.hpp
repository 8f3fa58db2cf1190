// integrator.hpp — C++ mirror of the reference's plugin surface for the `path` hot path, layered
// on the C-ABI (include/rustlight_amd.h).  Names, argument meaning and defaults follow rustlight:
//   trait Sampler / IndependentSampler      src/samplers/mod.rs:3-9, independent.rs:5-34
//   struct Scene                            src/scene.rs:16-30
//   struct BufferCollection ("primal")      src/integrators/mod.rs:48-216
//   trait Integrator::compute               src/integrators/mod.rs:219-233
//   struct IntegratorPathTracing            src/integrators/explicit/path.rs:14-20
//   IntegratorType::compute (BVH build is untimed, then "Elapsed Integrator")  mod.rs:274-338
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
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
        int rc = rl_scene_load_pbrt(path.c_str(), use_shading_normals ? 1 : 0, &h);
        if (rc != RL_OK) throw std::runtime_error(std::string("error on loading the scene: ") + rl_last_error());
        return new Scene(h);
    }
    void build_emitters() { if (rl_scene_build_emitters(handle) != RL_OK) throw std::runtime_error("build_emitters failed"); }
};

struct BufferCollection {   // only the "primal" buffer exists on this path
    uint32_t width = 0, height = 0;
    std::vector<float> primal;   // W*H*3, row-major, origin top-left
    void save(const std::string& /*name = "primal"*/, const std::string& path) const {
        if (rl_save_pfm(path.c_str(), primal.data(), width, height) != RL_OK) throw std::runtime_error("cannot write " + path);
    }
};

enum class IntegratorPathTracingStrategies { All = RL_STRATEGY_ALL, BSDF = RL_STRATEGY_BSDF, Emitter = RL_STRATEGY_EMITTER };

struct IntegratorPathTracing {
    std::optional<uint32_t> min_depth = 0, max_depth = std::nullopt, rr_depth = 0;
    IntegratorPathTracingStrategies strategy = IntegratorPathTracingStrategies::All;
    bool single_scattering = false;
    // MI355X-specific knobs (not in the reference)
    int device = 0;
    rl_stream_mode stream_mode = RL_STREAM_PER_SAMPLE;
    uint32_t shard_index = 0, shard_count = 1;
    rl_render_stats last_stats{};

    // IntegratorType::compute + Integrator::compute: builds the BVH (untimed), renders, returns the image
    BufferCollection compute(IndependentSampler& sampler, Scene& scene) {
        rl_context* ctx = nullptr;
        int rc = rl_context_create(scene.handle, device, &ctx);
        if (rc != RL_OK) throw std::runtime_error(std::string("rl_context_create: ") + rl_last_error());
        BufferCollection img;
        rl_scene_image_size(scene.handle, &img.width, &img.height);
        img.primal.assign((size_t)3 * img.width * img.height, 0.0f);
        rl_path_params p;
        rl_path_params_default(&p);
        p.spp = (uint32_t)scene.nb_samples;
        p.has_min_depth = min_depth.has_value(); p.min_depth = min_depth.value_or(0);
        p.has_max_depth = max_depth.has_value(); p.max_depth = max_depth.value_or(0);
        p.has_rr_depth = rr_depth.has_value(); p.rr_depth = rr_depth.value_or(0);
        p.strategy = (int)strategy;
        p.single_scattering = single_scattering;
        p.stream_mode = stream_mode;
        p.seed_variant = sampler.variant;
        p.shard_index = shard_index; p.shard_count = shard_count;
        std::vector<uint64_t> seeds(rl_block_count(img.width, img.height));
        rl_generate_block_seeds(&sampler.rnd, img.width, img.height, seeds.data(), seeds.size());   // generate_img_blocks
        rc = rl_render_path(ctx, &p, seeds.data(), seeds.size(), img.primal.data(), 0, nullptr, &last_stats);
        rl_context_destroy(ctx);
        if (rc != RL_OK) throw std::runtime_error(std::string("rl_render_path: ") + rl_last_error());
        return img;
    }
};

// struct IntegratorAO (src/integrators/ao.rs:4-7) / struct IntegratorDirect (src/integrators/direct.rs:5-8)
struct IntegratorMC {
    int device = 0;
    rl_stream_mode stream_mode = RL_STREAM_PER_SAMPLE;
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

}  // namespace rustlight
