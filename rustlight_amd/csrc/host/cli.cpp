// cli.cpp — `rustlight-amd`: the reference CLI's flags for the `path` subcommand (examples/cli.rs:
// global flags 106-145, `path` 162-169, medium 355-399, sampler 876-896, run/save 898-923).
//   rustlight-amd <scene.pbrt|scene.xml> -n SPP -o out.pfm [-r independent:SEED] [-m s[:a[:g]]] [-s SCALE] [-t N]
//                 [--device D] [--gpus N] [--frames-in-flight K] [--stream-mode reference|per-sample] [--numerics exact|fast] [--option name=value ...]
//                 path [-m MAX|inf] [-n MIN] [-r RR|inf] [-x] [-s all|bsdf|emitter]
//               | ao [-d DIST|inf] [-n]            (examples/cli.rs:149-154)
//               | direct [-b NB_BSDF] [-l NB_LIGHT] (examples/cli.rs:155-160)
// Note `-n` / `-m` / `-r` / `-s` mean spp / medium / sampler / scale before the subcommand and
// min-depth / max-depth / rr-depth / strategy after it, exactly as in the reference.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <sstream>

#include "integrator.hpp"

using namespace rustlight;

static std::optional<uint32_t> match_infinity(const std::string& s) {   // cli.rs:31-39
    if (s == "inf") return std::nullopt;
    char* e = nullptr;
    unsigned long v = std::strtoul(s.c_str(), &e, 10);
    if (!e || *e) { std::fprintf(stderr, "wrong input for inf type parameter\n"); std::exit(2); }
    return (uint32_t)v;
}

int main(int argc, char** argv) {
    std::string scene_path, output, medium = "0.0", rng = "independent", strategy = "all";
    std::string max_depth = "inf", min_depth = "0", rr_depth = "0";
    size_t nbsamples = 1;
    float scale_image = 1.0f;
    int device = 0, gpus = 1;
    bool single_scattering = false, have_cmd = false, use_ats = false, shading_normals = true;
    int light_override = RL_EMISSION_COLOR;      // -x hvs-light | texture-light
    std::string cmd, ao_distance = "1.0", average, equal_time;
    bool ao_normal_correction = false;
    size_t nb_bsdf = 1, nb_light = 1;
    rl_stream_mode mode = RL_STREAM_REFERENCE_ORDER;   // like rustlight; `--stream-mode per-sample` trades the seed-for-seed image for throughput
    uint32_t numerics = RL_NUMERICS_EXACT;
    int frames_in_flight = 1;
    std::vector<std::pair<std::string, std::string>> options;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (!have_cmd) {
            if (a == "path" || a == "ao" || a == "direct") { have_cmd = true; cmd = a; }
            else if (a == "-n" || a == "--nbsamples") nbsamples = std::strtoull(val().c_str(), nullptr, 10);
            else if (a == "-o" || a == "--output") output = val();
            else if (a == "-r" || a == "--random-number-generator") rng = val();
            else if (a == "-m" || a == "--medium") medium = val();
            else if (a == "-s" || a == "--scale-image") scale_image = std::strtof(val().c_str(), nullptr);
            else if (a == "-t" || a == "--threads") (void)val();   // host threads are irrelevant on the GPU path
            else if (a == "--device") device = std::atoi(val().c_str());
            else if (a == "--gpus") gpus = std::atoi(val().c_str());
            else if (a == "--frames-in-flight") frames_in_flight = std::max(1, std::atoi(val().c_str()));   // -a / -e: that many independent passes on the GPU at once (same images, more of the chip busy)
            else if (a == "--stream-mode") mode = val() == "reference" ? RL_STREAM_REFERENCE_ORDER : RL_STREAM_PER_SAMPLE;
            else if (a == "--numerics") numerics = val() == "fast" ? RL_NUMERICS_FAST : RL_NUMERICS_EXACT;
            else if (a == "--option") {   // an execution option of the device context(s): name=value (rl_context_set_option; none changes an image)
                const std::string o = val();
                const size_t eq = o.find('=');
                options.emplace_back(o.substr(0, eq), eq == std::string::npos ? std::string("1") : o.substr(eq + 1));
            }
            else if (a == "-a" || a == "--average") average = val();
            else if (a == "-e" || a == "--equal-time") equal_time = val();
            else if (a == "-x" || a == "--xtra-options") {   // ExtraOptions (cli.rs:41-50): ats | no-shading are honoured
                const std::string o = val();
                if (o == "ats") use_ats = true;
                else if (o == "no-shading") shading_normals = false;
                else if (o == "hvs-light") light_override = RL_EMISSION_HSV;               // ExtraOptions::HVSLight (cli.rs:327, 410-429)
                else if (o == "texture-light") light_override = RL_EMISSION_TEXTURE;       // ExtraOptions::TextureLight
                else { std::fprintf(stderr, "extra option %s is not supported by this drop-in (ats, no-shading, hvs-light, texture-light)\n", o.c_str()); return 2; }
            }
            else if (a == "-l" || a == "--log") (void)val();   // log file: nothing is logged on this path
            else if (a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
            else if (scene_path.empty()) scene_path = a;
            else { std::fprintf(stderr, "only the `path`, `ao` and `direct` subcommands are provided (got %s)\n", a.c_str()); return 2; }
        } else if (cmd == "ao") {
            if (a == "-d" || a == "--distance") ao_distance = val();
            else if (a == "-n" || a == "--normal-correction") ao_normal_correction = true;
            else { std::fprintf(stderr, "unknown ao option %s\n", a.c_str()); return 2; }
        } else if (cmd == "direct") {
            if (a == "-b" || a == "--nb-bsdf-samples") nb_bsdf = std::strtoull(val().c_str(), nullptr, 10);
            else if (a == "-l" || a == "--nb-light-samples") nb_light = std::strtoull(val().c_str(), nullptr, 10);
            else { std::fprintf(stderr, "unknown direct option %s\n", a.c_str()); return 2; }
        } else {
            if (a == "-m" || a == "--max-depth") max_depth = val();
            else if (a == "-n" || a == "--min-depth") min_depth = val();
            else if (a == "-r" || a == "--rr-depth") rr_depth = val();
            else if (a == "-x" || a == "--single-scattering") single_scattering = true;
            else if (a == "-s" || a == "--strategy") strategy = val();
            else { std::fprintf(stderr, "unknown path option %s\n", a.c_str()); return 2; }
        }
    }
    if (scene_path.empty() || output.empty() || !have_cmd) {
        std::fprintf(stderr, "usage: rustlight-amd <scene.pbrt|scene.xml> -n SPP -o out.pfm [-r independent:SEED] [-m s[:a[:g]]] path [-m max] [-n min] [-r rr] [-x] [-s all|bsdf|emitter]\n");
        return 2;
    }
    try {
        std::unique_ptr<Scene> scene(Scene::load(scene_path, shading_normals));
        scene->nb_samples = nbsamples;
        scene->output_img_path = output;
        {   // medium: sigma_s[:sigma_a[:g]] (cli.rs:355-399)
            std::vector<std::string> parts;
            std::stringstream ss(medium);
            for (std::string tok; std::getline(ss, tok, ':');) parts.push_back(tok);
            float sigma_s = parts.size() > 0 ? std::strtof(parts[0].c_str(), nullptr) : 0.0f;
            float sigma_a = parts.size() > 1 ? std::strtof(parts[1].c_str(), nullptr) : 0.0f;
            if (parts.size() > 3) { std::fprintf(stderr, "invalid medium_density\n"); return 2; }
            if (sigma_a + sigma_s != 0.0f) {
                float sa[3] = {sigma_a, sigma_a, sigma_a}, s3[3] = {sigma_s, sigma_s, sigma_s};
                float g = parts.size() > 2 ? std::strtof(parts[2].c_str(), nullptr) : 0.0f;
                if (rl_scene_set_medium(scene->handle, sa, s3, parts.size() > 2 ? RL_PHASE_HG : RL_PHASE_ISOTROPIC, g) != RL_OK) { std::fprintf(stderr, "invalid medium_density\n"); return 2; }
            }
        }
        if (scale_image != 1.0f && rl_scene_scale_image(scene->handle, scale_image) != RL_OK) { std::fprintf(stderr, "invalid image scale: %s\n", rl_last_error()); return 2; }
        if (light_override != RL_EMISSION_COLOR) {   // "Overide light is needed" (cli.rs:410-429): every light mesh becomes HSV { scale } / Texture { scale, butterfly.jpg }
            int bitmap = -1;
            if (light_override == RL_EMISSION_TEXTURE) {
                uint32_t bw = 0, bh = 0;
                if (rl_load_image("butterfly.jpg", &bw, &bh, nullptr, 0) != RL_OK) { std::fprintf(stderr, "texture-light: cannot read butterfly.jpg from the working directory (Bitmap::read(\"butterfly.jpg\"), cli.rs:423): %s\n", rl_last_error()); return 1; }
                std::vector<float> px((size_t)3 * bw * bh);
                if (rl_load_image("butterfly.jpg", &bw, &bh, px.data(), px.size()) != RL_OK) { std::fprintf(stderr, "texture-light: %s\n", rl_last_error()); return 1; }
                bitmap = rl_scene_add_bitmap(scene->handle, bw, bh, px.data());
                if (bitmap < 0) { std::fprintf(stderr, "texture-light: %s\n", rl_last_error()); return 1; }      // (a negative return is an error code, not a bitmap id)
            }
            if (rl_scene_override_light_emission(scene->handle, light_override, bitmap) != RL_OK) { std::fprintf(stderr, "-x %s: %s\n", light_override == RL_EMISSION_HSV ? "hvs-light" : "texture-light", rl_last_error()); return 1; }
        }
        scene->build_emitters(use_ats);      // scene.build_emitters(use_ats) (cli.rs:432)
        IntegratorPathTracing integrator;
        integrator.min_depth = match_infinity(min_depth);
        integrator.max_depth = match_infinity(max_depth);
        integrator.rr_depth = match_infinity(rr_depth);
        if (strategy == "all") integrator.strategy = IntegratorPathTracingStrategies::All;
        else if (strategy == "bsdf") integrator.strategy = IntegratorPathTracingStrategies::BSDF;
        else if (strategy == "emitter") integrator.strategy = IntegratorPathTracingStrategies::Emitter;
        else { std::fprintf(stderr, "invalid strategy: %s\n", strategy.c_str()); return 2; }
        integrator.single_scattering = single_scattering;
        integrator.device = device;
        integrator.n_gpus = gpus;      // --gpus N: blocks dealt round-robin over N devices, one RCCL reduce of the framebuffers over xGMI
        integrator.stream_mode = mode;
        integrator.numerics = numerics;
        integrator.frames_in_flight = frames_in_flight;
        integrator.options = options;
        uint64_t seed;
        if (rng == "independent") seed = std::random_device{}();   // IndependentSampler::default(): OS entropy
        else if (rng.rfind("independent:", 0) == 0) seed = std::strtoull(rng.c_str() + 12, nullptr, 10);
        else { std::fprintf(stderr, "Wrong sampler type provided %s (only independent[:seed])\n", rng.c_str()); return 2; }
        IndependentSampler sampler(seed);
        BufferCollection img;
        double elapsed_ms = 0.0;
        if (cmd == "ao") {
            IntegratorAO ao;
            ao.device = device; ao.stream_mode = mode; ao.normal_correction = ao_normal_correction;
            if (ao_distance == "inf") ao.max_distance = std::nullopt; else ao.max_distance = std::strtof(ao_distance.c_str(), nullptr);
            img = ao.compute(sampler, *scene); elapsed_ms = ao.last_stats.render_ms;
        } else if (cmd == "direct") {
            IntegratorDirect di;
            di.device = device; di.stream_mode = mode; di.nb_bsdf_samples = nb_bsdf; di.nb_light_samples = nb_light;
            img = di.compute(sampler, *scene); elapsed_ms = di.last_stats.render_ms;
        } else if (!equal_time.empty()) {        // cli.rs:898-907
            IntegratorEqualTime<IntegratorPathTracing> eq{integrator, std::strtod(equal_time.c_str(), nullptr) * 1000.0};
            img = eq.compute(sampler, *scene); elapsed_ms = integrator.last_stats.render_ms;
            std::fprintf(stderr, "INFO Number iter: %zu\nINFO Number spp: %zu\n", eq.iterations, eq.iterations * scene->nb_samples);
        } else if (!average.empty()) {           // cli.rs:908-917
            IntegratorAverage<IntegratorPathTracing> av{integrator, std::nullopt, true};
            if (average != "inf") av.time_out = (size_t)std::strtoull(average.c_str(), nullptr, 10);
            img = av.compute(sampler, *scene); elapsed_ms = integrator.last_stats.render_ms;
        } else { img = integrator.compute(sampler, *scene); elapsed_ms = integrator.last_stats.render_ms; }
        std::fprintf(stderr, "INFO Elapsed Integrator: %.0f ms\n", elapsed_ms);
        if (integrator.multi) {     // --gpus N: where the shards ran, how the framebuffers were merged, kernel ms per device
            std::vector<char> buf(1 << 16);
            if (rl_multi_describe(integrator.multi, buf.data(), buf.size()) == RL_OK) std::fprintf(stderr, "INFO Multi-GPU: %s\n", buf.data());
            for (int g = 0; g < gpus; g++) {
                int dev = -1; rl_render_stats st{};
                if (rl_multi_shard_stats(integrator.multi, g, &dev, &st) == RL_OK)
                    std::fprintf(stderr, "INFO shard %d on device %d: kernel %.2f ms (chain pass %.2f ms), %llu camera samples\n", g, dev, st.ms_other + st.ms_prepass, st.ms_prepass, (unsigned long long)st.camera_samples);
            }
        }
        std::fprintf(stderr, "INFO Save final image: %s\n", output.c_str());
        img.save("primal", output);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ERROR %s\n", e.what());
        return 1;
    }
    return 0;
}
