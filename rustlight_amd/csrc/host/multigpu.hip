// multigpu.cpp — several MI355X of one node behind ONE call: pixel-tile shards + a single RCCL framebuffer reduce over xGMI
// (the reference's merge step is `accumulate_bitmap` over the blocks' bitmaps, src/integrators/mod.rs:445-448).
//
// One process, one device context + one host thread per GPU (rl_render_path is blocking), every GPU renders the blocks
// b % N == g into its own zeroed W x H x 3 f32 buffer in HBM; then ONE ncclReduce(sum, root = GPU 0) inside a
// ncclGroupStart / ncclGroupEnd pair — device to device over xGMI, no per-GPU download, no host adds — and one download
// from the root.  Sums with zeros are exact, so the image is the 1-GPU image bit for bit.
//
// RCCL needs distinct devices per communicator rank.  When the caller asks for more shards than there are devices (the
// CLI's `--gpus 3` on a one-GPU box: a plumbing mode used by the tests) the shards that share a device are first added on
// that device (k_add on its stream, still exact) and the communicator spans the distinct devices only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rustlight_amd.h"

void rl_set_error(const std::string& s);

namespace {

__global__ void k_add_framebuffer(float* dst, const float* src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

// the calling thread's current HIP device is the caller's business: every entry point puts it back
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) { saved = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

}  // namespace

struct rl_multi {
    std::vector<rl_context*> ctxs;       // one per shard
    std::vector<int> device_of;          // shard -> HIP device
    std::vector<float*> fb;              // shard -> W*H*3 floats on its device
    std::vector<hipStream_t> stream;     // shard -> stream on its device
    std::vector<int> comm_devices;       // distinct devices, communicator rank order
    std::vector<int> leader;             // communicator rank -> the shard whose buffer carries that device's sum
    std::vector<ncclComm_t> comms;
    float* fb_sum = nullptr;             // on the root device: receive buffer of the reduce (the root's own shard stays intact, so a failed
                                         // collective can still be merged on the host from the per-device sums)
    std::vector<float> host_tmp;         // host merge only
    bool host_merge = false;             // the communicator could not be set up, or a reduce failed / timed out: merge on the host (reported)
    std::string merge_note;
    std::vector<rl_render_stats> last_stats;   // per shard, of the last render
    double last_reduce_ms = 0.0;
    uint32_t width = 0, height = 0;
    int rccl_version = 0;
};

#define MG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { rl_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); (void)hipGetLastError(); return RL_ERR_HIP; } \
    } while (0)
#define MG_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) { rl_set_error(std::string(#expr) + ": " + ncclGetErrorString(r_)); return RL_ERR_HIP; } \
    } while (0)

// strings that go into rl_multi_describe's JSON (device names, RCCL / HIP error texts, environment text) are escaped
static std::string json_escape(const std::string& in) {
    std::string o;
    for (unsigned char ch : in) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += (char)ch; }
        else if (ch < 0x20) { char b[8]; std::snprintf(b, sizeof(b), "\\u%04x", ch); o += b; }
        else o += (char)ch;
    }
    return o;
}

extern "C" void rl_multi_destroy(rl_multi* m) {
    if (!m) return;
    DeviceGuard guard;
    for (ncclComm_t c : m->comms) if (c) ncclCommDestroy(c);
    if (m->fb_sum && !m->leader.empty()) { hipSetDevice(m->device_of[m->leader[0]]); hipFree(m->fb_sum); }
    for (size_t g = 0; g < m->ctxs.size(); g++) {
        if (g < m->device_of.size()) hipSetDevice(m->device_of[g]);
        if (g < m->fb.size() && m->fb[g]) hipFree(m->fb[g]);
        if (g < m->stream.size() && m->stream[g]) hipStreamDestroy(m->stream[g]);
        if (m->ctxs[g]) rl_context_destroy(m->ctxs[g]);
    }
    delete m;
}

extern "C" int rl_multi_create(const rl_scene* scene, const int* devices, int n, rl_multi** out) {
    if (!scene || !out || n <= 0 || n > 1024) return RL_ERR_INVALID_ARGUMENT;
    int n_dev = 0;
    if (rl_device_count(&n_dev) != RL_OK || n_dev <= 0) { rl_set_error("no HIP device available; the MI355X path has no CPU fallback"); return RL_ERR_NO_DEVICE; }
    DeviceGuard guard;
    rl_multi* m = new rl_multi();
    int rc = RL_OK;
    rc = rl_scene_image_size(scene, &m->width, &m->height);
    const size_t n_floats = (size_t)3 * m->width * m->height;
    for (int g = 0; g < n && rc == RL_OK; g++) {
        const int dev = devices ? devices[g] : g % n_dev;     // default: round-robin over the visible devices
        if (dev < 0 || dev >= n_dev) { rl_set_error("device ordinal out of range"); rc = RL_ERR_NO_DEVICE; break; }
        rl_context* c = nullptr;
        if ((rc = rl_context_create(scene, dev, &c)) != RL_OK) break;
        m->ctxs.push_back(c);
        m->device_of.push_back(dev);
        m->fb.push_back(nullptr);
        m->stream.push_back(nullptr);
        if (hipSetDevice(dev) != hipSuccess || hipMalloc((void**)&m->fb.back(), n_floats * sizeof(float)) != hipSuccess ||
            hipStreamCreateWithFlags(&m->stream.back(), hipStreamNonBlocking) != hipSuccess) {
            rl_set_error("rl_multi_create: device buffer / stream allocation failed"); (void)hipGetLastError(); rc = RL_ERR_HIP; break;
        }
        if (std::find(m->comm_devices.begin(), m->comm_devices.end(), dev) == m->comm_devices.end()) { m->comm_devices.push_back(dev); m->leader.push_back(g); }
    }
    if (rc == RL_OK) {
        // ONE communicator clique over the distinct devices of this process (ncclCommInitAll = the single-process form of ncclCommInitRank)
        m->comms.assign(m->comm_devices.size(), nullptr);
        ncclResult_t r = getenv("RL_MULTI_FORCE_HOST_MERGE") ? ncclInternalError : ncclCommInitAll(m->comms.data(), (int)m->comm_devices.size(), m->comm_devices.data());
        ncclGetVersion(&m->rccl_version);
        if (r != ncclSuccess) {
            // The shards themselves are unaffected by a communicator that cannot be built (peer access, IPC mode, driver): keep rendering and merge
            // the per-device sums on the host instead — loudly (stderr, rl_multi_describe), unless RL_MULTI_NO_FALLBACK asks for the error.
            const std::string why = getenv("RL_MULTI_FORCE_HOST_MERGE") ? std::string("RL_MULTI_FORCE_HOST_MERGE") : std::string("ncclCommInitAll: ") + ncclGetErrorString(r);
            if (getenv("RL_MULTI_NO_FALLBACK")) { rl_set_error(why); rc = RL_ERR_HIP; }
            else {
                std::fprintf(stderr, "rustlight_amd: %s — the framebuffers of the %zu devices are merged on the host instead of with one RCCL reduce\n", why.c_str(), m->comm_devices.size());
                for (ncclComm_t& c : m->comms) c = nullptr;
                m->comms.clear();
                m->host_merge = true; m->merge_note = why;
            }
        }
        if (rc == RL_OK && hipSetDevice(m->device_of[m->leader[0]]) == hipSuccess && hipMalloc((void**)&m->fb_sum, n_floats * sizeof(float)) != hipSuccess) {
            rl_set_error("rl_multi_create: reduce buffer allocation failed"); (void)hipGetLastError(); rc = RL_ERR_HIP;
        }
    }
    if (rc != RL_OK) { rl_multi_destroy(m); return rc; }
    *out = m;
    return RL_OK;
}

extern "C" int rl_multi_info(const rl_multi* m, int* n_shards, int* n_comm_ranks, int* rccl_version) {
    if (!m) return RL_ERR_INVALID_ARGUMENT;
    if (n_shards) *n_shards = (int)m->ctxs.size();
    if (n_comm_ranks) *n_comm_ranks = (int)m->comms.size();
    if (rccl_version) *rccl_version = m->rccl_version;
    return RL_OK;
}

extern "C" int rl_multi_describe(const rl_multi* m, char* buf, size_t capacity) {
    if (!m || !buf || capacity == 0) return RL_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    std::string s = "{\"shards\": " + std::to_string(m->ctxs.size()) + ", \"rccl_version\": " + std::to_string(m->rccl_version) + ", \"comm_ranks\": " + std::to_string(m->comms.size()) +
                    ", \"merge\": \"" + (m->host_merge ? "host sum (" + json_escape(m->merge_note) + ")" : std::string("ncclReduce(sum) onto device ") + std::to_string(m->comm_devices.empty() ? -1 : m->comm_devices[0])) + "\", \"devices\": [";
    for (size_t r = 0; r < m->comm_devices.size(); r++) {
        hipDeviceProp_t prop{};
        const int d = m->comm_devices[r];
        (void)hipGetDeviceProperties(&prop, d);
        int shards_here = 0;
        for (int dv : m->device_of) shards_here += dv == d;
        s += std::string(r ? ", " : "") + "{\"device\": " + std::to_string(d) + ", \"name\": \"" + json_escape(prop.name) + "\", \"arch\": \"" + json_escape(prop.gcnArchName) + "\", \"cus\": " + std::to_string(prop.multiProcessorCount) + ", \"shards\": " + std::to_string(shards_here) + ", \"peer_access\": [";
        for (size_t q = 0; q < m->comm_devices.size(); q++) {      // hipDeviceCanAccessPeer: what the xGMI ring of the reduce rides on
            int can = d == m->comm_devices[q] ? 1 : 0;
            if (d != m->comm_devices[q] && hipDeviceCanAccessPeer(&can, d, m->comm_devices[q]) != hipSuccess) { can = -1; (void)hipGetLastError(); }
            s += std::string(q ? ", " : "") + std::to_string(can);
        }
        s += "]}";
    }
    s += "], \"last_render\": {\"reduce_ms\": " + std::to_string(m->last_reduce_ms) + ", \"kernel_ms\": [";
    for (size_t g = 0; g < m->last_stats.size(); g++) s += std::string(g ? ", " : "") + std::to_string(m->last_stats[g].ms_other + m->last_stats[g].ms_prepass + m->last_stats[g].ms_raygen + m->last_stats[g].ms_extend + m->last_stats[g].ms_shade + m->last_stats[g].ms_shadow);
    s += "]}}";
    if (s.size() + 1 > capacity) { rl_set_error("rl_multi_describe: buffer too small (" + std::to_string(s.size() + 1) + " bytes needed)"); return RL_ERR_INVALID_ARGUMENT; }
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return RL_OK;
}

extern "C" int rl_multi_shard_stats(const rl_multi* m, int shard, int* device, rl_render_stats* stats) {
    if (!m || shard < 0 || shard >= (int)m->ctxs.size()) return RL_ERR_INVALID_ARGUMENT;
    if (device) *device = m->device_of[shard];
    if (stats) { if ((size_t)shard < m->last_stats.size()) *stats = m->last_stats[shard]; else std::memset(stats, 0, sizeof(*stats)); }
    return RL_OK;
}

extern "C" int rl_multi_render_path(rl_multi* m, const rl_path_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb,
                                    rl_render_stats* stats) {
    if (!m || !params || !block_seeds || !out_rgb) return RL_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    const int n = (int)m->ctxs.size();
    const size_t n_floats = (size_t)3 * m->width * m->height;
    std::vector<rl_render_stats> st(n);
    std::vector<int> rcs(n, RL_OK);
    std::vector<std::string> errs(n);
    {   // shard renders: one host thread per GPU (the call blocks until its kernels are done)
        std::vector<std::thread> th;
        for (int g = 0; g < n; g++)
            th.emplace_back([&, g]() {
                try {
                    (void)hipSetDevice(m->device_of[g]);
                    rl_path_params q = *params;
                    q.shard_index = (uint32_t)g; q.shard_count = (uint32_t)n;
                    rcs[g] = rl_render_path(m->ctxs[g], &q, block_seeds, n_blocks, m->fb[g], 1, m->stream[g], &st[g]);
                    if (rcs[g] != RL_OK) errs[g] = rl_last_error();      // rl_last_error is thread-local: fetch it on this thread
                } catch (const std::exception& e) { rcs[g] = RL_ERR_HIP; errs[g] = e.what(); }
            });
        for (std::thread& t : th) t.join();
    }
    for (int g = 0; g < n; g++) if (rcs[g] != RL_OK) { rl_set_error("rl_render_path (shard " + std::to_string(g) + "): " + errs[g]); return rcs[g]; }
    // shards that share a device: add them into that device's leader buffer on the device (exact: disjoint blocks, zeros elsewhere)
    for (int g = 0; g < n; g++) {
        const int r = (int)(std::find(m->comm_devices.begin(), m->comm_devices.end(), m->device_of[g]) - m->comm_devices.begin());
        const int lead = m->leader[r];
        if (lead == g) continue;
        MG_HIP(hipSetDevice(m->device_of[g]));
        hipLaunchKernelGGL(k_add_framebuffer, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, m->stream[lead], m->fb[lead], m->fb[g], n_floats);
        MG_HIP(hipGetLastError());
    }
    m->last_stats = st;
    const int root = m->leader[0];
    auto t_merge = std::chrono::steady_clock::now();
    // merge on the host: download every device's sum and add them in device order (sums with zeros are exact: same bits as the reduce)
    auto host_merge = [&]() -> int {
        std::memset(out_rgb, 0, n_floats * sizeof(float));
        m->host_tmp.resize(n_floats);
        for (size_t r = 0; r < m->comm_devices.size(); r++) {
            const int lead = m->leader[r];
            MG_HIP(hipSetDevice(m->device_of[lead]));
            MG_HIP(hipMemcpyAsync(m->host_tmp.data(), m->fb[lead], n_floats * sizeof(float), hipMemcpyDeviceToHost, m->stream[lead]));
            MG_HIP(hipStreamSynchronize(m->stream[lead]));
            for (size_t i = 0; i < n_floats; i++) out_rgb[i] = out_rgb[i] + m->host_tmp[i];
        }
        return RL_OK;
    };
    bool merged = false;
    if (!m->host_merge) {
        // the single exchange step: one sum-reduce onto the root GPU (ring over xGMI); out of place on the root, so that its own shard survives a
        // collective that fails half-way and the host merge below still has every device's sum
        std::string nccl_err;
        ncclResult_t e = ncclGroupStart();
        if (e != ncclSuccess) nccl_err = std::string("ncclGroupStart: ") + ncclGetErrorString(e);
        for (size_t r = 0; r < m->comms.size() && nccl_err.empty(); r++) {
            const int lead = m->leader[r];
            e = ncclReduce(m->fb[lead], r == 0 ? m->fb_sum : m->fb[lead], n_floats, ncclFloat32, ncclSum, 0, m->comms[r], m->stream[lead]);
            if (e != ncclSuccess) nccl_err = std::string("ncclReduce: ") + ncclGetErrorString(e);
        }
        e = ncclGroupEnd();
        if (e != ncclSuccess && nccl_err.empty()) nccl_err = std::string("ncclGroupEnd: ") + ncclGetErrorString(e);
        if (nccl_err.empty()) {
            // wait for the reduce under a deadline: a collective that never completes (first contact of the devices over xGMI) must not hang the render
            const double limit_s = getenv("RL_MULTI_REDUCE_TIMEOUT_S") ? atof(getenv("RL_MULTI_REDUCE_TIMEOUT_S")) : 120.0;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(limit_s);
            for (size_t r = 0; r < m->comms.size() && nccl_err.empty(); r++) {
                (void)hipSetDevice(m->comm_devices[r]);
                for (;;) {
                    const hipError_t q = hipStreamQuery(m->stream[m->leader[r]]);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) { nccl_err = std::string("reduce stream: ") + hipGetErrorString(q); (void)hipGetLastError(); break; }
                    ncclResult_t ae = ncclSuccess;
                    if (ncclCommGetAsyncError(m->comms[r], &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) { nccl_err = std::string("RCCL async error: ") + ncclGetErrorString(ae); break; }
                    if (std::chrono::steady_clock::now() > deadline) { nccl_err = "ncclReduce did not complete within " + std::to_string((int)limit_s) + " s"; break; }
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
        }
        if (nccl_err.empty()) {
            MG_HIP(hipSetDevice(m->device_of[root]));
            MG_HIP(hipMemcpyAsync(out_rgb, m->fb_sum, n_floats * sizeof(float), hipMemcpyDeviceToHost, m->stream[root]));
            MG_HIP(hipStreamSynchronize(m->stream[root]));
            merged = true;
        } else {
            if (getenv("RL_MULTI_NO_FALLBACK")) { rl_set_error(nccl_err); return RL_ERR_HIP; }
            std::fprintf(stderr, "rustlight_amd: %s — aborting the communicator; this and later renders merge the framebuffers on the host\n", nccl_err.c_str());
            for (ncclComm_t& c : m->comms) if (c) { ncclCommAbort(c); c = nullptr; }
            m->comms.clear();
            m->host_merge = true; m->merge_note = nccl_err;
        }
    }
    if (!merged) { int r = host_merge(); if (r != RL_OK) return r; }
    m->last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_merge).count();
    if (stats) {
        *stats = st[0];
        for (int g = 1; g < n; g++) {
            stats->camera_samples += st[g].camera_samples; stats->vertices += st[g].vertices; stats->extension_rays += st[g].extension_rays;
            stats->shadow_rays += st[g].shadow_rays; stats->rng_draws += st[g].rng_draws; stats->kernel_launches += st[g].kernel_launches;
            stats->render_ms = std::max(stats->render_ms, st[g].render_ms);
            stats->ms_other = std::max(stats->ms_other, st[g].ms_other);
            stats->ms_prepass = std::max(stats->ms_prepass, st[g].ms_prepass);
            stats->ms_eval_span = std::max(stats->ms_eval_span, st[g].ms_eval_span);
            stats->chunks = std::max(stats->chunks, st[g].chunks);                 // (the shard that was cut into the most chunks; 1 = every shard overlapped)
            stats->overlapped = std::min(stats->overlapped, st[g].overlapped);
        }
    }
    return RL_OK;
}
