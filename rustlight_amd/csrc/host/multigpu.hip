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

}  // namespace

struct rl_multi {
    std::vector<rl_context*> ctxs;       // one per shard
    std::vector<int> device_of;          // shard -> HIP device
    std::vector<float*> fb;              // shard -> W*H*3 floats on its device
    std::vector<hipStream_t> stream;     // shard -> stream on its device
    std::vector<int> comm_devices;       // distinct devices, communicator rank order
    std::vector<int> leader;             // communicator rank -> the shard whose buffer carries that device's sum
    std::vector<ncclComm_t> comms;
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

extern "C" void rl_multi_destroy(rl_multi* m) {
    if (!m) return;
    for (ncclComm_t c : m->comms) if (c) ncclCommDestroy(c);
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
        ncclResult_t r = ncclCommInitAll(m->comms.data(), (int)m->comm_devices.size(), m->comm_devices.data());
        if (r != ncclSuccess) { rl_set_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); rc = RL_ERR_HIP; }
        ncclGetVersion(&m->rccl_version);
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

extern "C" int rl_multi_render_path(rl_multi* m, const rl_path_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb,
                                    rl_render_stats* stats) {
    if (!m || !params || !block_seeds || !out_rgb) return RL_ERR_INVALID_ARGUMENT;
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
    // the single exchange step: one sum-reduce onto GPU 0 (ring over xGMI), in place on the root
    MG_NCCL(ncclGroupStart());
    for (size_t r = 0; r < m->comms.size(); r++) {
        const int lead = m->leader[r];
        ncclResult_t e = ncclReduce(m->fb[lead], m->fb[lead], n_floats, ncclFloat32, ncclSum, 0, m->comms[r], m->stream[lead]);
        if (e != ncclSuccess) { ncclGroupEnd(); rl_set_error(std::string("ncclReduce: ") + ncclGetErrorString(e)); return RL_ERR_HIP; }
    }
    MG_NCCL(ncclGroupEnd());
    const int root = m->leader[0];
    MG_HIP(hipSetDevice(m->device_of[root]));
    MG_HIP(hipMemcpyAsync(out_rgb, m->fb[root], n_floats * sizeof(float), hipMemcpyDeviceToHost, m->stream[root]));
    for (size_t r = 0; r < m->comms.size(); r++) { MG_HIP(hipSetDevice(m->comm_devices[r])); MG_HIP(hipStreamSynchronize(m->stream[m->leader[r]])); }
    if (stats) {
        *stats = st[0];
        for (int g = 1; g < n; g++) {
            stats->camera_samples += st[g].camera_samples; stats->vertices += st[g].vertices; stats->extension_rays += st[g].extension_rays;
            stats->shadow_rays += st[g].shadow_rays; stats->rng_draws += st[g].rng_draws; stats->kernel_launches += st[g].kernel_launches;
            stats->render_ms = std::max(stats->render_ms, st[g].render_ms);
            stats->ms_other = std::max(stats->ms_other, st[g].ms_other);
        }
    }
    return RL_OK;
}
