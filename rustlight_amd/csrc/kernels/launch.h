// launch.h — host-side launchers of the kernel families that live in their own translation units (fused_lds.hip, fused_stream.hip,
// shade.hip, mc.hip), so that the families compile in parallel.  Launch errors are picked up by the caller's hipGetLastError().
#pragma once

namespace rl {

// IntegratorAO / IntegratorDirect parameters (ao.rs:4-7, direct.rs:5-8)
struct McConst {
    int has_max_distance; float max_distance; int normal_correction;
    unsigned nb_bsdf_samples, nb_light_samples;
};

// mat: the scene's one BSDF type, or -1 = run-time switch per vertex.  area_only: every emitter is a mesh area light and there is no light
// tree (the NEE code of the other emitter kinds is compiled out: same results, 84 -> 21 spilled VGPRs on the diffuse Cornell box)
void launch_fused_lds(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_fused_stream(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
// the same kernels built with RL_FAST_MATH (fused_*_fast.hip): rl_path_params.numerics = RL_NUMERICS_FAST
void launch_fused_lds_fast(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_fused_stream_fast(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
// the queue-fed form (exact build): the evaluation pass of reference-order streams beside the chain pass (fusedq_*.hip)
void launch_fusedq_lds(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_fusedq_stream(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
// k_stream_chain (chain.hip.h): first pass of reference-order streams — one lane per 16x16 block records the sampler state at the start of every sample
void launch_chain_lds(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_chain_stream(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_chain_lds_fast(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
void launch_chain_stream_fast(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc);
// k_stream_spec (spec.hip.h): the same first pass with every lane busy — windows of the stream walked speculatively per pixel, the chain threaded through them
void launch_spec_lds(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc);
void launch_spec_stream(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc);
void launch_trace_batch_fast(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const DeviceScene& ds, const StackConf& stc, unsigned n, const float* o, const float* d, float* t_out,
                             int* mesh_out, int* tri_out, int* steps_out);   // test hook: the tolerance build's BVH4 traversal, batched
void launch_shade_type(int type, bool medium, dim3 grid, dim3 block, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const Pool& pool);
void launch_shade_sorted(bool medium, unsigned chunks, dim3 grid, dim3 block, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const Pool& pool);
void launch_pixel_mc(int kind, bool lds_scene, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const McConst& mp);
void launch_mc_chain(int kind, bool lds_scene, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const McConst& mp);   // first pass of reference-order streams for ao / direct
void dump_stage_timers(bool lds_scene);   // dev-only (-DRL_STAGE_TIMERS)
void dump_stage_timers_stream();
void dump_chain_timers_lds();      // dev-only (-DRL_STAGE_TIMERS): cycle shares of k_stream_chain's stages
void dump_chain_timers_stream();

}  // namespace rl
