// vmem_calib.hip — what a wave64 vector load costs the CU's address path (TA / TCP) as a function of the live lanes, the width and the address
// pattern: the bound analysis of the streaming path-tracing kernel hinges on whether a sparsely filled `global_load_dwordx4`
// costs as much as a full one.
// Every CU runs `waves` waves per SIMD (256-thread workgroups, occupancy pinned with dynamic LDS); each live lane issues `iters` x 4 independent
// loads of 16 / 8 / 4 bytes at pseudo-random 64-byte records of a table (2 MB: L2-resident; 64 MB: Infinity Cache), i.e. the node fetch of the
// traversal loop without the arithmetic.  Patterns: every lane its own record; all live lanes the same record; scalar loads (s_load_dwordx16) of one
// record per wave.  Output: one JSON line per case with ns and cycles (at the measured clock) per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O2 scratch/vmem_calib.hip -o scratch/bin/vmem_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Pattern { PER_LANE = 0, SAME_RECORD = 1, SCALAR = 2 };

template <int WIDTH, int PATTERN>
__global__ void __launch_bounds__(256) k_vmem(const float4* __restrict__ tab, unsigned rec_mask, unsigned live, unsigned iters, float* sink) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    float acc = 0.0f;
    if (PATTERN == SCALAR) {
        unsigned s = __builtin_amdgcn_readfirstlane(wave * 2654435761u + 12345u);
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 __attribute__((address_space(4)))* ctab = (const f4 __attribute__((address_space(4)))*)tab;
        for (unsigned it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s = s * 1664525u + 1013904223u;
                const f4 __attribute__((address_space(4)))* q = ctab + 4u * ((s >> 8) & rec_mask);
                const f4 a = q[0], b = q[1], c = q[2], d = q[3];
                acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
            }
        }
    } else if (lane < live) {
        unsigned s = (PATTERN == SAME_RECORD ? wave : wave * 64u + lane) * 2654435761u + 12345u;
        for (unsigned it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s = s * 1664525u + 1013904223u;
                const float4* q = tab + 4u * ((s >> 8) & rec_mask);
                if (WIDTH == 16) { const float4 a = q[0], b = q[1], c = q[2], d = q[3]; acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w); }
                else if (WIDTH == 8) { const float2* r = reinterpret_cast<const float2*>(q); const float2 a = r[0], b = r[2], c = r[4], d = r[6]; acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y); }
                else { const float* r = reinterpret_cast<const float*>(q); acc += r[0] + r[4] + r[8] + r[12]; }
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int WIDTH, int PATTERN>
static void run(const char* name, const float4* tab, size_t table_bytes, unsigned live, int waves, unsigned iters, int cus, float* sink, double clock_ghz) {
    const unsigned rec_mask = (unsigned)(table_bytes / 64) - 1u;
    const size_t lds = waves >= 8 ? 0 : (size_t)(160 * 1024 / waves) - 1024;       // pins workgroups per CU
    const dim3 grid((unsigned)(cus * waves)), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_vmem<WIDTH, PATTERN>), grid, block, lds, 0, tab, rec_mask, live, iters / 8, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_vmem<WIDTH, PATTERN>), grid, block, lds, 0, tab, rec_mask, live, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per CU: waves/SIMD x 4 SIMDs x iters x 4 records x 4 loads per record (the scalar form: one s_load_dwordx16, or four x4, per record)
    const double per_cu = (double)waves * 4.0 * iters * 4.0 * 4.0;
    const double ns = ms * 1e6 / per_cu;
    std::printf("{\"case\": \"%s\", \"width_bytes\": %d, \"table_MB\": %.0f, \"live_lanes\": %u, \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_wave_load_per_cu\": %.3f, "
                "\"cycles_per_wave_load_per_cu\": %.2f, \"ns_per_record_per_wave\": %.1f}\n",
                name, WIDTH, table_bytes / 1048576.0, live, waves, ms, ns, ns * clock_ghz, ms * 1e6 / (iters * 4.0));
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const unsigned iters = argc > 1 ? (unsigned)std::atoi(argv[1]) : 4000u;
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const double clock_ghz = 2.4;
    const size_t big = 64u << 20;
    float4* tab; float* sink;
    CHECK(hipMalloc((void**)&tab, big));
    CHECK(hipMemset(tab, 0, big));
    CHECK(hipMalloc((void**)&sink, 64));
    const size_t sizes[2] = {2u << 20, big};
    const int lives[5] = {64, 32, 16, 4, 1};
    for (int w : {6, 2}) {
        for (size_t sz : sizes) {
            for (int lv : lives) run<16, PER_LANE>("dwordx4, a record per lane", tab, sz, lv, w, iters, cus, sink, clock_ghz);
            for (int lv : {64, 16, 1}) run<8, PER_LANE>("dwordx2, a record per lane", tab, sz, lv, w, iters, cus, sink, clock_ghz);
            for (int lv : {64, 16, 1}) run<4, PER_LANE>("dword, a record per lane", tab, sz, lv, w, iters, cus, sink, clock_ghz);
            for (int lv : {64, 16}) run<16, SAME_RECORD>("dwordx4, one record for the wave", tab, sz, lv, w, iters, cus, sink, clock_ghz);
            run<16, SCALAR>("s_load of the 64-byte record", tab, sz, 64, w, iters, cus, sink, clock_ghz);
        }
    }
    return 0;
}
