// valu_calib.hip — issue-cost calibration for the bound analysis of k_path_fused.
// For each instruction class: a loop of 64 inline-asm instructions (8 independent accumulators x 8, or one dependent
// chain), run by 1 / 2 / 4 / 8 waves per SIMD on every CU (256-thread workgroups = one wave per SIMD each; the
// workgroups-per-CU count is pinned with dynamic LDS), timed with s_memtime inside the kernel and HIP events outside.
// Output: one JSON object per (class, waves/SIMD) with cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 scratch/valu_calib.hip -o scratch/bin/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Op {
    FMA_INDEP, FMA_DEP, MUL_INDEP, ADD_INDEP, MULADD_DEP, CNDMASK, MAX3, PK_FMA, PK_MUL, RCP, SQRT, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32,
    FMA_F64, MUL_F64, ADD_F64, CMP_CNDMASK, VALU_SALU_MIX, DS_READ_B32, DS_READ2_B32, DS_READ_B64, DS_WRITE_B64, DIV_IEEE, SQRT_IEEE, NODE_STEP_LIKE, N_OPS
};
static const char* kNames[N_OPS] = {
    "v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_mul_f32 x8 independent", "v_add_f32 x8 independent", "v_mul_f32 -> v_add_f32 dependent chain",
    "v_cndmask_b32 x8", "v_max3_f32 x8", "v_pk_fma_f32 x8", "v_pk_mul_f32 x8", "v_rcp_f32 x8", "v_sqrt_f32 x8", "v_mul_lo_u32 x8", "v_mul_hi_u32 x8",
    "v_mad_u64_u32 x8", "v_fma_f64 x8", "v_mul_f64 x8", "v_add_f64 x8", "v_cmp_lt_f32 + v_cndmask_b32 pairs", "v_fma_f32 x8 with one s_add_u32 per VALU",
    "ds_read_b32 x8", "ds_read2_b32 x8", "ds_read_b64 x8", "ds_write_b64 x8", "IEEE f32 divide (compiled a / b, dependent)", "IEEE f32 sqrt (compiled, dependent)",
    "slab-test-like block (6 sub, 6 mul, 2 max3, 2 min3 style, 4 waves of ILP)"};
// instructions the loop body issues per iteration (for the compiled ops: "operations", the instruction count is read from the ISA separately)
static const int kPerIter[N_OPS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 128, 64, 64, 64, 64, 8, 8, 64};

#define REP8(X) X X X X X X X X
#define A8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int OP>
__global__ void __launch_bounds__(256) k_calib(unsigned iters, unsigned long long* cycles, float* sink) {
    extern __shared__ float lds[];
    const unsigned tid = threadIdx.x;
    float a0 = tid * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float x = 0.999f + tid * 1e-9f, y = 1e-3f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const double dx = 0.999, dy = 1e-3;
    unsigned u0 = tid + 1, u1 = tid + 2, u2 = tid + 3, u3 = tid + 4, u4 = tid + 5, u5 = tid + 6, u6 = tid + 7, u7 = tid + 8;
    unsigned long long q0 = tid, q1 = tid + 1, q2 = tid + 2, q3 = tid + 3, q4 = tid + 4, q5 = tid + 5, q6 = tid + 6, q7 = tid + 7;
    const unsigned ux = 2654435761u;
    unsigned s_cnt = 0;
    if (OP >= DS_READ_B32 && OP <= DS_WRITE_B64) { for (unsigned i = tid; i < 2048; i += 256) lds[i] = (float)i; __syncthreads(); }
    const unsigned laddr = tid * 8u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned it = 0; it < iters; it++) {
        if (OP == FMA_INDEP) {
#define I(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
#undef I
        } else if (OP == FMA_DEP) {
            REP8(asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(x), "v"(y));)
        } else if (OP == MUL_INDEP) {
#define I(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
#undef I
        } else if (OP == ADD_INDEP) {
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
#undef I
        } else if (OP == MULADD_DEP) {
            REP8(asm volatile("v_mul_f32 %0, %0, %1\nv_add_f32 %0, %0, %2\nv_mul_f32 %0, %0, %1\nv_add_f32 %0, %0, %2\n"
                              "v_mul_f32 %0, %0, %1\nv_add_f32 %0, %0, %2\nv_mul_f32 %0, %0, %1\nv_add_f32 %0, %0, %2\n" : "+v"(a0) : "v"(x), "v"(y));)
        } else if (OP == CNDMASK) {
#define I(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");)
#undef I
        } else if (OP == MAX3) {
#define I(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
#undef I
        } else if (OP == PK_FMA) {
#define I(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
            REP8(asm volatile(A8(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx), "v"(dy));)
#undef I
        } else if (OP == PK_MUL) {
#define I(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx));)
#undef I
        } else if (OP == RCP) {
#define I(k) "v_rcp_f32 %" #k ", %" #k "\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
#undef I
        } else if (OP == SQRT) {
#define I(k) "v_sqrt_f32 %" #k ", %" #k "\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
#undef I
        } else if (OP == MUL_LO_U32) {
#define I(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ux));)
#undef I
        } else if (OP == MUL_HI_U32) {
#define I(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ux));)
#undef I
        } else if (OP == MAD_U64_U32) {
#define I(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n"
            REP8(asm volatile(A8(I) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(ux), "v"(u0) : "vcc");)
#undef I
        } else if (OP == FMA_F64) {
#define I(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
            REP8(asm volatile(A8(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx), "v"(dy));)
#undef I
        } else if (OP == MUL_F64) {
#define I(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx));)
#undef I
        } else if (OP == ADD_F64) {
#define I(k) "v_add_f64 %" #k ", %" #k ", %8\n"
            REP8(asm volatile(A8(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dy));)
#undef I
        } else if (OP == CMP_CNDMASK) {
#define I(k) "v_cmp_lt_f32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
            REP8(asm volatile(I(0) I(1) I(2) I(3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");)
#undef I
        } else if (OP == VALU_SALU_MIX) {
#define I(k) "v_fma_f32 %" #k ", %" #k ", %9, %10\ns_add_u32 %8, %8, 1\n"
            REP8(asm volatile(A8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s_cnt) : "v"(x), "v"(y) : "scc");)
#undef I
        } else if (OP == DS_READ_B32) {
#define I(k) "ds_read_b32 %" #k ", %8 offset:" #k "*1024\n"
            REP8(asm volatile(A8(I) "s_waitcnt lgkmcnt(0)\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(laddr & 1023u) : "memory");)
#undef I
        } else if (OP == DS_READ2_B32) {
#define I(k) "ds_read2_b32 %" #k ", %8 offset0:" #k "*16 offset1:" #k "*16+6\n"
            REP8(asm volatile(A8(I) "s_waitcnt lgkmcnt(0)\n" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"((laddr & 1023u)) : "memory");)
#undef I
        } else if (OP == DS_READ_B64) {
#define I(k) "ds_read_b64 %" #k ", %8 offset:" #k "*2048\n"
            REP8(asm volatile(A8(I) "s_waitcnt lgkmcnt(0)\n" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(laddr) : "memory");)
#undef I
        } else if (OP == DS_WRITE_B64) {
#define I(k) "ds_write_b64 %8, %" #k " offset:" #k "*2048\n"
            REP8(asm volatile(A8(I) "s_waitcnt lgkmcnt(0)\n" : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(laddr) : "memory");)
#undef I
        } else if (OP == DIV_IEEE) {
            // the product's div_rn: plain `/` under -fhip-fp32-correctly-rounded-divide-sqrt (v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup)
            for (int k = 0; k < 8; k++) { a0 = a0 / x; asm volatile("" : "+v"(a0)); }
        } else if (OP == SQRT_IEEE) {
            for (int k = 0; k < 8; k++) { a0 = __builtin_sqrtf(a0) + y; asm volatile("" : "+v"(a0)); }
        } else if (OP == NODE_STEP_LIKE) {
            // the VALU shape of one BVH node step: 12 sub, 12 mul, 4 max3/min3-style, compares, selects (independent enough to dual-issue if the part could)
#define I(k) "v_sub_f32 %" #k ", %" #k ", %8\nv_mul_f32 %" #k ", %" #k ", %9\n"
            asm volatile(A8(I) A8(I) "v_max3_f32 %0, %0, %1, %2\nv_min3_f32 %3, %3, %4, %5\nv_max3_f32 %6, %6, %7, %0\nv_min3_f32 %1, %1, %2, %3\n"
                         "v_cmp_le_f32 vcc, %0, %3\nv_cndmask_b32 %4, %4, %5, vcc\nv_cmp_le_f32 vcc, %6, %1\nv_cndmask_b32 %7, %7, %2, vcc\n"
                         A8(I) "v_max3_f32 %0, %0, %1, %2\nv_min3_f32 %3, %3, %4, %5\nv_max3_f32 %6, %6, %7, %0\nv_min3_f32 %1, %1, %2, %3\n"
                         "v_cmp_le_f32 vcc, %0, %3\nv_cndmask_b32 %4, %4, %5, vcc\nv_cmp_le_f32 vcc, %6, %1\nv_cndmask_b32 %7, %7, %2, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(x) : "vcc");
#undef I
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((tid & 63u) == 0u) cycles[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) +
              (float)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) + (float)s_cnt;
    if (r == 12345.678f) sink[0] = r;
}

template <int OP>
static void run(unsigned iters, int cus, unsigned long long* d_cycles, float* d_sink, const int* filter, int n_filter) {
    if (n_filter) { bool ok = false; for (int i = 0; i < n_filter; i++) ok |= filter[i] == OP; if (!ok) return; }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_calib<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wps : {1, 2, 4, 8}) {
        // wps workgroups of 256 lanes per CU: LDS per workgroup sized so that exactly `wps` fit
        const size_t lds = (size_t)(160 * 1024 / wps) - 512;
        const int grid = cus * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_calib<OP>, dim3(grid), dim3(256), lds, 0, iters / 8, d_cycles, d_sink);   // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_calib<OP>, dim3(grid), dim3(256), lds, 0, iters, d_cycles, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)grid * 4);
        CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0, mx = 0;
        for (auto c : h) { sum += (double)c; mx = mx > (double)c ? mx : (double)c; }
        const double mean_cyc = sum / h.size();
        const double n_instr = (double)iters * kPerIter[OP];
        // one SIMD runs `wps` waves: cycles per wave-instruction per SIMD = elapsed / (instructions issued on that SIMD)
        std::printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"iters\": %u, \"per_iter\": %d, \"kernel_ms\": %.4f, \"memtime_ticks_mean\": %.0f, \"memtime_ticks_max\": %.0f, "
                    "\"ticks_per_op_per_simd\": %.3f, \"ns_per_op_per_simd\": %.4f, \"memtime_MHz\": %.1f}\n",
                    kNames[OP], wps, iters, kPerIter[OP], ms, mean_cyc, mx, mean_cyc / (n_instr * wps), ms * 1e6 / (n_instr * wps), mean_cyc / (ms * 1e3));
        std::fflush(stdout);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

template <int OP>
static void run_all(unsigned iters, int cus, unsigned long long* c, float* s, const int* f, int nf) {
    run<OP>(iters, cus, c, s, f, nf);
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(iters, cus, c, s, f, nf);
}

int main(int argc, char** argv) {
    unsigned iters = argc > 1 ? (unsigned)std::atoi(argv[1]) : 20000u;
    std::vector<int> filter;
    for (int i = 2; i < argc; i++) filter.push_back(std::atoi(argv[i]));
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int clk = 0;
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    std::fprintf(stderr, "CUs %d, clock attribute %d kHz\n", cus, clk);
    unsigned long long* d_cycles; float* d_sink;
    CHECK(hipMalloc((void**)&d_cycles, (size_t)cus * 8 * 4 * 8));
    CHECK(hipMalloc((void**)&d_sink, 64));
    run_all<0>(iters, cus, d_cycles, d_sink, filter.data(), (int)filter.size());
    return 0;
}
