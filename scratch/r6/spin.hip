// Dev experiment (round 6): a finite, lowest-priority VALU spinner — one 64-lane wave per SIMD (or per CU) issuing dependent v_fma-free multiply-adds for a fixed number of
// iterations, then exiting (nothing waits on anything).  Used to see whether SIMDs that are kept issuing make the long-lived waves of another process's kernel faster.
//   usage: spin <waves per CU: 1..8> <millions of iterations per wave> [priority 0..3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_spin(unsigned long long n, int prio, float* out) {
    if (prio == 0) __builtin_amdgcn_s_setprio(0); else if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    float v = (float)threadIdx.x;
    for (unsigned long long i = 0; i < n; i++) { v = v * 1.0000001f + 1.0f; asm volatile("" : "+v"(v)); }
    if (v == 123.456f) out[0] = v;
}
int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 4; const unsigned long long n = (argc > 2 ? atoll(argv[2]) : 1000) * 1000000ull; const int prio = argc > 3 ? atoi(argv[3]) : 0;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d; hipMalloc(&d, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k_spin, dim3(cus * wpc), dim3(64), 0, 0, n, prio, d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); std::printf("spin: %d waves per CU x %llu iterations, priority %d: %.0f ms\n", wpc, n, prio, ms);
    return 0;
}
