// coexec_probe.hip -- do a wave that issues only bf16 MFMAs and a wave that issues only plain
// vector instructions, resident on the SAME SIMD, run concurrently?  512-thread workgroups, one per
// CU: waves w and w + 4 share a SIMD.  role 0: every wave MFMA-only; role 1: every wave VALU-only;
// role 2: waves 0-3 MFMA-only, waves 4-7 VALU-only; role 3: every wave alternates 1 MFMA : R VALU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.hip -o tools/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int R>
__device__ __forceinline__ void mixed(int iters, v8bf a, v8bf b, v16f& c0, v16f& c1, float* x)
{
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < R; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(1.0001f), "v"(0.5f));
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < R; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(1.0001f), "v"(0.5f));
    }
}

__global__ void __launch_bounds__(512) probe(float* out, int iters, int role, int ratio, unsigned long long* st)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v8bf a, b;
    for (int k = 0; k < 8; k++) { a[k] = (__bf16)(1.0f + lane + k); b[k] = (__bf16)(0.5f * k); }
    v16f c0 = {0}, c1 = {0};
    float x[8];
    for (int k = 0; k < 8; k++) x[k] = lane + k;
    const bool do_mfma = role == 0 || (role == 2 && wave < 4);
    const bool do_valu = role == 1 || (role == 2 && wave >= 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 3) {
        if (ratio == 2) mixed<2>(iters, a, b, c0, c1, x);
        else if (ratio == 4) mixed<4>(iters, a, b, c0, c1, x);
        else if (ratio == 8) mixed<8>(iters, a, b, c0, c1, x);
        else mixed<16>(iters, a, b, c0, c1, x);
    } else if (do_mfma) {
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
    } else if (do_valu) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(1.0001f), "v"(0.5f));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[5];
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) atomicMax(st + (wave >= 4 ? 1 : 0), t1 - t0);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* d_out; unsigned long long* d_st;
    CHECK(hipMalloc(&d_out, (size_t)cus * 512 * 4));
    CHECK(hipMalloc(&d_st, 16));
    const int iters = 200000;
    const char* names[] = {"all waves MFMA only (2 per iter)", "all waves VALU only (16 per iter)", "waves 0-3 MFMA, waves 4-7 VALU", "every wave 1 MFMA : R VALU"};
    for (int role = 0; role < 4; role++)
        for (int ratio : {2, 4, 8, 16}) {
            if (role != 3 && ratio != 2) continue;
            hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, d_out, 1000, role, ratio, d_st);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(d_st, 0, 16));
            hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, d_out, iters, role, ratio, d_st);
            CHECK(hipDeviceSynchronize());
            unsigned long long st[2];
            CHECK(hipMemcpy(st, d_st, 16, hipMemcpyDeviceToHost));
            printf("%-36s R=%2d: waves 0-3 %7.1f cyc/iter, waves 4-7 %7.1f cyc/iter\n", names[role], ratio, (double)st[0] / iters, (double)st[1] / iters);
        }
    return 0;
}
