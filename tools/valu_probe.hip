// valu_probe.hip -- measures the issue rate of the vector instructions the sieve loop is made of
// (v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_alignbit_b32) and of broadcast
// ds_read_b128 beside them, at 1..8 waves per SIMD.  Build:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/valu_probe.hip -o tools/valu_probe
// Prints one line per (mode, waves/SIMD): instructions/clk/CU assuming the measured wall time
// and a 2.4 GHz clock, and the equivalent lane-results per second.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));          \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int NACC = 16;

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float c1, float c2, unsigned long long* cyc)
{
    const unsigned long long t0 = clock64();
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(c1, c2, c1, c2);
    __syncthreads();
    v2f x[NACC];
    unsigned m[NACC];
    for (int k = 0; k < NACC; k++) {
        x[k] = v2f{(float)threadIdx.x + k, (float)k};
        m[k] = threadIdx.x + k;
    }
    const v2f k1 = {c1, c1}, k2 = {c2, c2};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) {
            if constexpr (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].x) : "v"(k1.x), "v"(k2.x));
            } else if constexpr (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(k1), "v"(k2));
            } else if constexpr (MODE == 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(k1));
            } else if constexpr (MODE == 3) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(k2));
            } else if constexpr (MODE == 4) {
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m[k]) : "v"(k1.x));
            } else if constexpr (MODE == 5) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k].x) : "v"(k1.x));
            } else if constexpr (MODE == 6) {  // pk_fma + one broadcast ds_read_b128 per 4 of them
                if ((k & 3) == 0) {
                    const float4 v = lds[(i + k) & 255];
                    x[k] = __builtin_elementwise_fma(x[k], v2f{v.x, v.y}, v2f{v.z, v.w});
                } else {
                    x[k] = __builtin_elementwise_fma(x[k], k1, k2);
                }
            } else if constexpr (MODE == 7) {  // the sieve's mix: 12 pk + 2 alignbit + 2 ds_read_b128 per 14
                if (k == 0) {
                    const float4 v = lds[(2 * i) & 255], w = lds[(2 * i + 1) & 255];
                    x[0] = __builtin_elementwise_fma(x[0], v2f{v.x, v.y}, v2f{w.z, w.w});
                    x[1] = x[1] + v2f{v.z, v.w};
                    x[2] = x[2] * v2f{w.x, w.y};
                } else if (k >= 3 && k < 12) {
                    x[k] = __builtin_elementwise_fma(x[k], k1, k2);
                } else if (k >= 12 && k < 14) {
                    m[k] = __builtin_amdgcn_alignbit(m[k], __float_as_uint(x[k - 12].x), 31);
                }
            }
        }
    }
    float s = 0.f;
    unsigned t = 0;
    for (int k = 0; k < NACC; k++) {
        s += x[k].x + x[k].y;
        t ^= m[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
    const unsigned long long t1 = clock64();  // s_memtime: shader-clock cycles
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int MODE>
static void run(const char* name, int per_iter, float* d_out, int cus)
{
    const int iters = 200000;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_cyc, 8));
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1.0001f, 0.5f, d_cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(d_cyc, 0, 8));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double insts = (double)blocks * 4 * iters * per_iter;  // wave-instructions
        const double per_s = insts / (ms * 1e-3);
        unsigned long long cyc = 0;
        CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        const double wave_cyc = (double)cyc / ((double)blocks * 4);          // mean cycles a wave lived
        const double clk_ghz = wave_cyc / (ms * 1e-3) * 1e-9;                // shader clock while it ran
        const double cyc_per_inst_simd = wave_cyc / ((double)iters * per_iter * wps);  // SIMD cycles per wave-inst
        printf("%-28s waves/SIMD %d  %8.3f ms  %7.2f G wave-inst/s  clock %.2f GHz  %5.2f cyc/inst/SIMD\n", name, wps,
               ms, per_s * 1e-9, clk_ghz, cyc_per_inst_simd);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    run<0>("v_fma_f32", NACC, d_out, cus);
    run<1>("v_pk_fma_f32", NACC, d_out, cus);
    run<2>("v_pk_mul_f32", NACC, d_out, cus);
    run<3>("v_pk_add_f32", NACC, d_out, cus);
    run<4>("v_alignbit_b32", NACC, d_out, cus);
    run<5>("v_mul_f32", NACC, d_out, cus);
    run<6>("pk_fma + ds_read_b128/4", NACC, d_out, cus);
    run<7>("sieve mix 12pk+2align+2ds", 14, d_out, cus);
    CHECK(hipFree(d_out));
    return 0;
}
