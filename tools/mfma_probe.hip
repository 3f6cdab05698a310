// mfma_probe.hip -- can the sieve's two K=4 dot products (rays x spheres) run on the FP32 matrix
// pipe beside the vector ALU?  Measures, per 32-sphere x 64-ray chunk and per SIMD:
//   mode 0  the current sieve mix on the vector ALU alone (16 pairs x (7 v_pk_fma + 1 v_pk_add + 2 v_alignbit))
//   mode 1  8 x v_mfma_f32_32x32x2_f32 alone (2 ray halves x (s: 2, t: 2))
//   mode 2  mode 1 + the vector work that remains (per accumulator register one v_fma_f32 g = s*s + t and one
//           v_alignbit_b32, one lane-half swap) in the same wave
//   mode 3  mode 2 plus ~`extra` plain vector instructions per chunk standing in for walk + shading
// at 1..4 waves per SIMD, with the in-kernel clock (s_memtime / s_memrealtime).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                          \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

template <int MODE, int EXTRA>
__global__ void __launch_bounds__(256) probe(float* out, int chunks, float seed, unsigned long long* stamps)
{
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(seed + i, seed - i, 0.5f * i, 1.0f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float acc_out = 0.f;
    unsigned mask_out = 0;
    if constexpr (MODE == 0) {
        v2f nnx = {seed, seed}, nny = {seed * 2, seed * 2}, nnz = {seed * 3, seed * 3}, d0 = {1.f, 1.f};
        v2f p2x = {seed, seed}, p2y = {seed, seed}, p2z = {seed, seed}, no2 = {-seed, -seed};
        unsigned mask = 0;
        for (int c = 0; c < chunks; c++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float4 A = lds[(c * 32 + 2 * k) & 1023], B = lds[(c * 32 + 2 * k + 1) & 1023];
                const v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, w = {B.z, B.w};
                const v2f s = __builtin_elementwise_fma(cz, nnz, __builtin_elementwise_fma(cy, nny, __builtin_elementwise_fma(cx, nnx, d0)));
                const v2f t = __builtin_elementwise_fma(cz, p2z, __builtin_elementwise_fma(cy, p2y, __builtin_elementwise_fma(cx, p2x, no2 + w)));
                const v2f q = __builtin_elementwise_fma(s, s, t);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y), 31);
            }
            mask_out ^= mask;
        }
    } else if constexpr (MODE >= 4) {
        // the same products as three-term bf16 expansions: per 32x32 tile and dot product two
        // v_mfma_f32_32x32x16_bf16 (K = 4 dims x 8 partial products)
        v8bf bs1[2], bs2[2], bt1[2], bt2[2];
        v16f cinit[2];
        for (int g = 0; g < 2; g++) {
            for (int k = 0; k < 8; k++) {
                bs1[g][k] = (__bf16)(seed + lane + g + k); bs2[g][k] = (__bf16)(seed - lane + k);
                bt1[g][k] = (__bf16)(2.f * seed + g + k); bt2[g][k] = (__bf16)(1.0f + k);
            }
            for (int v = 0; v < 16; v++) cinit[g][v] = -seed * (lane + g);
        }
        float ex[8];
        for (int k = 0; k < 8; k++) ex[k] = seed + k;
        const v8bf* lb = reinterpret_cast<const v8bf*>(lds);
        for (int c = 0; c < chunks; c++) {
            const v8bf a1 = lb[(c * 128 + lane) & 1023], a2 = lb[(c * 128 + 64 + lane) & 1023];
            unsigned m[2] = {0u, 0u};
#pragma unroll
            for (int g = 0; g < 2; g++) {
                v16f S = {0};
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bs1[g], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bs2[g], S, 0, 0, 0);
                v16f T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bt1[g], cinit[g], 0, 0, 0);
                T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bt2[g], T, 0, 0, 0);
                if constexpr (MODE == 4) {
                    acc_out += S[0] + T[15];
                } else {
#pragma unroll
                    for (int v = 0; v < 16; v++) {
                        const float q = __builtin_fmaf(S[v], S[v], T[v]);
                        m[g] = __builtin_amdgcn_alignbit(m[g], __float_as_uint(q), 31);
                    }
                }
            }
            if constexpr (MODE >= 5) {
                const unsigned other = __shfl_xor((int)(lane < 32 ? m[1] : m[0]), 32, 64);
                mask_out ^= (lane < 32 ? m[0] : m[1]) << 16 | other;
            }
            if constexpr (MODE == 6) {
#pragma unroll
                for (int k = 0; k < EXTRA; k++) ex[k & 7] = __builtin_fmaf(ex[k & 7], 1.0001f, 0.5f);
            }
        }
        for (int k = 0; k < 8; k++) acc_out += ex[k];
    } else {
        // ray-side operands of the two ray halves (B matrices), constant over the sphere loop
        float bs1[2], bs2[2], bt1[2], bt2[2];
        v16f cinit[2];
        for (int g = 0; g < 2; g++) {
            bs1[g] = seed + lane + g; bs2[g] = seed - lane; bt1[g] = 2.f * seed + g; bt2[g] = 1.0f;
            for (int v = 0; v < 16; v++) cinit[g][v] = -seed * (lane + g);
        }
        float ex[8];
        for (int k = 0; k < 8; k++) ex[k] = seed + k;
        for (int c = 0; c < chunks; c++) {
            const float4 a = lds[(c * 64 + lane) & 1023];  // {a1, a2 for t, a2 for s, -}
            unsigned m[2] = {0u, 0u};
#pragma unroll
            for (int g = 0; g < 2; g++) {
                v16f S = {0};
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bs1[g], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bs2[g], S, 0, 0, 0);
                v16f T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bt1[g], cinit[g], 0, 0, 0);
                T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bt2[g], T, 0, 0, 0);
                if constexpr (MODE == 1) {
                    acc_out += S[0] + T[15];
                } else {
#pragma unroll
                    for (int v = 0; v < 16; v++) {
                        const float q = __builtin_fmaf(S[v], S[v], T[v]);
                        m[g] = __builtin_amdgcn_alignbit(m[g], __float_as_uint(q), 31);
                    }
                }
            }
            if constexpr (MODE >= 2) {
                // lanes 32..63 hand their half of ray group 0 down, lanes 0..31 theirs of group 1 up
                const unsigned other = __shfl_xor((int)(lane < 32 ? m[1] : m[0]), 32, 64);
                mask_out ^= (lane < 32 ? m[0] : m[1]) << 16 | other;
            }
            if constexpr (MODE == 3) {
#pragma unroll
                for (int k = 0; k < EXTRA; k++) ex[k & 7] = __builtin_fmaf(ex[k & 7], 1.0001f, 0.5f);
            }
        }
        for (int k = 0; k < 8; k++) acc_out += ex[k];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc_out + (float)mask_out;
    if (lane == 0) {
        atomicAdd(stamps + 0, c1 - c0);
        atomicAdd(stamps + 1, r1 - r0);
    }
}

template <int MODE, int EXTRA>
static void run(const char* name, float* d_out, unsigned long long* d_st, int cus)
{
    const int chunks = 20000;
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((probe<MODE, EXTRA>), dim3(blocks), dim3(256), 0, 0, d_out, 2000, 1.0001f, d_st);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(d_st, 0, 16));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE, EXTRA>), dim3(blocks), dim3(256), 0, 0, d_out, chunks, 1.0001f, d_st);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        CHECK(hipMemcpy(st, d_st, 16, hipMemcpyDeviceToHost));
        const double waves = (double)blocks * 4;
        const double clk_ghz = (double)st[0] / (double)st[1] * 0.1;  // s_memrealtime ticks at 100 MHz
        const double cyc_per_chunk_simd = (double)st[0] / waves / chunks;  // cycles a wave needs per chunk
        // one chunk = 32 spheres x 64 rays; per SIMD wps waves advance together
        const double tests_per_s = waves * chunks * 32.0 * 64.0 / (ms * 1e-3);
        printf("%-34s waves/SIMD %d  %8.3f ms  clock %.2f GHz  %7.1f cyc/chunk/wave  %6.1f cyc/chunk/SIMD  %.2f T tests/s\n",
               name, wps, ms, clk_ghz, cyc_per_chunk_simd, cyc_per_chunk_simd / wps, tests_per_s * 1e-12);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    float* d_out;
    unsigned long long* d_st;
    CHECK(hipMalloc(&d_out, (size_t)cus * 4 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_st, 16));
    run<0, 0>("valu sieve (pk_fma mix)", d_out, d_st, cus);
    run<1, 0>("8 mfma_f32_32x32x2 alone", d_out, d_st, cus);
    run<2, 0>("8 mfma + 32 fma + 32 alignbit", d_out, d_st, cus);
    run<3, 32>("  ... + 32 plain valu", d_out, d_st, cus);
    run<3, 64>("  ... + 64 plain valu", d_out, d_st, cus);
    run<3, 128>("  ... + 128 plain valu", d_out, d_st, cus);
    run<4, 0>("8 mfma_f32_32x32x16_bf16 alone", d_out, d_st, cus);
    run<5, 0>("8 bf16 mfma + 32 fma + 32 alignbit", d_out, d_st, cus);
    run<6, 32>("  ... + 32 plain valu", d_out, d_st, cus);
    run<6, 64>("  ... + 64 plain valu", d_out, d_st, cus);
    run<6, 128>("  ... + 128 plain valu", d_out, d_st, cus);
    return 0;
}
