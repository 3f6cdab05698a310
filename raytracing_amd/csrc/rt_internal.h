// rt_internal.h -- what the translation units of libraytracing_amd.so share besides the C ABI:
// the kernel launchers (rt_kernels.hip), the launch-geometry helpers (rt_api.cpp) and the
// thread-local error message.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_params.h"

// ---- rt_kernels.hip
size_t rt_stream_lds_bytes(int n_lds, int threads, bool mfma = false, int extra_float4 = 0, bool sg_wide = false, int n_table = 0);
hipError_t rt_launch_stream(const RtParams& P, bool fused, bool record, int mode, int blocks, int threads,
                            hipStream_t st);
hipError_t rt_launch_cull_audit(const RtParams& P, const float* rays, int n_rays, unsigned long long* out,
                                hipStream_t st);
hipError_t rt_launch_tile(const RtParams& P, bool fused, bool record, hipStream_t st);
hipError_t rt_launch_lite(const RtParams& P, bool fused, int threads, hipStream_t st);
size_t rt_lite_handover_bytes(unsigned waves, int ways, int split_frames);
hipError_t rt_launch_camera_hits(const RtParams& P, bool fused, uint2* hits, unsigned long long* live, hipStream_t st);
hipError_t rt_launch_sky_frames(const RtParams& P, bool fused, hipStream_t st);
hipError_t rt_launch_tile_order(const uint32_t* cost, const uint32_t* chain, uint32_t* order, unsigned n, unsigned nwaves,
                                unsigned heavy_den, hipStream_t st);
hipError_t rt_launch_stream_join(const RtParams& P, hipStream_t st);
hipError_t rt_launch_tile_units(uint32_t* order, uint4* units, unsigned n, unsigned cap, int ways, int frames, int n_frames,
                                hipStream_t st);
#ifdef RT_COUNT_PASSES
void rt_bvh_step_counts(unsigned long long out[4]);  // experiment build: traversal steps since the last call
#endif
hipError_t rt_launch_resolve(const float4* accum, uint32_t* color, int width, int x_max, int y0,
                             int y1, float divisor, hipStream_t st);
hipError_t rt_launch_sieve_audit(const RtParams& P, const float* rays, int n_rays, unsigned long long* out,
                                 hipStream_t st);
hipError_t rt_launch_math_probe(int op, const float* a, const float* b, const float* c, float* out,
                                int n, hipStream_t st);

// Development knobs.  The tuning experiments behind the measured choices in this library read
// RT_DEBUG_* environment variables, but only in a build with -DRT_DEV_KNOBS (tools/build_variant.sh
// makes one); the shipped library ignores the environment, every knob is its default.
#ifdef RT_DEV_KNOBS
#include <cstdlib>
inline int rt_dev_knob(const char* name, int fallback)
{
    const char* e = getenv(name);
    return e ? atoi(e) : fallback;
}
#else
inline int rt_dev_knob(const char*, int fallback) { return fallback; }
#endif

// ---- rt_api.cpp
// Sets the calling thread's error message (printf style) and returns `code`.
int rt_fail(int code, const char* fmt, ...);
// Derived kernel arguments: call after sphere_count, n_pad, n_frames and the tile rectangle are set.
void rt_finish_params(RtParams& P);
// Workgroup count and size of the stream kernel on a device of `cu_count` compute units;
// want_blocks / want_threads: 0 = choose.
void rt_stream_geometry(int cu_count, int want_blocks, int want_threads, const RtParams& P, int* blocks,
                        int* threads);
constexpr size_t RT_LDS_PER_CU = 160 * 1024;

// Makes `device` current for the lifetime of the object and puts the caller's device back after.
class RtDeviceGuard {
public:
    explicit RtDeviceGuard(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        err_ = prev_ == device ? hipSuccess : hipSetDevice(device);
        if (prev_ == device) prev_ = -1;  // nothing to put back
    }
    ~RtDeviceGuard()
    {
        if (prev_ >= 0) (void)hipSetDevice(prev_);
    }
    RtDeviceGuard(const RtDeviceGuard&) = delete;
    RtDeviceGuard& operator=(const RtDeviceGuard&) = delete;
    hipError_t status() const { return err_; }

private:
    int prev_ = -1;
    hipError_t err_ = hipSuccess;
};
