// rt_static.hip -- rt_dispatch_static: one dispatch of the path-trace kernel that allocates
// nothing, for callers that cannot call hipMalloc (code under HIP graph capture, tight serving
// loops, harnesses that forbid it).  It runs kernels rt_dispatch_frames runs too; what a
// context keeps in device memory of its own between calls -- span counters, ray counters, the
// repacked sieve geometry and radii -- lives here in one file-scope device array, `scratch`,
// that every GPU the library runs on holds until the process ends.
//
// Every call on a GPU uses that GPU's one copy of the array: calls that overlap on one GPU
// (two streams, two threads) race on it and must be serialised by the caller.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "rt_internal.h"

namespace {

// Layout of the scratch array (bytes).  The two counter blocks come first and are zeroed by one
// memset at the start of every call; the geometry is rewritten by the pack kernel of every call.
constexpr size_t kWorkAt = 0;
constexpr size_t kRaysAt = kWorkAt + RT_WORK_BYTES;
constexpr size_t kGeomAt = kRaysAt + RT_RAY_BYTES;
constexpr size_t kRadiusAt = kGeomAt + (size_t)RT_STATIC_MAX_SPHERES * sizeof(float4);
constexpr size_t kScratchBytes = kRadiusAt + (size_t)RT_STATIC_MAX_SPHERES * sizeof(float);
static_assert(kGeomAt % 256 == 0 && kRadiusAt % 256 == 0, "scratch blocks must stay 256-byte aligned");
static_assert(RT_STATIC_MAX_SPHERES % RT_CHUNK == 0 && RT_STATIC_MAX_SPHERES <= RT_MAX_LDS_SPHERES,
              "the static entry point stages all geometry in LDS");

}  // namespace

// 128 KiB per GPU.  In host code this name is only a handle for hipGetSymbolAddress: the
// address the kernels get is the one looked up per device in static_setup().
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

// AoS spheres -> pair-packed sieve geometry and radii, the repack rt_upload_spheres does on the
// host: for the pair (2k, 2k+1)  {cx0, cx1, cy0, cy1}, {cz0, cz1, w0, w1} with
// w = r*r - c.c + margin*(c.c + r*r) evaluated in double and rounded UP to float.  Slots from
// n_spheres to n_pad are zero spheres, as in the host's padded copy.
__global__ void rt_static_pack_kernel(const RtSphereAos* aos, int n_spheres, int n_pad, float* geom, float* radius)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float cx = 0.f, cy = 0.f, cz = 0.f, r = 0.f;
    if (i < n_spheres) {
        const float4 g = *reinterpret_cast<const float4*>(aos + i);
        cx = g.x; cy = g.y; cz = g.z; r = g.w;
    }
    const double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
    const double rr = (double)r * r;
    const double w = rr - cc + (double)RT_SIEVE_MARGIN * (cc + rr);
    float wf = (float)w;  // round to nearest, then one step up if that fell below w
    if ((double)wf < w) {
        const uint32_t b = __float_as_uint(wf);
        wf = wf == 0.0f ? __uint_as_float(1u) : __uint_as_float(wf > 0.0f ? b + 1u : b - 1u);
    }
    float* f = geom + (size_t)(i >> 1) * 8 + (i & 1);
    f[0] = cx; f[2] = cy; f[4] = cz; f[6] = wf;
    radius[i] = r;
}

// Sum of the ray-counter slots -> *out (one wave).
__global__ void rt_static_sum_rays_kernel(const unsigned long long* slots, unsigned long long* out)
{
    unsigned long long v = 0ull;
    for (unsigned i = threadIdx.x; i < RT_RAY_SLOTS; i += 64u) v += slots[i * RT_RAY_SLOT_STRIDE];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) *out = v;
}

constexpr int kMaxDevices = 64;

struct StaticDevice {
    unsigned char* base = nullptr;  // this device's copy of `scratch`
    int cu_count = 0;
};

std::mutex g_mutex;
StaticDevice g_devices[kMaxDevices];

// Once per device, with that device current: the device address of its copy of the array.
int static_setup(int device, StaticDevice* out)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    StaticDevice& d = g_devices[device];
    if (d.base == nullptr) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr)
            return rt_fail(RT_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess || cus <= 0)
            return rt_fail(RT_E_HIP, "compute-unit count of device %d: %s", device, hipGetErrorString(e));
        d.cu_count = cus;
        d.base = static_cast<unsigned char*>(p);
    }
    *out = d;
    return RT_OK;
}

}  // namespace

extern "C" int rt_dispatch_static(int device, void* hip_stream, const void* spheres_aos32_dev, int n_spheres,
                                  const void* ubo80, int width, int height, int n_frames, int max_depth,
                                  int profile, void* accum_dev, void* color_dev, uint64_t* rays_dev)
{
    if (!ubo80 || !accum_dev || !color_dev) return rt_fail(RT_E_INVALID, "null argument");
    if (width <= 0 || height <= 0) return rt_fail(RT_E_INVALID, "bad image size %dx%d", width, height);
    if (n_frames < 1) return rt_fail(RT_E_INVALID, "n_frames must be >= 1 (got %d)", n_frames);
    if (max_depth < 1 || max_depth > 65536)
        return rt_fail(RT_E_INVALID, "max_depth %d out of range [1,65536]", max_depth);
    if (profile != RT_PROFILE_STRICT && profile != RT_PROFILE_FUSED)
        return rt_fail(RT_E_INVALID, "unknown float profile %d", profile);
    if (n_spheres < 0 || (n_spheres > 0 && !spheres_aos32_dev)) return rt_fail(RT_E_INVALID, "bad sphere buffer");
    if (n_spheres > RT_STATIC_MAX_SPHERES)
        return rt_fail(RT_E_INVALID,
                       "rt_dispatch_static holds at most %d spheres in its static device buffer (got %d); "
                       "use a context (rt_create / rt_upload_spheres / rt_dispatch_frames) for larger scenes",
                       RT_STATIC_MAX_SPHERES, n_spheres);
    int n_devices = 0;
    hipError_t e = hipGetDeviceCount(&n_devices);
    if (e != hipSuccess || n_devices <= 0)
        return rt_fail(RT_E_HIP, "no HIP device available (%s); this library has no CPU path",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n_devices || device >= kMaxDevices)
        return rt_fail(RT_E_INVALID, "device %d out of range [0,%d)", device, std::min(n_devices, kMaxDevices));

    RtDeviceGuard guard(device);
    if (guard.status() != hipSuccess)
        return rt_fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard.status()));
    StaticDevice dev;
    int rc = static_setup(device, &dev);
    if (rc != RT_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);

    RtParams P;
    memset(&P, 0, sizeof P);
    const unsigned char* u = static_cast<const unsigned char*>(ubo80);  // includes/scene.hpp:22-29
    memcpy(P.fwd, u + 0, 12);
    memcpy(P.right, u + 16, 12);
    memcpy(P.up, u + 32, 12);
    memcpy(P.pos, u + 48, 12);
    memcpy(&P.sphere_count, u + 60, 4);
    memcpy(&P.frame0, u + 64, 4);
    if (P.sphere_count < 0 || P.sphere_count > n_spheres)
        return rt_fail(RT_E_INVALID, "ubo.sphereCount = %d but %d spheres were passed", P.sphere_count, n_spheres);
    P.n_pad = (P.sphere_count + RT_CHUNK - 1) / RT_CHUNK * RT_CHUNK;
    P.width = width;
    P.height = height;
    P.tile_col0 = 0;
    P.tile_cols = width / 8;  // floor(W/8) x floor(H/8) groups, compute_pipeline.cpp:429-432
    P.tile_row0 = 0;
    P.tile_rows = height / 8;
    P.n_frames = n_frames;
    rt_finish_params(P);
    P.max_depth = max_depth;
    P.frame_stride = 1;
    P.clear_mode = RT_CLEAR_REFERENCE;
    P.cam_reuse = 1;  // a pixel's camera ray is traced in its first frame only (no buffer involved)
    P.spheres = static_cast<const RtSphereAos*>(spheres_aos32_dev);
    P.geom = reinterpret_cast<const float4*>(dev.base + kGeomAt);
    P.radius = reinterpret_cast<const float*>(dev.base + kRadiusAt);
    P.accum = static_cast<float4*>(accum_dev);
    P.color = static_cast<uint32_t*>(color_dev);
    P.work_counter = reinterpret_cast<unsigned*>(dev.base + kWorkAt);
    P.rays = reinterpret_cast<unsigned long long*>(dev.base + kRaysAt);
    if ((unsigned long long)P.tile_cols * P.tile_cols * P.tile_rows >= 0x100000000ull)
        return rt_fail(RT_E_INVALID, "image of %d x %d tiles is too large for the stream kernel", P.tile_cols,
                       P.tile_rows);

#define STATIC_TRY(expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return rt_fail(RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

    // the array keeps the previous call's contents: both counter blocks start every call at zero
    STATIC_TRY(hipMemsetAsync(dev.base + kWorkAt, 0, RT_WORK_BYTES + RT_RAY_BYTES, st));
    if (P.total_pixels != 0) {
        if (P.n_pad > 0) {
            hipLaunchKernelGGL(rt_static_pack_kernel, dim3((P.n_pad + 255) / 256), dim3(256), 0, st, P.spheres,
                               n_spheres, P.n_pad, reinterpret_cast<float*>(dev.base + kGeomAt),
                               reinterpret_cast<float*>(dev.base + kRadiusAt));
            STATIC_TRY(hipGetLastError());
        }
        // Only kernels that need no scratch memory (private segment) are launched from here: the
        // runtime allocates that lazily, at the launch of the first kernel that wants more of it.
        // Those are the lite kernel, up to 64 traced spheres, and the stream kernel with its sieve
        // geometry in LDS (tests/test_abi.py reads the private segment sizes off the built library).
        if (P.n_pad <= 2 * RT_CHUNK) {
            STATIC_TRY(rt_launch_lite(P, profile == RT_PROFILE_FUSED, 256, st));
        } else {
            int blocks = 0, threads = 0;
            rt_stream_geometry(dev.cu_count, 0, 0, P, &blocks, &threads);
            if (rt_stream_lds_bytes(P.n_lds, threads) > RT_LDS_PER_CU)
                return rt_fail(RT_E_INVALID, "workgroup of %d threads needs more than 160 KiB of LDS", threads);
            STATIC_TRY(rt_launch_stream(P, profile == RT_PROFILE_FUSED, false, 0, blocks, threads, st));
        }
    }
    if (rays_dev) {
        hipLaunchKernelGGL(rt_static_sum_rays_kernel, dim3(1), dim3(64), 0, st, P.rays,
                           reinterpret_cast<unsigned long long*>(rays_dev));
        STATIC_TRY(hipGetLastError());
    }
#undef STATIC_TRY
    return RT_OK;
}
