// rt_api.cpp -- the rt_context half of the C ABI (include/rt_hip.h): device memory, the
// geometry repack at upload, launch geometry and the dispatch entry points.  It stands in for
// the reference's ComputePipeline (src/engine/src/compute_pipeline.cpp) without any of its
// Vulkan objects: one HIP stream, four device buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rt_internal.h"

namespace {

thread_local std::string g_error;

}  // namespace

int rt_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

namespace {

#define fail rt_fail

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr size_t kLdsPerCu = RT_LDS_PER_CU;

// Every entry point that touches the device runs with the context's device current and puts the
// calling thread's own device back on return.
#define RT_ON_DEVICE(ctx)                                                                   \
    RtDeviceGuard guard_((ctx)->device);                                                    \
    if (guard_.status() != hipSuccess)                                                      \
        return fail(RT_E_HIP, "hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(guard_.status()))

}  // namespace

void rt_set_error(const char* msg) { g_error = msg ? msg : ""; }

struct rt_context {
    int device = 0;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_done[2] = {nullptr, nullptr};  // recorded after every dispatch, alternating (rt_sync_dispatch)
    unsigned long long n_dispatch = 0;           // dispatches issued so far
    hipEvent_t ev_order[4] = {nullptr, nullptr, nullptr, nullptr};  // rt_wait_stream / rt_stream_wait, used in turn
    unsigned n_order = 0;
    int width = 0, height = 0;

    float4* accum_own = nullptr;
    uint32_t* color_own = nullptr;
    float4* accum = nullptr;  // bound (own or caller's)
    uint32_t* color = nullptr;

    RtSphereAos* d_spheres = nullptr;  // AoS copy
    float4* d_geom = nullptr;  // pair-packed sieve geometry
    float* d_radius = nullptr; // radii, padded
    uint4* d_frags = nullptr;  // sphere-side operands of the matrix-core sieve, room for RT_MAX_MFMA_FAR_SPHERES
    bool have_frags = false;   // d_frags describes the uploaded spheres (false: too many of them)
    float2* d_materials = nullptr;  // material extension: {kind, param} per sphere
    int material_cap = 0;
    bool have_materials = false;    // true when some sphere has a kind other than 0
    int n_spheres = 0;   // spheres in the buffer (Scene::spheres().size())
    int sphere_cap = 0;  // allocated entries (multiple of RT_CHUNK)

    // RT_KERNEL_STREAM_CULLED: the first `cull_count` spheres sorted in space (built on demand)
    std::vector<RtSphereAos> h_spheres;  // host copy of the upload, the source of the sorted copy
    float4* d_cgeom = nullptr;    // pair-packed sieve geometry of the sorted list
    float* d_cradius = nullptr;   // radii of the sorted list
    float4* d_bounds = nullptr;   // pair-packed bounding spheres of the groups of RT_CULL_GROUP slots
    // lite kernel: rays per tile of its last dispatch, and the tile order made from them for the next one
    uint32_t* d_tile_cost = nullptr;
    uint32_t* d_tile_order = nullptr;
    uint32_t* d_tile_chain = nullptr;  // stream kernel: the most rays a pixel of the tile took (behind d_tile_cost, one allocation)
    unsigned tile_cap = 0;       // tiles both arrays have room for
    unsigned cost_tiles = 0;     // tiles d_tile_cost describes (0: nothing yet), for the region below
    uint2* d_cam_hits = nullptr;       // per pixel, the nearest hit of its camera ray (use_camera_hits)
    unsigned hits_cap = 0;             // tiles d_cam_hits has room for
    bool hits_valid = false;           // d_cam_hits describes hits_key
    float hits_key[12 + 8] = {0};      // camera, image size, region, sphere count and float profile of that
    unsigned long long lite_paths = 0;  // paths since the counters were reset whose camera ray was not traced against the
                                        // whole sphere list: the lite kernel's, the stream kernel's with cam_hits
    float4* d_split = nullptr;   // lights of the tiles whose frames several waves share (use_frame_split)
    size_t split_bytes = 0;
    unsigned order_tiles = 0;    // tiles d_tile_order orders (0: nothing yet), for the same region
    int order_uses = 0;          // dispatches that have used d_tile_order since it was sorted
    unsigned order_serial = 0;   // sorts so far
    uint4* d_units = nullptr;    // stream kernel: units of work made of d_tile_order (use_stream_split)
    size_t units_cap = 0;
    unsigned units_key[5] = {0}; // tiles, runs per heavy tile, frames per run, frames and order_serial d_units was made for
    int cost_region[4] = {0, 0, 0, 0};  // tile_col0, tile_cols, tile_row0, tile_rows of that dispatch
    float4* d_boxes = nullptr;    // per-lane traversal: boxes of the groups, then of the blocks (3 float4 per pair)
    int* d_perm = nullptr;        // sorted slot -> index in d_spheres
    int cull_cap = 0;             // slots allocated
    int cull_count = -1;          // sphere_count the sorted copy was built for; -1 = none / stale
    uint32_t kd_axes = 0;         // CulledScene::kd_axes of that copy

    int last_blocks = 0, last_threads = 0; // geometry of the last stream-kernel launch
    int last_mode = -1;                    // and its sieve: the values of rt_get_sieve_mode
    int last_split[3] = {0, 0, 0};         // and how it shared heavy tiles: RtParams::split_ways, split_frames, split_tiles
    const uint32_t* last_split_count = nullptr;  // ... and the device word with the tiles it split
    unsigned* d_work = nullptr;           // RT_WORK_BYTES: span counters + finished-workgroup count
    unsigned long long* d_rays = nullptr; // RT_RAY_BYTES: ray counters, summed by rt_get_counters
    uint64_t paths = 0;

    unsigned char ubo[80] = {0};
    bool have_ubo = false;
    int max_depth = 50;
    int profile = RT_PROFILE_STRICT;
    int kernel = RT_KERNEL_STREAM;
    int row0 = 0, row1 = 0;
    int frame_stride = 1;
    int clear_mode = RT_CLEAR_REFERENCE;
    int blocks = 0, threads = 0;  // 0 = automatic
    bool camera_shortcuts = true;   // RT_OPTION_CAMERA_SHORTCUTS
    bool fail_next_launch = false;  // rt_debug_fail_next_launch
    // rt_dispatch_frames_timed: event pairs around dispatches not yet read by rt_timed_total
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed_pending, timed_free;
};

namespace {

struct UboView {
    float fwd[3], right[3], up[3], pos[3];
    int sphere_count;
    unsigned frame_count;
};

UboView read_ubo(const unsigned char* u)
{
    UboView v;
    memcpy(v.fwd, u + 0, 12);
    memcpy(v.right, u + 16, 12);
    memcpy(v.up, u + 32, 12);
    memcpy(v.pos, u + 48, 12);
    memcpy(&v.sphere_count, u + 60, 4);
    memcpy(&v.frame_count, u + 64, 4);
    return v;
}

void free_images(rt_context* c)
{
    if (c->accum_own) (void)hipFree(c->accum_own);
    if (c->color_own) (void)hipFree(c->color_own);
    c->accum_own = nullptr;
    c->color_own = nullptr;
}

// New W x H images.  The context is touched only once both allocations have succeeded, so a
// failure leaves it rendering into the images it had.
int alloc_images(rt_context* c, int w, int h)
{
    const size_t px = (size_t)w * h;
    float4* accum = nullptr;
    uint32_t* color = nullptr;
    hipError_t e = hipMalloc(&accum, px * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&color, px * sizeof(uint32_t));
    // The reference leaves a new accumulation image undefined
    // (compute_pipeline.cpp:178-269); zero is one legal instance of that.
    if (e == hipSuccess) e = hipMemsetAsync(accum, 0, px * sizeof(float4), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(color, 0, px * sizeof(uint32_t), c->stream);
    if (e != hipSuccess) {
        if (accum) (void)hipFree(accum);
        if (color) (void)hipFree(color);
        return fail(RT_E_HIP, "allocating %d x %d images: %s", w, h, hipGetErrorString(e));
    }
    free_images(c);
    c->accum_own = c->accum = accum;
    c->color_own = c->color = color;
    c->width = w;
    c->height = h;
    return RT_OK;
}

// ceil(2^32 / d), or 0 for d <= 1 (the kernel then skips the division).  With m = ceil(2^32/d),
// __umulhi(t, m) == t / d whenever t * d < 2^32; launch() checks that for the tile indices.
unsigned tile_magic(int d)
{
    if (d <= 1) return 0u;
    return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d);
}

}  // namespace

void rt_finish_params(RtParams& P)
{
    P.total_pixels = (unsigned)P.tile_cols * (unsigned)P.tile_rows * 64u;
    P.tile_cols_magic = tile_magic(P.tile_cols);
    P.n_lds = P.n_pad <= RT_MAX_LDS_SPHERES ? P.n_pad : 0;
    const int rem = P.sphere_count - (P.n_pad - RT_CHUNK);  // spheres in the last chunk
    P.last_chunk_valid = (P.n_pad == 0 || rem >= RT_CHUNK) ? 0xffffffffu : ~(0xffffffffu >> rem);
    P.last_chunk_count = P.n_pad == 0 ? 0 : std::min(rem, (int)RT_CHUNK);
    // A span of a short dispatch (few frames, few spheres) is used up in microseconds, about as
    // long as the answer to a request for the next one takes: such dispatches ask one span ahead.
    // Long spans do not: a span held in reserve is a span an idle wave cannot take at the end.
    P.prefetch = (long long)P.n_frames * std::max(P.n_pad, 32) <= 1024 ? 1 : 0;
}

namespace {

// Fill the launch-independent part of the kernel arguments; validates state.
int make_params(rt_context* c, int n_frames, RtParams* out)
{
    if (!c->have_ubo) return fail(RT_E_STATE, "rt_set_uniforms has not been called");
    const UboView u = read_ubo(c->ubo);
    if (u.sphere_count < 0 || u.sphere_count > c->n_spheres)
        return fail(RT_E_INVALID, "ubo.sphereCount = %d but %d spheres are uploaded", u.sphere_count,
                    c->n_spheres);
    if (n_frames < 1) return fail(RT_E_INVALID, "n_frames must be >= 1 (got %d)", n_frames);
    RtParams P;
    memset(&P, 0, sizeof P);
    memcpy(P.fwd, u.fwd, 12);
    memcpy(P.right, u.right, 12);
    memcpy(P.up, u.up, 12);
    memcpy(P.pos, u.pos, 12);
    P.sphere_count = u.sphere_count;
    P.n_pad = (u.sphere_count + RT_CHUNK - 1) / RT_CHUNK * RT_CHUNK;
    P.width = c->width;
    P.height = c->height;
    const int rows = c->height / 8;
    const int r0 = std::max(0, c->row0);
    const int r1 = (c->row1 > 0 && c->row1 < rows) ? c->row1 : rows;
    P.tile_col0 = 0;
    P.tile_cols = c->width / 8;
    P.tile_row0 = r0;
    P.tile_rows = std::max(0, r1 - r0);
    P.n_frames = n_frames;
    rt_finish_params(P);
    P.max_depth = c->max_depth;
    P.frame0 = u.frame_count;
    P.frame_stride = c->frame_stride;
    P.clear_mode = c->clear_mode;
    P.spheres = c->d_spheres;
    P.geom = c->d_geom;
    P.radius = c->d_radius;
    P.materials = c->have_materials ? c->d_materials : nullptr;
    P.accum = c->accum;
    P.color = c->color;
    P.work_counter = c->d_work;
    P.rays = c->d_rays;
    *out = P;
    return RT_OK;
}

// w of the sieve geometry: r*r - c.c + margin*(c.c + r*r), evaluated in double and rounded UP to
// float so the margin is never eroded (see sieve_pair in rt_kernels.hip).
float sieve_w(double cx, double cy, double cz, double r, double margin = RT_SIEVE_MARGIN)
{
    const double cc = cx * cx + cy * cy + cz * cz;
    const double rr = r * r;
    const double w = rr - cc + margin * (cc + rr);
    float wf = (float)w;
    if ((double)wf < w) wf = std::nextafter(wf, INFINITY);
    return wf;
}

// Sphere-side operands of the sieve on the matrix cores ("sieve on the matrix cores" in
// rt_kernels.hip has the layout): per chunk of 32 spheres and MFMA step s, lane l = (row r = l & 31,
// half h = l >> 5) holds eight bf16: the three-term split a1 + a2 + a3 of coordinate d = 2 s + h of
// the sphere in row r, ordered a_p[j], p = 1 1 2 2 1 3 2 3; coordinate 3 is {w1, w2, w3, 1, 1, 1, 0, 0}.
// Row r holds sphere 16 ((r >> 2) & 1) + (r & 3) + 4 (r >> 3) of the chunk.  A sphere with a
// non-finite centre, radius or w can never be hit by the shader's test (its t is never a finite
// positive number), so it gets operands that are always rejected, like the padding.
void build_mfma_frags(const std::vector<RtSphereAos>& padded, std::vector<uint32_t>* out)
{
    const int cap = (int)padded.size();
    out->assign((size_t)cap * 16, 0u);
    auto bits = [](float x) { uint32_t u; memcpy(&u, &x, 4); return u; };
    auto flt = [](uint32_t u) { float x; memcpy(&x, &u, 4); return x; };
    auto split3 = [&](float x, uint32_t t[3]) {  // high halves of the three terms
        t[0] = bits(x) & 0xffff0000u;
        const float r1 = x - flt(t[0]);
        t[1] = bits(r1) & 0xffff0000u;
        const float r2 = r1 - flt(t[1]);
        t[2] = bits(r2) & 0xffff0000u;
        for (int k = 0; k < 3; k++) t[k] >>= 16;
    };
    const uint32_t one = 0x3f80u;
    for (int ch = 0; ch < cap / RT_CHUNK; ch++)
        for (int step = 0; step < 2; step++)
            for (int lane = 0; lane < 64; lane++) {
                const int r = lane & 31, h = lane >> 5, d = 2 * step + h;
                const RtSphereAos& sp = padded[(size_t)ch * RT_CHUNK + 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3)];
                float w = sieve_w(sp.cx, sp.cy, sp.cz, sp.radius, RT_SIEVE_MARGIN_MFMA);
                float c[3] = {sp.cx, sp.cy, sp.cz};
                if (!std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2]) || !std::isfinite(w)) {
                    c[0] = c[1] = c[2] = 0.f;
                    w = -3.0e38f;
                }
                uint32_t e[8];
                uint32_t t[3];
                if (d < 3) {
                    split3(c[d], t);
                    const int p_of[8] = {0, 0, 1, 1, 0, 2, 1, 2};
                    for (int j = 0; j < 8; j++) e[j] = t[p_of[j]];
                } else {
                    split3(w, t);
                    e[0] = t[0]; e[1] = t[1]; e[2] = t[2]; e[3] = e[4] = e[5] = one; e[6] = e[7] = 0u;
                }
                uint32_t* dst = out->data() + (((size_t)ch * 2 + step) * 64 + lane) * 4;
                for (int k = 0; k < 4; k++) dst[k] = e[2 * k] | (e[2 * k + 1] << 16);
            }
}

// The sorted copy of the first `count` spheres for RT_KERNEL_STREAM_CULLED.
//
// Order: spheres much larger than the typical one (and any with a non-finite entry) first, the
// rest by recursive median splits of their centres, so that every block of RT_BVH_BLOCK and every
// group of RT_CULL_GROUP consecutive slots is compact.  Each group gets a bounding sphere (c_b, R_b); the kernel skips a
// group when its sieve, run on the bounding sphere with the margin RT_CULL_MARGIN_O on o.o,
// rejects it for every ray of the wave.
//
// Why a skipped group cannot hold a hit (u = 2^-24, first order in u; notation of sieve_pair):
// the computed g of any sphere (c, r) satisfies |g - (F + m)| <= err, F = r^2 - D^2, D the
// distance from c to the ray's line up to 4u|c-o|^2, m = 2^-16 (o.o + c.c + r^2), err <= 62u(o.o +
// c.c) + 7u r^2.  A sphere i of the group is certainly rejected by its own sieve (and therefore,
// the sieve being conservative, cannot be hit by the shader's test) once
//     dist_i >= r_i + kappa (|o| + |c_i| + r_i),   kappa = sqrt(326 u) = 0.004408,
// because then dist_i^2 >= r_i^2 + m_i + err_i + 4u|c_i-o|^2.  With rho_i = |c_i - c_b| and
//     A = max_i ( rho_i + r_i + kappa (|c_i| + r_i) ),        R_b = A sqrt(1 + 2^-6) (rounded up),
// dist_i >= dist_b - rho_i makes dist_b >= A + kappa|o| sufficient, and since 2 A kappa |o| <=
// 2^-6 A^2 + 64 kappa^2 |o|^2 that follows from dist_b^2 >= R_b^2 + 65 * 326u |o|^2 = R_b^2 + 21190u
// |o|^2.  The group test being negative means dist_b^2 > R_b^2 + m_b - err_b - 4u|c_b-o|^2 >= R_b^2 +
// (2^-9 - 70u)|o|^2 = R_b^2 + 32698u |o|^2.  Hence RT_CULL_MARGIN_O = 2^-9.
// Entries of the box array beyond the groups: blocks (8 per superblock, at least the RT_BVH_BLOCKS
// the LDS-resident traversal tests in one go) and superblocks (tested 4 at a time).
int block_boxes(int n_pad)
{
    return std::max(RT_BVH_BLOCKS, ((n_pad + RT_BVH_BLOCK - 1) / RT_BVH_BLOCK + 7) / 8 * 8);
}
int super_boxes(int n_pad) { return std::max(4, ((n_pad + RT_BVH_SUPER - 1) / RT_BVH_SUPER + 3) / 4 * 4); }

struct CulledScene {
    std::vector<float> geom;    // 4 floats per slot, pair-packed
    std::vector<float> radius;  // per slot
    std::vector<float> bounds;  // 4 floats per group, pair-packed, padded to 64 groups
    std::vector<float> boxes;   // per-lane traversal: 6 floats per group, then per block (see build_boxes)
    std::vector<int> perm;      // per slot
    uint32_t kd_axes = 0;       // two bits per bit 3.. of a slot number: the axis along which the halves of the list that
                                // differ in that bit were split (the one most spheres were split by), see build_culled
};

void build_culled(const std::vector<RtSphereAos>& src, int count, int n_pad, CulledScene* out)
{
    const double kKappa = 0.00441;  // >= sqrt(326 * 2^-24)
    const float inf = INFINITY;
    auto finite = [](const RtSphereAos& s) {
        return std::isfinite(s.cx) && std::isfinite(s.cy) && std::isfinite(s.cz) && std::isfinite(s.radius);
    };
    // typical radius = median of the finite ones
    std::vector<float> radii;
    for (int i = 0; i < count; i++)
        if (finite(src[i])) radii.push_back(std::fabs(src[i].radius));
    float median = 0.f;
    if (!radii.empty()) {
        std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
        median = radii[radii.size() / 2];
    }
    // big ones (and any with a non-finite entry) first, in index order
    std::vector<std::pair<uint32_t, int>> keyed;
    keyed.reserve(count);
    std::vector<int> rest;
    for (int i = 0; i < count; i++) {
        if (!finite(src[i]) || std::fabs(src[i].radius) > 4.0f * median)
            keyed.push_back({0u, i});
        else
            rest.push_back(i);
    }
    // The rest: median splits, the lower half first, cut at slot boundaries: always at the multiple of RT_CULL_GROUP
    // inside a range whose slot number has the most trailing zero bits, so that superblocks (RT_BVH_SUPER slots), blocks
    // (RT_BVH_BLOCK) and groups (RT_CULL_GROUP) each are a compact cell of a k-d partition AND the two halves of a split
    // differ in one bit of the slot number.  All splits that differ in the same bit are along the same axis -- the one
    // with the largest extent summed over them, weighted by their spheres -- so the traversal turns "nearer half first"
    // into flipping that bit for a ray that travels towards lower coordinates along that axis (kd_axes).  (Runs of a
    // Morton curve, the first version, are not compact: 64 consecutive points of a 22 x 22 grid came out as rectangles
    // up to 22 x 8 that overlapped one another.)
    {
        const int base = (int)keyed.size();  // slot of rest[0]
        auto coord = [&](int i, int axis) { return axis == 0 ? src[i].cx : axis == 1 ? src[i].cy : src[i].cz; };
        struct Range { int lo, hi; };
        std::vector<Range> ranges{{0, (int)rest.size()}};
        out->kd_axes = 0;
        for (int tz = 30; tz >= 3; tz--) {
            const long long step = 1ll << tz;
            // the ranges with a boundary of this alignment inside (at most one: a second would make one of a higher)
            std::vector<std::pair<size_t, int>> cut;  // index into ranges, position of the boundary in `rest`
            double extent[3] = {0.0, 0.0, 0.0};
            for (size_t n = 0; n < ranges.size(); n++) {
                const Range r = ranges[n];
                if (r.hi - r.lo <= 1) continue;
                const long long s0 = ((long long)(base + r.lo) / step + 1) * step;
                if (s0 >= base + r.hi) continue;
                cut.push_back({n, (int)(s0 - base)});
                for (int a = 0; a < 3; a++) {
                    float lo = inf, hi = -inf;
                    for (int k = r.lo; k < r.hi; k++) {
                        lo = std::min(lo, coord(rest[k], a));
                        hi = std::max(hi, coord(rest[k], a));
                    }
                    if (hi >= lo && std::isfinite(hi - lo)) extent[a] += (double)(hi - lo) * (r.hi - r.lo);
                }
            }
            if (cut.empty()) continue;
            int axis = 0;
            for (int a = 1; a < 3; a++)
                if (extent[a] > extent[axis]) axis = a;
            if (tz - 3 < 16) out->kd_axes |= (uint32_t)axis << (2 * (tz - 3));
            for (const auto& c : cut) {
                const Range r = ranges[c.first];
                const int mid = c.second;
                std::nth_element(rest.begin() + r.lo, rest.begin() + mid, rest.begin() + r.hi, [&](int x, int y) {
                    const float cx = coord(x, axis), cy = coord(y, axis);
                    return cx < cy || (cx == cy && x < y);
                });
                ranges[c.first] = {r.lo, mid};
                ranges.push_back({mid, r.hi});
            }
        }
        for (const Range& r : ranges) std::sort(rest.begin() + r.lo, rest.begin() + r.hi);  // inside a group: index order
        for (int i : rest) keyed.push_back({1u, i});
    }

    const int ngroups = n_pad / RT_CULL_GROUP;
    const int nbounds = (ngroups + 63) / 64 * 64;
    out->geom.assign((size_t)n_pad * 4, 0.f);
    out->radius.assign((size_t)n_pad, 0.f);
    out->perm.assign((size_t)n_pad, 0);
    out->bounds.assign((size_t)nbounds * 4, 0.f);
    auto put = [](std::vector<float>& v, int slot, float x, float y, float z, float w) {
        const size_t base = (size_t)(slot >> 1) * 8 + (slot & 1);
        v[base + 0] = x; v[base + 2] = y; v[base + 4] = z; v[base + 6] = w;
    };
    for (int k = 0; k < n_pad; k++) {
        if (k < count) {
            const RtSphereAos& s = src[keyed[k].second];
            put(out->geom, k, s.cx, s.cy, s.cz, sieve_w(s.cx, s.cy, s.cz, s.radius));
            out->radius[k] = s.radius;
            out->perm[k] = keyed[k].second;
        } else {
            put(out->geom, k, 0.f, 0.f, 0.f, -inf);  // padding: rejected whatever the ray
        }
    }
    for (int g = 0; g < nbounds; g++) {
        const int first = g * RT_CULL_GROUP, end = std::min(count, first + RT_CULL_GROUP);
        if (first >= count) {
            put(out->bounds, g, 0.f, 0.f, 0.f, -inf);  // no spheres: never needed
            continue;
        }
        bool ok = true;
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (int k = first; k < end; k++) {
            const RtSphereAos& s = src[keyed[k].second];
            ok = ok && finite(s);
            const double c[3] = {s.cx, s.cy, s.cz};
            for (int a = 0; a < 3; a++) {
                blo[a] = std::min(blo[a], c[a]);
                bhi[a] = std::max(bhi[a], c[a]);
            }
        }
        if (!ok) {
            put(out->bounds, g, 0.f, 0.f, 0.f, inf);  // a non-finite member: always evaluated
            continue;
        }
        // the centre as the float the kernel will see; everything else relative to that float
        const float cb[3] = {(float)(0.5 * (blo[0] + bhi[0])), (float)(0.5 * (blo[1] + bhi[1])),
                             (float)(0.5 * (blo[2] + bhi[2]))};
        double A = 0.0;
        for (int k = first; k < end; k++) {
            const RtSphereAos& s = src[keyed[k].second];
            const double dx = (double)s.cx - cb[0], dy = (double)s.cy - cb[1], dz = (double)s.cz - cb[2];
            const double rho = std::sqrt(dx * dx + dy * dy + dz * dz);
            const double r = std::fabs((double)s.radius);
            const double cn = std::sqrt((double)s.cx * s.cx + (double)s.cy * s.cy + (double)s.cz * s.cz);
            A = std::max(A, rho + r + kKappa * (cn + r));
        }
        const double Rb = A * std::sqrt(1.0 + 1.0 / 64.0) * (1.0 + 1e-9);
        put(out->bounds, g, cb[0], cb[1], cb[2], sieve_w(cb[0], cb[1], cb[2], Rb));
    }

    // Boxes for the per-lane traversal (RT_KERNEL_BVH; trace_bvh in rt_kernels.hip has the test and
    // why it is conservative): entries 0 .. nbounds-1 the groups of RT_CULL_GROUP slots, then
    // block_boxes(n_pad) blocks of RT_BVH_BLOCK slots, then super_boxes(n_pad) superblocks of RT_BVH_SUPER.  The box of a node holds
    // every member sphere with its radius enlarged by RT_BOX_PAD (|c| + r); the kernel adds
    // RT_BOX_PAD |o| for its ray.  Stored as centre and half-width, pair-packed per axis: entries
    // (2k, 2k+1) = {cx0, cx1, hx0, hx1}, {cy0, cy1, hy0, hy1}, {cz0, cz1, hz0, hz1}.  A node with a
    // non-finite member gets half-width +inf (always entered), a node without spheres -inf (never).
    const int nblocks = block_boxes(n_pad), nsupers = super_boxes(n_pad);
    const int nboxes = nbounds + nblocks + nsupers;
    out->boxes.assign((size_t)nboxes * 6, 0.f);
    for (int g = 0; g < nboxes; g++) {
        const int level = g < nbounds ? 0 : g < nbounds + nblocks ? 1 : 2;
        const int size = level == 0 ? RT_CULL_GROUP : level == 1 ? RT_BVH_BLOCK : RT_BVH_SUPER;
        const int first = (g - (level == 0 ? 0 : level == 1 ? nbounds : nbounds + nblocks)) * size;
        const int end = std::min(count, first + size);
        float ctr[3] = {0.f, 0.f, 0.f}, half[3] = {-inf, -inf, -inf};
        if (first < count) {
            bool ok = true;
            double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
            for (int k = first; k < end; k++) {
                const RtSphereAos& s = src[keyed[k].second];
                ok = ok && finite(s);
                const double c[3] = {s.cx, s.cy, s.cz};
                const double r = std::fabs((double)s.radius);
                const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                const double reach = r + (double)RT_BOX_PAD * (cn + r);
                for (int a = 0; a < 3; a++) {
                    blo[a] = std::min(blo[a], c[a] - reach);
                    bhi[a] = std::max(bhi[a], c[a] + reach);
                }
            }
            for (int a = 0; a < 3; a++) {
                if (!ok) {
                    half[a] = inf;
                    continue;
                }
                ctr[a] = (float)(0.5 * (blo[a] + bhi[a]));
                const double h = std::max(bhi[a] - (double)ctr[a], (double)ctr[a] - blo[a]) * (1.0 + 1e-6);
                half[a] = (float)h;
                if ((double)half[a] < h) half[a] = std::nextafter(half[a], inf);
                if (!std::isfinite(ctr[a]) || !std::isfinite(half[a])) { ctr[a] = 0.f; half[a] = inf; }
            }
        }
        float* dst = out->boxes.data() + (size_t)(g >> 1) * 12 + (g & 1);
        for (int a = 0; a < 3; a++) {
            dst[4 * a + 0] = ctr[a];
            dst[4 * a + 2] = half[a];
        }
    }
}

void stream_geometry(const rt_context* c, const RtParams& P, int* blocks, int* threads)
{
    rt_stream_geometry(c->cu_count, c->blocks, c->threads, P, blocks, threads);
}

// Whether this dispatch runs the sieve on the matrix cores (mode 3 of rt_launch_stream).
bool wants_mfma(const rt_context* c, const RtParams& P)
{
    // up to RT_MAX_MFMA_SPHERES with the operands in LDS, beyond with the operands in global memory
    static const bool far_ok = rt_dev_knob("RT_DEBUG_MFMA_FAR", 1) != 0;  // 0 = scenes above RT_MAX_MFMA_SPHERES stay on the vector ALU
    const int limit = far_ok || c->kernel == RT_KERNEL_STREAM_MFMA ? RT_MAX_MFMA_FAR_SPHERES : RT_MAX_MFMA_SPHERES;
    const bool can = c->have_frags && P.n_pad > 0 && P.n_pad <= limit;
    static const int min_pad = rt_dev_knob("RT_DEBUG_MFMA_MIN_SPHERES", 64);  // smallest padded count RT_KERNEL_STREAM hands to the matrix cores
    return can && (c->kernel == RT_KERNEL_STREAM_MFMA || (c->kernel == RT_KERNEL_STREAM && P.n_pad >= min_pad));
}

}  // namespace

// Workgroup size and count of the persistent kernel.  The kernel is bound by VALU issue, which
// two waves per SIMD already saturate; more waves only add start-up and drain cost (every wave
// ends with a few half-empty iterations).  Measured on MI355X: 12 waves per CU is the best or
// within noise of it from 2 to 16 frames per dispatch, 8 when a dispatch gives a wave fewer than
// 16 lane-fills of paths (one 1080p frame).  Bounded by the LDS one workgroup needs (geometry +
// per-wave candidate masks).
void rt_stream_geometry(int cu_count, int want_blocks, int want_threads, const RtParams& P, int* blocks,
                        int* threads)
{
    const bool far = P.frags != nullptr && P.n_pad > RT_MAX_MFMA_SPHERES;  // matrix-core sieve, operands left in global memory
    const bool mfma = P.frags != nullptr && !far;  // the operand fragments are staged in LDS too: larger workgroups share them
    // matrix-core sieve with its operands in LDS: one workgroup of 768 threads per CU, three waves per SIMD
    // sharing one copy of the operands (measured at 484 spheres, since camera rays go by tile lists and the
    // kernel no longer fills the issue port: 19.7-20.3 Grays/s against 18.7-19.3 for two workgroups of 256);
    // 512 or 256 threads when 768 do not fit in LDS
    static const int t_env = rt_dev_knob("RT_DEBUG_MFMA_THREADS", RT_MFMA_MAX_THREADS);
    int t_mfma = std::max(256, std::min(RT_MFMA_MAX_THREADS, t_env / 256 * 256));
    while (t_mfma > 256 && rt_stream_lds_bytes(P.n_lds, t_mfma, true, 0, P.sg_wide) > kLdsPerCu) t_mfma -= 256;
    if (t_mfma == 256 && rt_stream_lds_bytes(P.n_lds, 256, true, 0, P.sg_wide) * 2 > kLdsPerCu) t_mfma = 512;
    int t = want_threads > 0 ? want_threads : (mfma ? t_mfma : P.n_lds <= 1024 ? 256 : RT_STREAM_MAX_THREADS);
    t = std::max(64, std::min(mfma ? RT_MFMA_MAX_THREADS : RT_STREAM_MAX_THREADS, t / 64 * 64));
    const size_t lds = rt_stream_lds_bytes(P.n_lds, t, mfma, 0, P.sg_wide);
    const unsigned long long paths = (unsigned long long)P.total_pixels * (unsigned)P.n_frames;
    int waves_per_cu = paths < 16ull * 64 * 12 * (unsigned)cu_count ? 8 : 12;
    if (mfma) waves_per_cu = t == 768 ? 12 : 8;
    int per_cu = std::min<int>(std::max(1, waves_per_cu / (t / 64)), (int)(kLdsPerCu / std::max<size_t>(lds, 1)));
    per_cu = std::max(1, per_cu);
    int b = want_blocks > 0 ? want_blocks : cu_count * per_cu;
    const unsigned lanes_needed = (P.total_pixels + (unsigned)t - 1) / (unsigned)t;
    b = (int)std::max(1u, std::min((unsigned)b, lanes_needed));
    *blocks = b;
    *threads = t;
}

namespace {

// Build and upload the sorted copy of the first `count` spheres unless the device already holds it.
int ensure_culled(rt_context* c, int count, int n_pad)
{
    if (c->cull_count == count) return RT_OK;
    CulledScene cs;
    try {
        build_culled(c->h_spheres, count, n_pad, &cs);
    } catch (const std::bad_alloc&) {
        return fail(RT_E_NOMEM, "out of host memory sorting %d spheres", count);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // no launch may still read the old copy
    c->cull_count = -1;
    if (n_pad > c->cull_cap) {
        if (c->d_cgeom) HIP_TRY(hipFree(c->d_cgeom));
        if (c->d_cradius) HIP_TRY(hipFree(c->d_cradius));
        if (c->d_bounds) HIP_TRY(hipFree(c->d_bounds));
        if (c->d_boxes) HIP_TRY(hipFree(c->d_boxes));
        if (c->d_perm) HIP_TRY(hipFree(c->d_perm));
        c->d_cgeom = nullptr; c->d_cradius = nullptr; c->d_bounds = nullptr; c->d_boxes = nullptr; c->d_perm = nullptr;
        c->cull_cap = 0;
        const size_t nbounds = ((size_t)n_pad / RT_CULL_GROUP + 63) / 64 * 64;
        HIP_TRY(hipMalloc(&c->d_cgeom, (size_t)n_pad * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->d_cradius, (size_t)n_pad * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_bounds, nbounds * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->d_boxes, (nbounds + block_boxes(n_pad) + super_boxes(n_pad)) * 6 * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_perm, (size_t)n_pad * sizeof(int)));
        c->cull_cap = n_pad;
    }
    HIP_TRY(hipMemcpy(c->d_cgeom, cs.geom.data(), cs.geom.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_cradius, cs.radius.data(), cs.radius.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_bounds, cs.bounds.data(), cs.bounds.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_boxes, cs.boxes.data(), cs.boxes.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_perm, cs.perm.data(), cs.perm.size() * sizeof(int), hipMemcpyHostToDevice));
    c->cull_count = count;
    c->kd_axes = cs.kd_axes;
    return RT_OK;
}

// The culled kernel's view of the scene: the sorted copies in place of geom/radius, plus bounds
// and perm.  False when the scene does not qualify: empty, too large to stage in LDS, or (unless
// `force`, which the audit uses) so small that the group test costs more than it saves -- measured,
// the 40-sphere default scene runs 12 % slower culled.
int use_culled(rt_context* c, RtParams* P, bool* on, bool force = false, int max_spheres = RT_MAX_LDS_SPHERES)
{
    *on = false;
    if (P->n_pad == 0 || P->n_pad > max_spheres) return RT_OK;
    if (!force && P->n_pad < 128) return RT_OK;
    int rc = ensure_culled(c, P->sphere_count, P->n_pad);
    if (rc != RT_OK) return rc;
    P->geom = c->d_cgeom;
    P->radius = c->d_cradius;
    P->bounds = c->d_bounds;
    P->boxes = c->d_boxes;
    P->n_bounds = (P->n_pad / RT_CULL_GROUP + 63) / 64 * 64;
    P->n_block_boxes = block_boxes(P->n_pad);
    for (int a = 0; a < 3; a++) {
        P->kd_flip[a] = 0u;
        for (int k = 0; k < 16; k++)
            if (((c->kd_axes >> (2 * k)) & 3u) == (unsigned)a) P->kd_flip[a] |= 1u << k;
    }
    P->perm = c->d_perm;
    *on = true;
    return RT_OK;
}

// The span counters are only back at zero when a stream kernel has run to its end.  After a
// launch that failed, or a stream that reported an error, they are zeroed from the host so that
// the next dispatch does not find its queues "already empty" and render nothing.
void reset_work(rt_context* c)
{
    if (c->d_work) (void)hipMemsetAsync(c->d_work, 0, RT_WORK_BYTES, c->stream);
}

// Development knob: what RT_KERNEL_STREAM runs for scenes of at most 64 traced spheres: 1 = the lite
// kernel (the default), 0 = the stream kernel itself.
int small_scene_kernel()
{
    static const int which = rt_dev_knob("RT_DEBUG_LITE", -1);
    return which;
}

int env_int(const char* name, int fallback) { return rt_dev_knob(name, fallback); }

// Tiles in descending order of their cost in an earlier dispatch of the same region (the lite kernel, which hands
// out whole tiles: the tile's rays; the stream kernel, whose lanes take single pixels: the most rays any pixel of the
// tile took, zero_cost below), for the kernels that hand out whole tiles or spans of pixels
// (rt_tile_order_kernel; nwaves and heavy_den: how many of the leading tiles it counts as heavy): fills P.tile_order
// when an order is at hand, and P.tile_cost (where this dispatch leaves its costs) when the next
// dispatch will want to sort.  An order serves 8 dispatches: the sort is one workgroup's work
// (measured under rocprofv3: 60 us for the 32 400 tiles of a 1080p image, 215 us for 4K) and what
// it sorts by changes little from one dispatch of a region to the next -- and the order only says
// who goes first, never what is computed.  Measured, an order used 1 / 4 / 8 times: default scene
// 1080p x 64 frames 83.4 / 84.9 / 85.1 Grays/s, 4096 spheres 1.96 / 1.99 / 2.00.  zero_cost: the kernel adds to the
// costs pixel by pixel instead of storing them tile by tile, so they are zeroed first, and
// recorded only by the dispatch before a sort (an atomic per pixel is not free); stored costs are
// recorded every time.  Does nothing (image order) when the arrays cannot be allocated.
void use_tile_order(rt_context* c, RtParams& P, bool order_now, bool zero_cost, int uses = 8, unsigned nwaves = 0,
                    unsigned heavy_den = 1)
{
    P.tile_chain = nullptr;
    static const int uses_env = env_int("RT_DEBUG_LPT_USES", 0);  // development knob
    if (uses_env > 0) uses = uses_env;
    const unsigned ntiles = P.total_pixels / 64u;
    const int region[4] = {P.tile_col0, P.tile_cols, P.tile_row0, P.tile_rows};
    P.tile_cost = nullptr;
    P.tile_order = nullptr;
    if (ntiles > c->tile_cap) {
        if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
        if (c->d_tile_order) (void)hipFree(c->d_tile_order);
        c->d_tile_cost = c->d_tile_order = nullptr;
        c->tile_cap = c->cost_tiles = c->order_tiles = 0;
        if (hipMalloc(&c->d_tile_cost, (size_t)ntiles * 8) == hipSuccess &&  // (costs, then chains)
            hipMalloc(&c->d_tile_order, ((size_t)ntiles + 2) * 4) == hipSuccess) {  // (+ the numbers of heavy tiles and of units)
            c->tile_cap = ntiles;
            c->d_tile_chain = c->d_tile_cost + ntiles;
        } else {
            if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
            c->d_tile_cost = c->d_tile_order = nullptr;
            (void)hipGetLastError();
            return;
        }
    }
    if (memcmp(region, c->cost_region, sizeof region) != 0) c->cost_tiles = c->order_tiles = 0;  // another region
    memcpy(c->cost_region, region, sizeof region);
    if (order_now) {
        if (c->order_tiles == ntiles && c->order_uses < uses) {
            c->order_uses++;
            P.tile_order = c->d_tile_order;
        } else if (c->cost_tiles == ntiles &&
                   rt_launch_tile_order(c->d_tile_cost, zero_cost ? c->d_tile_chain : nullptr, c->d_tile_order, ntiles, nwaves,
                                        heavy_den, c->stream) == hipSuccess) {
            c->order_tiles = ntiles;
            c->order_uses = 1;
            c->order_serial++;
            P.tile_order = c->d_tile_order;
        } else {
            c->order_tiles = 0;
        }
    }
    c->cost_tiles = 0;
    // will the next dispatch of this region sort?  (it may also not ask for an order at all: then the costs are unused)
    const bool wanted = c->order_tiles != ntiles || c->order_uses >= uses;
    // (the stream kernel: costs and chains, one behind the other while the capacity is what this region needs)
    if (zero_cost && (!wanted || hipMemsetAsync(c->d_tile_cost, 0, (size_t)c->tile_cap * 8, c->stream) != hipSuccess))
        return;
    P.tile_cost = c->d_tile_cost;
    P.tile_chain = c->d_tile_chain;
    c->cost_tiles = ntiles;  // valid once this dispatch has run (the stream is in order)
}

// Stream kernel: the nearest hit of every pixel's camera ray, found once (rt_camera_hits_kernel;
// RtParams::cam_hits) and used by every frame of every dispatch until the camera, the image, the
// region, the spheres, the materials or the float profile change (~0.15 ms for 3840x2160 and 484
// spheres; 8 bytes of device memory per pixel of the region).  Does nothing when RT_OPTION_CAMERA_SHORTCUTS
// is off or the buffer cannot be allocated.
int use_camera_hits(rt_context* c, RtParams& P, bool fused)
{
    P.cam_hits = nullptr;
    P.cam_live = nullptr;
    if (!c->camera_shortcuts) return RT_OK;
    const unsigned ntiles = P.total_pixels / 64u;
    if (ntiles > c->hits_cap) {
        if (c->d_cam_hits) (void)hipFree(c->d_cam_hits);
        c->d_cam_hits = nullptr;
        c->hits_cap = 0;
        c->hits_valid = false;
        // (the hits of the pixels, then a word per tile: the pixels that hit a sphere)
        if (hipMalloc(&c->d_cam_hits, (size_t)ntiles * (64u * sizeof(uint2) + sizeof(unsigned long long))) != hipSuccess) {
            (void)hipGetLastError();
            return RT_OK;
        }
        c->hits_cap = ntiles;
    }
    float key[12 + 8];
    memcpy(key, P.fwd, 12 * sizeof(float));  // fwd, right, up, pos: contiguous in RtParams
    const int ints[8] = {P.width, P.height, P.tile_col0, P.tile_cols, P.tile_row0, P.tile_rows, P.sphere_count, fused ? 1 : 0};
    memcpy(key + 12, ints, sizeof ints);
    if (!c->hits_valid || memcmp(key, c->hits_key, sizeof key) != 0) {
        c->hits_valid = false;
        const hipError_t e = rt_launch_camera_hits(P, fused, c->d_cam_hits,
                                                   reinterpret_cast<unsigned long long*>(c->d_cam_hits + (size_t)c->hits_cap * 64u), c->stream);
        if (e != hipSuccess) return fail(RT_E_HIP, "camera hit kernel launch: %s", hipGetErrorString(e));
        memcpy(c->hits_key, key, sizeof key);
        c->hits_valid = true;
    }
    P.cam_hits = c->d_cam_hits;
    P.cam_live = reinterpret_cast<const unsigned long long*>(c->d_cam_hits + (size_t)c->hits_cap * 64u);
    return RT_OK;
}

// Many frames of a small scene: the heavier half of the tiles of the order (as many as 1 GiB of lights
// allows) are traced by several waves each, a run of frames per wave, and joined in frame order
// afterwards (rt_lite_join_kernel).  A pixel between two mirrors bounces to the depth limit in every
// frame, and all its frames in one lane are n_frames x max_depth passes one after the other: 1.2 ms of
// a 1.45 ms dispatch at 1280x720 x 64 frames (depth 8), whatever else the chip does meanwhile.
// Measured, runs per tile 2 / 4 / 8 (Grays/s): 640x360 x 64 frames 27 / 49 / 71 (unsplit 14), 1280x720 x 64
// (depth 8) 89 / 99 / 97 (53), 1920x1080 x 64 112 / 107 / 104 (96), 3840x2160 x 16 114 / 108 / 97 -- the fewer
// tiles a wave slot of the chip gets, the more runs.  How many tiles: with the order by a tile's longest lane (which is
// what a wave lasts by), a third of them up to five tiles a wave slot, a sixteenth beyond -- measured at 64 frames,
// one tile in 2 / 3 / 4 / 8 / 16 / 32 (ms): 640x360 0.227 / 0.228 / 0.33 / 0.31 / 0.59 / 0.86, 1280x720 (depth 8) 0.524 /
// 0.511 / 0.59 / 0.59 / 0.61 / 0.83, 1920x1080 1.06 / 1.04 / 1.06 / 1.02 / 1.01 / 1.11, 2560x1440 1.77 / 1.73 / 1.71 /
// 1.69 / 1.67 / 1.68, 3840x2160 x 16 frames 1.08 / 1.06 / 1.04 / 1.03 / 1.03 / 1.02.  Does nothing without an order or
// when the lights cannot be allocated.
void use_frame_split(rt_context* c, RtParams& P)
{
    static const int share_env = env_int("RT_DEBUG_SPLIT_SHARE", -1);  // development knob: one tile in N; 0 = none; -1: as above
    static const int ways_env = env_int("RT_DEBUG_SPLIT_WAYS", 0);  // development knob: 2, 4 or 8
    P.split_tiles = 0;
    P.split_ways = 0;
    P.split_frames = 0;
    P.split_lds = 0;
    P.split_lights = nullptr;
    P.split_count = nullptr;
    P.split_units = nullptr;
    if (share_env == 0 || P.tile_order == nullptr || P.n_frames < 16) return;
    const unsigned ntiles = P.total_pixels / 64u;
    const unsigned slots = 20u * (unsigned)c->cu_count;  // waves of this kernel that the chip holds at a time
    const int share = share_env > 0 ? share_env : ntiles < 5u * slots ? 3 : 16;
    int ways = 2 * ntiles < 3 * slots ? 8 : 2 * ntiles < 9 * slots ? 4 : 2;
    // two runs whose lights would have to be parked in device memory, where four runs' lights fit in LDS (below):
    // four, up to images of 20 tiles per wave slot (measured at 64 frames, 2 ways parked against 4 through LDS:
    // 1920x1080 119 -> 124 Grays/s, 2560x1440 127 -> 132, 3840x2160 146 -> 141)
    auto fits_lds = [&](int w) {
        return rt_lite_handover_bytes(4u, w, (P.n_frames + w - 1) / w) <= (size_t)9216 * 4u;
    };
    if (ways == 2 && !fits_lds(2) && fits_lds(4) && ntiles < 20u * slots) ways = 4;
    if (ways_env == 2 || ways_env == 4 || ways_env == 8) ways = ways_env;
    const int frames = (P.n_frames + ways - 1) / ways;
    // Where the lights of a tile's later runs fit in LDS beside four workgroups' worth of everything else (9 KiB a
    // wave: 64 frames four ways), the runs are waves of one workgroup and the first of them joins the lights at the
    // end of the kernel: no parked lights in device memory, no join kernel.  Measured at 1280x720 x 64 frames: the
    // parked lights were 354 MB of a dispatch's 387 MB of HBM traffic, the join kernel 7 % of its time.
    static const int lds_on = rt_dev_knob("RT_DEBUG_SPLIT_LDS", 1);
    const unsigned waves = (unsigned)std::max(4, ways);
    // (eight ways would want workgroups of eight waves: measured 6 % slower than the parked lights at 640x360 x 64 frames)
    if (lds_on && ways <= 4 && rt_lite_handover_bytes(waves, ways, frames) <= (size_t)9216 * waves) {
        P.split_tiles = (int)(ntiles / (unsigned)share);
        P.split_ways = ways;
        P.split_frames = frames;
        P.split_lds = 1;
        return;
    }
    const size_t per_tile = (size_t)(ways - 1) * (size_t)frames * 64u * sizeof(float4);
    const size_t tiles = std::min<size_t>(ntiles / (unsigned)share, ((size_t)1 << 30) / per_tile);
    if (tiles == 0) return;
    if (tiles * per_tile > c->split_bytes) {
        if (c->d_split) (void)hipFree(c->d_split);
        c->d_split = nullptr;
        c->split_bytes = 0;
        if (hipMalloc(&c->d_split, tiles * per_tile) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        c->split_bytes = tiles * per_tile;
    }
    P.split_tiles = (int)tiles;
    P.split_ways = ways;
    P.split_frames = frames;
    P.split_lights = c->d_split;
}

// Stream kernel, several frames per dispatch: the tiles that rt_tile_order_kernel counted as heavy (the first
// order[ntiles] of the order, at most an eighth of the tiles) are traced as `ways` units of work each, a run of
// frames per unit, and joined in frame order afterwards (rt_lite_join_kernel).  The units are listed on the device
// (rt_tile_units_kernel), once per order and shape of dispatch; the lights of the later runs wait in device memory,
// 1 KiB per tile and frame.  Does nothing without an order or when the buffers cannot be allocated.
void use_stream_split(rt_context* c, RtParams& P, int ways_max)
{
    P.split_tiles = 0;
    P.split_ways = 0;
    P.split_frames = 0;
    P.split_lds = 0;
    P.split_lights = nullptr;
    P.split_count = nullptr;
    P.split_units = nullptr;
    if (ways_max < 2 || P.tile_order == nullptr || P.n_frames < 2) return;
    const unsigned ntiles = P.total_pixels / 64u;
    int frames = 1;  // a power of two: the kernel finds the end of a run by its low bits
    while (frames * std::min(ways_max, P.n_frames) < P.n_frames) frames *= 2;
    const int ways = (P.n_frames + frames - 1) / frames;  // every run holds a frame
    if (ways < 2) return;
    const size_t per_tile = (size_t)ways * (size_t)frames * 64u * sizeof(float4);
    const size_t tiles = std::min<size_t>(ntiles / 8u, ((size_t)1 << 29) / per_tile);
    if (tiles == 0) return;
    if (tiles * per_tile > c->split_bytes) {
        if (c->d_split) (void)hipFree(c->d_split);
        c->d_split = nullptr;
        c->split_bytes = 0;
        if (hipMalloc(&c->d_split, tiles * per_tile) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        c->split_bytes = tiles * per_tile;
    }
    const size_t n_units = (size_t)ntiles + tiles * (size_t)(ways - 1);
    if (n_units > c->units_cap) {
        if (c->d_units) (void)hipFree(c->d_units);
        c->d_units = nullptr;
        c->units_cap = 0;
        if (hipMalloc(&c->d_units, n_units * sizeof(uint4)) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        c->units_cap = n_units;
        c->units_key[0] = 0;
    }
    const unsigned key[5] = {ntiles, (unsigned)ways, (unsigned)frames, (unsigned)P.n_frames, c->order_serial};
    if (memcmp(key, c->units_key, sizeof key) != 0) {
        c->units_key[0] = 0;
        if (rt_launch_tile_units(c->d_tile_order, c->d_units, ntiles, (unsigned)tiles, ways, frames, P.n_frames, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        memcpy(c->units_key, key, sizeof key);
    }
    P.split_tiles = (int)tiles;
    P.split_ways = ways;
    P.split_frames = frames;
    P.split_count = c->d_tile_order + ntiles;
    P.split_units = c->d_units;
    P.split_lights = c->d_split;
}

int launch(rt_context* c, const RtParams& P_in, bool record)
{
    RtParams P = P_in;
    if (P.total_pixels == 0) return RT_OK;
    const bool fused = c->profile == RT_PROFILE_FUSED;
    const bool tile = c->kernel == RT_KERNEL_TILE;
    // at most 64 traced spheres: the lite kernel (one wave per tile with the sieve)
    const bool small = !record && P.n_pad <= 2 * RT_CHUNK;
    int which = c->kernel == RT_KERNEL_LITE ? 1 : 0;
    if (c->kernel == RT_KERNEL_STREAM) {
        const int env = small_scene_kernel();
        which = env >= 0 ? env : 1;
    }
    if (small && which == 1) {
        static const int threads_env = env_int("RT_DEBUG_LITE_THREADS", 0);  // development knob
        const int threads = threads_env >= 64 && threads_env <= 1024 ? threads_env / 64 * 64 : 256;
        hipError_t e = hipSuccess;
        if (c->fail_next_launch) {  // test hook; this kernel keeps no state between launches
            c->fail_next_launch = false;
            e = hipErrorLaunchFailure;
        } else {
            // Longest tiles first, from the costs the previous dispatch of the same region left behind
            // (use_tile_order).  Measured on the default scene:
            // 1920x1080 x 64 frames 2.98 -> 2.47 ms; sorted before every dispatch, 3840x2160 x 8 frames 1.27 -> 1.37 ms.
            // Dispatches of a few frames (the interactive loop: one) end with the tiles that hold a chain of
            // mirror bounces, wherever in the image they are; the chains sit where the geometry puts them, frame
            // after frame, so there an order serves 512 dispatches and its sort costs next to nothing: 1080p x 1
            // frame 61.7 -> 56.6-57.6 us with 128 (an order for 8 dispatches: 61.7 -> 61.9; for 32: 58.0); with the
            // scripted camera of the interactive benchmark 128 / 512 / 2048 dispatches: 43.08 / 42.74 / 43.0 us a frame.
            static const int lpt_min_frames = env_int("RT_DEBUG_LPT_MIN_FRAMES", 1);  // development knob; < 0: never
            if (lpt_min_frames >= 0) use_tile_order(c, P, P.n_frames >= lpt_min_frames, false, P.n_frames >= 16 ? 8 : 512);
            use_frame_split(c, P);
            P.cam_reuse = c->camera_shortcuts;  // the bundled camera pass of every frame
            e = rt_launch_lite(P, fused, threads, c->stream);
        }
        if (e != hipSuccess) return fail(RT_E_HIP, "lite kernel launch (%d threads): %s", threads, hipGetErrorString(e));
        if (c->camera_shortcuts)  // every camera ray is traced against its tile's candidates
            c->lite_paths += (unsigned long long)P.total_pixels * (unsigned)P.n_frames;
        c->last_mode = 4;
        c->last_split[0] = c->last_split[1] = c->last_split[2] = 0;
        c->last_split_count = nullptr;
        return RT_OK;
    }
    if (tile) {
        HIP_TRY(rt_launch_tile(P, fused, record, c->stream));
        c->last_mode = -1;
        c->last_split[0] = c->last_split[1] = c->last_split[2] = 0;
        c->last_split_count = nullptr;
        return RT_OK;
    }
    const bool mfma = wants_mfma(c, P);
    P.frags = mfma ? c->d_frags : nullptr;
    int blocks, threads;
    const bool far = mfma && P.n_pad > RT_MAX_MFMA_SPHERES;
    // matrix-core sieve over more than 512 spheres that the walk reads from LDS: candidate walks every 1024
    // spheres (RT_SG_CHUNKS_WIDE), if the masks still fit
    static const int sg_wide = env_int("RT_DEBUG_SG_WIDE", 1);  // development knob: 0 = never
    P.sg_wide = sg_wide && mfma && P.n_lds != 0 && P.n_pad > RT_SG_CHUNKS * RT_CHUNK;
    stream_geometry(c, P, &blocks, &threads);
    if (P.sg_wide && rt_stream_lds_bytes(P.n_lds, threads, mfma && !far, 0, true) > kLdsPerCu) {
        P.sg_wide = 0;
        stream_geometry(c, P, &blocks, &threads);
    }
    if (rt_stream_lds_bytes(P.n_lds, threads, mfma && !far, 0, P.sg_wide) > kLdsPerCu)
        return fail(RT_E_INVALID, "workgroup of %d threads needs more than 160 KiB of LDS", threads);
    if (P.width > 0xffff || P.height > 0xffff)  // a lane holds its pixel as x | y << 16
        return fail(RT_E_INVALID, "image of %d x %d pixels is too large for the stream kernel", P.width, P.height);
    if ((unsigned long long)P.tile_cols * P.tile_cols * P.tile_rows >= 0x100000000ull)
        return fail(RT_E_INVALID, "image of %d x %d tiles is too large for the stream kernel", P.tile_cols,
                    P.tile_rows);
    // d_work (span counters, workgroups finished) is zeroed at creation, and the kernel's last
    // workgroup zeroes it again on its way out, so no launch needs a memset in front of it; a
    // launch or a synchronisation that fails zeroes it here (reset_work).
    // The scalar cache serves the sieve faster than LDS while the geometry stream fits in it.
    int mode = c->kernel == RT_KERNEL_STREAM_SGPR || c->kernel == RT_KERNEL_STREAM_CULLED || c->kernel == RT_KERNEL_BVH ||
                       (c->kernel == RT_KERNEL_STREAM && P.n_pad <= RT_MAX_SGPR_SPHERES)
                   ? 1 : 0;
    if (mfma) mode = far ? 5 : 3;
    if (c->kernel == RT_KERNEL_STREAM_CULLED || c->kernel == RT_KERNEL_BVH) {
        bool on = false;
        const bool bvh = c->kernel == RT_KERNEL_BVH;
        const int rc = use_culled(c, &P, &on, false, bvh ? RT_BVH_FAR_MAX_SPHERES : RT_MAX_LDS_SPHERES);
        if (rc != RT_OK) return rc;
        // per-lane traversal: out of LDS while the scene fits there, out of global memory beyond
        static const int far_from = env_int("RT_DEBUG_BVH_FAR_FROM", RT_BVH_BLOCK * RT_BVH_BLOCKS + 1);  // development knob
        if (on) mode = !bvh ? 2 : P.n_lds != 0 && P.n_pad < far_from ? 4 : 6;
        if (mode == 6 && P.n_lds != 0) {  // nothing staged: geometry for workgroup size and LDS as for a scene beyond LDS
            P.n_lds = 0;
            stream_geometry(c, P, &blocks, &threads);
        }
        if (mode == 4 && rt_stream_lds_bytes(P.n_lds, threads, false, P.n_bounds * 3 / 2) > kLdsPerCu)
            return fail(RT_E_INVALID, "workgroup of %d threads needs more than 160 KiB of LDS", threads);
    }
    // Large scenes: spans of the heaviest tiles first.  The end of a dispatch is a few deep pixels taken late,
    // each a chain of (frames x depth) passes, and a pass grows with the scene.  Measured at 3840x2160, brute
    // force: 4096 spheres 16.5 -> 15.5 ms, 1000 spheres 13.4 -> 11.6 ms; but 484 spheres 9.63 -> 9.90 ms and
    // 200 spheres 4.82 -> 5.13 ms (the sort of 129 600 tiles and the cost atomics outweigh the shorter tail).
    // Culled and bvh kernels out of LDS: 1000 spheres 13.3 -> 9.1 ms (bvh) and 19.3 -> 13.8 ms (culled), 4096
    // spheres 5.8 -> 4.7 ms and 20.9 -> 16.3 ms.  The bvh kernel out of global memory is left as it is
    // (16 384 spheres 4.35 -> 4.75 ms: its dispatch is short and its tail is not what bounds it).
    static const int lpt_min_spheres = env_int("RT_DEBUG_LPT_STREAM_SPHERES", 0);  // development knob; < 0: never
    if (!record && mode != 6 && lpt_min_spheres >= 0 && P.sphere_count >= lpt_min_spheres &&
        (long long)P.total_pixels * P.n_frames >= (1 << 22)) {
        // The order is by the longest pixel of a tile (rt_tile_order_kernel: a lane takes a pixel at a time, and a pixel's
        // frames x bounces are one chain).  Several frames a dispatch: the heavy tiles in runs of frames
        // (use_stream_split): the tiles with a pixel whose rays in the dispatch before were more than 1 / split_den of the
        // passes of an average wave.  Measured, order by the tile's rays / by its longest pixel / and heavy tiles in
        // runs (ms): 484 spheres x 8 frames 1920x1080 2.31 / 2.40 / 1.60, 1280x720 x 32 frames - / 7.95 / 3.85; 1000
        // spheres 1920x1080 x 8 6.7 / 6.3 / 3.75; 4096 spheres 1920x1080 x 4 12.9 / 12.8 / 3.7, 2560x1440 x 2 7.0 / 6.9 /
        // 3.75, 3840x2160 x 2 7.6 / 7.5 / 6.55; 100 spheres 1280x720 x 16 0.73 / 0.74 / 0.44, 1920x1080 x 32 1.73 / 1.65 /
        // 1.37; 200 spheres 3840x2160 x 8 2.52 / 2.20 / 2.26.  1 / 4 and 1 / 8 measure the same, 1 / 2 less; 8 runs a tile
        // as 4, or up to 3 % better.
        static const int split_den = rt_dev_knob("RT_DEBUG_STREAM_SPLIT_DEN", 4);   // 0: never
        static const int split_ways = rt_dev_knob("RT_DEBUG_STREAM_SPLIT_WAYS", 8);
        const unsigned nwaves = (unsigned)blocks * (unsigned)threads / 64u;
        // (images of many tiles per wave end on a short note by themselves, and the instantiation that can park lights is 2 %
        // slower: 484 spheres 3840x2160 x 8 frames, 42 tiles per wave, 5.28 -> 5.36 ms; but 4096 spheres, 63 tiles per
        // wave of the kernel that streams its operands from L2, 7.50 -> 6.55 ms)
        static const int split_tiles = rt_dev_knob("RT_DEBUG_STREAM_SPLIT_TILES", 24);
        const bool split = split_den > 0 && P.n_frames >= 2 && (mode == 0 || mode == 1 || mode == 3 || mode == 5) &&
                           (mode == 5 || P.total_pixels / 64u <= (unsigned)split_tiles * nwaves);
        use_tile_order(c, P, true, true, 8, split ? nwaves : 0u, (unsigned)std::max(1, split_den));
        if (split) use_stream_split(c, P, split_ways);
    }
    P.cam_reuse = !record && c->camera_shortcuts;
    if (!record) {
        const int rc = use_camera_hits(c, P, fused);
        if (rc != RT_OK) return rc;
    }
    // The shading reads the winning sphere's record (centre, radius, albedo): out of LDS where the workgroups of
    // a CU leave room for a copy, else out of global memory (a gather that a wave of this kernel, three to a
    // SIMD, has little else to cover with).
    P.n_table = 0;
    if (!record && P.n_pad > 0) {
        const size_t per_cu = ((size_t)blocks + (size_t)c->cu_count - 1) / (size_t)c->cu_count;
        const size_t with_table = rt_stream_lds_bytes(P.n_lds, threads, mode == 3, mode == 4 ? P.n_bounds * 3 / 2 : 0,
                                                      mode != 2 && mode != 4 && mode != 6 && P.sg_wide, P.n_pad);
        if (with_table * per_cu <= kLdsPerCu) P.n_table = P.n_pad;
    }
    hipError_t e = hipSuccess;
    if (c->fail_next_launch) {  // test hook: a launch that died half way, its span counters left as they were
        c->fail_next_launch = false;
        (void)hipMemsetAsync(c->d_work, 0xff, RT_WORK_BYTES, c->stream);
        e = hipErrorLaunchFailure;
    } else {
        // the pixels that see the sky, all their frames; the stream kernel has the others (disjoint: any order)
        if (P.cam_live != nullptr) e = rt_launch_sky_frames(P, fused, c->stream);
        if (e == hipSuccess) e = rt_launch_stream(P, fused, record, mode, blocks, threads, c->stream);
        if (e == hipSuccess && P.split_count != nullptr) e = rt_launch_stream_join(P, c->stream);
    }
    if (e != hipSuccess) {
        reset_work(c);
        return fail(RT_E_HIP, "stream kernel launch (%d x %d): %s", blocks, threads, hipGetErrorString(e));
    }
    if (P.cam_hits != nullptr)  // every camera ray's hit was looked up
        c->lite_paths += (unsigned long long)P.total_pixels * (unsigned)P.n_frames;
    c->last_blocks = blocks;
    c->last_threads = threads;
    c->last_split[0] = P.split_count ? P.split_ways : 0;
    c->last_split[1] = P.split_count ? P.split_frames : 0;
    c->last_split[2] = P.split_count ? P.split_tiles : 0;
    c->last_split_count = P.split_count;
    c->last_mode = mode == 4 || mode == 6 ? 6 : mode == 5 ? 3 : mode;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_create(int device, int width, int height, rt_context** out)
{
    if (!out) return fail(RT_E_INVALID, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0) return fail(RT_E_INVALID, "bad image size %dx%d", width, height);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RT_E_HIP, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RT_E_INVALID, "device %d out of range [0,%d)", device, n);
    RtDeviceGuard guard_(device);
    if (guard_.status() != hipSuccess)
        return fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard_.status()));
    rt_context* c = new (std::nothrow) rt_context();
    if (!c) return fail(RT_E_NOMEM, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete c;
        return fail(RT_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    c->cu_count = prop.multiProcessorCount;
    auto bail = [&](int rc) {
        rt_destroy(c);
        return rc;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(RT_E_HIP, "hipStreamCreate failed"));
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done[1], hipEventDisableTiming) != hipSuccess)
        return bail(fail(RT_E_HIP, "hipEventCreate failed"));
    for (hipEvent_t& e2 : c->ev_order)
        if (hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return bail(fail(RT_E_HIP, "hipEventCreate failed"));
    if (hipMalloc(&c->d_work, RT_WORK_BYTES) != hipSuccess || hipMalloc(&c->d_rays, RT_RAY_BYTES) != hipSuccess)
        return bail(fail(RT_E_HIP, "hipMalloc failed"));
    if (hipMemsetAsync(c->d_work, 0, RT_WORK_BYTES, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_rays, 0, RT_RAY_BYTES, c->stream) != hipSuccess)
        return bail(fail(RT_E_HIP, "hipMemsetAsync failed"));
    int rc = alloc_images(c, width, height);
    if (rc != RT_OK) return bail(rc);
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_context* c)
{
    if (!c) return RT_OK;
    RtDeviceGuard guard_(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_images(c);
    if (c->d_spheres) (void)hipFree(c->d_spheres);
    if (c->d_geom) (void)hipFree(c->d_geom);
    if (c->d_radius) (void)hipFree(c->d_radius);
    if (c->d_frags) (void)hipFree(c->d_frags);
    if (c->d_materials) (void)hipFree(c->d_materials);
    if (c->d_cgeom) (void)hipFree(c->d_cgeom);
    if (c->d_cradius) (void)hipFree(c->d_cradius);
    if (c->d_bounds) (void)hipFree(c->d_bounds);
    if (c->d_boxes) (void)hipFree(c->d_boxes);
    if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
    if (c->d_tile_order) (void)hipFree(c->d_tile_order);
    if (c->d_units) (void)hipFree(c->d_units);
    if (c->d_split) (void)hipFree(c->d_split);
    if (c->d_cam_hits) (void)hipFree(c->d_cam_hits);
    if (c->d_perm) (void)hipFree(c->d_perm);
    if (c->d_work) (void)hipFree(c->d_work);
    if (c->d_rays) (void)hipFree(c->d_rays);
    for (auto* list : {&c->timed_pending, &c->timed_free})
        for (auto& ev : *list) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->ev_done)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_order)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_resize(rt_context* c, int width, int height)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (width <= 0 || height <= 0) return fail(RT_E_INVALID, "bad image size %dx%d", width, height);
    RT_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));  // vkDeviceWaitIdle, compute_pipeline.cpp:53
    return alloc_images(c, width, height);  // also unbinds caller-owned buffers
}

int rt_upload_spheres(rt_context* c, const void* aos, int count)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (count < 0 || (count > 0 && !aos)) return fail(RT_E_INVALID, "bad sphere buffer");
    RT_ON_DEVICE(c);
    const int cap = std::max(RT_CHUNK, (count + RT_CHUNK - 1) / RT_CHUNK * RT_CHUNK);
    HIP_TRY(hipStreamSynchronize(c->stream));  // no launch may still read the old buffers
    if (cap > c->sphere_cap) {
        c->n_spheres = 0;  // until the new buffers are filled there is nothing to trace
        c->cull_count = -1;
        c->hits_valid = false;
        c->have_materials = false;
        if (c->d_spheres) HIP_TRY(hipFree(c->d_spheres));
        if (c->d_geom) HIP_TRY(hipFree(c->d_geom));
        if (c->d_radius) HIP_TRY(hipFree(c->d_radius));
        c->d_spheres = nullptr;
        c->d_geom = nullptr;
        c->d_radius = nullptr;
        c->sphere_cap = 0;
        HIP_TRY(hipMalloc(&c->d_spheres, (size_t)cap * sizeof(RtSphereAos)));
        HIP_TRY(hipMalloc(&c->d_geom, (size_t)cap * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->d_radius, (size_t)cap * sizeof(float)));
        c->sphere_cap = cap;
    }
    // AoS copy (zero padded), the radii, and the pair-packed sieve geometry: for the pair
    // (2k, 2k+1)   float4 {cx0, cx1, cy0, cy1}, float4 {cz0, cz1, w0, w1},
    // w = r*r - c.c + margin*(c.c + r*r), evaluated in double and rounded UP to float so the
    // margin is never eroded (see sieve_chunk in rt_kernels.hip).
    std::vector<RtSphereAos> host;
    std::vector<float> geom, radius;
    try {
        host.resize(cap);
        geom.resize((size_t)cap * 4);
        radius.resize((size_t)cap);
    } catch (const std::bad_alloc&) {
        return fail(RT_E_NOMEM, "out of host memory staging %d spheres", count);
    }
    memset(host.data(), 0, host.size() * sizeof(RtSphereAos));
    if (count) memcpy(host.data(), aos, (size_t)count * sizeof(RtSphereAos));
    for (int i = 0; i < cap; i++) {
        const RtSphereAos& s = host[i];
        const size_t base = (size_t)(i >> 1) * 8 + (i & 1);
        geom[base + 0] = s.cx;
        geom[base + 2] = s.cy;
        geom[base + 4] = s.cz;
        geom[base + 6] = sieve_w(s.cx, s.cy, s.cz, s.radius);
        radius[i] = s.radius;
    }
    HIP_TRY(hipMemcpy(c->d_spheres, host.data(), (size_t)cap * sizeof(RtSphereAos), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_geom, geom.data(), (size_t)cap * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_radius, radius.data(), (size_t)cap * sizeof(float), hipMemcpyHostToDevice));
    c->have_frags = false;
    if (cap <= RT_MAX_MFMA_FAR_SPHERES) {
        if (!c->d_frags) HIP_TRY(hipMalloc(&c->d_frags, (size_t)RT_MAX_MFMA_FAR_SPHERES * 4 * sizeof(uint4)));
        std::vector<uint32_t> frags;
        try {
            build_mfma_frags(host, &frags);
        } catch (const std::bad_alloc&) {
            return fail(RT_E_NOMEM, "out of host memory staging %d spheres", count);
        }
        HIP_TRY(hipMemcpy(c->d_frags, frags.data(), frags.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->have_frags = true;
    }
    c->n_spheres = count;
    c->have_materials = false;  // materials belong to a sphere list: upload them again
    c->cull_count = -1;         // so does the sorted copy of the culled kernel
    c->hits_valid = false;      // and the hits of the camera rays
    host.resize(count);
    c->h_spheres.swap(host);
    return RT_OK;
}

int rt_upload_materials(rt_context* c, const float* kind_param, int count)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hits_valid = false;  // a dielectric sphere is also hit from inside: the camera rays' hits may change
    if (count == 0 || !kind_param) {  // back to the reference's single surface response
        c->have_materials = false;
        return RT_OK;
    }
    if (count != c->n_spheres)
        return fail(RT_E_INVALID, "%d materials for %d uploaded spheres", count, c->n_spheres);
    bool any = false;
    for (int i = 0; i < count; i++) {
        const float kind = kind_param[2 * i], param = kind_param[2 * i + 1];
        if (!(kind == 0.0f || kind == 1.0f || kind == 2.0f || kind == 3.0f))
            return fail(RT_E_INVALID, "material %d: unknown kind %g", i, (double)kind);
        if (kind == 3.0f && !(param > 0.0f))
            return fail(RT_E_INVALID, "material %d: a dielectric needs a positive index of refraction", i);
        any = any || kind != 0.0f;
    }
    if (count > c->material_cap) {
        if (c->d_materials) HIP_TRY(hipFree(c->d_materials));
        c->d_materials = nullptr;
        c->material_cap = 0;
        HIP_TRY(hipMalloc(&c->d_materials, (size_t)count * sizeof(float2)));
        c->material_cap = count;
    }
    HIP_TRY(hipMemcpy(c->d_materials, kind_param, (size_t)count * sizeof(float2), hipMemcpyHostToDevice));
    c->have_materials = any;
    return RT_OK;
}

int rt_set_uniforms(rt_context* c, const void* ubo80)
{
    if (!c || !ubo80) return fail(RT_E_INVALID, "null argument");
    memcpy(c->ubo, ubo80, 80);
    c->have_ubo = true;
    return RT_OK;
}

int rt_set_max_depth(rt_context* c, int d)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (d < 1 || d > 65536) return fail(RT_E_INVALID, "max_depth %d out of range [1,65536]", d);
    c->max_depth = d;
    return RT_OK;
}

int rt_set_profile(rt_context* c, int profile)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (profile != RT_PROFILE_STRICT && profile != RT_PROFILE_FUSED)
        return fail(RT_E_INVALID, "unknown float profile %d", profile);
    c->profile = profile;
    return RT_OK;
}

int rt_set_kernel(rt_context* c, int kernel)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (kernel < RT_KERNEL_STREAM || kernel > RT_KERNEL_BVH || kernel == 7 /* a kernel since removed */)
        return fail(RT_E_INVALID, "unknown kernel %d", kernel);
    c->kernel = kernel;
    return RT_OK;
}

int rt_set_option(rt_context* c, int option, int value)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    switch (option) {
        case RT_OPTION_CAMERA_SHORTCUTS:
            if (value != 0 && value != 1) return fail(RT_E_INVALID, "RT_OPTION_CAMERA_SHORTCUTS takes 0 or 1, got %d", value);
            c->camera_shortcuts = value != 0;
            return RT_OK;
        default:
            return fail(RT_E_INVALID, "unknown option %d", option);
    }
}

int rt_set_tile_rows(rt_context* c, int row0, int row1)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (row0 < 0 || (row1 > 0 && row1 < row0)) return fail(RT_E_INVALID, "bad tile rows [%d,%d)", row0, row1);
    c->row0 = row0;
    c->row1 = row1;
    return RT_OK;
}

int rt_set_frame_schedule(rt_context* c, int stride, int clear_mode)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (stride < 1) return fail(RT_E_INVALID, "frame stride %d must be >= 1", stride);
    if (clear_mode < RT_CLEAR_REFERENCE || clear_mode > RT_CLEAR_NEVER)
        return fail(RT_E_INVALID, "unknown clear mode %d", clear_mode);
    c->frame_stride = stride;
    c->clear_mode = clear_mode;
    return RT_OK;
}

int rt_set_launch(rt_context* c, int blocks, int threads)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (blocks < 0 || threads < 0 || threads > 1024 || (threads % 64) != 0)
        return fail(RT_E_INVALID, "bad launch geometry %d x %d", blocks, threads);
    c->blocks = blocks;
    c->threads = threads;
    return RT_OK;
}

int rt_get_info(rt_context* c, int* device, int* cu_count, int* blocks, int* threads)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (device) *device = c->device;
    if (cu_count) *cu_count = c->cu_count;
    // the geometry of the last stream launch, or what a one-frame dispatch would use
    int b = c->last_blocks, t = c->last_threads;
    RtParams P;
    if (b == 0 && c->have_ubo && make_params(c, 1, &P) == RT_OK) stream_geometry(c, P, &b, &t);
    if (blocks) *blocks = b;
    if (threads) *threads = t;
    return RT_OK;
}

int rt_debug_split_info(rt_context* c, int out[4])
{
    if (!c || !out) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    out[0] = c->last_split[0];
    out[1] = c->last_split[1];
    out[2] = c->last_split[2];
    out[3] = 0;
    if (c->last_split_count != nullptr) {
        uint32_t heavy = 0;
        HIP_TRY(hipMemcpyAsync(&heavy, c->last_split_count, sizeof heavy, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        out[3] = (int)std::min<uint32_t>(heavy, (uint32_t)c->last_split[2]);
    }
    return RT_OK;
}

int rt_get_sieve_mode(rt_context* c, int* mode)
{
    if (!c || !mode) return fail(RT_E_INVALID, "null argument");
    *mode = c->last_mode;
    return RT_OK;
}

int rt_dispatch_frames(rt_context* c, int n_frames)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    RtParams P;
    int rc = make_params(c, n_frames, &P);
    if (rc != RT_OK) return rc;
    rc = launch(c, P, false);
    if (rc != RT_OK) return rc;
    c->paths += (uint64_t)P.total_pixels * (uint64_t)n_frames;
    HIP_TRY(hipEventRecord(c->ev_done[c->n_dispatch & 1], c->stream));
    c->n_dispatch++;
    return RT_OK;
}

int rt_dispatch(rt_context* c) { return rt_dispatch_frames(c, 1); }

int rt_sync_dispatch(rt_context* c, int back)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (back < 0 || back > 1) return fail(RT_E_INVALID, "back must be 0 (the last dispatch) or 1 (the one before), got %d", back);
    RT_ON_DEVICE(c);
    if (c->n_dispatch <= (unsigned long long)back) return RT_OK;  // no such dispatch yet: nothing to wait for
    const hipError_t e = hipEventSynchronize(c->ev_done[(c->n_dispatch - 1 - (unsigned)back) & 1]);
    if (e != hipSuccess) {
        reset_work(c);
        return fail(RT_E_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
    }
    return RT_OK;
}

int rt_bind_buffers_next(rt_context* c, void* accum_dev, void* color_dev)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    c->accum = accum_dev ? static_cast<float4*>(accum_dev) : c->accum_own;
    c->color = color_dev ? static_cast<uint32_t*>(color_dev) : c->color_own;
    return RT_OK;
}

// An event recorded on `from` that `to` waits for: what was queued on `from` before this call finishes
// before anything queued on `to` after it starts; neither stream waits for the host.
static int order_streams(rt_context* c, hipStream_t from, hipStream_t to)
{
    hipEvent_t ev = c->ev_order[c->n_order++ % 4u];
    HIP_TRY(hipEventRecord(ev, from));
    HIP_TRY(hipStreamWaitEvent(to, ev, 0));
    return RT_OK;
}

int rt_wait_stream(rt_context* c, void* hip_stream)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    return order_streams(c, static_cast<hipStream_t>(hip_stream), c->stream);
}

int rt_stream_wait(rt_context* c, void* hip_stream)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    return order_streams(c, c->stream, static_cast<hipStream_t>(hip_stream));
}

static_assert(sizeof(hipIpcMemHandle_t) == RT_IPC_HANDLE_BYTES, "rt_hip.h promises a 64-byte handle");

int rt_ipc_alloc(int device, size_t bytes, void** dev_ptr, void* handle64)
{
    if (!dev_ptr || !handle64 || bytes == 0) return fail(RT_E_INVALID, "bad argument");
    *dev_ptr = nullptr;
    RtDeviceGuard guard_(device);
    if (guard_.status() != hipSuccess) return fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard_.status()));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    hipIpcMemHandle_t h;
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(RT_E_HIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    }
    memcpy(handle64, &h, sizeof h);
    *dev_ptr = p;
    return RT_OK;
}

int rt_ipc_open(int device, const void* handle64, void** dev_ptr)
{
    if (!dev_ptr || !handle64) return fail(RT_E_INVALID, "bad argument");
    *dev_ptr = nullptr;
    RtDeviceGuard guard_(device);
    if (guard_.status() != hipSuccess) return fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard_.status()));
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    HIP_TRY(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return RT_OK;
}

int rt_ipc_close(int device, void* dev_ptr)
{
    if (!dev_ptr) return RT_OK;
    RtDeviceGuard guard_(device);
    if (guard_.status() != hipSuccess) return fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard_.status()));
    HIP_TRY(hipIpcCloseMemHandle(dev_ptr));
    return RT_OK;
}

int rt_ipc_free(int device, void* dev_ptr)
{
    if (!dev_ptr) return RT_OK;
    RtDeviceGuard guard_(device);
    if (guard_.status() != hipSuccess) return fail(RT_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard_.status()));
    HIP_TRY(hipFree(dev_ptr));
    return RT_OK;
}

int rt_dispatch_frames_timed(rt_context* c, int n_frames)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (!c->timed_free.empty()) {
        ev = c->timed_free.back();
        c->timed_free.pop_back();
    } else {
        HIP_TRY(hipEventCreate(&ev.first));
        const hipError_t e = hipEventCreate(&ev.second);
        if (e != hipSuccess) {
            (void)hipEventDestroy(ev.first);
            return fail(RT_E_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    hipError_t e = hipEventRecord(ev.first, c->stream);
    int rc = e == hipSuccess ? rt_dispatch_frames(c, n_frames) : fail(RT_E_HIP, "hipEventRecord: %s", hipGetErrorString(e));
    if (rc == RT_OK) {
        e = hipEventRecord(ev.second, c->stream);
        if (e != hipSuccess) rc = fail(RT_E_HIP, "hipEventRecord: %s", hipGetErrorString(e));
    }
    try {
        // room for every pair in either list, so that rt_timed_total can move them without allocating
        const size_t pairs = c->timed_pending.size() + c->timed_free.size() + 1;
        c->timed_pending.reserve(pairs);
        c->timed_free.reserve(pairs);
        (rc == RT_OK ? c->timed_pending : c->timed_free).push_back(ev);
    } catch (const std::bad_alloc&) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
        return rc == RT_OK ? fail(RT_E_NOMEM, "out of host memory") : rc;
    }
    return rc;
}

int rt_timed_total(rt_context* c, float* ms, int* dispatches)
{
    if (!c || !ms) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0.0;
    int n = 0;
    hipError_t bad = hipSuccess;
    while (!c->timed_pending.empty()) {
        const auto ev = c->timed_pending.back();
        c->timed_pending.pop_back();
        c->timed_free.push_back(ev);  // rt_dispatch_frames_timed reserved the room: no allocation here
        float t = 0.f;
        const hipError_t e = hipEventElapsedTime(&t, ev.first, ev.second);
        if (e != hipSuccess) {  // the remaining pairs are still retired: nothing is left to be summed into a later total
            bad = e;
            continue;
        }
        total += t;
        n++;
    }
    if (bad != hipSuccess) return fail(RT_E_HIP, "hipEventElapsedTime: %s", hipGetErrorString(bad));
    *ms = (float)total;
    if (dispatches) *dispatches = n;
    return RT_OK;
}

int rt_sync(rt_context* c)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        reset_work(c);
        return fail(RT_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    }
    return RT_OK;
}

int rt_read_color(rt_context* c, uint8_t* rgba8)
{
    if (!c || !rgba8) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(rgba8, c->color, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_accum(rt_context* c, float* rgba32f)
{
    if (!c || !rgba32f) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(rgba32f, c->accum, (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_write_accum(rt_context* c, const float* rgba32f)
{
    if (!c || !rgba32f) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(c->accum, rgba32f, (size_t)c->width * c->height * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_device_buffers(rt_context* c, void** accum_dev, void** color_dev)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    if (accum_dev) *accum_dev = c->accum;
    if (color_dev) *color_dev = c->color;
    return RT_OK;
}

int rt_bind_buffers(rt_context* c, void* accum_dev, void* color_dev)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->accum = accum_dev ? static_cast<float4*>(accum_dev) : c->accum_own;
    c->color = color_dev ? static_cast<uint32_t*>(color_dev) : c->color_own;
    return RT_OK;
}

int rt_resolve(rt_context* c, float divisor)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    const int rows = c->height / 8;
    const int r0 = std::max(0, c->row0);
    const int r1 = (c->row1 > 0 && c->row1 < rows) ? c->row1 : rows;
    HIP_TRY(rt_launch_resolve(c->accum, c->color, c->width, (c->width / 8) * 8, r0 * 8, r1 * 8, divisor,
                              c->stream));
    return RT_OK;
}

int rt_trace_crop(rt_context* c, int n_frames, int x0, int y0, int w, int h, rt_record* records,
                  float* light)
{
    if (!c || !records) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    RtParams P;
    int rc = make_params(c, n_frames, &P);
    if (rc != RT_OK) return rc;
    const int xmax = (c->width / 8) * 8, ymax = (c->height / 8) * 8;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > xmax || y0 + h > ymax)
        return fail(RT_E_INVALID, "crop [%d,%d)+(%d,%d) outside the rendered %dx%d", x0, y0, w, h, xmax, ymax);
    // tile-aligned bounding box of the crop
    P.tile_col0 = x0 / 8;
    P.tile_cols = (x0 + w + 7) / 8 - P.tile_col0;
    P.tile_row0 = y0 / 8;
    P.tile_rows = (y0 + h + 7) / 8 - P.tile_row0;
    rt_finish_params(P);
    P.rec_x0 = x0;
    P.rec_y0 = y0;
    P.rec_w = w;
    P.rec_h = h;
    const size_t npix = (size_t)n_frames * w * h;
    const size_t nrec = npix * (size_t)c->max_depth;
    if (nrec > ((size_t)1 << 27))  // 2 GiB of records: a crop is a test probe, not an image
        return fail(RT_E_INVALID, "crop of %d frames x %dx%d x depth %d holds too many records", n_frames, w, h,
                    c->max_depth);
    std::vector<rt_record> init;
    try {
        init.resize(nrec);
    } catch (const std::bad_alloc&) {
        return fail(RT_E_NOMEM, "out of host memory for %zu records", nrec);
    }
    for (auto& r : init) {
        r.hit = -2;
        r.hash[0] = r.hash[1] = r.hash[2] = 0;
    }
    rt_record* d_rec = nullptr;
    float* d_light = nullptr;
    HIP_TRY(hipMalloc(&d_rec, nrec * sizeof(rt_record)));
    hipError_t e = hipMemcpy(d_rec, init.data(), nrec * sizeof(rt_record), hipMemcpyHostToDevice);
    if (e == hipSuccess && light) e = hipMalloc(&d_light, npix * 3 * sizeof(float));
    // on the context's stream: it does not wait for work queued on the default stream
    if (e == hipSuccess && light) e = hipMemsetAsync(d_light, 0, npix * 3 * sizeof(float), c->stream);
    if (e != hipSuccess) {
        (void)hipFree(d_rec);
        if (d_light) (void)hipFree(d_light);
        return fail(RT_E_HIP, "record buffer setup: %s", hipGetErrorString(e));
    }
    P.records = d_rec;
    P.rec_light = d_light;
    rc = launch(c, P, true);
    if (rc == RT_OK) {
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(records, d_rec, nrec * sizeof(rt_record), hipMemcpyDeviceToHost);
        if (e == hipSuccess && light) e = hipMemcpy(light, d_light, npix * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_E_HIP, "record readback: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_rec);
    if (d_light) (void)hipFree(d_light);
    return rc;
}

int rt_get_counters(rt_context* c, rt_counters* out)
{
    if (!c || !out) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    static_assert(sizeof(unsigned long long) == 8, "ray counters are 64-bit");
    unsigned long long slots[RT_RAY_SLOTS * RT_RAY_SLOT_STRIDE];
    HIP_TRY(hipMemcpyAsync(slots, c->d_rays, RT_RAY_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long rays = 0;
    for (unsigned i = 0; i < RT_RAY_SLOTS; i++) rays += slots[i * RT_RAY_SLOT_STRIDE];
    unsigned long long tests = 0;
    for (unsigned i = 0; i < RT_RAY_SLOTS; i++) tests += slots[i * RT_RAY_SLOT_STRIDE + 2];
#ifdef RT_COUNT_PASSES
    unsigned long long passes = 0;
    for (unsigned i = 0; i < RT_RAY_SLOTS; i++) passes += slots[i * RT_RAY_SLOT_STRIDE + 1];
    unsigned long long passlanes = 0;
    for (unsigned i = 0; i < RT_RAY_SLOTS; i++) passlanes += slots[i * RT_RAY_SLOT_STRIDE + 3];
    fprintf(stderr, "passes %llu rays %llu lanes per pass %.2f\n", passes, rays, passes ? (double)passlanes / passes : 0.0);
    unsigned long long bvh[4];
    rt_bvh_step_counts(bvh);
    if (bvh[0])
        fprintf(stderr, "per-lane traversal: %.2f steps per pass, %.2f of 64 lanes at work in a step; per ray %.2f blocks opened, "
                        "%.2f groups sieved, %.2f literal tests\n",
                (double)bvh[0] / passes, (double)bvh[1] / bvh[0], (double)bvh[2] / rays, (double)(bvh[1] - bvh[2]) / rays,
                (double)bvh[3] / rays);
#endif
#ifdef RT_PHASE_TIMERS
    {
        unsigned long long t[5] = {0, 0, 0, 0, 0};
        for (unsigned i = 0; i < RT_RAY_SLOTS; i++)
            for (int k = 0; k < 5; k++) t[k] += slots[i * RT_RAY_SLOT_STRIDE + 3 + k];
        fprintf(stderr, "wave cycles: refill %llu sieve %llu walk %llu shade %llu; iterations %llu; per iteration %.0f %.0f %.0f %.0f\n",
                t[0], t[1], t[2], t[3], t[4], (double)t[0] / t[4], (double)t[1] / t[4], (double)t[2] / t[4],
                (double)t[3] / t[4]);
        unsigned long long w[6] = {0, 0, 0, 0, 0, 0};
        for (unsigned i = 0; i < RT_RAY_SLOTS; i++) {
            const unsigned long long* q = slots + i * RT_RAY_SLOT_STRIDE + 8;
            w[0] += q[0]; w[1] += q[1]; w[2] = std::max(w[2], q[2]); w[3] = std::max(w[3], q[3]);
            w[4] = std::max(w[4], q[4]); w[5] += q[5];
        }
        unsigned long long sp[2] = {0, 0};
        for (unsigned i = 0; i < RT_RAY_SLOTS; i++) {
            sp[0] += slots[i * RT_RAY_SLOT_STRIDE + 14];
            sp[1] += slots[i * RT_RAY_SLOT_STRIDE + 15];
        }
        fprintf(stderr, "span fetches: %llu, %.0f cycles each, %.1f %% of the refill\n", sp[1], (double)sp[0] / std::max(1ull, sp[1]),
                100.0 * sp[0] / std::max(1ull, t[0]));
        const double nw = (double)std::max(1ull, w[5]);
        fprintf(stderr, "wave lifetime (us): mean %.1f, shortest %.1f, longest %.1f; after the queues ran dry: mean %.1f, longest %.1f\n",
                w[0] * 0.01 / nw, ~w[3] * 0.01, w[2] * 0.01, w[1] * 0.01 / nw, w[4] * 0.01);
    }
#endif
    unsigned long long list_rays = 0;
#ifndef RT_PHASE_TIMERS
    for (unsigned i = 0; i < RT_RAY_SLOTS; i++) list_rays += slots[i * RT_RAY_SLOT_STRIDE + 4];
#endif
    list_rays += c->lite_paths;  // every camera ray of those dispatches (an atomic per wave would cost the lite kernel 6 %)
    out->rays = rays;
    out->sphere_tests = tests;
    out->paths = c->paths;
    out->list_rays = list_rays;
    return RT_OK;
}

int rt_reset_counters(rt_context* c)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    HIP_TRY(hipMemsetAsync(c->d_rays, 0, RT_RAY_BYTES, c->stream));
    c->paths = 0;
    c->lite_paths = 0;
    return RT_OK;
}

int rt_timer_begin(rt_context* c)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    RT_ON_DEVICE(c);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return RT_OK;
}

int rt_timer_end(rt_context* c, float* ms)
{
    if (!c || !ms) return fail(RT_E_INVALID, "null argument");
    RT_ON_DEVICE(c);
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return RT_OK;
}

// Test hook: the next stream-kernel launch of this context is not made; the dispatch returns
// RT_E_HIP and finds the span counters as a kernel that died half way would have left them.
int rt_debug_fail_next_launch(rt_context* c)
{
    if (!c) return fail(RT_E_INVALID, "ctx is null");
    c->fail_next_launch = true;
    return RT_OK;
}

// Device arithmetic probe for the parity tests: out[i] = op(a[i], b[i], c[i]) evaluated on the
// GPU with the kernels' compile flags (0 sqrt, 1 divide, 2 fma, 3 mul-then-add, 4 hash->float,
// 5 1/sqrt, 6 unorm8, 7 wang_hash on the bit pattern).
int rt_debug_math(rt_context* c, int op, const float* a, const float* b, const float* cc, float* out, int n)
{
    if (!c || !a || !b || !cc || !out || n <= 0) return fail(RT_E_INVALID, "bad argument");
    RT_ON_DEVICE(c);
    float* d = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    HIP_TRY(hipMalloc(&d, 4 * bytes));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)n, cc, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt_launch_math_probe(op, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_E_HIP, "math probe: %s", hipGetErrorString(e));
    return RT_OK;
}

// Sieve audit for the parity tests: runs the stream kernel's sieve on n_rays caller-supplied rays
// (6 floats each: origin, direction) against the uploaded spheres.  out[0], out[1]: spheres
// rejected by the sieve whose literal discriminant (STRICT, FUSED) is not negative -- must be 0;
// out[2]: spheres with a non-negative STRICT discriminant; out[3]: spheres the sieve passed.
int rt_debug_sieve_audit(rt_context* c, const float* rays, int n_rays, uint64_t out[4])
{
    if (!c || !rays || !out || n_rays <= 0) return fail(RT_E_INVALID, "bad argument");
    RT_ON_DEVICE(c);
    RtParams P;
    int rc = make_params(c, 1, &P);
    if (rc != RT_OK) return rc;
    if (P.n_lds == 0 && P.n_pad) return fail(RT_E_INVALID, "too many spheres for the sieve audit");
    P.frags = wants_mfma(c, P) ? c->d_frags : nullptr;  // audits the sieve the context's kernel would run
    float* d_rays = nullptr;
    unsigned long long* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_rays, (size_t)n_rays * 6 * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n_rays * 6 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt_launch_sieve_audit(P, d_rays, n_rays, d_out, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    unsigned long long host[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, d_out, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_E_HIP, "sieve audit: %s", hipGetErrorString(e));
    for (int i = 0; i < 4; i++) out[i] = host[i];
    return RT_OK;
}

// Audit of the culled kernel's group test (see rt_cull_audit_kernel).
int rt_debug_cull_audit(rt_context* c, const float* rays, int n_rays, uint64_t out[7])
{
    if (!c || !rays || !out || n_rays <= 0) return fail(RT_E_INVALID, "bad argument");
    RT_ON_DEVICE(c);
    RtParams P;
    int rc = make_params(c, 1, &P);
    if (rc != RT_OK) return rc;
    bool on = false;
    rc = use_culled(c, &P, &on, true, RT_BVH_FAR_MAX_SPHERES);
    if (rc != RT_OK) return rc;
    if (!on) return fail(RT_E_INVALID, "the culled kernel does not apply to this scene (%d spheres)", P.sphere_count);
    float* d_rays = nullptr;
    unsigned long long* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_rays, (size_t)n_rays * 6 * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, 7 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, 7 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n_rays * 6 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt_launch_cull_audit(P, d_rays, n_rays, d_out, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    unsigned long long host[7] = {0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, d_out, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_E_HIP, "cull audit: %s", hipGetErrorString(e));
    for (int i = 0; i < 7; i++) out[i] = host[i];
    return RT_OK;
}

}  // extern "C"
