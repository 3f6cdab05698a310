// rt_kernels.hip -- the path-trace kernels for gfx950 (CDNA4, wave64).
//
// What is computed is the reference compute shader, res/shaders/shader.comp:72-133
// with the RNG of res/shaders/def.glsl:40-62: per pixel and frame one camera ray,
// up to max_depth brute-force nearest-hit searches over the sphere list, a tinted
// near-mirror bounce, and a float4 accumulate + rgba8 resolve.
//
// Build with -ffp-contract=off: the only fused multiply-adds are the explicit ones of
// the FUSED float profile, so every result is reproducible bit for bit on the CPU.
//
// Three kernels, same results:
//   rt_tile_kernel   one wave per 8x8 tile, one lane per pixel, the sphere loop exactly as
//                    the shader writes it.  The plain mapping, kept as the in-GPU cross-check.
//   rt_stream_kernel persistent waves.  A lane whose pixel is finished takes the next pixel
//                    from its wave's span (64-pixel spans drawn from per-XCD work queues), so all
//                    64 lanes carry a live ray in almost every pass over the spheres, whatever
//                    the spread of bounce counts.  The sphere loop is split in two:
//                      sieve: for 32 spheres at a time a conservative form of the discriminant's
//                             sign is shifted into a per-lane mask, no branch.  On the vector ALU,
//                             packed-f32 math on sphere PAIRS: 5 instructions per ray-sphere test
//                             (sieve_pair); on the matrix cores, the two dot products as exact
//                             bf16-split MFMAs: 2 vector instructions per test (sieve_chunks_mfma);
//                      walk : each lane then visits only its own candidates in index order and
//                             evaluates the shader's expressions literally (sqrt, divide, t > 0,
//                             t < best), so ties and rounding are the shader's.
//                    MODE picks the sieve and where its operands come from (scalar cache, LDS,
//                    matrix-core fragments in LDS or streamed from global memory), or, not brute
//                    force any more, a spatially sorted list with wave-level group culling or a
//                    per-lane walk down a hierarchy of boxes, nearer nodes first (trace_bvh).
//   rt_lite_kernel   at most 64 spheres: one wave per tile, no queues, sieve + walk from LDS,
//                    lanes run their pixel's frames back to back.
// and their helpers: rt_camera_hits_kernel (the hit of every pixel's camera ray, once per camera and scene),
// rt_sky_frames_kernel (all frames of the pixels that see the sky), rt_tile_order_kernel and rt_tile_units_kernel
// (the tiles by descending cost; the units of work of a dispatch whose heavy tiles are traced in runs of frames),
// rt_lite_join_kernel (the lights of such runs, added in frame order), rt_resolve_kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rt_internal.h"

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define RT_FLT_MAX 3.402823466e+38f  // def.glsl:27

// ----------------------------------------------------------------- arithmetic

// The shader's `a*b + c` under the two float profiles.
template <bool FUSED>
__device__ __forceinline__ float mad(float a, float b, float c)
{
    if constexpr (FUSED)
        return __builtin_fmaf(a, b, c);
    else
        return a * b + c;
}

template <bool FUSED>
__device__ __forceinline__ v2f mad2(v2f a, v2f b, v2f c)
{
    if constexpr (FUSED)
        return __builtin_elementwise_fma(a, b, c);
    else
        return a * b + c;
}

template <bool FUSED>
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return mad<FUSED>(az, bz, mad<FUSED>(ay, by, ax * bx));
}

// def.glsl:40-48
__device__ __forceinline__ uint32_t wang_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

// def.glsl:52-54: the row stride is the literal 1920 whatever the image width.
__device__ __forceinline__ uint32_t rng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t bounce)
{
    return px + 1920u * py + frame * (1920u * 1080u) + bounce * (1920u * 1080u * 256u);
}

// def.glsl:55: v_cvt_f32_u32 rounds to nearest even; the scale by 2^-32 is exact.
__device__ __forceinline__ float hash_to_float(uint32_t h)
{
    return (float)h * 2.3283064365386963e-10f;
}

// rgba8 image store of shader.comp:132.
__device__ __forceinline__ uint32_t unorm8(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)floorf(v * 255.0f + 0.5f);
}

__device__ __forceinline__ bool clear_at(const RtParams& P, int j, uint32_t frame)
{
    return P.clear_mode == RT_CLEAR_REFERENCE ? frame == 1u
         : P.clear_mode == RT_CLEAR_FIRST     ? j == 0
                                              : false;
}

// ----------------------------------------------------------------- path state

struct Path {
    float ox, oy, oz;
    float dx, dy, dz;
    float cr, cg, cb;  // contribution
    float lr, lg, lb;  // light
    int bounce;
};

// shader.comp:73-76,82-90: direction of the camera ray of pixel (px,py).  It depends on the
// pixel and the camera only, not on the frame, so a pixel computes it once per dispatch.
template <bool FUSED>
__device__ __forceinline__ void primary_direction(const RtParams& P, uint32_t px, uint32_t py,
                                                  float& dx, float& dy, float& dz)
{
    const float fw = (float)P.width;
    const float hc = ((float)px * 2.0f - fw) / fw;
    const float vc = ((float)py * 2.0f - (float)P.height) / fw;  // divides by WIDTH
    const float x = mad<FUSED>(vc, P.up[0], mad<FUSED>(hc, P.right[0], P.fwd[0]));
    const float y = mad<FUSED>(vc, P.up[1], mad<FUSED>(hc, P.right[1], P.fwd[1]));
    const float z = mad<FUSED>(vc, P.up[2], mad<FUSED>(hc, P.right[2], P.fwd[2]));
    const float inv = 1.0f / sqrtf(dot3<FUSED>(x, y, z, x, y, z));
    dx = x * inv; dy = y * inv; dz = z * inv;
}

// shader.comp:88-94: state of a path at its start.
__device__ __forceinline__ void path_reset(Path& p, const RtParams& P, float dx, float dy, float dz)
{
    p.ox = P.pos[0]; p.oy = P.pos[1]; p.oz = P.pos[2];
    p.dx = dx; p.dy = dy; p.dz = dz;
    p.cr = p.cg = p.cb = 0.15f;
    p.lr = p.lg = p.lb = 0.0f;
    p.bounce = 0;
}

// {cos, sin} of 2*pi*u for u in [0,1]: quadrant reduction and the classic single-precision
// minimax polynomials on [-pi/4, pi/4], every step an explicit fma, so that a CPU evaluating
// the same sequence gets the same bits (the device's own sinf/cosf could not promise that).
__device__ __forceinline__ void sincos_2pi(float u, float& c, float& s)
{
    const float k = floorf(__builtin_fmaf(u, 4.0f, 0.5f));
    const float f = __builtin_fmaf(k, -0.25f, u);
    const float th = f * 6.283185307179586f;
    const float z = th * th;
    const float sp = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(sp * z, th, th);
    const float cp = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                                    4.166664568298827e-2f);
    const float cn = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    c = q == 0 ? cn : q == 1 ? -sn : q == 2 ? -cn : sn;
    s = q == 0 ? sn : q == 1 ? cn : q == 2 ? -sn : -cn;
}

// Material extension (no counterpart in the reference, whose one surface response is kind 0):
// scatter at a hit for kind 1 (lambertian), 2 (metal, param = fuzz) and 3 (dielectric, param =
// index of refraction), built on the sampling primitive of res/shaders/def.glsl:65-73
// (random_unit_vector: z = 2*rand(i)-1, a = 2*pi*rand(i+1)).  h = hit point, n = unit normal,
// r0/r1 drive the unit vector, r2 the reflect/refract choice.  Returns false when the ray is
// absorbed (the path ends without a sky term).
template <bool FUSED>
__device__ __forceinline__ bool scatter(Path& p, int kind, float param, float hx, float hy, float hz,
                                        float nx, float ny, float nz, float r0, float r1, float r2)
{
    float cs, sn;
    sincos_2pi(r1, cs, sn);
    const float z = mad<FUSED>(r0, 2.0f, -1.0f);
    const float rr = sqrtf(fmaxf(mad<FUSED>(-z, z, 1.0f), 0.0f));
    const float vx = rr * cs, vy = rr * sn, vz = z;  // random_unit_vector
    const float dinv = 1.0f / sqrtf(dot3<FUSED>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz));
    const float ux = p.dx * dinv, uy = p.dy * dinv, uz = p.dz * dinv;
    if (kind == 1) {  // lambertian: normalize(n + unit vector)
        float dx = nx + vx, dy = ny + vy, dz = nz + vz;
        const float dd = dot3<FUSED>(dx, dy, dz, dx, dy, dz);
        if (dd < 1e-12f) {
            dx = nx; dy = ny; dz = nz;
        } else {
            const float inv = 1.0f / sqrtf(dd);
            dx *= inv; dy *= inv; dz *= inv;
        }
        p.ox = mad<FUSED>(nx, 0.0001f, hx); p.oy = mad<FUSED>(ny, 0.0001f, hy); p.oz = mad<FUSED>(nz, 0.0001f, hz);
        p.dx = dx; p.dy = dy; p.dz = dz;
        return true;
    }
    if (kind == 2) {  // metal: reflect(unit d, n) + fuzz * unit vector; absorbed below the surface
        const float k2 = 2.0f * dot3<FUSED>(nx, ny, nz, ux, uy, uz);
        const float dx = mad<FUSED>(param, vx, mad<FUSED>(-k2, nx, ux));
        const float dy = mad<FUSED>(param, vy, mad<FUSED>(-k2, ny, uy));
        const float dz = mad<FUSED>(param, vz, mad<FUSED>(-k2, nz, uz));
        p.ox = mad<FUSED>(nx, 0.0001f, hx); p.oy = mad<FUSED>(ny, 0.0001f, hy); p.oz = mad<FUSED>(nz, 0.0001f, hz);
        if (!(dot3<FUSED>(dx, dy, dz, nx, ny, nz) > 0.0f)) return false;
        p.dx = dx; p.dy = dy; p.dz = dz;
        return true;
    }
    // dielectric: Schlick reflectance, Snell refraction
    const float dn = dot3<FUSED>(ux, uy, uz, nx, ny, nz);
    const bool front = dn < 0.0f;
    const float fx = front ? nx : -nx, fy = front ? ny : -ny, fz = front ? nz : -nz;
    const float ri = front ? 1.0f / param : param;
    const float cosv = fminf(-dot3<FUSED>(ux, uy, uz, fx, fy, fz), 1.0f);
    const float sinv = sqrtf(fmaxf(mad<FUSED>(-cosv, cosv, 1.0f), 0.0f));
    const float q = (1.0f - ri) / (1.0f + ri);
    const float q2 = q * q;
    const float x = 1.0f - cosv;
    const float x2 = x * x;
    const float x5 = (x2 * x2) * x;
    const float reflectance = mad<FUSED>(1.0f - q2, x5, q2);
    if (ri * sinv > 1.0f || reflectance > r2) {
        const float k2 = 2.0f * dot3<FUSED>(fx, fy, fz, ux, uy, uz);
        p.dx = mad<FUSED>(-k2, fx, ux); p.dy = mad<FUSED>(-k2, fy, uy); p.dz = mad<FUSED>(-k2, fz, uz);
        p.ox = mad<FUSED>(fx, 0.0001f, hx); p.oy = mad<FUSED>(fy, 0.0001f, hy); p.oz = mad<FUSED>(fz, 0.0001f, hz);
    } else {
        const float ex = ri * mad<FUSED>(cosv, fx, ux), ey = ri * mad<FUSED>(cosv, fy, uy),
                    ez = ri * mad<FUSED>(cosv, fz, uz);
        const float par = sqrtf(fabsf(1.0f - dot3<FUSED>(ex, ey, ez, ex, ey, ez)));
        p.dx = mad<FUSED>(-par, fx, ex); p.dy = mad<FUSED>(-par, fy, ey); p.dz = mad<FUSED>(-par, fz, ez);
        p.ox = mad<FUSED>(fx, -0.0001f, hx); p.oy = mad<FUSED>(fy, -0.0001f, hy); p.oz = mad<FUSED>(fz, -0.0001f, hz);
    }
    return true;
}

// shader.comp:97-118 after one Trace(): returns true when the path has ended.
// idx/t are the nearest hit (idx < 0: miss).  table = the AoS spheres as float4 pairs {centre,
// radius}, {albedo, padding}: P.spheres, or a workgroup's copy of it in LDS.
template <bool FUSED, bool RECORD>
__device__ __forceinline__ bool path_bounce(Path& p, const RtParams& P, int idx, float t,
                                            uint32_t px, uint32_t py, uint32_t frame,
                                            rt_record* rec, const float4* table)
{
    if (idx < 0) {
        p.lr = mad<FUSED>(0.6f, p.cr, p.lr);  // light += sky_color * contribution
        p.lg = mad<FUSED>(0.7f, p.cg, p.lg);
        p.lb = mad<FUSED>(0.9f, p.cb, p.lb);
        if constexpr (RECORD) if (rec) rec[p.bounce].hit = -1;
        return true;
    }
    const float4* sp = table + 2 * idx;
    const float4 g = sp[0];   // centre, radius
    const float4 col = sp[1]; // albedo

    // shader.comp:56-57 for the winning sphere
    const float hx = mad<FUSED>(t, p.dx, p.ox);
    const float hy = mad<FUSED>(t, p.dy, p.oy);
    const float hz = mad<FUSED>(t, p.dz, p.oz);
    const float pcx = hx - g.x, pcy = hy - g.y, pcz = hz - g.z;
    const float ninv = 1.0f / sqrtf(dot3<FUSED>(pcx, pcy, pcz, pcx, pcy, pcz));
    const float nx = pcx * ninv, ny = pcy * ninv, nz = pcz * ninv;

    const uint32_t s = rng_seed(px, py, frame, (uint32_t)p.bounce);
    if (P.materials != nullptr) {  // wave-uniform: the extension costs nothing when it is off
        const float2 m = P.materials[idx];
        const int kind = (int)m.x;
        if (kind != 0) {
            // rand(i), rand(i+1) for the unit vector (def.glsl:67-68), rand at pixel+(1,0) for the choice
            const uint32_t m0 = wang_hash(s), m1 = wang_hash(s + 1920u * 1080u * 256u), m2 = wang_hash(s + 1u);
            if constexpr (RECORD) if (rec) {
                rec[p.bounce].hit = idx;
                rec[p.bounce].hash[0] = m0;
                rec[p.bounce].hash[1] = m1;
                rec[p.bounce].hash[2] = m2;
            }
            p.cr *= col.x; p.cg *= col.y; p.cb *= col.z;
            if (idx <= 2) {
                p.lr = mad<FUSED>(2.0f, col.x, p.lr);
                p.lg = mad<FUSED>(2.0f, col.y, p.lg);
                p.lb = mad<FUSED>(2.0f, col.z, p.lb);
            }
            if (!scatter<FUSED>(p, kind, m.y, hx, hy, hz, nx, ny, nz, hash_to_float(m0), hash_to_float(m1),
                                hash_to_float(m2)))
                return true;
            p.bounce++;
            return p.bounce >= P.max_depth;
        }
    }
    const uint32_t h0 = wang_hash(s), h1 = wang_hash(s + 1u), h2 = wang_hash(s + 1920u);
    if constexpr (RECORD) if (rec) {
        rec[p.bounce].hit = idx;
        rec[p.bounce].hash[0] = h0;
        rec[p.bounce].hash[1] = h1;
        rec[p.bounce].hash[2] = h2;
    }
    const float r0 = hash_to_float(h0), r1 = hash_to_float(h1), r2 = hash_to_float(h2);
    const float rough = mad<FUSED>(r0, 0.02f - 0.0f, 0.0f);  // shader.comp:105

    p.cr *= col.x; p.cg *= col.y; p.cb *= col.z;  // :109
    if (idx <= 2) {                               // :110-112 (idx >= 0 here)
        p.lr = mad<FUSED>(2.0f, col.x, p.lr);
        p.lg = mad<FUSED>(2.0f, col.y, p.lg);
        p.lb = mad<FUSED>(2.0f, col.z, p.lb);
    }
    p.ox = mad<FUSED>(nx, 0.0001f, hx);           // :115
    p.oy = mad<FUSED>(ny, 0.0001f, hy);
    p.oz = mad<FUSED>(nz, 0.0001f, hz);

    const float vx = mad<FUSED>(r0, 2.0f, -1.0f), vy = mad<FUSED>(r1, 2.0f, -1.0f),
                vz = mad<FUSED>(r2, 2.0f, -1.0f);
    const float vinv = 1.0f / sqrtf(dot3<FUSED>(vx, vy, vz, vx, vy, vz));
    const float npx = mad<FUSED>(rough, vx * vinv, nx);  // :118 normal, not unit length
    const float npy = mad<FUSED>(rough, vy * vinv, ny);
    const float npz = mad<FUSED>(rough, vz * vinv, nz);
    const float k2 = 2.0f * dot3<FUSED>(npx, npy, npz, p.dx, p.dy, p.dz);
    p.dx = mad<FUSED>(-k2, npx, p.dx);                   // reflect(); not re-normalised
    p.dy = mad<FUSED>(-k2, npy, p.dy);
    p.dz = mad<FUSED>(-k2, npz, p.dz);

    p.bounce++;
    return p.bounce >= P.max_depth;
}

// shader.comp:44-60 for one sphere, literally.  The square root and the divide are skipped when
// b >= 0: then -b - sqrt(disc) <= 0, so t > 0 is false whatever the root is -- same outcome.
// far_root (material extension, dielectric spheres only): when the near root is not positive the
// far one is taken, so a ray that was refracted into the sphere meets its far wall.
// perm (culled kernel only): i and idx are slots of the sorted sphere list, visited in slot order;
// the shader's strict `<` in index order keeps, among equal t, the lowest ORIGINAL index.
template <bool FUSED>
__device__ __forceinline__ void test_sphere(const Path& p, float a, float cx, float cy, float cz,
                                            float r, int i, float& best, int& idx, bool far_root = false,
                                            const int* perm = nullptr)
{
    const float ocx = p.ox - cx, ocy = p.oy - cy, ocz = p.oz - cz;
    const float b = 2.0f * dot3<FUSED>(ocx, ocy, ocz, p.dx, p.dy, p.dz);
    const float c = mad<FUSED>(-r, r, dot3<FUSED>(ocx, ocy, ocz, ocx, ocy, ocz));  // dot(oc,oc) - r*r
    const float disc = mad<FUSED>(b, b, -((4.0f * a) * c));
    if (!(disc < 0.0f) && (b < 0.0f || far_root)) {
        const float sq = sqrtf(disc);
        float t = (-b - sq) / (2.0f * a);
        if (far_root && !(t > 0.0f)) t = (-b + sq) / (2.0f * a);
        bool closer = t < best;
        if (perm != nullptr && t == best && idx >= 0) closer = perm[i] < perm[idx];
        if (t > 0.0f && closer) {
            best = t;
            idx = i;
        }
    }
}

__device__ __forceinline__ rt_record* record_slot(const RtParams& P, uint32_t px, uint32_t py, int j)
{
    const int rx = (int)px - P.rec_x0, ry = (int)py - P.rec_y0;
    if (rx < 0 || ry < 0 || rx >= P.rec_w || ry >= P.rec_h) return nullptr;
    const size_t pix = ((size_t)j * P.rec_h + ry) * P.rec_w + rx;
    return P.records + pix * (size_t)P.max_depth;
}

__device__ __forceinline__ void record_light(const RtParams& P, uint32_t px, uint32_t py, int j,
                                             const Path& p)
{
    const int rx = (int)px - P.rec_x0, ry = (int)py - P.rec_y0;
    if (!P.rec_light || rx < 0 || ry < 0 || rx >= P.rec_w || ry >= P.rec_h) return;
    const size_t pix = ((size_t)j * P.rec_h + ry) * P.rec_w + rx;
    P.rec_light[3 * pix + 0] = p.lr;
    P.rec_light[3 * pix + 1] = p.lg;
    P.rec_light[3 * pix + 2] = p.lb;
}

// shader.comp:126-132 for the last frame of a dispatch.
__device__ __forceinline__ void store_pixel(const RtParams& P, uint32_t px, uint32_t py, float4 acc)
{
    const size_t at = (size_t)py * P.width + px;
    P.accum[at] = acc;
    const uint32_t last = P.frame0 + (uint32_t)((P.n_frames - 1) * P.frame_stride);
    const float div = (float)(int)(last + 1u);
    P.color[at] = unorm8(acc.x / div) | (unorm8(acc.y / div) << 8) | (unorm8(acc.z / div) << 16) |
                  0xff000000u;
}

__device__ __forceinline__ unsigned mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// One wave instruction per wave adds its rays (word 0 of a slot) and its ray-sphere tests, rays x
// the sphere count of this dispatch (word 2), into one of RT_RAY_SLOTS slots a cache line apart
// (the host adds them up): the waves of a launch finish together, and on a single address their
// additions would complete one after the other.
// nlist: those of the rays that were traced against a tile's candidate list instead of the whole
// sphere list (word 4 of the slot; not in the builds that stamp phase times into words 3..7).
__device__ __forceinline__ unsigned long long add_rays(const RtParams& P, unsigned nrays, unsigned slot, unsigned nlist = 0)
{
    unsigned long long v = nrays, w = nlist;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off, 64);
        w += __shfl_down(w, off, 64);
    }
    v = __shfl(v, 0, 64);
    w = __shfl(w, 0, 64);
    const unsigned lane = threadIdx.x & 63;
    if ((lane == 0 || lane == 2) && v != 0)
        atomicAdd(P.rays + (slot % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + lane,
                  lane == 0 ? v : v * (unsigned long long)P.sphere_count);
#ifndef RT_PHASE_TIMERS
    if (lane == 4 && w != 0) atomicAdd(P.rays + (slot % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + 4, w);
#endif
    return v;  // the wave's rays
}

// ---------------------------------------------------------------- tile kernel

// One wave per 8x8 tile (the shader's local_size, shader.comp:2), one lane per pixel, the
// sphere loop as written.  grid = (tile_cols, tile_rows).
template <bool FUSED, bool RECORD>
__global__ void __launch_bounds__(64) rt_tile_kernel(const RtParams P)
{
    const uint32_t px = (uint32_t)(P.tile_col0 + (int)blockIdx.x) * 8 + (threadIdx.x & 7);
    const uint32_t py = (uint32_t)(P.tile_row0 + (int)blockIdx.y) * 8 + (threadIdx.x >> 3);
    const size_t at = (size_t)py * P.width + px;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!RECORD)
        if (!clear_at(P, 0, P.frame0)) acc = P.accum[at];
    unsigned nrays = 0;
    float d0x, d0y, d0z;
    primary_direction<FUSED>(P, px, py, d0x, d0y, d0z);

    for (int j = 0; j < P.n_frames; j++) {
        const uint32_t frame = P.frame0 + (uint32_t)(j * P.frame_stride);
        if (j > 0 && clear_at(P, j, frame)) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        rt_record* rec = nullptr;
        if constexpr (RECORD) rec = record_slot(P, px, py, j);
        Path p;
        path_reset(p, P, d0x, d0y, d0z);
        bool alive = true;
        // per-wavefront bounce termination: the wave leaves when its last path has ended
        while (__ballot(alive) != 0ull) {
            const float a = dot3<FUSED>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
            float best = RT_FLT_MAX;
            int idx = -1;
            for (int i = 0; i < P.sphere_count; i++) {
                const float4 g = *reinterpret_cast<const float4*>(P.spheres + i);
                const bool far_root = P.materials != nullptr && P.materials[i].x == 3.0f;
                test_sphere<FUSED>(p, a, g.x, g.y, g.z, g.w, i, best, idx, far_root);
            }
            if (alive) {
                nrays++;
                if (path_bounce<FUSED, RECORD>(p, P, idx, best, px, py, frame, rec, reinterpret_cast<const float4*>(P.spheres))) alive = false;
            }
        }
        if constexpr (RECORD) record_light(P, px, py, j, p);
        acc.x = p.lr + acc.x;  // vec4(light, 1.0) + accumulated
        acc.y = p.lg + acc.y;
        acc.z = p.lb + acc.z;
        acc.w = 1.0f + acc.w;
    }
    if constexpr (!RECORD) {
        store_pixel(P, px, py, acc);
        add_rays(P, nrays, blockIdx.x + blockIdx.y);
    }
}

// -------------------------------------------------------------- stream kernel

// Rays in flight in a wave of rt_stream_kernel at or below which it traces them one at a time with the
// spheres dealt to its lanes (brute-force modes with the geometry in LDS).
#ifndef RT_STREAM_TEAM_MAX
#define RT_STREAM_TEAM_MAX 4
#endif

// Per-ray constants of the sieve.
struct SieveRay {
    v2f nnx, nny, nnz;     // -d / |d|
    v2f d0;                // (d / |d|) . o
    v2f p2ox, p2oy, p2oz;  // 2 o
    v2f no2;               // -(o . o) * (1 - RT_SIEVE_MARGIN)
};

__device__ __forceinline__ SieveRay sieve_ray(const Path& p, float a, bool alive, float margin = RT_SIEVE_MARGIN)
{
    const float inv = __builtin_amdgcn_rsqf(a);  // 1 ulp is plenty: the margin absorbs it
    const float nx = p.dx * inv, ny = p.dy * inv, nz = p.dz * inv;
    const float d0 = __builtin_fmaf(nz, p.oz, __builtin_fmaf(ny, p.oy, nx * p.ox));
    const float o2 = __builtin_fmaf(p.oz, p.oz, __builtin_fmaf(p.oy, p.oy, p.ox * p.ox));
    // an idle lane gets -inf here, which makes every sphere's g = -inf: rejected
    const float no2 = alive ? o2 * (margin - 1.0f) : -__builtin_inff();
    SieveRay r;
    r.nnx = v2f{-nx, -nx}; r.nny = v2f{-ny, -ny}; r.nnz = v2f{-nz, -nz};
    r.d0 = v2f{d0, d0};
    r.p2ox = v2f{2.0f * p.ox, 2.0f * p.ox};
    r.p2oy = v2f{2.0f * p.oy, 2.0f * p.oy};
    r.p2oz = v2f{2.0f * p.oz, 2.0f * p.oz};
    r.no2 = v2f{no2, no2};
    return r;
}

// Sieve of one chunk (32 spheres = 16 pairs).  g points at the chunk's 32 float4s in LDS: pair k
// is {cx0,cx1,cy0,cy1}, {cz0,cz1,w0,w1} with w = r*r - c.c + margin*(c.c + r*r), rounded up.
//
// With n = d/|d| the shader's discriminant is disc = 4|d|^2 (s^2 - C), s = n.(o-c),
// C = |o-c|^2 - r^2.  The sieve evaluates g = s^2 - C + m in the expanded form
//     s = n.o - n.c,   -C + m = 2 o.c - o.o(1-margin) + w
// with fused multiply-adds (8 packed instructions per sphere pair) and rejects a sphere only
// when g < 0.  m = margin*(o.o + c.c + r*r), margin = 2^-16 = 256 ulp, exceeds the rounding
// error of this form plus that of the shader's own expression (each below 64 ulp of
// |d|^2 ((|o|+|c|)^2 + r^2)), so a sphere whose literal discriminant is >= 0 is never
// rejected; a few near-tangent spheres pass needlessly and are dismissed by the literal test of
// the walk.  Returns a mask whose bit (31-n) is SET when sphere n of the chunk is rejected: the
// sign bit of g is shifted in from the right, sphere 0 first.
__device__ __forceinline__ void sieve_pair(uint32_t& mask, v4f A, v4f B, const SieveRay& r)
{
    const v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, w = {B.z, B.w};
    const v2f s = __builtin_elementwise_fma(
        cz, r.nnz, __builtin_elementwise_fma(cy, r.nny, __builtin_elementwise_fma(cx, r.nnx, r.d0)));
    const v2f t = __builtin_elementwise_fma(
        cz, r.p2oz, __builtin_elementwise_fma(cy, r.p2oy, __builtin_elementwise_fma(cx, r.p2ox, r.no2 + w)));
    const v2f q = __builtin_elementwise_fma(s, s, t);
    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x), 31);
    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y), 31);
}

// The loads of group n+1 (RT_SIEVE_GROUP pairs) are issued before the arithmetic of group n,
// so their latency is covered by a group's worth of packed math.
template <int RT_SIEVE_GROUP, class GeomPtr>
__device__ __forceinline__ uint32_t sieve_chunk(GeomPtr g, const SieveRay& r)
{
    constexpr int NG = RT_CHUNK / 2 / RT_SIEVE_GROUP;
    uint32_t mask = 0;
    v4f cur[2 * RT_SIEVE_GROUP], nxt[2 * RT_SIEVE_GROUP];
#pragma unroll
    for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) cur[k] = g[k];
#pragma unroll
    for (int n = 0; n < NG; n++) {
        if (n + 1 < NG) {
#pragma unroll
            for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) nxt[k] = g[(n + 1) * 2 * RT_SIEVE_GROUP + k];
        }
#ifdef RT_SIEVE_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k = 0; k < RT_SIEVE_GROUP; k++) sieve_pair(mask, cur[2 * k], cur[2 * k + 1], r);
#ifdef RT_SIEVE_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) cur[k] = nxt[k];
    }
    return mask;
}

// The scene's last chunk when it holds fewer than 32 spheres: only the first `ngroups` groups of
// RT_SIEVE_GROUP pairs are evaluated (no prefetch: one chunk per trace does not pay for it).  The
// bits of the spheres left out come back as 0, "not rejected"; the caller masks them.
template <int RT_SIEVE_GROUP, class GeomPtr>
__device__ __forceinline__ uint32_t sieve_chunk_partial(GeomPtr g, const SieveRay& r, int ngroups)
{
    uint32_t mask = 0;
#pragma unroll 1
    for (int n = 0; n < ngroups; n++) {
        v4f cur[2 * RT_SIEVE_GROUP];
#pragma unroll
        for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) cur[k] = g[n * 2 * RT_SIEVE_GROUP + k];
#pragma unroll
        for (int k = 0; k < RT_SIEVE_GROUP; k++) sieve_pair(mask, cur[2 * k], cur[2 * k + 1], r);
    }
    return mask << (RT_CHUNK - 2 * RT_SIEVE_GROUP * ngroups);
}

// As sieve_chunk_partial, software-pipelined by hand over two register buffers: the loads of a
// group are issued before the arithmetic of the group in front of it, in a loop the compiler
// does not unroll (unrolled, it issues all loads of a chunk first and runs out of registers).
// Groups are evaluated two at a time: with an odd `ngroups` one group beyond is evaluated too
// (the caller masks its bits; the memory behind it must be readable).
template <int RT_SIEVE_GROUP, class GeomPtr>
__device__ __forceinline__ uint32_t sieve_chunk_piped(GeomPtr g, const SieveRay& r, int ngroups)
{
    uint32_t mask = 0;
    v4f a[2 * RT_SIEVE_GROUP], b[2 * RT_SIEVE_GROUP];
#pragma unroll
    for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) a[k] = g[k];
    int n = 0;
#pragma unroll 1
    for (; n < ngroups; n += 2) {
#pragma unroll
        for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) b[k] = g[(n + 1) * 2 * RT_SIEVE_GROUP + k];
#pragma unroll
        for (int k = 0; k < RT_SIEVE_GROUP; k++) sieve_pair(mask, a[2 * k], a[2 * k + 1], r);
#pragma unroll
        for (int k = 0; k < 2 * RT_SIEVE_GROUP; k++) a[k] = g[(n + 2) * 2 * RT_SIEVE_GROUP + k];
#pragma unroll
        for (int k = 0; k < RT_SIEVE_GROUP; k++) sieve_pair(mask, b[2 * k], b[2 * k + 1], r);
    }
    return n ? mask << (RT_CHUNK - 2 * RT_SIEVE_GROUP * n) : 0u;
}

// Culled kernel: a chunk of which only the groups of RT_CULL_GROUP spheres named in `need`
// (bit 3 = first group) are evaluated; the spheres of the other groups come back as rejected.
template <class GeomPtr>
__device__ __forceinline__ uint32_t sieve_chunk_culled(GeomPtr g, const SieveRay& r, uint32_t need)
{
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < RT_CHUNK / RT_CULL_GROUP; j++) {
        if (need & (8u >> j)) {  // wave-uniform
            v4f cur[RT_CULL_GROUP];
#pragma unroll
            for (int k = 0; k < RT_CULL_GROUP; k++) cur[k] = g[j * RT_CULL_GROUP + k];
#pragma unroll
            for (int k = 0; k < RT_CULL_GROUP / 2; k++) sieve_pair(mask, cur[2 * k], cur[2 * k + 1], r);
        } else {
            mask = (mask << RT_CULL_GROUP) | ((1u << RT_CULL_GROUP) - 1u);
        }
    }
    return mask;
}

typedef const v4f __attribute__((address_space(4)))* ConstGeomPtr;
#ifndef RT_GROUP_SGPR
#define RT_GROUP_SGPR 2
#endif
#ifndef RT_GROUP_LDS
#define RT_GROUP_LDS 4
#endif

// Minimum of v over the 64 lanes of a wave whose lanes are all active, as a wave-uniform value:
// two quad permutations and the two row mirrors leave every row of 16 lanes with its minimum (DPP,
// no LDS traffic), and the four rows meet on the scalar side.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    return ~wave_min_u32(~v);
}

__device__ __forceinline__ float lane_value(float v, int from)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from));
}

// Paths still running in a wave at or below which rt_lite_kernel traces them one at a time with
// the whole wave (below).
#ifndef RT_LITE_TEAM_MAX
#define RT_LITE_TEAM_MAX 4
#endif

// ---------------------------------------------------------------- lite kernel

// Scenes of at most 64 traced spheres (the reference's default scene traces 40).  A pass over so
// few spheres is a few hundred instructions, and then what the persistent kernel spends per
// pixel on its work queues and on refilling lanes costs more than the tracing: measured, its
// loop takes 115 us for one 1080p frame of a scene with NO spheres, the plain tile mapping 18 us.
// So here the mapping is the shader's own -- one wave per 8x8 tile, handed out by the hardware
// dispatcher, no queues -- with the two things that matter kept from the stream kernel:
//   - the sieve (packed-f32) and the candidate walk replace the literal loop over the spheres;
//     the candidate masks of <= 2 chunks stay in registers;
//   - the first pass of every frame is the camera rays of the tile's 64 pixels: one origin,
//     directions within half a degree of one another.  Lane i tests sphere i against the cone that
//     holds them all (tile_cone_keeps, once per tile), the ballot is the tile's candidate list, and
//     every lane runs the literal test on just those, in index order and without divergence -- in
//     place of 40 sieve tests and a walk as long as the busiest lane's.  Three rays in four are
//     camera rays, so the lanes of a wave start every frame TOGETHER: a lane whose path has ended
//     waits for the slowest path of the tile.  (Letting every lane go on to its pixel's next frame
//     at once keeps the lanes busier but mixes camera rays with bounce rays in every pass, and
//     then every pass pays for the sieve: measured 85 against 100 Grays/s at 1080p x 64 frames.)
//   - when only a few lanes are left with a path (RT_LITE_TEAM_MAX; the mirror chains of up to
//     max_depth bounces that a frame ends with), a pass is no longer sieve + walk per lane -- some
//     400 dependent instructions for a lone ray -- but the literal test of every sphere at once:
//     lane i holds sphere i, the ray is broadcast, every lane evaluates shader.comp:44-60 for its
//     sphere and the wave takes the minimum t, lowest index among equal ones -- what the shader's
//     strict `<` in index order keeps.
// The workgroup first copies the sphere records and the sieve geometry into LDS and reads them
// from there, the sieve in a loop over groups of sphere pairs that is not unrolled (unrolled,
// the compiler keeps the whole geometry in registers).  Through the scalar cache, as rt_stream_kernel reads the geometry, this kernel
// has too few scalar registers: about half of the geometry was spilled into vector registers
// lane by lane, and even a dispatch of one frame, where a wave makes one or two passes, came out
// slower (79 against 70 us for 1080p).
// blockDim.x = a multiple of 64 (one tile per wave); grid = ceil(tiles / waves per workgroup);
// dynamic LDS = rt_lite_lds_bytes().
// Can a ray from o whose unit direction lies within the cone (axis ax, ay, az of length 1, half
// angle theta < 90 degrees given by its sine and cosine) meet the sphere (c, r) as far as the
// shader's floating-point test can tell?  false only when surely not.
// The centre's distance from the cone's surface is at least p cos(theta) - z sin(theta), z = L.a the
// depth along the axis and p = |L x a| the distance from it (L = c - o): the distance from a line
// through the apex, which the apex itself is not nearer than.  The sphere is kept unless that exceeds
// sqrt(r^2 + 2^-14 (o.o + c.c + r^2)) -- four times the sieve's margin, which already exceeds the
// rounding of the shader's discriminant (sieve_pair) -- plus 10^-4 (|z| + p + r) for the rounding
// here (a few ulp each).  Any NaN on the way compares false: kept.
__device__ __forceinline__ bool tile_cone_keeps(float ox, float oy, float oz, float ax, float ay, float az,
                                                float sin_t, float cos_t, float4 sphere)
{
    const float lx = sphere.x - ox, ly = sphere.y - oy, lz = sphere.z - oz;
    const float z = __builtin_fmaf(lz, az, __builtin_fmaf(ly, ay, lx * ax));
    const float qx = __builtin_fmaf(ly, az, -(lz * ay)), qy = __builtin_fmaf(lz, ax, -(lx * az)),
                qz = __builtin_fmaf(lx, ay, -(ly * ax));
    const float pd = __builtin_sqrtf(__builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx)));
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float cc = __builtin_fmaf(sphere.z, sphere.z, __builtin_fmaf(sphere.y, sphere.y, sphere.x * sphere.x));
    const float rr = sphere.w * sphere.w;
    const float reach = __builtin_sqrtf(__builtin_fmaf(6.103515625e-5f, oo + cc + rr, rr)) +
                        1e-4f * (__builtin_fabsf(z) + pd + __builtin_fabsf(sphere.w));
    const float away = __builtin_fmaf(pd, cos_t, -(z * sin_t));
    return !(away > reach);
}

// MANY = false: the instantiation for dispatches of one frame (the interactive loop), without the
// code for going from frame to frame and for runs of frames (measured: 48 against 52 us per 1080p frame).
// (one frame per dispatch is bound by latency, not by issue: its instantiation is held to 80 registers, six waves
// per SIMD instead of five -- 1920x1080 x 1 frame 47.5 -> 45.4 us; the many-frame one would spill at that and is
// held to 96 registers, five waves)
template <bool FUSED, bool MANY>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(MANY ? 4 : 6, MANY ? 5 : 6))) rt_lite_kernel(const RtParams P)
{
    extern __shared__ float4 smem[];
    float4* lgeom = smem + 2 * 2 * RT_CHUNK;
    // A wave of a one-frame dispatch lives a few microseconds, most of them waiting for memory: the loads of the
    // workgroup's copy of the scene are issued first, then the wave's tile (through the scalar cache: the
    // slot is wave-uniform) and its texel, and only then is anything waited for.
    const float4* scene4 = reinterpret_cast<const float4*>(P.spheres);
    const bool stage0 = (int)threadIdx.x < 2 * P.sphere_count, stage1 = (int)threadIdx.x < P.n_pad;
    float4 staged0 = make_float4(0.f, 0.f, 0.f, 0.f), staged1 = staged0;
    if (stage0) staged0 = scene4[threadIdx.x];
    if (stage1) staged1 = P.geom[threadIdx.x];
    uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // The first split_tiles tiles of the order, the heaviest, go to split_ways waves each (below): wave
    // `run` of a tile traces the frames [run, run + 1) * split_frames of the dispatch.  run < 0: all frames.
    int run = -1;
    const uint32_t ways = MANY ? (uint32_t)P.split_ways : 0u;
    if constexpr (MANY) {
        const uint32_t split_slots = (uint32_t)P.split_tiles * ways;
        if (slot < split_slots) {
            run = (int)(slot % ways);
            slot /= ways;
        } else {
            slot -= split_slots - (uint32_t)P.split_tiles;
        }
    }
    // (a wave beyond the last tile stays for the barrier of the runs' hand-over through LDS and does nothing else)
    const bool valid = slot < (uint32_t)P.tile_cols * (uint32_t)P.tile_rows;  // wave-uniform
    if (!valid) {
        slot = 0;
        run = -1;
    }
    // heaviest tiles first, when the previous dispatch left their costs (rt_tile_order_kernel)
    const uint32_t tile = P.tile_order ? P.tile_order[slot] : slot;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ty = P.tile_cols_magic ? __umulhi(tile, P.tile_cols_magic) : tile;
    const uint32_t tx = tile - ty * (uint32_t)P.tile_cols;
    const uint32_t px = ((uint32_t)P.tile_col0 + tx) * 8 + (lane & 7u);
    const uint32_t py = ((uint32_t)P.tile_row0 + ty) * 8 + (lane >> 3);
    const size_t at = (size_t)py * P.width + px;
    const int j_begin = MANY && run > 0 ? run * P.split_frames : 0;
    const int j_end = !MANY ? 1 : run >= 0 ? min(P.n_frames, j_begin + P.split_frames) : P.n_frames;
    // runs after the first leave their lights here, a frame after the other, for rt_lite_join_kernel
    // ... in global memory, or, where the runs of a tile are waves of one workgroup and their lights fit, in LDS:
    // per tile of the workgroup (ways - 1) * split_frames entries of three planes of 64 floats
    const bool lds_handover = MANY && P.split_lds != 0;  // wave-uniform
    float4* lights = MANY && run > 0 && !lds_handover
                         ? P.split_lights + ((size_t)slot * (ways - 1u) + (size_t)(run - 1)) * (size_t)P.split_frames * 64u + lane
                         : nullptr;
    float* lds_lights = nullptr;  // this tile's entries, at this lane
    if (lds_handover && run >= 0)
        lds_lights = reinterpret_cast<float*>(lgeom + 2 * RT_CHUNK) +
                     (size_t)((threadIdx.x >> 6) / (MANY ? ways : 1u)) * (ways - 1u) * (size_t)P.split_frames * 192u + lane;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (run <= 0 && !clear_at(P, 0, P.frame0)) acc = P.accum[at];
    float d0x, d0y, d0z;
    primary_direction<FUSED>(P, px, py, d0x, d0y, d0z);

    // the workgroup's copy of the sphere records and of the sieve geometry
    if (stage0) smem[threadIdx.x] = staged0;
    if (stage1) lgeom[threadIdx.x] = staged1;
    for (int i = threadIdx.x + blockDim.x; i < 2 * P.sphere_count; i += blockDim.x) smem[i] = scene4[i];
    for (int i = threadIdx.x + blockDim.x; i < P.n_pad; i += blockDim.x) lgeom[i] = P.geom[i];
    __syncthreads();
    if (!valid && (!MANY || P.split_lds == 0)) return;
    const bool two = P.n_pad > RT_CHUNK;                   // one or two chunks of 32 spheres
    const uint32_t valid0 = two ? 0xffffffffu : P.n_pad ? P.last_chunk_valid : 0u;  // empty scene: no candidates
    const uint32_t valid1 = two ? P.last_chunk_valid : 0u;
    constexpr int PGROUP = 2;                              // the ragged last chunk goes 2 pairs at a time
    const int last_groups = (P.last_chunk_count + 2 * PGROUP - 1) / (2 * PGROUP);
    const int groups0 = two ? RT_CHUNK / 2 / PGROUP : last_groups;
    const int groups1 = two ? last_groups : 0;

    // the lane's own sphere, for the passes that the whole wave spends on one ray
    const bool own_valid = (int)lane < P.sphere_count;
    const float4 own = smem[2 * lane];
    const bool own_far = own_valid && P.materials != nullptr && P.materials[lane].x == 3.0f;

    int j = j_begin;
    uint32_t frame = P.frame0 + (uint32_t)(j * P.frame_stride);
    unsigned nrays = 0;
    Path p;
    path_reset(p, P, d0x, d0y, d0z);
    bool alive = valid && j < j_end;

    // the spheres that the camera rays of this tile can meet: the cone around the ray of pixel (3, 3)
    // that holds the 64 directions as they were computed (sine of the largest angle, a little more)
    unsigned long long tile_spheres;
    {
        const float ax = lane_value(p.dx, 27), ay = lane_value(p.dy, 27), az = lane_value(p.dz, 27);
        const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
        const float ux = ax * inv, uy = ay * inv, uz = az * inv;                     // the axis, length 1
        const float dn = __builtin_amdgcn_rsqf(__builtin_fmaf(p.dz, p.dz, __builtin_fmaf(p.dy, p.dy, p.dx * p.dx)));
        const float cx = __builtin_fmaf(p.dy, uz, -(p.dz * uy)), cy = __builtin_fmaf(p.dz, ux, -(p.dx * uz)),
                    cz = __builtin_fmaf(p.dx, uy, -(p.dy * ux));
        const float sine = __builtin_sqrtf(__builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx))) * dn;
        const float along = __builtin_fmaf(p.dz, uz, __builtin_fmaf(p.dy, uy, p.dx * ux));
        // |sine| as bits: NaN sorts above every number, so a direction that is not a number ends as "keep all"
        const float sin_t = __builtin_fmaf(__uint_as_float(wave_max_u32(__float_as_uint(__builtin_fabsf(sine)))), 1.0001f, 1e-6f);
        const float cos_t = __builtin_sqrtf(__builtin_fmaf(-sin_t, sin_t, 1.0f));   // NaN beyond 90 degrees
        const bool narrow = __ballot(!(along > 0.0f)) == 0ull && sin_t < 0.5f;        // wave-uniform
        const bool keep = !narrow || tile_cone_keeps(p.ox, p.oy, p.oz, ux, uy, uz, sin_t, cos_t, own);
        tile_spheres = __ballot(keep && own_valid);
    }
    const bool bundle = P.cam_reuse != 0;  // off: camera rays go through the sieve like every other ray
    bool camera_pass = bundle;  // every live lane is on the camera ray of its pixel: the first pass of every frame

    bool have_cam = false;  // wave-uniform: cam_idx, cam_t hold the hit of the lane's camera ray
    int cam_idx = -1;
    float cam_t = RT_FLT_MAX;
    // The end of a path: its light goes to the texel (or, in a run after a tile's first, to where the first run
    // will find it), and the lane goes on to its next frame or is done.
#define RT_LITE_END_PATH                                                                                        \
    {                                                                                                           \
        if (MANY && lights != nullptr) {                                                                        \
            lights[(size_t)(j - j_begin) * 64u] = make_float4(p.lr, p.lg, p.lb, 0.f);                            \
        } else if (MANY && lds_handover && run > 0) {                                                           \
            float* at_l = lds_lights + (size_t)(j - P.split_frames) * 192u; /* run r, frame k: (r - 1) * split_frames + k */ \
            at_l[0] = p.lr;                                                                                     \
            at_l[64] = p.lg;                                                                                    \
            at_l[128] = p.lb;                                                                                   \
        } else {                                                                                                \
            acc.x = p.lr + acc.x; /* vec4(light, 1.0) + accumulated */                                          \
            acc.y = p.lg + acc.y;                                                                               \
            acc.z = p.lb + acc.z;                                                                               \
            acc.w = 1.0f + acc.w;                                                                               \
        }                                                                                                       \
        j++;                                                                                                    \
        if (!MANY || j == j_end) {                                                                              \
            alive = false;                                                                                      \
        } else {                                                                                                \
            frame = P.frame0 + (uint32_t)(j * P.frame_stride);                                                  \
            if (clear_at(P, j, frame)) acc = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
            path_reset(p, P, d0x, d0y, d0z);                                                                    \
        }                                                                                                       \
    }
#ifdef RT_COUNT_PASSES
    unsigned lite_passes = 0, lite_pass_lanes = 0;  // experiment: passes of this wave, lanes with a ray in them
#endif
    for (;;) {
        // ---- a lane at the start of a frame whose camera hit is known shades it and joins the others with its
        // first bounce ray: after the wave's first pass no lane waits for another, each runs through its frames at
        // its own pace (a pass of sieve and walk costs the same whatever the number of rays in it, and per frame
        // the slowest path of 64 is several times the average one).  A pixel that sees the sky is finished at
        // once, all its frames.  (More than one turn of the inner loop only at depth 1.)
        if constexpr (MANY) {
            if (have_cam) {  // wave-uniform
                while (__ballot(alive && p.bounce == 0) != 0ull) {
                    if (alive && p.bounce == 0) {
                        if (cam_idx < 0) {
                            (void)path_bounce<FUSED, false>(p, P, -1, RT_FLT_MAX, px, py, frame, nullptr, smem);
                            const float sr = p.lr, sg = p.lg, sb = p.lb;
                            nrays += (unsigned)(j_end - j);
                            while (alive) {
                                p.lr = sr; p.lg = sg; p.lb = sb;
                                RT_LITE_END_PATH
                            }
                        } else {
                            nrays++;
                            if (path_bounce<FUSED, false>(p, P, cam_idx, cam_t, px, py, frame, nullptr, smem)) RT_LITE_END_PATH
                        }
                    }
                }
            }
        }
        const unsigned long long alive_mask = __ballot(alive);
        if (alive_mask == 0ull) break;
#ifdef RT_COUNT_PASSES
        lite_passes++;
        lite_pass_lanes += (unsigned)__popcll(alive_mask);
#endif
        const float a = dot3<FUSED>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
        float best = RT_FLT_MAX;
        int idx = -1;
        if (camera_pass) {  // wave-uniform
            // ---- Trace() of the camera rays: the literal test of the tile's candidates, lowest index first --
            // in the wave's first frame; a pixel's camera ray is the same ray in every frame, and so is its hit
            camera_pass = false;
            for (unsigned long long todo = tile_spheres; todo != 0ull; todo &= todo - 1ull) {
                const int i = (int)__builtin_ctzll(todo);
                bool far_root = false;
                if (P.materials != nullptr) far_root = P.materials[i].x == 3.0f;
                const float4 g = smem[2 * i];
                if (alive) test_sphere<FUSED>(p, a, g.x, g.y, g.z, g.w, i, best, idx, far_root);
            }
            cam_t = best;
            cam_idx = idx;
            have_cam = true;
        } else if (__popcll(alive_mask) <= RT_LITE_TEAM_MAX) {  // wave-uniform
            // ---- Trace() of the few rays left, one after the other, a sphere per lane
            for (unsigned long long todo = alive_mask; todo != 0ull; todo &= todo - 1ull) {
                const int from = (int)__builtin_ctzll(todo);
                Path q;
                q.ox = lane_value(p.ox, from); q.oy = lane_value(p.oy, from); q.oz = lane_value(p.oz, from);
                q.dx = lane_value(p.dx, from); q.dy = lane_value(p.dy, from); q.dz = lane_value(p.dz, from);
                float t = RT_FLT_MAX;  // stays there unless this lane's sphere is hit at 0 < t < FLT_MAX
                int hit = -1;
                if (own_valid)
                    test_sphere<FUSED>(q, lane_value(a, from), own.x, own.y, own.z, own.w, (int)lane, t, hit, own_far);
                const uint32_t nearest = wave_min_u32(__float_as_uint(t));  // t > 0: ordered as its bits
                const unsigned long long at_nearest = __ballot(__float_as_uint(t) == nearest);
                if ((int)lane == from && nearest != __float_as_uint(RT_FLT_MAX)) {
                    best = __uint_as_float(nearest);
                    idx = (int)__builtin_ctzll(at_nearest);
                }
            }
        } else {
            // ---- Trace(): sieve + candidate walk (rt_stream_kernel has the commentary)
            const SieveRay sr = sieve_ray(p, a, alive);
            // read from LDS anew in every pass: the offset is hidden from the optimiser, which would
            // otherwise keep the whole geometry in registers across passes
            unsigned goff = 0;
            asm volatile("" : "+v"(goff));
            const v4f* geom = reinterpret_cast<const v4f*>(lgeom) + goff;
            const uint32_t cand0 = ~sieve_chunk_partial<PGROUP>(geom, sr, groups0) & valid0;
            uint32_t cand1 = 0u;
            if (two) cand1 = ~sieve_chunk_partial<PGROUP>(geom + RT_CHUNK, sr, groups1) & valid1;
            uint32_t cur = cand0;
            int base = 0;
            for (;;) {
                if (cur == 0u && base == 0) {
                    cur = cand1;
                    base = RT_CHUNK;
                }
                if (__ballot(cur != 0u) == 0ull) break;
                if (cur != 0u) {
                    const int b = __clz(cur);
                    cur &= ~(0x80000000u >> b);
                    const int i = base + b;
                    bool far_root = false;
                    if (P.materials != nullptr) far_root = P.materials[i].x == 3.0f;  // wave-uniform branch
                    const float4 g = smem[2 * i];
                    test_sphere<FUSED>(p, a, g.x, g.y, g.z, g.w, i, best, idx, far_root);
                }
            }
        }
        // ---- shade; a lane whose path has ended goes on to its next frame
        if (alive) {
            nrays++;
            if (path_bounce<FUSED, false>(p, P, idx, best, px, py, frame, nullptr, smem)) RT_LITE_END_PATH
        }
    }
#undef RT_LITE_END_PATH
    if (lds_handover) {
        // the runs of a tile meet here: the first one adds the others' lights in frame order, as one wave tracing
        // all frames would have added them (float addition does not associate), and stores the pixel
        __syncthreads();
        if (run == 0) {
            for (int jj = P.split_frames; jj < P.n_frames; jj++) {
                if (clear_at(P, jj, P.frame0 + (uint32_t)(jj * P.frame_stride))) acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* at_l = lds_lights + (size_t)(jj - P.split_frames) * 192u;
                acc.x = at_l[0] + acc.x;  // vec4(light, 1.0) + accumulated
                acc.y = at_l[64] + acc.y;
                acc.z = at_l[128] + acc.z;
                acc.w = 1.0f + acc.w;
            }
        }
        if (valid && run <= 0) store_pixel(P, px, py, acc);
    } else {
        if (run < 0) store_pixel(P, px, py, acc);
        if (run == 0) P.accum[at] = acc;  // the sum so far; rt_lite_join_kernel adds the other runs' lights and resolves
    }
    if (!valid) return;
    // (every camera ray of this kernel is traced against its tile's list: the host counts them, a path each)
    (void)add_rays(P, nrays, tile);
#ifdef RT_COUNT_PASSES
    if (lane == 0) atomicAdd(P.rays + (tile % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + 1, (unsigned long long)lite_passes);
    if (lane == 1) atomicAdd(P.rays + (tile % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + 3, (unsigned long long)lite_pass_lanes);
#endif
    // The tile's cost for the next order: the rays of its longest lane -- the lanes of a wave go through their frames
    // independently, so the wave lasts as long as the pixel that bounces most (by the sum of the lanes' rays, which is
    // what the order went by before: 1280x720 x 64 frames 0.635 -> 0.522 ms, 1920x1080 x 64 frames 1.148 -> 1.065 ms).
    // (A split tile's: its first run's times the runs, near enough for an order.)
    const uint32_t longest = wave_max_u32(nrays);
    if (P.tile_cost != nullptr && lane == 0 && run <= 0)
        P.tile_cost[tile] = (uint32_t)min((unsigned long long)longest * (run == 0 ? ways : 1u), 0xffffffffull);
}

// The pixels of the tiles that rt_lite_kernel traced in split_ways runs of frames: the first run left
// its sum in the accumulation image, the others the light of every frame; added here in frame order, as
// one wave tracing all frames would have added them (float addition does not associate), then resolved.
// Why split at all: a pixel between two mirrors bounces to the depth limit in every frame, and all its
// frames in one lane are n_frames x max_depth passes one after the other -- 1.2 ms of a 1.45 ms dispatch
// at 1280x720 x 64 frames, whatever else the chip does meanwhile.
// One wave per split tile; blockDim.x = a multiple of 64.
__global__ void __launch_bounds__(256) rt_lite_join_kernel(const RtParams P)
{
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // (stream kernel: the number of split tiles is the order kernel's, at most split_tiles)
    const uint32_t count = P.split_count != nullptr ? min(*P.split_count, (uint32_t)P.split_tiles) : (uint32_t)P.split_tiles;
    if (slot >= count) return;
    const uint32_t tile = P.tile_order[slot];
    const uint32_t lane = threadIdx.x & 63u;
    if (P.cam_live != nullptr && ((P.cam_live[tile] >> lane) & 1ull) == 0ull) return;  // a sky pixel: no run has traced it
    const uint32_t ty = P.tile_cols_magic ? __umulhi(tile, P.tile_cols_magic) : tile;
    const uint32_t tx = tile - ty * (uint32_t)P.tile_cols;
    const uint32_t px = ((uint32_t)P.tile_col0 + tx) * 8 + (lane & 7u);
    const uint32_t py = ((uint32_t)P.tile_row0 + ty) * 8 + (lane >> 3);
    float4 acc = P.accum[(size_t)py * P.width + px];
    // lite kernel: run r, frame k of it at (r - 1) * split_frames + k of the tile's lights; stream kernel: frame j at j
    // (the first split_frames places stay empty)
    const float4* lights = P.split_lights + lane +
                           (P.split_count != nullptr
                                ? (ptrdiff_t)slot * P.split_ways * (ptrdiff_t)P.split_frames * 64
                                : ((ptrdiff_t)slot * (P.split_ways - 1) - 1) * (ptrdiff_t)P.split_frames * 64);
    for (int j = P.split_frames; j < P.n_frames; j++) {
        const uint32_t frame = P.frame0 + (uint32_t)(j * P.frame_stride);
        if (clear_at(P, j, frame)) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 l = lights[(size_t)j * 64u];
        acc.x = l.x + acc.x;  // vec4(light, 1.0) + accumulated
        acc.y = l.y + acc.y;
        acc.z = l.z + acc.z;
        acc.w = 1.0f + acc.w;
    }
    store_pixel(P, px, py, acc);
}

// Order of the tiles for the NEXT dispatch of rt_lite_kernel: descending cost (rays of the tile in
// the dispatch before), by a counting sort over 256 cost classes.  A tile with all its frames is
// the unit the hardware dispatcher hands out, between 64 and several hundred passes long; handed
// out in image order, the heaviest tiles start whenever their turn comes and the dispatch ends
// waiting for them (measured: 25 % of the vector ALU idle at 1080p x 64 frames).  Longest first,
// they start at once and the light ones fill in behind.  The order only says who goes first: the
// bits of the image cannot depend on it.  One workgroup of 1024 threads; n = number of tiles.
// chain != null (stream kernel): the order is by chain[tile] instead, the most rays any pixel of the tile took in that
// dispatch: there a lane takes one pixel at a time, and what a dispatch ends waiting for is a pixel's chain of
// frames x bounces, started late.  order[n] = the number of leading tiles of that order that are HEAVY: whose
// longest pixel is more passes (a ray a pass) than 1 / heavy_den of what an average wave of `nwaves` makes in the whole
// dispatch, the sum of `cost` / (64 nwaves).  The stream kernel shares the frames of those tiles among several units
// of work (RtParams::split_*).  0 with nwaves == 0.
__global__ void __launch_bounds__(1024) rt_tile_order_kernel(const uint32_t* cost, const uint32_t* chain, uint32_t* order,
                                                             unsigned n, unsigned nwaves, unsigned heavy_den)
{
    __shared__ unsigned s_max;
    __shared__ unsigned s_bin[256];
    __shared__ unsigned long long s_sum;
    __shared__ unsigned s_heavy;
    const uint32_t* key = chain != nullptr ? chain : cost;
    if (threadIdx.x == 0) {
        s_sum = 0ull;
        s_heavy = 0u;
        s_max = 1u;
    }
    if (threadIdx.x < 256) s_bin[threadIdx.x] = 0u;
    __syncthreads();
    unsigned m = 0;
    unsigned long long sum = 0ull;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        m = max(m, key[i]);
        sum += cost[i];
    }
    atomicMax(&s_max, m);
    if (nwaves != 0u) atomicAdd(&s_sum, sum);
    __syncthreads();
    const float scale = 255.0f / (float)s_max;
    auto cls = [&](unsigned c) { return 255u - min(255u, (unsigned)((float)c * scale)); };  // class 0 = heaviest
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_bin[cls(key[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix: where each class begins
        unsigned at = 0;
        for (int b = 0; b < 256; b++) {
            const unsigned c = s_bin[b];
            s_bin[b] = at;
            at += c;
        }
    }
    __syncthreads();
    // heavy: chain passes >= (sum / 64 / nwaves) / heavy_den -- by cost class, so that the heavy tiles lead the order
    const unsigned long long heavy_from =
        nwaves != 0u && chain != nullptr ? s_sum / (64ull * nwaves * heavy_den) + 1ull : ~0ull;
    const unsigned heavy_cls = heavy_from <= (unsigned long long)s_max ? cls((unsigned)heavy_from) : 0u;
    unsigned heavy = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned c = cls(key[i]);
        order[atomicAdd(&s_bin[c], 1u)] = i;
        heavy += heavy_from <= (unsigned long long)s_max && c < heavy_cls ? 1u : 0u;
    }
    if (heavy != 0u) atomicAdd(&s_heavy, heavy);
    __syncthreads();
    if (threadIdx.x == 0) order[n] = s_heavy;
}

// Lane state `park` of the instantiations that share the frames of heavy tiles among several units of work: where the
// lights of the lane's run of frames go.  A pixel of a tile that is not split: RT_PARK_NONE (all frames in this lane,
// the texel stored at the end); the first run of a split tile: RT_PARK_FIRST (its sum goes to the accumulation image
// and rt_lite_join_kernel adds the other runs); a later run: RT_PARK_LATER + 4 x the index in RtParams::split_lights
// of the light of the pixel's frame 0, the light of frame j 64 j further on.
#define RT_PARK_LATER 0u
#define RT_PARK_FIRST 1u
#define RT_PARK_NONE 2u

// The units of work of a dispatch of the stream kernel that shares the frames of heavy tiles, in the order they are
// handed out: `ways` runs of `frames` frames (a power of two) for each of the first S tiles of the order (S = order[n],
// the heavy tiles rt_tile_order_kernel counted, at most `cap`), then the other tiles with all their frames.  units[u] =
// {tile, first frame, end frame, RT_PARK_*: for a run after a tile's first, 4 x the index in the parked lights of the light of
// lane 0's frame 0}; order[n + 1] = the number of units.  One thread per unit, n + cap * (ways - 1) at least.
__global__ void __launch_bounds__(256) rt_tile_units_kernel(uint32_t* order, uint4* units, unsigned n, unsigned cap, unsigned ways,
                                                            unsigned frames, unsigned n_frames)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t heavy = min(order[n], cap);
    const uint32_t n_units = n + heavy * (ways - 1u);
    if (u == 0u) order[n + 1] = n_units;
    if (u >= n_units) return;
    if (u < heavy * ways) {
        const uint32_t slot = u / ways, run = u - slot * ways;
        const uint32_t j0 = run * frames;
        units[u] = make_uint4(order[slot], j0, min(n_frames, j0 + frames),
                              run == 0u ? RT_PARK_FIRST : RT_PARK_LATER + ((slot * ways * frames * 64u) << 2));
    } else {
        units[u] = make_uint4(order[u - heavy * (ways - 1u)], 0u, n_frames, RT_PARK_NONE);
    }
}

// Nearest hit of the camera ray of every pixel, for rt_stream_kernel.  The camera ray of a pixel is the
// same ray in every frame (shader.comp:73-90: origin and direction depend on camera and pixel only), so
// its Trace() gives the same sphere and the same distance in every frame and every dispatch until camera,
// image, spheres or float profile change.  One wave per 8x8 tile, a lane per pixel: the cone test of
// rt_lite_kernel (tile_cone_keeps) over the whole sphere list, 64 spheres at a time with a sphere per
// lane, leaves the few spheres that any camera ray of the tile can meet; every lane then evaluates the
// shader's own test (test_sphere) on those, in index order.  hits[tile * 64 + lane] = {sphere index or
// -1, bits of t}; live[tile] = the lanes whose camera ray hits a sphere.  blockDim.x = a multiple of 64.
template <bool FUSED>
__global__ void __launch_bounds__(256) rt_camera_hits_kernel(const RtParams P, uint2* hits, unsigned long long* live)
{
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= (uint32_t)P.tile_cols * (uint32_t)P.tile_rows) return;  // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ty = P.tile_cols_magic ? __umulhi(tile, P.tile_cols_magic) : tile;
    const uint32_t tx = tile - ty * (uint32_t)P.tile_cols;
    const uint32_t px = ((uint32_t)P.tile_col0 + tx) * 8 + (lane & 7u);
    const uint32_t py = ((uint32_t)P.tile_row0 + ty) * 8 + (lane >> 3);
    float dx, dy, dz;
    primary_direction<FUSED>(P, px, py, dx, dy, dz);
    // the cone around the ray of pixel (3, 3) that holds the 64 directions as they were computed
    const float ax = lane_value(dx, 27), ay = lane_value(dy, 27), az = lane_value(dz, 27);
    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
    const float ux = ax * inv, uy = ay * inv, uz = az * inv;
    const float dn = __builtin_amdgcn_rsqf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    const float cx = __builtin_fmaf(dy, uz, -(dz * uy)), cy = __builtin_fmaf(dz, ux, -(dx * uz)),
                cz = __builtin_fmaf(dx, uy, -(dy * ux));
    const float sine = __builtin_sqrtf(__builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx))) * dn;
    const float along = __builtin_fmaf(dz, uz, __builtin_fmaf(dy, uy, dx * ux));
    const float sin_t = __builtin_fmaf(__uint_as_float(wave_max_u32(__float_as_uint(__builtin_fabsf(sine)))), 1.0001f, 1e-6f);
    const float cos_t = __builtin_sqrtf(__builtin_fmaf(-sin_t, sin_t, 1.0f));
    const bool narrow = __ballot(!(along > 0.0f)) == 0ull && sin_t < 0.5f;  // wave-uniform
    Path p;
    path_reset(p, P, dx, dy, dz);
    const float a = dot3<FUSED>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
    float best = RT_FLT_MAX;
    int idx = -1;
    for (int base = 0; base < P.sphere_count; base += 64) {
        const int i = base + (int)lane;
        bool keep = false;
        if (i < P.sphere_count) {
            const float4 sphere = *reinterpret_cast<const float4*>(P.spheres + i);
            keep = !narrow || tile_cone_keeps(P.pos[0], P.pos[1], P.pos[2], ux, uy, uz, sin_t, cos_t, sphere);
        }
        for (unsigned long long todo = __ballot(keep); todo != 0ull; todo &= todo - 1ull) {
            const int k = base + (int)__builtin_ctzll(todo);
            bool far_root = false;
            if (P.materials != nullptr) far_root = P.materials[k].x == 3.0f;
            const float4 g = *reinterpret_cast<const float4*>(P.spheres + k);
            test_sphere<FUSED>(p, a, g.x, g.y, g.z, g.w, k, best, idx, far_root);
        }
    }
    hits[(size_t)tile * 64u + lane] = make_uint2((uint32_t)idx, __float_as_uint(best));
    const unsigned long long mask = __ballot(idx >= 0);
    if (lane == 0) live[tile] = mask;
}

// The pixels whose camera ray goes to the sky (P.cam_hits says which), all frames of the dispatch: every frame is
// one Trace() call and light = sky_color * (0.15, 0.15, 0.15) (shader.comp:94-102), added frame by frame in the
// shader's order with the clears where they fall.  rt_stream_kernel takes only the other pixels (P.cam_live).
// A wave per tile at a time, tiles strided over the grid; blockDim.x = a multiple of 64.
template <bool FUSED>
__global__ void __launch_bounds__(256) rt_sky_frames_kernel(const RtParams P)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_tiles = (uint32_t)P.tile_cols * (uint32_t)P.tile_rows;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    unsigned nrays = 0;
    for (uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += waves) {
        if ((P.cam_live[tile] >> lane) & 1ull) continue;
        const uint32_t ty = P.tile_cols_magic ? __umulhi(tile, P.tile_cols_magic) : tile;
        const uint32_t tx = tile - ty * (uint32_t)P.tile_cols;
        const uint32_t px = ((uint32_t)P.tile_col0 + tx) * 8 + (lane & 7u);
        const uint32_t py = ((uint32_t)P.tile_row0 + ty) * 8 + (lane >> 3);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!clear_at(P, 0, P.frame0)) acc = P.accum[(size_t)py * P.width + px];
        float dx, dy, dz;
        primary_direction<FUSED>(P, px, py, dx, dy, dz);
        Path p;
        path_reset(p, P, dx, dy, dz);
        (void)path_bounce<FUSED, false>(p, P, -1, RT_FLT_MAX, px, py, 0u, nullptr, nullptr);
        for (int j = 0; j < P.n_frames; j++) {
            if (j > 0 && clear_at(P, j, P.frame0 + (uint32_t)(j * P.frame_stride))) acc = make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x = p.lr + acc.x;  // vec4(light, 1.0) + accumulated
            acc.y = p.lg + acc.y;
            acc.z = p.lb + acc.z;
            acc.w = 1.0f + acc.w;
        }
        nrays += (unsigned)P.n_frames;
        store_pixel(P, px, py, acc);
    }
    add_rays(P, nrays, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}

// ------------------------------------------------- sieve on the matrix cores
//
// The same conservative test g = s^2 + t (sieve_pair above) with its two dot products
//     s = c . (-n) + d0,        t = c . (2 o) + w - o.o (1 - margin)
// evaluated for 32 spheres x 32 rays at a time by v_mfma_f32_32x32x16_bf16: a dense contraction
// over K = 4 coordinates.  Every f32 operand x is split exactly into three bf16 terms
// x = x1 + x2 + x3 (truncation: each term holds the next 8 significant bits), and a product
// a b is the sum of the eight partial products a_p b_q other than a3 b3 (which is below
// 2^-32 |a b|); each partial product of two bf16 is exact in f32, the matrix core accumulates in
// f32.  So one dot product is a K = 4 x 8 = 32 contraction = two MFMA steps per 32 x 32 tile:
//   step s, lane half h  <->  coordinate d = 2 s + h   (0: x, 1: y, 2: z, 3: the constants)
//   element j of a fragment <-> partial product (p, q)[j] = (1,1) (1,2) (2,1) (2,2) (1,3) (3,1) (2,3) (3,2)
// Sphere side (A operand, from LDS, built by the host: rt_api.cpp build_mfma_frags): elements
// a_p[j] of coordinate d; for d = 3  {w1, w2, w3, 1, 1, 1, 0, 0}.  Ray side (B operand): b_q[j];
// for d = 3 the s-product carries {0, 0, 0, e1, e2, e3, 0, 0}, e = d0, and the t-product
// {1, 1, 1, k1, k2, k3, 0, 0}, k = -o.o (1 - margin): the constants ride in the spare slots.
// Rows of a tile are spheres: row (v & 3) + 8 (v >> 2) + 4 h holds sphere 16 h + v of the chunk,
// so that after the products the 16 accumulator registers of a lower-half lane are spheres
// 0..15 and those of an upper-half lane spheres 16..31 of the lane's column (= ray).
// Two tiles cover the wave's 64 rays (ray group g = rays 32 g .. 32 g + 31); a lane owns ray
// `lane`, so for each group one half of the lanes holds bits of rays the other half owns, and one
// v_permlane32_swap puts them right.  What is left for the vector ALU is one v_fma_f32 and one
// v_alignbit_b32 per accumulator register: 2 instructions per ray-sphere test instead of 5, the
// matrix cores working beside them.
//
// Error: with u = 2^-24, every accumulation step of a 16-term MFMA is taken to err by at most
// u times the sum of the magnitudes it adds (f32 accumulation, any order), i.e. |s - S| <=
// 32u (|o| + |c|) and |t - T| <= 32u ((|o| + |c|)^2 + r^2), against 10.5u and 8u for the fma chains
// of sieve_pair; together with the shader's own error that stays below
// 250u (o.o + c.c) + 50u r^2, and this path uses the margin RT_SIEVE_MARGIN_MFMA = 2^-15 = 512u.
// rt_debug_sieve_audit runs this very code on caller-supplied rays and must find no miss.

// The three bf16 terms of x as f32 values whose low 16 bits are zero (x = x1 + x2 + x3 exactly
// for finite x; terms below the bf16 normal range may be flushed by the matrix core).
__device__ __forceinline__ void split3(float x, uint32_t& x1, uint32_t& x2, uint32_t& x3)
{
    x1 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(x1);
    x2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(x2);
    x3 = __float_as_uint(r2) & 0xffff0000u;
}

// {lo, hi} -> one register holding bf16(lo) in its low half and bf16(hi) in its high half
__device__ __forceinline__ uint32_t pack_bf(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

// Ray-side fragment of one coordinate: elements b_q[j], q = 1 2 1 2 3 1 3 2.
__device__ __forceinline__ v4u ray_frag(float x)
{
    uint32_t b1, b2, b3;
    split3(x, b1, b2, b3);
    const uint32_t r01 = pack_bf(b1, b2);
    return v4u{r01, r01, pack_bf(b3, b1), pack_bf(b3, b2)};
}

// Ray-side fragment of coordinate 3: {lead x 3, c1, c2, c3, 0, 0}, lead = 0 (s-product) or 1 (t-product).
__device__ __forceinline__ v4u ray_frag_const(float c, bool ones)
{
    uint32_t c1, c2, c3;
    split3(c, c1, c2, c3);
    const uint32_t one = ones ? 0x3f800000u : 0u;
    return v4u{pack_bf(one, one), pack_bf(one, c1), pack_bf(c2, c3), 0u};
}

struct MfmaRay {
    v8bf s0[2], s1[2];  // s-product, step 0 / 1, ray group 0 / 1
    v8bf t0[2], t1[2];  // t-product
};

// Own-ray fragments of coordinates (lo, hi) -> the operands of the two ray groups for one step:
// group 0 wants {lower lanes: own lo, upper lanes: hi of the lane 32 below}, group 1 the mirror
// image; v_permlane32_swap exchanges the upper half of its first operand with the lower half of
// its second.
__device__ __forceinline__ void swap_halves(v4u lo, v4u hi, v8bf& g0, v8bf& g1)
{
    v4u a, b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const v2u r = __builtin_amdgcn_permlane32_swap(lo[k], hi[k], false, false);
        a[k] = r.x;
        b[k] = r.y;
    }
    g0 = __builtin_bit_cast(v8bf, a);
    g1 = __builtin_bit_cast(v8bf, b);
}

__device__ __forceinline__ MfmaRay mfma_ray(const Path& p, float a)
{
    const float inv = __builtin_amdgcn_rsqf(a);  // 1 ulp is plenty: the margin absorbs it
    const float nx = p.dx * inv, ny = p.dy * inv, nz = p.dz * inv;
    const float d0 = __builtin_fmaf(nz, p.oz, __builtin_fmaf(ny, p.oy, nx * p.ox));
    const float o2 = __builtin_fmaf(p.oz, p.oz, __builtin_fmaf(p.oy, p.oy, p.ox * p.ox));
    MfmaRay r;
    swap_halves(ray_frag(-nx), ray_frag(-ny), r.s0[0], r.s0[1]);
    swap_halves(ray_frag(-nz), ray_frag_const(d0, false), r.s1[0], r.s1[1]);
    swap_halves(ray_frag(2.0f * p.ox), ray_frag(2.0f * p.oy), r.t0[0], r.t0[1]);
    swap_halves(ray_frag(2.0f * p.oz), ray_frag_const(o2 * (RT_SIEVE_MARGIN_MFMA - 1.0f), true), r.t1[0], r.t1[1]);
    return r;
}

// Sign bits of g = s^2 + t over a tile's 16 accumulator registers, first register in bit 15.
// BEHIND: g = t - s |s| instead, which is the same for s < 0 and rejects as well what lies behind
// the ray's origin as seen from outside the sphere -- the abs and neg modifiers of v_fma_f32 cost
// nothing.  s = n.(o - c) has the sign of the shader's b = 2 dot(oc, d), and the shader takes a hit
// only when b < 0 (test_sphere; with the material extension a dielectric sphere also when b >= 0,
// so then BEHIND is off).  A sphere the shader can take has b < 0 as computed, hence s < 40u (|o| +
// |c|) as computed here (u = 2^-24; both roundings), and where 0 <= s is that small t - s^2 differs
// from t + s^2 by 2 s^2 < 10^-11 (|o| + |c|)^2, five orders of magnitude inside the margin's
// slack.  With the origin inside the sphere t > s^2 anyway and nothing changes.
template <bool BEHIND>
__device__ __forceinline__ uint32_t tile_mask(const v16f& S, const v16f& T)
{
    uint32_t mask = 0;
    // (one v_fma_f32 per accumulator register: measured, v_pk_fma_f32 on register pairs is 5 % slower here)
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const float g = BEHIND ? __builtin_fmaf(S[v], -__builtin_fabsf(S[v]), T[v]) : __builtin_fmaf(S[v], S[v], T[v]);
        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(g), 31);
    }
    return mask;
}

// Sieve of one chunk: frag points at the chunk's two steps of 64 lane fragments in LDS.  Same
// result convention as sieve_chunk: bit (31-n) SET when sphere n of the chunk is rejected.
template <bool BEHIND>
__device__ __forceinline__ uint32_t sieve_chunk_mfma(const v8bf* frag, int lane, const MfmaRay& r)
{
    const v8bf a0 = frag[lane], a1 = frag[64 + lane];
    uint32_t m[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const v16f zero = {0};
        v16f S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, r.s0[g], zero, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, r.s1[g], S, 0, 0, 0);
        v16f T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, r.t0[g], zero, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, r.t1[g], T, 0, 0, 0);
        m[g] = tile_mask<BEHIND>(S, T);  // spheres 16 h .. 16 h + 15 of ray 32 g + (lane & 31), h = lane >> 5
    }
    // lower lanes own group 0, upper lanes group 1: m[0] keeps its lower half and takes m[1]'s
    // lower half (spheres 0..15 of the upper lanes' own rays) into its upper half, and so on
    const v2u own = __builtin_amdgcn_permlane32_swap(m[0], m[1], false, false);
    return (own.x << 16) | own.y;
}

// Vector instructions the scheduler is asked to place behind each MFMA of the pipelined sieve: a
// tile's accumulators take 16 v_fma_f32 and 16 v_alignbit_b32 beside its 4 MFMAs.
#ifndef RT_MFMA_VALU_GROUP
#define RT_MFMA_VALU_GROUP 8
#endif

// The four MFMAs of one 32 x 32 tile (ray group g of a chunk).
__device__ __forceinline__ void mfma_tile(v8bf a0, v8bf a1, const MfmaRay& r, int g, v16f& S, v16f& T)
{
    const v16f zero = {0};
    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, r.s0[g], zero, 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, r.t0[g], zero, 0, 0, 0);
    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, r.s1[g], S, 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, r.t1[g], T, 0, 0, 0);
}

// Matrix-core sieve of `nsg` consecutive chunks (frag = the first one's fragments in LDS),
// software-pipelined by one tile: the MFMAs of the next tile are issued before the vector
// instructions that read the current tile's accumulators, so that a wave keeps both pipes busy
// by itself.  Writes the candidate masks of the chunks to wmask[c * 64 + lane] and returns the
// summary (bit 31 - c set when chunk c has a candidate for this lane).
// PIPE = false: one tile at a time, 32 accumulator registers instead of 64 -- for the launch with three
// waves per SIMD (168 registers a lane), where the other waves cover the wait for the matrix pipe and the
// pipelined loop made the compiler park a lane's per-pixel state in scratch memory around it (measured at
// 484 spheres: 24.4 against 23.9 Grays/s; with two waves per SIMD, 1000 spheres, the pipelined loop wins by 3 %).
template <bool BEHIND, bool PIPE>
__device__ __forceinline__ uint32_t sieve_chunks_mfma(const v8bf* frag, int nsg, int lane, const MfmaRay& r,
                                                      bool alive, bool has_last, uint32_t last_valid,
                                                      uint32_t* wmask)
{
    uint32_t summary = 0;
    if constexpr (!PIPE) {
        // two chunks per trip, their fragments in two register sets that take turns (no copies from one to the
        // other: 484 spheres 24.0 -> 24.7 Grays/s); an odd last chunk on its own
        auto chunk = [&](int c, v8bf a0, v8bf a1) {
            v16f S, T;
            mfma_tile(a0, a1, r, 0, S, T);
            const uint32_t m0 = tile_mask<BEHIND>(S, T);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(a0, a1, r, 1, S, T);
            const uint32_t m1 = tile_mask<BEHIND>(S, T);
            __builtin_amdgcn_sched_barrier(0);
            const v2u own = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
            uint32_t cand = ~((own.x << 16) | own.y);
            if (has_last && c == nsg - 1) cand &= last_valid;
            if (!alive) cand = 0u;
            wmask[c * 64 + lane] = cand;
            summary = (summary << 1) | (cand != 0u ? 1u : 0u);
        };
        v8bf a0 = frag[lane], a1 = frag[64 + lane];
        for (int c = 0; c + 1 < nsg; c += 2) {
            const int c2 = min(c + 2, nsg - 1);
            const v8bf b0 = frag[(c + 1) * 128 + lane], b1 = frag[(c + 1) * 128 + 64 + lane];
            chunk(c, a0, a1);
            a0 = frag[c2 * 128 + lane];
            a1 = frag[c2 * 128 + 64 + lane];
            chunk(c + 1, b0, b1);
        }
        if (nsg & 1) chunk(nsg - 1, a0, a1);  // a0, a1: the pairs' last load, or the first chunk when it is the only one
        return summary << (32 - nsg);
    }
    v16f SA, TA, SB, TB;
    v8bf a0 = frag[lane], a1 = frag[64 + lane];
    mfma_tile(a0, a1, r, 0, SA, TA);
    for (int c = 0; c < nsg; c++) {
        // the next chunk's fragments are on their way while this chunk is worked on (the last
        // chunk "prefetches" itself: its extra tile is computed and dropped, which keeps the loop
        // body free of branches for the instruction scheduler)
        const int cn = min(c + 1, nsg - 1);
        const v8bf n0 = frag[cn * 128 + lane], n1 = frag[cn * 128 + 64 + lane];
        mfma_tile(a0, a1, r, 1, SB, TB);
        const uint32_t m0 = tile_mask<BEHIND>(SA, TA);
        // one MFMA, then the eight vector instructions of four accumulator registers: the matrix
        // pipe takes 32 cycles per MFMA, the vector instructions fit beside it
#pragma unroll
        for (int k = 0; k < 4; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, RT_MFMA_VALU_GROUP, 0);
        }
        mfma_tile(n0, n1, r, 0, SA, TA);
        const uint32_t m1 = tile_mask<BEHIND>(SB, TB);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, RT_MFMA_VALU_GROUP, 0);
        }
        a0 = n0;
        a1 = n1;
        // lower lanes own ray group 0, upper lanes group 1: after the swap .x holds spheres 0..15
        // and .y spheres 16..31 of the lane's own ray
        const v2u own = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
        uint32_t cand = ~((own.x << 16) | own.y);
        if (has_last && c == nsg - 1) cand &= last_valid;
        if (!alive) cand = 0u;  // an idle lane has no candidates
        wmask[c * 64 + lane] = cand;
        summary = (summary << 1) | (cand != 0u ? 1u : 0u);
    }
    return summary << (32 - nsg);
}

// As sieve_chunks_mfma for scenes whose sphere-side fragments do not fit in LDS beside the rest
// (more than RT_MAX_MFMA_SPHERES spheres): `frag` points into GLOBAL memory (the fragments of a
// 4096-sphere scene are 256 KiB, resident in L2), every wave streams them for itself, and the
// loads run AHEAD chunks in front of the arithmetic through a ring of AHEAD register pairs.  The
// loop is unrolled AHEAD times so that the ring is addressed statically; the chunks are taken in
// runs of AHEAD, a run that overshoots `nsg` recomputes the last chunk and drops the result.
template <int AHEAD, bool BEHIND>
__device__ __forceinline__ uint32_t sieve_chunks_mfma_far(const v8bf* __restrict__ frag, int nsg, int lane,
                                                          const MfmaRay& r, bool alive, bool has_last,
                                                          uint32_t last_valid, uint32_t* wmask)
{
    uint32_t summary = 0;
    v16f SA, TA, SB, TB;
    v8bf q0[AHEAD], q1[AHEAD];
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        const int cn = min(k, nsg - 1);
        q0[k] = frag[cn * 128 + lane];
        q1[k] = frag[cn * 128 + 64 + lane];
    }
    mfma_tile(q0[0], q1[0], r, 0, SA, TA);
    for (int c0 = 0; c0 < nsg; c0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            const int c = c0 + k;
            constexpr int dummy = 0;
            (void)dummy;
            const v8bf a0 = q0[k], a1 = q1[k];
            const int cn = min(c + AHEAD, nsg - 1);
            mfma_tile(a0, a1, r, 1, SB, TB);
            // slot k is free once its second tile has been issued: chunk c + AHEAD goes there
            q0[k] = frag[cn * 128 + lane];
            q1[k] = frag[cn * 128 + 64 + lane];
            const uint32_t m0 = tile_mask<BEHIND>(SA, TA);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, RT_MFMA_VALU_GROUP, 0);
            }
            mfma_tile(q0[(k + 1) % AHEAD], q1[(k + 1) % AHEAD], r, 0, SA, TA);
            const uint32_t m1 = tile_mask<BEHIND>(SB, TB);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, RT_MFMA_VALU_GROUP, 0);
            }
            const v2u own = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
            uint32_t cand = ~((own.x << 16) | own.y);
            if (has_last && c == nsg - 1) cand &= last_valid;
            if (!alive) cand = 0u;
            if (c < nsg) {  // wave-uniform
                wmask[c * 64 + lane] = cand;
                summary = (summary << 1) | (cand != 0u ? 1u : 0u);
            }
        }
    }
    return summary << (32 - nsg);
}

// ------------------------------------------------- per-lane traversal (MODE 4)
//
// Trace() over the spatially sorted sphere list as a walk down an implicit 8-ary hierarchy, every
// lane on its own: blocks of RT_BVH_BLOCK = 64 slots, their 8 groups of RT_CULL_GROUP = 8 slots,
// the 8 spheres of a group.  Blocks and groups carry axis-aligned boxes (build_culled() in
// rt_api.cpp), spheres are tested with the sieve:
//   - top: the whole wave sieves group 0 (the largest spheres), then tests the boxes of the blocks
//     (at most 64) from the scalar cache, one 64-bit mask of blocks per lane;
//   - then a lane takes its work one NODE at a time: with no group open it opens its next block --
//     reads the boxes of the block's 8 groups from LDS at its own address and tests them -- and
//     with a group open it reads the group's 8 spheres and sieves them; the spheres that pass get
//     the shader's literal test at once.
// What the wave-level OR of the culled kernel gives away to incoherent rays stays here: a lane
// visits only what its own ray may touch, and of that only what lies in front of its origin and
// not beyond the nearest hit it has found so far.  Ties go to the lowest ORIGINAL index
// (test_sphere with the slot -> index table), whatever the order of the visits.
//
// Why nothing that can be hit is dropped (u = 2^-24, first order in u).  The shader's own t of a
// sphere (c, r), times |d|, lies between the exact entry and exit distances of the sphere inflated
// by E = 21u|o-c|^2 + 7u r^2 -- the error bound of its discriminant, see sieve_pair -- up to a few u
// of |o - c|; so the hit point lies within sqrt(r^2 + E) <= r + 0.0012 (|o| + |c|) + 0.0007 r of c.
//   Boxes: the host's box of a node holds every member with its radius enlarged by RT_BOX_PAD
//   (|c| + r), the test below widens it by RT_BOX_PAD (|ox| + |oy| + |oz|) >= RT_BOX_PAD |o| for the ray at
//   hand, RT_BOX_PAD = 0.002: the hit point is inside by at least 0.0008 (|o| + |c|) + 0.0013 r on
//   every axis, against rounding in the slab arithmetic of a few u (|o| + |c| + h + t).  So the
//   interval [tmin, tmax] the test computes holds t |d| with that much to spare at both ends, and
//   "tmax < 0", "tmin > tmax" and "tmin > best |d|" (best = the nearest literal t so far, a tie
//   included) are all false for a node with such a hit.  Direction components below 2^-40 are
//   given the reciprocal 2^40: the interval then still covers every t up to 2^40 times the
//   padding, far beyond the scene.
//   Spheres: bvh_pair below.
// rt_debug_cull_audit evaluates these very functions on caller-supplied rays.

// Per-ray constants of the box test: unit direction n = d/|d|, t counted in units of n.
struct BoxRay {
    v2f ix, iy, iz;     // 1 / n, clamped to +-2^40
    v2f ax, ay, az;     // |1 / n|
    v2f ox, oy, oz;     // -o / n
    v2f ex, ey, ez;     // RT_BOX_PAD (|ox| + |oy| + |oz|) |1 / n|
};

__device__ __forceinline__ BoxRay box_ray(const Path& p, float a)
{
    const float inv = __builtin_amdgcn_rsqf(a);
    const float big = 1099511627776.0f;  // 2^40
    const float pad = RT_BOX_PAD * (fabsf(p.ox) + fabsf(p.oy) + fabsf(p.oz));
    BoxRay r;
    const float ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(p.dx * inv), -big, big);
    const float iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(p.dy * inv), -big, big);
    const float iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(p.dz * inv), -big, big);
    r.ix = v2f{ix, ix}; r.iy = v2f{iy, iy}; r.iz = v2f{iz, iz};
    r.ax = v2f{fabsf(ix), fabsf(ix)}; r.ay = v2f{fabsf(iy), fabsf(iy)}; r.az = v2f{fabsf(iz), fabsf(iz)};
    r.ox = v2f{-p.ox * ix, -p.ox * ix}; r.oy = v2f{-p.oy * iy, -p.oy * iy}; r.oz = v2f{-p.oz * iz, -p.oz * iz};
    r.ex = v2f{pad * fabsf(ix), pad * fabsf(ix)};
    r.ey = v2f{pad * fabsf(iy), pad * fabsf(iy)};
    r.ez = v2f{pad * fabsf(iz), pad * fabsf(iz)};
    return r;
}

// The verdict on one pair of boxes, X Y Z = {c0, c1, h0, h1} per axis: per axis the ray is inside
// the slab for t in [tc - th, tc + th], tc = (c - o) / n, th = (h + pad) / |n|; the sign bit shifted
// into `mask` is SET (rejected) when the three intervals have no common point, when they end
// before t = 0, or when they begin after bh = best |d|.
__device__ __forceinline__ void bvh_box_pair(uint32_t& mask, v4f X, v4f Y, v4f Z, const BoxRay& r, float bh)
{
    const v2f tcx = __builtin_elementwise_fma(v2f{X.x, X.y}, r.ix, r.ox), thx = __builtin_elementwise_fma(v2f{X.z, X.w}, r.ax, r.ex);
    const v2f tcy = __builtin_elementwise_fma(v2f{Y.x, Y.y}, r.iy, r.oy), thy = __builtin_elementwise_fma(v2f{Y.z, Y.w}, r.ay, r.ey);
    const v2f tcz = __builtin_elementwise_fma(v2f{Z.x, Z.y}, r.iz, r.oz), thz = __builtin_elementwise_fma(v2f{Z.z, Z.w}, r.az, r.ez);
    const v2f lx = tcx - thx, hx = tcx + thx, ly = tcy - thy, hy = tcy + thy, lz = tcz - thz, hz = tcz + thz;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(lx[k], ly[k]), lz[k]);
        const float tmax = __builtin_fminf(__builtin_fminf(hx[k], hy[k]), hz[k]);
        const uint32_t verdict = __float_as_uint(tmax - tmin) | __float_as_uint(tmax) | __float_as_uint(bh - tmin);
        mask = __builtin_amdgcn_alignbit(mask, verdict, 31);
    }
}

// The verdict on one pair of spheres for one ray: the sign bit of the sieve's g as in sieve_pair,
// OR two tests on the same intermediate values that the brute-force kernels have no use for.
// With s = n.(o - c) and t = r^2 - |o - c|^2 + m (n = d/|d|, m the margin), g = s^2 + t, and
// bh = (best t so far) * |d|:
//   - behind:  s > 0 and t < 0 -- the origin lies outside the sphere (inflated by m) and the centre
//     behind it: both roots are negative;
//   - beyond:  s + bh < 0 and bh (bh + 2 s) - t > 0 -- the ray enters the inflated sphere only after
//     bh: it cannot be hit as near as the hit already found (a tie included).
// Safe because the sieve inflates by m = 256u (o.o + c.c + r.r) >= 6 E (E as above), and all rounding
// in s, t, bh and the expression above stays below 111u (o.o + c.c + r.r) of the 213u that difference
// leaves.  bh = +inf (no hit yet) prunes nothing.
__device__ __forceinline__ void bvh_pair(uint32_t& mask, v4f A, v4f B, const SieveRay& r, v2f bh)
{
    const v2f cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, w = {B.z, B.w};
    const v2f s = __builtin_elementwise_fma(
        cz, r.nnz, __builtin_elementwise_fma(cy, r.nny, __builtin_elementwise_fma(cx, r.nnx, r.d0)));
    const v2f t = __builtin_elementwise_fma(
        cz, r.p2oz, __builtin_elementwise_fma(cy, r.p2oy, __builtin_elementwise_fma(cx, r.p2ox, r.no2 + w)));
    const v2f q = __builtin_elementwise_fma(s, s, t);
    const v2f h = s + bh;
    const v2f f = __builtin_elementwise_fma(bh, s + h, -t);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t sb = __float_as_uint(s[k]), tb = __float_as_uint(t[k]), qb = __float_as_uint(q[k]);
        const uint32_t hb = __float_as_uint(h[k]), fb = __float_as_uint(f[k]);
        const uint32_t verdict = qb | (~sb & tb) | (~fb & hb);  // sign bit set: rejected
        mask = __builtin_amdgcn_alignbit(mask, verdict, 31);
    }
}

// bh of bvh_pair and bvh_box_pair for a ray: best * |d|, with |d| = a * rsq(a) as the sieve's
// normalisation has it.
__device__ __forceinline__ float bvh_reach(float best, float a)
{
    return best * (a * __builtin_amdgcn_rsqf(a));
}

// A mask of 32 nodes, bit (31 - n) for node n, rearranged so that bit (31 - m) stands for node m ^ f (f < 32; levels:
// wave-uniform, the bits of f that can be set).  Taking the bits of the result from the top then visits the nodes in
// ascending order of (number ^ f): with f = the bits whose split a ray crosses downwards, nearer nodes first.
__device__ __forceinline__ uint32_t bvh_rearrange(uint32_t m, uint32_t f, int levels)
{
    if (f & 1u) m = ((m & 0x55555555u) << 1) | ((m >> 1) & 0x55555555u);
    if (f & 2u) m = ((m & 0x33333333u) << 2) | ((m >> 2) & 0x33333333u);
    if (f & 4u) m = ((m & 0x0f0f0f0fu) << 4) | ((m >> 4) & 0x0f0f0f0fu);
    if (levels > 3) {
        if (f & 8u) m = ((m & 0x00ff00ffu) << 8) | ((m >> 8) & 0x00ff00ffu);
        if (f & 16u) m = (m << 16) | (m >> 16);
    }
    return m;
}

// The bits of a slot number whose split a ray crosses from the higher coordinates to the lower: bit k of the result
// stands for bit 3 + k of the slot number; RtParams::kd_flip[a] holds the bits whose split is along axis a.
__device__ __forceinline__ uint32_t bvh_flip_bits(const Path& p, const RtParams& P, int n_bits)
{
    const uint32_t flip = (p.dx < 0.0f ? P.kd_flip[0] : 0u) | (p.dy < 0.0f ? P.kd_flip[1] : 0u) | (p.dz < 0.0f ? P.kd_flip[2] : 0u);
    return flip & ((1u << n_bits) - 1u);
}

// (The traversal is a chain of dependent LDS gathers and gains from every wave it can get: 13.9 /
// 18.9 / 22.9 / 26.3 Grays/s at 4 / 6 / 8 / 12 waves per CU on the 484-sphere scene.  Its ~150
// registers allow 12; capped at 128 registers for 16 waves it spills 46 of them.)

#ifdef RT_COUNT_PASSES
static __device__ unsigned long long g_bvh_count[4];  // experiment: wave-level steps, lane-level steps; of the latter: blocks opened, literal tests
#endif

// lds_geom: the sorted pair-packed sieve geometry; lds_boxes: the boxes of the groups (P.n_bounds
// entries, 3 float4 per pair); both in LDS.  The boxes of the blocks are read from P.boxes.
template <bool FUSED>
__device__ __forceinline__ void trace_bvh(const Path& p, float a, bool alive, const RtParams& P, const float4* lds_geom,
                                          const float* lds_radius, const float4* lds_boxes, float& best, int& idx)
{
#ifdef RT_COUNT_PASSES
    unsigned bvh_steps_ = 0, bvh_inner_ = 0, bvh_tests_ = 0;
#endif
    const SieveRay sr = sieve_ray(p, a, alive);
    const BoxRay br = box_ray(p, a);
    const v4f* geom = reinterpret_cast<const v4f*>(lds_geom);
    const v4f* boxes = reinterpret_cast<const v4f*>(lds_boxes);
    const float* gf = reinterpret_cast<const float*>(lds_geom);
    // Group 0 first, by the whole wave: the sorted list begins with the scene's largest spheres (a
    // ground sphere, say), and a hit on one of them lets the distance test drop whole blocks below.
    {
        const v2f none2 = {__builtin_inff(), __builtin_inff()};
        uint32_t rej = 0;
#pragma unroll
        for (int k = 0; k < RT_CULL_GROUP / 2; k++) bvh_pair(rej, geom[2 * k], geom[2 * k + 1], sr, none2);
        uint32_t cand = (~rej) << (32 - RT_CULL_GROUP);
        while (__ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int i = __clz(cand);
                cand &= ~(0x80000000u >> i);
                bool far_root = false;
                if (P.materials != nullptr) far_root = P.materials[P.perm[i]].x == 3.0f;  // wave-uniform branch
                const float* f = gf + (i >> 1) * 8 + (i & 1);
                test_sphere<FUSED>(p, a, f[0], f[2], f[4], lds_radius[i], i, best, idx, far_root, P.perm);
            }
        }
    }
    // block boxes, 4 at a time from the scalar cache
    const int nblk = (P.n_pad + RT_BVH_BLOCK - 1) / RT_BVH_BLOCK;
    const int n4 = (nblk + 3) >> 2;
    ConstGeomPtr bb = (ConstGeomPtr)(const void*)(P.boxes) + (P.n_bounds >> 1) * 3;
    const float reach0 = bvh_reach(best, a);
    uint32_t rej0 = 0u, rej1 = 0u;
#pragma unroll 1
    for (int n = 0; n < min(n4, 8); n++) {
        bvh_box_pair(rej0, bb[6 * n], bb[6 * n + 1], bb[6 * n + 2], br, reach0);
        bvh_box_pair(rej0, bb[6 * n + 3], bb[6 * n + 4], bb[6 * n + 5], br, reach0);
    }
#pragma unroll 1
    for (int n = 8; n < n4; n++) {
        bvh_box_pair(rej1, bb[6 * n], bb[6 * n + 1], bb[6 * n + 2], br, reach0);
        bvh_box_pair(rej1, bb[6 * n + 3], bb[6 * n + 4], bb[6 * n + 5], br, reach0);
    }
    // bit (31 - n): block n (bm0) / block 32 + n (bm1) may be touched; what was not evaluated has no spheres
    uint32_t bm0 = n4 > 0 && alive ? ~rej0 << (32 - 4 * min(n4, 8)) : 0u;
    uint32_t bm1 = n4 > 8 && alive ? ~rej1 << (32 - 4 * (n4 - 8)) : 0u;
    uint32_t gm = 0;  // bit (31 - k): group k of the open block is still to be visited
    int blk = 0;
    // Nearer blocks and groups first.  The sorted list is a k-d partition whose two halves at every split differ in
    // one bit of the slot number, the half with the lower coordinates holding the 0 (build_culled): a ray that travels
    // towards lower coordinates along the axis of a bit meets the half with the 1 first.  flip: those bits; a lane
    // visits blocks in ascending order of (block number ^ its flip bits) and the groups of a block likewise.  The sooner
    // a near hit is known, the more the distance test drops.  Measured at 3840x2160, slot order / nearer first (ms a
    // dispatch): 4096 spheres 4.68 / 3.0, 8192 (out of global memory, trace_bvh_far) 3.9 / 2.35, 16 384 2.40 / 1.76,
    // 65 536 2.29 / 1.99, 1000 spheres 8.5 / 7.1, 484 spheres on a plane 4.79 / 4.26 (4.11 with one bit for the whole
    // list, by the sign of dx + dy + dz: there the ~90 instructions a pass that the bits cost outweigh what they add).
    const int levels = nblk > 32 ? 6 : nblk > 16 ? 5 : nblk > 8 ? 4 : 3;  // bits that number this scene's blocks
    const uint32_t flip = bvh_flip_bits(p, P, 3 + levels);
    const uint32_t flip_g = flip & 7u, flip_b = flip >> 3;  // of the group within its block, of the block
    // the masks of the blocks, bit (31 - n) for block n, rearranged so that bit (31 - m) stands for block m ^ flip_b
    auto rearrange = [](uint32_t m, uint32_t f, int levels) { return bvh_rearrange(m, f, levels); };
    bm0 = rearrange(bm0, flip_b, levels);
    if (levels > 5) {
        bm1 = rearrange(bm1, flip_b, levels);
        if (flip_b & 32u) {
            const uint32_t t_ = bm0;
            bm0 = bm1;
            bm1 = t_;
        }
    }
    for (;;) {
        // ---- lanes without an open group open their next block
        const bool open = gm == 0u && (bm0 | bm1) != 0u;
        if (__ballot(open) != 0ull) {
            if (open) {
                if (bm0 != 0u) {
                    blk = __clz(bm0);
                    bm0 &= ~(0x80000000u >> blk);
                } else {
                    const int k = __clz(bm1);
                    bm1 &= ~(0x80000000u >> k);
                    blk = 32 + k;
                }
                blk ^= (int)flip_b;
                const v4f* g = boxes + blk * (RT_BVH_BLOCK / RT_CULL_GROUP / 2 * 3);  // 8 groups = 4 pairs = 12 float4
                const float reach = bvh_reach(best, a);
                uint32_t rej = 0;
#pragma unroll
                for (int half = 0; half < 2; half++) {  // (six registers of boxes at a time: this kernel has none to spare)
                    v4f e[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) e[k] = g[6 * half + k];
#pragma unroll
                    for (int k = 0; k < 2; k++) bvh_box_pair(rej, e[3 * k], e[3 * k + 1], e[3 * k + 2], br, reach);
                }
                gm = (~rej) << (32 - RT_CULL_GROUP);
                if (blk == 0) gm &= 0x7fffffffu;  // group 0 has been dealt with
                gm = rearrange(gm, flip_g, 3);      // bit (31 - m): group m ^ flip_g
            }
#ifdef RT_COUNT_PASSES
            bvh_steps_ += open ? 0x10001u : 0x10000u;  // experiment: wave steps << 16 | this lane's steps
            bvh_inner_ += open ? 1u : 0u;
#endif
        }
        // ---- lanes with an open group sieve its next group of spheres
        const bool leaf = gm != 0u;
        if (__ballot(leaf) == 0ull) {
            if (__ballot((bm0 | bm1) != 0u) == 0ull) break;
            continue;
        }
#ifdef RT_COUNT_PASSES
        bvh_steps_ += leaf ? 0x10001u : 0x10000u;
#endif
        int grp = 0;
        uint32_t cand = 0u;
        if (leaf) {
            const int k = __clz(gm);
            gm &= ~(0x80000000u >> k);
            grp = blk * (RT_BVH_BLOCK / RT_CULL_GROUP) + (k ^ (int)flip_g);
            const v4f* g = geom + grp * RT_CULL_GROUP;
            v4f e[RT_CULL_GROUP];
#pragma unroll
            for (int k2 = 0; k2 < RT_CULL_GROUP; k2++) e[k2] = g[k2];
            const float reach = bvh_reach(best, a);
            const v2f bh = {reach, reach};
            uint32_t rej = 0;
#pragma unroll
            for (int k2 = 0; k2 < RT_CULL_GROUP / 2; k2++) bvh_pair(rej, e[2 * k2], e[2 * k2 + 1], sr, bh);
            cand = (~rej) << (32 - RT_CULL_GROUP);  // bit (31 - k): sphere k of the group passed
        }
        while (__ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int b = __clz(cand);
                cand &= ~(0x80000000u >> b);
                const int i = grp * RT_CULL_GROUP + b;
                bool far_root = false;
                if (P.materials != nullptr) far_root = P.materials[P.perm[i]].x == 3.0f;  // wave-uniform branch
                const float* f = gf + (i >> 1) * 8 + (i & 1);
                test_sphere<FUSED>(p, a, f[0], f[2], f[4], lds_radius[i], i, best, idx, far_root, P.perm);
#ifdef RT_COUNT_PASSES
                bvh_tests_++;
#endif
            }
        }
    }
#ifdef RT_COUNT_PASSES
    unsigned long long lanes = bvh_steps_ & 0xffffu, inner = bvh_inner_, tests = bvh_tests_;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lanes += __shfl_down(lanes, off, 64);
        inner += __shfl_down(inner, off, 64);
        tests += __shfl_down(tests, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_bvh_count[0], (unsigned long long)(bvh_steps_ >> 16));
        atomicAdd(&g_bvh_count[1], lanes);
        atomicAdd(&g_bvh_count[2], inner);
        atomicAdd(&g_bvh_count[3], tests);
    }
#endif
}

// The same walk for scenes too large for LDS (MODE 6; up to RT_BVH_FAR_MAX_SPHERES spheres): one level
// more -- superblocks of RT_BVH_SUPER = 512 slots, whose boxes the whole wave tests from the scalar
// cache, a 128-bit mask per lane -- and every node read from global memory (the sorted geometry and
// the boxes stay resident in L2) at the lane's own address.
template <bool FUSED>
__device__ __forceinline__ void trace_bvh_far(const Path& p, float a, bool alive, const RtParams& P, float& best, int& idx)
{
    const SieveRay sr = sieve_ray(p, a, alive);
    const BoxRay br = box_ray(p, a);
    const v4f* geom = reinterpret_cast<const v4f*>(P.geom);
    const float* gf = reinterpret_cast<const float*>(P.geom);
    const v4f* boxes = reinterpret_cast<const v4f*>(P.boxes);
    const v4f* block_boxes = boxes + (P.n_bounds >> 1) * 3;
    // group 0, the largest spheres, first (see trace_bvh)
    {
        ConstGeomPtr g0 = (ConstGeomPtr)(const void*)(P.geom);
        const v2f none2 = {__builtin_inff(), __builtin_inff()};
        uint32_t rej = 0;
#pragma unroll
        for (int k = 0; k < RT_CULL_GROUP / 2; k++) bvh_pair(rej, g0[2 * k], g0[2 * k + 1], sr, none2);
        uint32_t cand = (~rej) << (32 - RT_CULL_GROUP);
        while (__ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int i = __clz(cand);
                cand &= ~(0x80000000u >> i);
                bool far_root = false;
                if (P.materials != nullptr) far_root = P.materials[P.perm[i]].x == 3.0f;  // wave-uniform branch
                const float* f = gf + (i >> 1) * 8 + (i & 1);
                test_sphere<FUSED>(p, a, f[0], f[2], f[4], P.radius[i], i, best, idx, far_root, P.perm);
            }
        }
    }
    // superblock boxes, 4 at a time from the scalar cache: bit (31 - n) of sm[w] = superblock 32 w + n
    const int nsb = (P.n_pad + RT_BVH_SUPER - 1) / RT_BVH_SUPER;
    const int n4 = (nsb + 3) >> 2;
    ConstGeomPtr sb = (ConstGeomPtr)(const void*)(P.boxes) + ((P.n_bounds + P.n_block_boxes) >> 1) * 3;
    const float reach0 = bvh_reach(best, a);
    uint32_t sm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int quads = min(8, n4 - 8 * w);  // wave-uniform
        if (quads <= 0) continue;
        uint32_t rej = 0u;
#pragma unroll 1
        for (int n = 8 * w; n < 8 * w + quads; n++) {
            bvh_box_pair(rej, sb[6 * n], sb[6 * n + 1], sb[6 * n + 2], br, reach0);
            bvh_box_pair(rej, sb[6 * n + 3], sb[6 * n + 4], sb[6 * n + 5], br, reach0);
        }
        sm[w] = alive ? ~rej << (32 - 4 * quads) : 0u;
    }
    // nearer nodes first (see trace_bvh): groups of a block, blocks of a superblock, superblocks
    int sup_bits = 0;
    while ((1 << sup_bits) < nsb) sup_bits++;
    const uint32_t flip = bvh_flip_bits(p, P, 6 + sup_bits);
    const uint32_t flip_g = flip & 7u, flip_k = (flip >> 3) & 7u, flip_s = flip >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) sm[w] = bvh_rearrange(sm[w], flip_s & 31u, sup_bits);
    if (flip_s & 32u) {
        uint32_t t_ = sm[0]; sm[0] = sm[1]; sm[1] = t_;
        t_ = sm[2]; sm[2] = sm[3]; sm[3] = t_;
    }
    if (flip_s & 64u) {
        uint32_t t_ = sm[0]; sm[0] = sm[2]; sm[2] = t_;
        t_ = sm[1]; sm[1] = sm[3]; sm[3] = t_;
    }
    uint32_t bm = 0, gm = 0;  // blocks of the open superblock / groups of the open block still to be visited
    int sup = 0, blk = 0;
    auto test_boxes = [&](const v4f* g) -> uint32_t {  // 8 boxes = 4 pairs = 12 float4 at the lane's own address
        v4f e[12];
#pragma unroll
        for (int k = 0; k < 12; k++) e[k] = g[k];
        const float reach = bvh_reach(best, a);
        uint32_t rej = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) bvh_box_pair(rej, e[3 * k], e[3 * k + 1], e[3 * k + 2], br, reach);
        return (~rej) << 24;
    };
    for (;;) {
        // ---- lanes with nothing open below open their next superblock
        const bool open_sup = gm == 0u && bm == 0u && (sm[0] | sm[1] | sm[2] | sm[3]) != 0u;
        if (__ballot(open_sup) != 0ull) {
            if (open_sup) {
                int w = sm[0] ? 0 : sm[1] ? 1 : sm[2] ? 2 : 3;
                uint32_t word = w == 0 ? sm[0] : w == 1 ? sm[1] : w == 2 ? sm[2] : sm[3];
                const int k = __clz(word);
                word &= ~(0x80000000u >> k);
                sm[0] = w == 0 ? word : sm[0]; sm[1] = w == 1 ? word : sm[1];
                sm[2] = w == 2 ? word : sm[2]; sm[3] = w == 3 ? word : sm[3];
                sup = (32 * w + k) ^ (int)flip_s;
                bm = bvh_rearrange(test_boxes(block_boxes + sup * 12), flip_k, 3);
            }
        }
        // ---- lanes without an open group open their next block
        const bool open_blk = gm == 0u && bm != 0u;
        if (__ballot(open_blk) != 0ull) {
            if (open_blk) {
                const int k = __clz(bm);
                bm &= ~(0x80000000u >> k);
                blk = sup * (RT_BVH_SUPER / RT_BVH_BLOCK) + (k ^ (int)flip_k);
                gm = test_boxes(boxes + blk * 12);
                if (blk == 0) gm &= 0x7fffffffu;  // group 0 has been dealt with
                gm = bvh_rearrange(gm, flip_g, 3);
            }
        }
        // ---- lanes with an open group sieve its next group of spheres
        const bool leaf = gm != 0u;
        if (__ballot(leaf) == 0ull) {
            if (__ballot((bm | sm[0] | sm[1] | sm[2] | sm[3]) != 0u) == 0ull) break;
            continue;
        }
        int grp = 0;
        uint32_t cand = 0u;
        if (leaf) {
            const int k = __clz(gm);
            gm &= ~(0x80000000u >> k);
            grp = blk * (RT_BVH_BLOCK / RT_CULL_GROUP) + (k ^ (int)flip_g);
            const v4f* g = geom + grp * RT_CULL_GROUP;
            v4f e[RT_CULL_GROUP];
#pragma unroll
            for (int k2 = 0; k2 < RT_CULL_GROUP; k2++) e[k2] = g[k2];
            const float reach = bvh_reach(best, a);
            const v2f bh = {reach, reach};
            uint32_t rej = 0;
#pragma unroll
            for (int k2 = 0; k2 < RT_CULL_GROUP / 2; k2++) bvh_pair(rej, e[2 * k2], e[2 * k2 + 1], sr, bh);
            cand = (~rej) << (32 - RT_CULL_GROUP);
        }
        while (__ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int b = __clz(cand);
                cand &= ~(0x80000000u >> b);
                const int i = grp * RT_CULL_GROUP + b;
                bool far_root = false;
                if (P.materials != nullptr) far_root = P.materials[P.perm[i]].x == 3.0f;  // wave-uniform branch
                const float* f = gf + (i >> 1) * 8 + (i & 1);
                test_sphere<FUSED>(p, a, f[0], f[2], f[4], P.radius[i], i, best, idx, far_root, P.perm);
            }
        }
    }
}

// Persistent kernel.  blockDim.x is a multiple of 64 (<= RT_STREAM_MAX_THREADS); dynamic LDS =
// rt_stream_lds_bytes(): n_pad*16 B sieve geometry + n_pad*4 B radii + per mode the matrix-core
// operands or the group bounds + (blockDim.x/64) * RT_SG_CHUNKS (or RT_SG_CHUNKS_WIDE) * 64*4 B masks.
//
// MODE selects where the sieve reads the geometry from: 0 = the LDS copy (wave-uniform
// ds_read_b128), 1 = global memory through the scalar cache (s_load into SGPRs, which the
// packed instructions take directly as their one scalar operand).  The walk reads LDS either way.
// 2 = as 1, over the spatially sorted sphere list with group culling: per 512 spheres the wave
// first sieves the 64 bounding spheres of their groups of 8, ORs the result over its lanes, and
// skips the groups no lane can touch.  3 = the sieve on the matrix cores.  4 = per-lane traversal
// of the sorted list (trace_bvh).  5 = as 3 for scenes of more than RT_MAX_MFMA_SPHERES spheres: the
// sphere-side operands stay in global memory.  6 = as 4 for scenes too large for LDS (trace_bvh_far).
// shader.comp:96-125 for the ray that a lane of rt_stream_kernel has just traced, and what follows from
// the end of a path.  A macro over the kernel's locals rather than a lambda: the closure cost the
// kernel a private segment and a third of its scalar registers in spills.
#define RT_STREAM_SHADE(best_, idx_)                                                                                        \
    {                                                                                                                       \
        const uint32_t px = pxy & 0xffffu, py = pxy >> 16;                                                                  \
        nrays++;                                                                                                            \
        if (path_bounce<FUSED, RECORD>(p, P, idx_, best_, px, py, P.frame0 + (uint32_t)(j * P.frame_stride), rec, table)) { \
            if constexpr (RECORD) record_light(P, px, py, j, p);                                                            \
            /* the end of the lane's frames: the last frame, or the last of its run of a split tile */                      \
            const int jn = j + 1;                                                                                           \
            /* (a run of a split tile is split_frames frames, a power of two, from a multiple of it) */                    \
            const bool last = jn == P.n_frames || (SPLIT && park != RT_PARK_NONE && (jn & (P.split_frames - 1)) == 0);      \
            if (SPLIT && (park & 3u) == RT_PARK_LATER) { /* a run after a split tile's first: the light waits for the join */ \
                P.split_lights[(park >> 2) + (uint32_t)j * 64u] = make_float4(p.lr, p.lg, p.lb, 0.f);                       \
            } else {                                                                                                        \
                /* (the frame's clear, if one falls on it: nothing has touched the texel since the frame began) */          \
                if (j > 0 && clear_at(P, j, P.frame0 + (uint32_t)(j * P.frame_stride))) acc = make_float4(0.f, 0.f, 0.f, 0.f); \
                acc.x = p.lr + acc.x; /* vec4(light, 1.0) + accumulated */                                                  \
                acc.y = p.lg + acc.y;                                                                                       \
                acc.z = p.lb + acc.z;                                                                                       \
                acc.w = 1.0f + acc.w;                                                                                       \
                if constexpr (!RECORD) {                                                                                    \
                    if (last) {                                                                                             \
                        if (!SPLIT || park == RT_PARK_NONE)                                                                 \
                            store_pixel(P, px, py, acc);                                                                    \
                        else /* a split tile's first run: the join adds the rest */                                         \
                            P.accum[(size_t)py * P.width + px] = acc;                                                       \
                    }                                                                                                       \
                }                                                                                                           \
            }                                                                                                               \
            j = jn;                                                                                                         \
            if (last) {                                                                                                     \
                if constexpr (!RECORD) {                                                                                    \
                    if (P.tile_cost != nullptr) { /* the pixel's rays: its tile's cost and chain, for the next order */    \
                        const uint32_t t_ = ((py >> 3) - (uint32_t)P.tile_row0) * (uint32_t)P.tile_cols +                   \
                                            ((px >> 3) - (uint32_t)P.tile_col0);                                            \
                        atomicAdd(P.tile_cost + t_, nrays - pixel_rays0);                                                   \
                        /* (a run of a split tile: as many rays as the whole pixel would have taken, near enough) */        \
                        atomicMax(P.tile_chain + t_, (nrays - pixel_rays0) * (SPLIT && park != RT_PARK_NONE ? (uint32_t)P.split_ways : 1u)); \
                    }                                                                                                       \
                }                                                                                                           \
                alive = false;                                                                                              \
            } else {                                                                                                        \
                start = true;                                                                                               \
            }                                                                                                               \
        }                                                                                                                   \
    }

// The frames j .. n_frames - 1 of a pixel whose camera ray goes to the sky, all at once: every one of them is
// one Trace() call and light = sky_color * (0.15, 0.15, 0.15) (shader.comp:94-102), added frame by frame in
// the shader's order with the clears where they fall.  The pixel is stored and its lane is free.
#define RT_STREAM_SKY_FRAMES                                                                                        \
    {                                                                                                               \
        const uint32_t px = pxy & 0xffffu, py = pxy >> 16;                                                          \
        path_reset(p, P, d0x, d0y, d0z);                                                                            \
        (void)path_bounce<FUSED, false>(p, P, -1, RT_FLT_MAX, px, py, 0u, nullptr, nullptr);                        \
        const int jstop = SPLIT && park != RT_PARK_NONE ? min(P.n_frames, j + P.split_frames) : P.n_frames;         \
        const unsigned left = (unsigned)(jstop - j);                                                                \
        for (; j < jstop; j++) {                                                                                    \
            if (SPLIT && (park & 3u) == RT_PARK_LATER) {                                                            \
                P.split_lights[(park >> 2) + (uint32_t)j * 64u] = make_float4(p.lr, p.lg, p.lb, 0.f);               \
                continue;                                                                                           \
            }                                                                                                       \
            if (j > 0 && clear_at(P, j, P.frame0 + (uint32_t)(j * P.frame_stride))) acc = make_float4(0.f, 0.f, 0.f, 0.f); \
            acc.x = p.lr + acc.x;                                                                                   \
            acc.y = p.lg + acc.y;                                                                                   \
            acc.z = p.lb + acc.z;                                                                                   \
            acc.w = 1.0f + acc.w;                                                                                   \
        }                                                                                                           \
        nrays += left;                                                                                              \
        if (!SPLIT || park == RT_PARK_NONE) store_pixel(P, px, py, acc);                                            \
        if (SPLIT && park == RT_PARK_FIRST) P.accum[(size_t)py * P.width + px] = acc;                               \
        if (P.tile_cost != nullptr)                                                                                 \
            atomicAdd(P.tile_cost + ((py >> 3) - (uint32_t)P.tile_row0) * (uint32_t)P.tile_cols +                   \
                          ((px >> 3) - (uint32_t)P.tile_col0),                                                      \
                      nrays - pixel_rays0);                                                                         \
        alive = false;                                                                                              \
        start = false;                                                                                              \
    }

#define RT_CAM_UNKNOWN (-2)
// COLD instantiations keep what only the shading of a ray needs in LDS while the ray is traced, not in registers:
// ten words per lane, cold_w[k * (most threads of the instantiation) + thread] (fixed strides: the offsets go into the
// instructions).
#define RT_COLD_CR 0  // what the path has gathered (Path::cr .. lb)
#define RT_COLD_CG 1
#define RT_COLD_CB 2
#define RT_COLD_LR 3
#define RT_COLD_LG 4
#define RT_COLD_LB 5
#define RT_COLD_BOUNCE 6
#define RT_COLD_J 7
#define RT_COLD_PXY 8
#define RT_COLD_NRAYS 9
#define RT_COLD_WORDS 10

// MODE_ = the sieve (MODE below), + RT_STREAM_SPLIT for the instantiation that shares the frames of heavy tiles.
#define RT_STREAM_SPLIT 16
template <bool FUSED, bool RECORD, int MODE_>
__global__ void __launch_bounds__((MODE_ & 15) == 7 ? RT_MFMA_MAX_THREADS : RT_STREAM_MAX_THREADS,
                                   (MODE_ & 15) == 4 && !RECORD ? 3 : 1)  // (the per-lane traversal lives on its waves: three per SIMD)
    rt_stream_kernel(const RtParams P)
{
    constexpr int MODE = MODE_ & 15;
    constexpr bool SPLIT = (MODE_ & RT_STREAM_SPLIT) != 0 && !RECORD;
    constexpr bool SGEOM = MODE == 1 || MODE == 2;
    constexpr bool CULL = MODE == 2;
    constexpr bool MFMA = MODE == 3 || MODE == 5 || MODE == 7;
    constexpr bool PIPE = MODE != 7;  // 7 = 3 for workgroups of more than RT_STREAM_MAX_THREADS threads (three waves per SIMD)
    constexpr bool COLD = MODE == 7;
    constexpr int COLD_STRIDE = RT_MFMA_MAX_THREADS;  // ... which has no register to spare over the sieve
    constexpr bool FAR_FRAGS = MODE == 5;  // matrix-core sieve with its sphere-side operands left in global memory
    constexpr bool BVH = MODE == 4;
    constexpr bool BVH_FAR = MODE == 6;  // per-lane traversal out of global memory (nothing staged: P.n_lds == 0)
    constexpr bool SORTED = CULL || BVH || BVH_FAR;  // slots of the sorted list: P.perm maps them to the shader's indices
    extern __shared__ float4 smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // Geometry is staged in LDS when it fits (P.n_lds == n_pad); a scene too large for that
    // (P.n_lds == 0, SGEOM only) is read from global memory by the sieve and the walk alike.
    float* radius = reinterpret_cast<float*>(smem + P.n_lds);
    for (int i = threadIdx.x; i < P.n_lds; i += blockDim.x) {
        smem[i] = P.geom[i];
        radius[i] = P.radius[i];
    }
    // matrix-core sieve: the sphere-side operand fragments follow (128 x 16 bytes per chunk)
    uint4* frags = reinterpret_cast<uint4*>(radius + P.n_lds);
    // matrix-core sieve: the sphere-side operands; per-lane traversal: the boxes of the groups
    const int n_frag = MFMA && !FAR_FRAGS ? P.n_lds * 4 : BVH ? P.n_bounds * 3 / 2 : 0;
    for (int i = threadIdx.x; i < n_frag; i += blockDim.x)
        frags[i] = BVH ? reinterpret_cast<const uint4*>(P.boxes)[i] : P.frags[i];
    // the sphere records (centre, radius, albedo) that the shading reads, when there is room for them
    float4* lds_table = reinterpret_cast<float4*>(frags + n_frag);
    for (int i = threadIdx.x; i < 2 * P.n_table; i += blockDim.x) lds_table[i] = reinterpret_cast<const float4*>(P.spheres)[i];
    const float4* table = P.n_table ? lds_table : reinterpret_cast<const float4*>(P.spheres);
    __syncthreads();  // the only barrier: waves run independently from here on
    const int sg_chunks = !SORTED && P.sg_wide ? RT_SG_CHUNKS_WIDE : RT_SG_CHUNKS;  // chunks between candidate walks
    uint32_t* wmask = reinterpret_cast<uint32_t*>(lds_table + 2 * P.n_table) + wave * (sg_chunks * 64);
    const float* gf = reinterpret_cast<const float*>(smem);
    const int nch = P.n_pad / RT_CHUNK;
    uint32_t* cold_w = reinterpret_cast<uint32_t*>(lds_table + 2 * P.n_table) + (blockDim.x >> 6) * (sg_chunks * 64);
    (void)cold_w;

    // wave-uniform work span.  Spans of RT_SPAN pixels are dealt round-robin into RT_QUEUES
    // queues, each with a counter on a cache line of its own: atomics on one address complete one
    // at a time (~10 ns each).  A wave draws from the queue of its workgroup (workgroups go
    // round-robin over the XCDs, so that is one queue per XCD) and moves on to another queue when
    // its own runs dry.
    // SPLIT: heavy tiles (rt_tile_order_kernel counts them) are several units of work each, a run of frames per unit: a
    // tile in the crevice between two spheres bounces to the depth limit in every pixel and frame, and all its frames
    // in one lane are a chain of frames x depth passes that nothing shortens but sharing the frames.  The units are
    // listed in P.split_units (rt_tile_units_kernel), the runs of the heavy tiles first.
    const uint32_t nspans = SPLIT ? P.split_count[1] : (P.total_pixels + RT_SPAN - 1) / RT_SPAN;
    uint32_t queue = blockIdx.x % RT_QUEUES;
    // the wave's unit of work: the pixels of its tile not yet handed to a lane, the tile, SPLIT: its run of frames
    unsigned long long avail = 0ull;
    uint32_t span_tile = 0;
    int span_j0 = 0;
    uint32_t span_park = RT_PARK_NONE;
    (void)span_j0; (void)span_park;
    bool exhausted = false;
    auto queue_len = [&](uint32_t q) -> uint32_t { return (nspans + RT_QUEUES - 1 - q) / RT_QUEUES; };
    // Lane 0 asks a queue for its next span; the answer is left in a vector register so that
    // nobody waits for it before it is needed.
    auto ask = [&](uint32_t q) -> uint32_t {
        uint32_t k = 0xffffffffu;
        if (lane == 0) k = atomicAdd(P.work_counter + q * RT_QUEUE_STRIDE, 1u);
        return k;
    };
    // The next queue after `q` that still holds spans, or RT_QUEUES when all are dry: lanes
    // 0..RT_QUEUES-1 look at one counter each (one load for the wave, no atomics).
    auto next_queue = [&](uint32_t q) -> uint32_t {
        bool has = false;
        if (lane < (int)RT_QUEUES)
            has = __hip_atomic_load(P.work_counter + lane * RT_QUEUE_STRIDE, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) < queue_len((uint32_t)lane);
        const unsigned long long m = __ballot(has);
        if (m == 0ull) return RT_QUEUES;
        const unsigned long long after = m & ~((2ull << q) - 1ull);  // queues q+1 .. : first choice
        return (uint32_t)__builtin_ctzll(after ? after : m);
    };
    uint32_t ahead = 0xffffffffu;
    bool have_ahead = false;
#ifdef RT_PREFETCH_FORCE
    const bool prefetch = RT_PREFETCH_FORCE != 0;
#else
    const bool prefetch = P.prefetch != 0;
#endif

    // lane state
    bool alive = false, start = false;
    uint32_t pxy = 0;  // the lane's pixel: x | y << 16
    int j = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned nrays = 0, pixel_rays0 = 0;  // rays of this lane so far; their count when its pixel began
    uint32_t park = RT_PARK_NONE;   // SPLIT: see RT_PARK_NONE
#ifdef RT_COUNT_PASSES
    unsigned npasses = 0;  // experiment: passes over the sphere list, added to the word after the ray counter
    unsigned npasslanes = 0;  // and the lanes that held a ray in them (word 3)
#endif
    float d0x = 0.f, d0y = 0.f, d0z = 0.f;  // camera-ray direction of the lane's pixel
    // Nearest hit of the pixel's camera ray, which is the same in every frame of a dispatch (origin and
    // direction depend on camera and pixel only, shader.comp:73-90): RT_CAM_UNKNOWN until it has been
    // traced once, then the sphere index (-1: sky) and in cam_t the distance.  Only with P.cam_reuse.
    int cam = RT_CAM_UNKNOWN;
    float cam_t = RT_FLT_MAX;
    rt_record* rec = nullptr;
    Path p;
    p.ox = p.oy = p.oz = 0.f; p.dx = p.dy = p.dz = 0.f;
    p.cr = p.cg = p.cb = 0.f; p.lr = p.lg = p.lb = 0.f; p.bounce = 0;

#ifdef RT_PHASE_TIMERS
    // experiment: shader-clock cycles this wave spends in each phase of its loop, added to words
    // 3..7 of its counter slot (refill, sieve, walk, shade, iterations)
    unsigned long long t_refill = 0, t_sieve = 0, t_walk = 0, t_shade = 0, n_iter = 0, t_span = 0, n_span = 0;
    unsigned long long t_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock
    unsigned long long rt_exhausted = 0;
#define RT_MARK(acc)                                                  \
    {                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        acc += t_ - t_mark;                                           \
        t_mark = t_;                                                  \
    }
#else
#define RT_MARK(acc)
#endif
    for (;;) {
        // ---- refill: free lanes take the next pixels of the wave's span
        unsigned long long freem = __ballot(!alive);
        while (freem != 0ull && !exhausted) {
            if (avail == 0ull) {
#ifdef RT_PHASE_TIMERS
                const unsigned long long t_span0 = __builtin_amdgcn_s_memtime();
                n_span++;
#endif
                // spans q, q + RT_QUEUES, ... belong to queue q
                const uint32_t k = __builtin_amdgcn_readfirstlane(have_ahead ? ahead : ask(queue));
                have_ahead = false;
                if (k >= queue_len(queue)) {
                    queue = next_queue(queue);
                    if (queue == RT_QUEUES) {
                        exhausted = true;
                        break;
                    }
                    continue;
                }
                if (prefetch) {
                    // short spans: ask for the next one now and read the answer when this one is used up
                    ahead = ask(queue);
                    have_ahead = true;
                }
                const uint32_t slot = k * RT_QUEUES + queue;  // (a span is the 64 pixels of a tile)
                if constexpr (SPLIT) {  // the unit: a tile and a run of its frames
                    const uint4 unit = P.split_units[slot];
                    span_tile = __builtin_amdgcn_readfirstlane(unit.x);
                    span_j0 = (int)__builtin_amdgcn_readfirstlane(unit.y);
                    span_park = __builtin_amdgcn_readfirstlane(unit.w);
                } else {  // heaviest tiles of the previous dispatch first when their costs are known
                    span_tile = P.tile_order != nullptr ? __builtin_amdgcn_readfirstlane(P.tile_order[slot]) : slot;
                }
                // the pixels whose camera ray meets a sphere, when that is known: rt_sky_frames_kernel has the others
                avail = ~0ull;
                if constexpr (!RECORD) {
                    if (P.cam_live != nullptr) {
                        const unsigned long long live = P.cam_live[span_tile];
                        avail = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(live >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)live);
                    }
                }
#ifdef RT_PHASE_TIMERS
                t_span += __builtin_amdgcn_s_memtime() - t_span0;
#endif
                if (avail == 0ull) continue;
            }
            // free lane number r takes the r-th pixel left: lane p, where the r-th set bit of avail is, sends p to
            // lane r, and the r-th free lane fetches it there
            const uint32_t ntake = min((uint32_t)__popcll(freem), (uint32_t)__popcll(avail));
            const uint32_t rank = mbcnt64(freem);
            const bool took = !alive && rank < ntake;
            const uint32_t arank = mbcnt64(avail);
            const bool given = ((avail >> lane) & 1ull) != 0ull && arank < ntake;
            const unsigned long long givenm = __ballot(given);
            // (every lane sends, to a lane of its own: the others fill the places ntake .. 63)
            const int nth = __builtin_amdgcn_ds_permute((int)((given ? arank : ntake + mbcnt64(~givenm)) << 2), lane);
            const uint32_t l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank << 2), nth);
            avail &= ~givenm;
            if (took) {
                const uint32_t tile = span_tile;
                j = 0;
                if constexpr (SPLIT) {
                    j = span_j0;
                    park = span_park + ((span_park & 3u) == RT_PARK_LATER ? l << 2 : 0u);
                }
                // tile / tile_cols by multiplication: exact for tile < 2^24 (checked on the host)
                const uint32_t ty = P.tile_cols_magic ? __umulhi(tile, P.tile_cols_magic) : tile;
                const uint32_t tx = tile - ty * (uint32_t)P.tile_cols;
                const uint32_t px = ((uint32_t)P.tile_col0 + tx) * 8 + (l & 7u);
                const uint32_t py = ((uint32_t)P.tile_row0 + ty) * 8 + (l >> 3);
                pxy = px | py << 16;
                primary_direction<FUSED>(P, px, py, d0x, d0y, d0z);
                pixel_rays0 = nrays;
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!RECORD)
                    if ((!SPLIT || (park & 3u) != RT_PARK_LATER) && !clear_at(P, 0, P.frame0)) acc = P.accum[(size_t)py * P.width + px];
                alive = true;
                start = true;
                cam = RT_CAM_UNKNOWN;
                if constexpr (!RECORD) {
                    if (P.cam_hits != nullptr) {  // wave-uniform: the hit of the pixel's camera ray is known (never the sky)
                        const uint2 h = P.cam_hits[(size_t)tile * 64u + l];
                        cam = (int)h.x;
                        cam_t = __uint_as_float(h.y);
                    }
                }
            }
            freem = __ballot(!alive);
        }
        if (__ballot(alive) == 0ull) break;
        RT_MARK(t_refill)
#ifdef RT_PHASE_TIMERS
        n_iter++;
        if (exhausted && rt_exhausted == 0) rt_exhausted = __builtin_amdgcn_s_memrealtime();
#endif

        // ---- a lane at the start of a path whose camera ray's hit is known shades it at once and goes into
        // the sieve below with its first bounce ray: every trip of the sieve is for 64 rays.  (More than one
        // turn of this loop only at depth 1, where the camera ray is the whole path.)
        if constexpr (!RECORD) {
            while (__ballot(alive && start && cam >= 0) != 0ull) {
                if (alive && start && cam >= 0) {
                    path_reset(p, P, d0x, d0y, d0z);
                    start = false;
                    RT_STREAM_SHADE(cam_t, cam)
                }
            }
        }
        // ---- the others generate their camera ray
        if (start) {
            if constexpr (RECORD) rec = record_slot(P, pxy & 0xffffu, pxy >> 16, j);
            path_reset(p, P, d0x, d0y, d0z);
            start = false;
        }
        const bool tracing = alive;  // every live lane holds a ray now
        if (__ballot(tracing) == 0ull) continue;

#ifdef RT_COUNT_PASSES
        npasses++;
        npasslanes += (unsigned)__popcll(__ballot(tracing));
#endif
        // ---- Trace(): sieve + candidate walk
        if constexpr (COLD) {  // what the path has gathered so far waits in LDS: the shading is the next to look at it
            cold_w[RT_COLD_CR * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.cr);
            cold_w[RT_COLD_CG * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.cg);
            cold_w[RT_COLD_CB * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.cb);
            cold_w[RT_COLD_LR * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.lr);
            cold_w[RT_COLD_LG * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.lg);
            cold_w[RT_COLD_LB * COLD_STRIDE + threadIdx.x] = __float_as_uint(p.lb);
            cold_w[RT_COLD_BOUNCE * COLD_STRIDE + threadIdx.x] = (uint32_t)p.bounce;
            cold_w[RT_COLD_J * COLD_STRIDE + threadIdx.x] = (uint32_t)j;
            cold_w[RT_COLD_PXY * COLD_STRIDE + threadIdx.x] = pxy;
            cold_w[RT_COLD_NRAYS * COLD_STRIDE + threadIdx.x] = nrays;
        }
        const float a = dot3<FUSED>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
        float best = RT_FLT_MAX;
        int idx = -1;
        if constexpr (BVH) {
            trace_bvh<FUSED>(p, a, tracing, P, smem, radius, reinterpret_cast<const float4*>(frags), best, idx);
            RT_MARK(t_walk)
        } else if constexpr (BVH_FAR) {
            trace_bvh_far<FUSED>(p, a, tracing, P, best, idx);
            RT_MARK(t_walk)
        } else if (!SORTED && P.n_lds != 0 && __popcll(__ballot(tracing)) <= RT_STREAM_TEAM_MAX) {  // wave-uniform
            // ---- the few rays a wave is left with when the queues are dry (the mirror chains of up to max_depth
            // bounces that a dispatch ends with), one after the other with the spheres dealt to the lanes: lane l
            // evaluates shader.comp:44-60 for spheres l, l + 64, ... and the wave takes the minimum t, lowest index
            // among equal ones -- what the shader's strict `<` in index order keeps.  A pass of sieve and walk
            // costs a lone ray the same few thousand instructions as 64 of them.
            for (unsigned long long todo = __ballot(tracing); todo != 0ull; todo &= todo - 1ull) {
                const int from = (int)__builtin_ctzll(todo);
                Path q;
                q.ox = lane_value(p.ox, from); q.oy = lane_value(p.oy, from); q.oz = lane_value(p.oz, from);
                q.dx = lane_value(p.dx, from); q.dy = lane_value(p.dy, from); q.dz = lane_value(p.dz, from);
                const float qa = lane_value(a, from);
                float t = RT_FLT_MAX;  // stays there unless one of this lane's spheres is hit at 0 < t < FLT_MAX
                int hit = -1;
                for (int i = lane; i < P.sphere_count; i += 64) {
                    bool far_root = false;
                    if (P.materials != nullptr) far_root = P.materials[i].x == 3.0f;
                    const float* f = gf + (i >> 1) * 8 + (i & 1);
                    test_sphere<FUSED>(q, qa, f[0], f[2], f[4], radius[i], i, t, hit, far_root);
                }
                const uint32_t nearest = wave_min_u32(__float_as_uint(t));  // t > 0: ordered as its bits
                const uint32_t lowest = wave_min_u32(__float_as_uint(t) == nearest ? (uint32_t)hit : 0xffffffffu);
                if (lane == from && nearest != __float_as_uint(RT_FLT_MAX)) {
                    best = __uint_as_float(nearest);
                    idx = (int)lowest;
                }
            }
            RT_MARK(t_walk)
        } else {
            SieveRay sr;
            MfmaRay mr;
            if constexpr (MFMA)
                mr = mfma_ray(p, a);
            else
                sr = sieve_ray(p, a, tracing);
            for (int sg = 0; sg < nch; sg += sg_chunks) {
                const int nsg = min(sg_chunks, nch - sg);
                uint32_t summary = 0;
                // culled kernel: bit (63 - n) of `wneed` = some lane's ray may touch group n of
                // this supergroup (4 groups of 8 spheres per chunk)
                unsigned long long wneed = ~0ull;
                if constexpr (CULL) {
                    const SieveRay srb = sieve_ray(p, a, tracing, RT_CULL_MARGIN_O);
                    ConstGeomPtr bnd = (ConstGeomPtr)(const void*)(P.bounds) + sg * (RT_CHUNK / RT_CULL_GROUP);
                    uint32_t rej0, rej1 = 0xffffffffu;
                    if (nsg == RT_SG_CHUNKS) {
                        rej0 = sieve_chunk<RT_GROUP_SGPR>(bnd, srb);
                        rej1 = sieve_chunk<RT_GROUP_SGPR>(bnd + RT_CHUNK, srb);
                    } else {  // nsg chunks = nsg groups of 4 bounding spheres; the rest stay "rejected"
                        const int n0 = min(nsg, 8), n1 = nsg - n0;
                        rej0 = sieve_chunk_partial<2>(bnd, srb, n0) | (n0 < 8 ? 0xffffffffu >> (4 * n0) : 0u);
                        if (n1 > 0) rej1 = sieve_chunk_partial<2>(bnd + RT_CHUNK, srb, n1) | (0xffffffffu >> (4 * n1));
                    }
                    uint32_t need0 = ~rej0, need1 = ~rej1;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        need0 |= (uint32_t)__shfl_xor((int)need0, off, 64);
                        need1 |= (uint32_t)__shfl_xor((int)need1, off, 64);
                    }
                    wneed = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)need0) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)need1);
                }
                // (matrix-core sieve: without the material extension it also rejects what lies behind the ray)
                if constexpr (FAR_FRAGS) {
                    const v8bf* fr = reinterpret_cast<const v8bf*>(P.frags) + sg * 128;
                    summary = P.materials == nullptr
                                  ? sieve_chunks_mfma_far<RT_MFMA_AHEAD, true>(fr, nsg, lane, mr, tracing, sg + nsg == nch,
                                                                               P.last_chunk_valid, wmask)
                                  : sieve_chunks_mfma_far<RT_MFMA_AHEAD, false>(fr, nsg, lane, mr, tracing, sg + nsg == nch,
                                                                                P.last_chunk_valid, wmask);
                } else if constexpr (MFMA) {
                    const v8bf* fr = reinterpret_cast<const v8bf*>(frags) + sg * 128;
                    summary = P.materials == nullptr
                                  ? sieve_chunks_mfma<true, PIPE>(fr, nsg, lane, mr, tracing, sg + nsg == nch, P.last_chunk_valid, wmask)
                                  : sieve_chunks_mfma<false, PIPE>(fr, nsg, lane, mr, tracing, sg + nsg == nch, P.last_chunk_valid, wmask);
                } else {
                    for (int c = 0; c < nsg; c++) {
                        constexpr int GROUP = SGEOM ? RT_GROUP_SGPR : RT_GROUP_LDS;
                        const bool last = sg + c == nch - 1;
                        constexpr int PGROUP = 2;  // the ragged last chunk goes 2 pairs at a time
                        const int ngroups = (P.last_chunk_count + 2 * PGROUP - 1) / (2 * PGROUP);
                        uint32_t rej;
                        if constexpr (CULL) {
                            const uint32_t need = (uint32_t)(wneed >> (60 - 4 * c)) & 0xfu;
                            if (need == 0u) {  // nobody's candidate: the walk never reads this chunk's masks
                                summary <<= 1;
                                continue;
                            }
                            ConstGeomPtr g = (ConstGeomPtr)(const void*)(P.geom) + (sg + c) * RT_CHUNK;
                            rej = sieve_chunk_culled(g, sr, need);
                        } else if constexpr (SGEOM) {
                            ConstGeomPtr g = (ConstGeomPtr)(const void*)(P.geom) + (sg + c) * RT_CHUNK;
                            rej = last && ngroups < RT_CHUNK / 2 / PGROUP ? sieve_chunk_partial<PGROUP>(g, sr, ngroups)
                                                                         : sieve_chunk<GROUP>(g, sr);
                        } else {
                            const v4f* g = reinterpret_cast<const v4f*>(smem) + (sg + c) * RT_CHUNK;
                            rej = last && ngroups < RT_CHUNK / 2 / PGROUP ? sieve_chunk_partial<PGROUP>(g, sr, ngroups)
                                                                         : sieve_chunk<GROUP>(g, sr);
                        }
                        // Idle lanes carry -inf in the sieve constants and reject everything through
                        // the arithmetic; only the last chunk has spheres beyond sphere_count to mask.
                        uint32_t cand = ~rej;
                        if (last) cand &= P.last_chunk_valid;
                        wmask[c * 64 + lane] = cand;
                        summary = (summary << 1) | (cand != 0u ? 1u : 0u);
                    }
                    summary <<= (32 - nsg);
                }
                RT_MARK(t_sieve)
                // walk: every lane visits its own candidates in increasing sphere index
                uint32_t cur = 0;
                int curbase = 0;
                for (;;) {
                    if (cur == 0u && summary != 0u) {
                        const int k = __clz(summary);
                        summary &= ~(0x80000000u >> k);
                        cur = wmask[k * 64 + lane];
                        curbase = (sg + k) * RT_CHUNK;
                    }
                    if (__ballot(cur != 0u) == 0ull) break;
                    if (cur != 0u) {
                        const int b = __clz(cur);
                        cur &= ~(0x80000000u >> b);
                        const int i = curbase + b;
                        bool far_root = false;
                        if (P.materials != nullptr)  // wave-uniform branch
                            far_root = P.materials[CULL ? P.perm[i] : i].x == 3.0f;
                        if (P.n_lds) {
                            const float* f = gf + (i >> 1) * 8 + (i & 1);
                            test_sphere<FUSED>(p, a, f[0], f[2], f[4], radius[i], i, best, idx, far_root,
                                               CULL ? P.perm : nullptr);
                        } else {  // never the culled kernel: its scenes fit in LDS
                            const float4 g = *reinterpret_cast<const float4*>(P.spheres + i);
                            test_sphere<FUSED>(p, a, g.x, g.y, g.z, g.w, i, best, idx, far_root);
                        }
                    }
                }
                RT_MARK(t_walk)
            }
        }

        if constexpr (SORTED)
            if (idx >= 0) idx = P.perm[idx];  // slot of the sorted list -> the shader's sphere index

        // ---- shade
        if constexpr (COLD) {
            p.cr = __uint_as_float(cold_w[RT_COLD_CR * COLD_STRIDE + threadIdx.x]);
            p.cg = __uint_as_float(cold_w[RT_COLD_CG * COLD_STRIDE + threadIdx.x]);
            p.cb = __uint_as_float(cold_w[RT_COLD_CB * COLD_STRIDE + threadIdx.x]);
            p.lr = __uint_as_float(cold_w[RT_COLD_LR * COLD_STRIDE + threadIdx.x]);
            p.lg = __uint_as_float(cold_w[RT_COLD_LG * COLD_STRIDE + threadIdx.x]);
            p.lb = __uint_as_float(cold_w[RT_COLD_LB * COLD_STRIDE + threadIdx.x]);
            p.bounce = (int)cold_w[RT_COLD_BOUNCE * COLD_STRIDE + threadIdx.x];
            j = (int)cold_w[RT_COLD_J * COLD_STRIDE + threadIdx.x];
            pxy = cold_w[RT_COLD_PXY * COLD_STRIDE + threadIdx.x];
            nrays = cold_w[RT_COLD_NRAYS * COLD_STRIDE + threadIdx.x];
        }
        if (tracing) {
            // a camera ray that went through the sieve (no list for its tile): its hit serves the pixel's other frames
            const bool first = !RECORD && P.cam_reuse != 0 && p.bounce == 0;
            if (first) {
                cam = idx;
                cam_t = best;
            }
            RT_STREAM_SHADE(best, idx)
            if constexpr (!RECORD)
                if (first && idx < 0 && alive) RT_STREAM_SKY_FRAMES
        }
        RT_MARK(t_shade)
    }
#ifdef RT_PHASE_TIMERS
    if (lane == 0) {
        unsigned long long* slot =
            P.rays + ((blockIdx.x * (blockDim.x >> 6) + (unsigned)wave) % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE;
        atomicAdd(slot + 3, t_refill);
        atomicAdd(slot + 4, t_sieve);
        atomicAdd(slot + 5, t_walk);
        atomicAdd(slot + 6, t_shade);
        atomicAdd(slot + 7, n_iter);
        // wall-clock marks in the slot's words 8..: sum of wave lifetimes, sum of the time waves
        // spend after the queues ran dry, longest and shortest (stored inverted) lifetime,
        // longest time after the queues ran dry
        const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (rt_exhausted == 0) rt_exhausted = rt_end;
        atomicAdd(slot + 8, rt_end - rt_start);
        atomicAdd(slot + 9, rt_end - rt_exhausted);
        atomicMax(slot + 10, rt_end - rt_start);
        atomicMax(slot + 11, ~(rt_end - rt_start));
        atomicMax(slot + 12, rt_end - rt_exhausted);
        atomicAdd(slot + 13, 1ull);
        atomicAdd(slot + 14, t_span);
        atomicAdd(slot + 15, n_span);
    }
#endif
    // (the camera rays served by P.cam_hits are a path each: the host counts them)
    if constexpr (!RECORD) add_rays(P, nrays, blockIdx.x * (blockDim.x >> 6) + (unsigned)wave);
#ifdef RT_COUNT_PASSES
    if (lane == 0)
        atomicAdd(P.rays + ((blockIdx.x * (blockDim.x >> 6) + (unsigned)wave) % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + 1,
                  (unsigned long long)npasses);
    if (lane == 1)
        atomicAdd(P.rays + ((blockIdx.x * (blockDim.x >> 6) + (unsigned)wave) % RT_RAY_SLOTS) * RT_RAY_SLOT_STRIDE + 3,
                  (unsigned long long)npasslanes);
#endif

    // The last workgroup to get here puts the work counters back to zero for the next launch:
    // every other workgroup has already seen them run out, so nobody reads them again.  "Last"
    // is found in two steps, per queue and then over the queues, to keep the additions of
    // workgroups that finish together on different addresses.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned q = blockIdx.x % RT_QUEUES;
        const unsigned in_queue = (gridDim.x - q + RT_QUEUES - 1) / RT_QUEUES;  // workgroups of this queue
        unsigned* all_done = P.work_counter + RT_QUEUES * RT_QUEUE_STRIDE;
        unsigned* queue_done = all_done + (1 + q) * RT_QUEUE_STRIDE;
        if (atomicAdd(queue_done, 1u) == in_queue - 1) {
            const unsigned nqueues = min(gridDim.x, (unsigned)RT_QUEUES);
            if (atomicAdd(all_done, 1u) == nqueues - 1)
                for (unsigned i = 0; i <= 2 * RT_QUEUES; i++) atomicExch(P.work_counter + i * RT_QUEUE_STRIDE, 0u);
        }
    }
}

// ---------------------------------------------------------------- sieve audit

// Test kernel: for caller-supplied rays, runs the sieve exactly as rt_stream_kernel does and
// compares every bit with the shader's literal discriminant under both float profiles.
// out[0] / out[1]: spheres the sieve rejected although the STRICT / FUSED discriminant was not
// negative (must stay 0); out[2]: spheres with a non-negative STRICT discriminant; out[3]: spheres
// the sieve let through.  The matrix-core sieve without the material extension also rejects what
// lies behind the ray: there "discriminant not negative" reads "discriminant not negative and
// b < 0", what test_sphere goes on to take the root of.
// One lane per ray; blockDim.x = 256; dynamic LDS = n_pad*16 bytes.
__global__ void __launch_bounds__(256) rt_sieve_audit_kernel(const RtParams P, const float* rays, int n_rays,
                                                            unsigned long long* out)
{
    extern __shared__ float4 smem[];
    const bool mfma = P.frags != nullptr;  // audit the matrix-core sieve instead
    const bool far = mfma && P.n_pad > RT_MAX_MFMA_SPHERES;  // fragments read from global memory, as MODE 5 does
    const bool behind = mfma && P.materials == nullptr;      // as rt_stream_kernel chooses
    uint4* frags = reinterpret_cast<uint4*>(smem + P.n_pad);
    for (int i = threadIdx.x; i < P.n_pad; i += blockDim.x) smem[i] = P.geom[i];
    for (int i = threadIdx.x; i < (mfma && !far ? P.n_pad * 4 : 0); i += blockDim.x) frags[i] = P.frags[i];
    __syncthreads();
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = ray < n_rays;
    Path p;
    p.ox = live ? rays[6 * ray + 0] : 0.f; p.oy = live ? rays[6 * ray + 1] : 0.f; p.oz = live ? rays[6 * ray + 2] : 0.f;
    p.dx = live ? rays[6 * ray + 3] : 1.f; p.dy = live ? rays[6 * ray + 4] : 0.f; p.dz = live ? rays[6 * ray + 5] : 0.f;
    unsigned bad_s = 0, bad_f = 0, n_exact = 0, n_sieve = 0;
    const float a_s = dot3<false>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
    const float a_f = dot3<true>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
    const SieveRay sr = sieve_ray(p, a_s, true);
    const MfmaRay mr = mfma_ray(p, a_s);
    for (int ch = 0; ch < P.n_pad / RT_CHUNK; ch++) {
        const v8bf* fr = reinterpret_cast<const v8bf*>(far ? P.frags : frags) + ch * 128;
        const uint32_t rej = behind ? sieve_chunk_mfma<true>(fr, threadIdx.x & 63, mr)
                             : mfma ? sieve_chunk_mfma<false>(fr, threadIdx.x & 63, mr)
                                    : sieve_chunk<RT_GROUP_LDS>(reinterpret_cast<const v4f*>(smem) + ch * RT_CHUNK, sr);
        for (int n = 0; n < RT_CHUNK; n++) {
            const int i = ch * RT_CHUNK + n;
            if (i >= P.sphere_count || !live) break;
            const float4 g = *reinterpret_cast<const float4*>(P.spheres + i);
            const float ocx = p.ox - g.x, ocy = p.oy - g.y, ocz = p.oz - g.z;
            const float bs = 2.0f * dot3<false>(ocx, ocy, ocz, p.dx, p.dy, p.dz);
            const float cs = mad<false>(-g.w, g.w, dot3<false>(ocx, ocy, ocz, ocx, ocy, ocz));
            const float ds = mad<false>(bs, bs, -((4.0f * a_s) * cs));
            const float bf = 2.0f * dot3<true>(ocx, ocy, ocz, p.dx, p.dy, p.dz);
            const float cf = mad<true>(-g.w, g.w, dot3<true>(ocx, ocy, ocz, ocx, ocy, ocz));
            const float df = mad<true>(bf, bf, -((4.0f * a_f) * cf));
            const bool rejected = (rej >> (31 - n)) & 1u;
            if (!(ds < 0.0f) && (bs < 0.0f || !behind)) { n_exact++; if (rejected) bad_s++; }
            if (!(df < 0.0f) && (bf < 0.0f || !behind) && rejected) bad_f++;
            if (!rejected) n_sieve++;
        }
    }
    if (bad_s) atomicAdd(out + 0, (unsigned long long)bad_s);
    if (bad_f) atomicAdd(out + 1, (unsigned long long)bad_f);
    if (n_exact) atomicAdd(out + 2, (unsigned long long)n_exact);
    if (n_sieve) atomicAdd(out + 3, (unsigned long long)n_sieve);
}

// Test kernel for the culled stream kernel: one lane per ray evaluates the group test exactly as
// the kernel does (per lane, before the OR over the wave) and, for every group it would cull,
// the literal discriminant of the group's spheres under both float profiles.
// out[0]: (ray, sphere) pairs culled with a non-negative discriminant (must stay 0);
// out[1]: (ray, group) pairs culled; out[2]: (ray, group) pairs tested; out[3], out[4]: (ray, node)
// pairs the per-lane traversal's box test drops / examines, over groups, blocks and superblocks (a node dropped
// although a member has a literal hit counts into out[0]); out[5]: spheres with a literal hit that
// the traversal's distance test would prune, itself or by the box of its group or block, when the
// nearest hit so far is that very hit (must stay 0); out[6]: literal hits so examined.
// blockDim.x = 256.
__global__ void __launch_bounds__(256) rt_cull_audit_kernel(const RtParams P, const float* rays, int n_rays,
                                                           unsigned long long* out)
{
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = ray < n_rays;
    Path p;
    p.ox = live ? rays[6 * ray + 0] : 0.f; p.oy = live ? rays[6 * ray + 1] : 0.f; p.oz = live ? rays[6 * ray + 2] : 0.f;
    p.dx = live ? rays[6 * ray + 3] : 1.f; p.dy = live ? rays[6 * ray + 4] : 0.f; p.dz = live ? rays[6 * ray + 5] : 0.f;
    unsigned bad = 0, culled = 0, tested = 0;
    const float a_s = dot3<false>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
    const float a_f = dot3<true>(p.dx, p.dy, p.dz, p.dx, p.dy, p.dz);
    const SieveRay srb = sieve_ray(p, a_s, true, RT_CULL_MARGIN_O);
    const int ngroups = P.n_pad / RT_CULL_GROUP;
    for (int g0 = 0; g0 < ngroups; g0 += RT_CHUNK) {  // 32 bounding spheres at a time
        const uint32_t rej = sieve_chunk<RT_GROUP_SGPR>((ConstGeomPtr)(const void*)(P.bounds) + g0, srb);
        for (int n = 0; n < RT_CHUNK && g0 + n < ngroups && live; n++) {
            const int first = (g0 + n) * RT_CULL_GROUP;
            if (first >= P.sphere_count) break;
            tested++;
            if (!((rej >> (31 - n)) & 1u)) continue;
            culled++;
            for (int k = first; k < first + RT_CULL_GROUP && k < P.sphere_count; k++) {
                const float4 g = *reinterpret_cast<const float4*>(P.spheres + P.perm[k]);
                const float ocx = p.ox - g.x, ocy = p.oy - g.y, ocz = p.oz - g.z;
                const float bs = 2.0f * dot3<false>(ocx, ocy, ocz, p.dx, p.dy, p.dz);
                const float cs = mad<false>(-g.w, g.w, dot3<false>(ocx, ocy, ocz, ocx, ocy, ocz));
                const float ds = mad<false>(bs, bs, -((4.0f * a_s) * cs));
                const float bf = 2.0f * dot3<true>(ocx, ocy, ocz, p.dx, p.dy, p.dz);
                const float cf = mad<true>(-g.w, g.w, dot3<true>(ocx, ocy, ocz, ocx, ocy, ocz));
                const float df = mad<true>(bf, bf, -((4.0f * a_f) * cf));
                if (!(ds < 0.0f) || !(df < 0.0f)) bad++;
            }
        }
    }
    // The boxes of the per-lane traversal, groups and blocks alike, evaluated as trace_bvh does
    // (no hit yet: bh = +inf).  A box is also dropped when it lies behind the origin, so what
    // counts here is a hit: either root positive, the far one being what a dielectric sphere may take.
    unsigned bculled = 0, btested = 0;
    const BoxRay br_s = box_ray(p, a_s), br_f = box_ray(p, a_f);
    const v4f* box = reinterpret_cast<const v4f*>(P.boxes);
    const int nblk = (P.n_pad + RT_BVH_BLOCK - 1) / RT_BVH_BLOCK;
    auto box_rejects = [&](int entry, const BoxRay& br, float bh) {
        uint32_t m = 0;
        bvh_box_pair(m, box[(entry >> 1) * 3], box[(entry >> 1) * 3 + 1], box[(entry >> 1) * 3 + 2], br, bh);
        return ((m >> (1 - (entry & 1))) & 1u) != 0u;
    };
    const int nsup = (P.n_pad + RT_BVH_SUPER - 1) / RT_BVH_SUPER;
    for (int node = 0; node < ngroups + nblk + nsup && live; node++) {
        const int level = node < ngroups ? 0 : node < ngroups + nblk ? 1 : 2;
        const int at = node - (level == 0 ? 0 : level == 1 ? ngroups : ngroups + nblk);
        const int size = level == 0 ? RT_CULL_GROUP : level == 1 ? RT_BVH_BLOCK : RT_BVH_SUPER;
        const int first = at * size;
        if (first >= P.sphere_count) continue;
        const int entry = level == 0 ? at : level == 1 ? P.n_bounds + at : P.n_bounds + P.n_block_boxes + at;
        btested++;
        const bool rej_s = box_rejects(entry, br_s, __builtin_inff()), rej_f = box_rejects(entry, br_f, __builtin_inff());
        if (!rej_s && !rej_f) continue;
        bculled++;
        for (int k = first; k < first + size && k < P.sphere_count; k++) {
            const float4 g = *reinterpret_cast<const float4*>(P.spheres + P.perm[k]);
            float ts = RT_FLT_MAX, tf = RT_FLT_MAX;
            int is = -1, jf = -1;
            test_sphere<false>(p, a_s, g.x, g.y, g.z, g.w, k, ts, is, true);
            test_sphere<true>(p, a_f, g.x, g.y, g.z, g.w, k, tf, jf, true);
            if ((rej_s && is >= 0) || (rej_f && jf >= 0)) bad++;
        }
    }
    // The distance test of the per-lane traversal: with "best" set to a sphere's own literal t,
    // neither the sphere (bvh_pair) nor the box of its group or of its block (bvh_box_pair) may be
    // rejected -- the sharpest case of "nothing as near as the hit found so far is pruned", a tie.
    // Both float profiles.
    unsigned pruned = 0, hits = 0;
    const SieveRay sr_s = sieve_ray(p, a_s, true), sr_f = sieve_ray(p, a_f, true);
    const v4f* geom = reinterpret_cast<const v4f*>(P.geom);
    for (int k = 0; k < P.sphere_count && live; k++) {
        const float4 g = *reinterpret_cast<const float4*>(P.spheres + P.perm[k]);
        float ts = RT_FLT_MAX, tf = RT_FLT_MAX;
        int is = -1, jf = -1;
        const bool far = (k & 1) != 0;  // every other sphere as a dielectric one would be tested
        test_sphere<false>(p, a_s, g.x, g.y, g.z, g.w, k, ts, is, far);
        test_sphere<true>(p, a_f, g.x, g.y, g.z, g.w, k, tf, jf, far);
        for (int prof = 0; prof < 2; prof++) {
            if ((prof ? jf : is) < 0) continue;
            hits++;
            const float reach = bvh_reach(prof ? tf : ts, prof ? a_f : a_s);
            const v2f bh = {reach, reach};
            uint32_t m = 0;
            bvh_pair(m, geom[(k >> 1) * 2], geom[(k >> 1) * 2 + 1], prof ? sr_f : sr_s, bh);
            bool rejected = ((m >> (1 - (k & 1))) & 1u) != 0u;
            rejected = rejected || box_rejects(k / RT_CULL_GROUP, prof ? br_f : br_s, reach);
            rejected = rejected || box_rejects(P.n_bounds + k / RT_BVH_BLOCK, prof ? br_f : br_s, reach);
            rejected = rejected || box_rejects(P.n_bounds + P.n_block_boxes + k / RT_BVH_SUPER, prof ? br_f : br_s, reach);
            if (rejected) pruned++;
        }
    }
    if (pruned) atomicAdd(out + 5, (unsigned long long)pruned);
    if (hits) atomicAdd(out + 6, (unsigned long long)hits);
    if (bad) atomicAdd(out + 0, (unsigned long long)bad);
    if (culled) atomicAdd(out + 1, (unsigned long long)culled);
    if (tested) atomicAdd(out + 2, (unsigned long long)tested);
    if (bculled) atomicAdd(out + 3, (unsigned long long)bculled);
    if (btested) atomicAdd(out + 4, (unsigned long long)btested);
}

// ------------------------------------------------------------------- resolve

// colour = unorm8(accum.rgb / divisor) over the rendered region (shader.comp:129-132).
__global__ void rt_resolve_kernel(const float4* accum, uint32_t* color, int width, int x_max,
                                  int y0, int y1, float divisor)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = y0 + blockIdx.y;
    if (x >= x_max || y >= y1) return;
    const size_t at = (size_t)y * width + x;
    const float4 a = accum[at];
    color[at] = unorm8(a.x / divisor) | (unorm8(a.y / divisor) << 8) | (unorm8(a.z / divisor) << 16) |
                0xff000000u;
}

// ---------------------------------------------------------------- math probe

// Device arithmetic exposed for the parity tests: out[i] = op(a[i], b[i], c[i]).
__global__ void rt_math_probe_kernel(int op, const float* a, const float* b, const float* c,
                                     float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.f;
    switch (op) {
        case 0: r = sqrtf(a[i]); break;
        case 1: r = a[i] / b[i]; break;
        case 2: r = __builtin_fmaf(a[i], b[i], c[i]); break;
        case 3: r = a[i] * b[i] + c[i]; break;  // must stay two roundings
        case 4: r = hash_to_float(__float_as_uint(a[i])); break;
        case 5: r = 1.0f / sqrtf(a[i]); break;
        case 6: r = __uint_as_float(unorm8(a[i])); break;
        case 7: r = __uint_as_float(wang_hash(__float_as_uint(a[i]))); break;
    }
    out[i] = r;
}

// ------------------------------------------------------------------ launchers

template <bool FUSED, bool RECORD, int MODE>
static hipError_t launch_stream(const RtParams& P, int blocks, int threads, size_t lds, hipStream_t st)
{
    if constexpr (!RECORD && (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 5 || MODE == 7)) {
        if (P.split_count != nullptr) {  // heavy tiles in runs of frames
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_stream_kernel<FUSED, RECORD, MODE + RT_STREAM_SPLIT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((rt_stream_kernel<FUSED, RECORD, MODE + RT_STREAM_SPLIT>), dim3(blocks), dim3(threads), lds, st, P);
            return hipGetLastError();
        }
    }
    if (P.split_count != nullptr) return hipErrorInvalidValue;  // no such instantiation
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_stream_kernel<FUSED, RECORD, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rt_stream_kernel<FUSED, RECORD, MODE>), dim3(blocks), dim3(threads), lds, st, P);
    return hipGetLastError();
}

template <bool FUSED, bool RECORD>
static hipError_t launch_tile(const RtParams& P, hipStream_t st)
{
    hipLaunchKernelGGL((rt_tile_kernel<FUSED, RECORD>), dim3(P.tile_cols, P.tile_rows), dim3(64), 0, st, P);
    return hipGetLastError();
}

size_t rt_stream_lds_bytes(int n_lds, int threads, bool mfma, int extra_float4, bool sg_wide, int n_table)
{
    return (size_t)n_lds * (sizeof(float4) + sizeof(float) + (mfma ? 4 * sizeof(uint4) : 0)) +
           (size_t)extra_float4 * sizeof(float4) + (size_t)n_table * 2 * sizeof(float4) +
           (size_t)(threads / 64) * (sg_wide ? RT_SG_CHUNKS_WIDE : RT_SG_CHUNKS) * 64 * sizeof(uint32_t) +
           // the instantiation for three waves per SIMD keeps ten words of every lane here while the lane's ray is traced
           (mfma && threads > RT_STREAM_MAX_THREADS ? (size_t)RT_MFMA_MAX_THREADS * RT_COLD_WORDS * sizeof(uint32_t) : 0);
}

// mode: 0 = sieve from LDS, 1 = sieve through the scalar cache, 2 = culled (P.bounds/P.perm set, P.n_lds != 0),
// 3 = sieve on the matrix cores (P.frags set, P.n_lds != 0), 4 = per-lane traversal (P.boxes/P.perm/P.n_bounds
// set, P.n_lds != 0), 5 = as 3 with the operands read from global memory (P.frags set, P.n_lds != 0),
// 6 = per-lane traversal out of global memory (P.boxes/P.perm/P.n_bounds/P.n_block_boxes set)
hipError_t rt_launch_stream(const RtParams& P, bool fused, bool record, int mode, int blocks, int threads,
                            hipStream_t st)
{
    if (threads < 64 || threads % 64 != 0 || threads > (mode == 3 ? RT_MFMA_MAX_THREADS : RT_STREAM_MAX_THREADS))
        return hipErrorInvalidConfiguration;  // beyond the instantiation's launch bounds
    const bool sorted = mode == 2 || mode == 4 || mode == 6;
    size_t lds = rt_stream_lds_bytes(P.n_lds, threads, mode == 3, mode == 4 ? P.n_bounds * 3 / 2 : 0, !sorted && P.sg_wide, P.n_table);
    static const size_t pad = (size_t)rt_dev_knob("RT_DEBUG_LDS_PAD", 0);  // extra LDS per workgroup, to cap workgroups per CU
    lds += pad;
    if (mode == 2) {
        if (record)
            return fused ? launch_stream<true, true, 2>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 2>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 2>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 2>(P, blocks, threads, lds, st);
    }
    if (mode == 6) {
        if (record)
            return fused ? launch_stream<true, true, 6>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 6>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 6>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 6>(P, blocks, threads, lds, st);
    }
    if (mode == 5) {
        if (record)
            return fused ? launch_stream<true, true, 5>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 5>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 5>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 5>(P, blocks, threads, lds, st);
    }
    if (mode == 4) {
        if (record)
            return fused ? launch_stream<true, true, 4>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 4>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 4>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 4>(P, blocks, threads, lds, st);
    }
    if (mode == 3 && threads > RT_STREAM_MAX_THREADS) {  // three waves per SIMD: the instantiation that keeps within 168 registers
        if (record)
            return fused ? launch_stream<true, true, 7>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 7>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 7>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 7>(P, blocks, threads, lds, st);
    }
    if (mode == 3) {
        if (record)
            return fused ? launch_stream<true, true, 3>(P, blocks, threads, lds, st)
                         : launch_stream<false, true, 3>(P, blocks, threads, lds, st);
        return fused ? launch_stream<true, false, 3>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 3>(P, blocks, threads, lds, st);
    }
    if (P.n_lds == 0) mode = 1;  // nothing staged: the sieve must read global memory
    if (record && P.n_lds)  // the debug twin takes the LDS path when there is one
        return fused ? launch_stream<true, true, 0>(P, blocks, threads, lds, st)
                     : launch_stream<false, true, 0>(P, blocks, threads, lds, st);
    if (record)
        return fused ? launch_stream<true, true, 1>(P, blocks, threads, lds, st)
                     : launch_stream<false, true, 1>(P, blocks, threads, lds, st);
    if (mode == 1)
        return fused ? launch_stream<true, false, 1>(P, blocks, threads, lds, st)
                     : launch_stream<false, false, 1>(P, blocks, threads, lds, st);
    return fused ? launch_stream<true, false, 0>(P, blocks, threads, lds, st)
                 : launch_stream<false, false, 0>(P, blocks, threads, lds, st);
}

hipError_t rt_launch_cull_audit(const RtParams& P, const float* rays, int n_rays, unsigned long long* out,
                                hipStream_t st)
{
    hipLaunchKernelGGL(rt_cull_audit_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, st, P, rays, n_rays, out);
    return hipGetLastError();
}

#ifdef RT_COUNT_PASSES
void rt_bvh_step_counts(unsigned long long out[4])
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_count), 32);
    const unsigned long long zero[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_count), zero, 32);
}
#endif

size_t rt_lite_lds_bytes() { return 3 * 2 * RT_CHUNK * sizeof(float4); }

// LDS for the lights that the runs of the split tiles of a workgroup of `waves` waves hand to their first run.
size_t rt_lite_handover_bytes(unsigned waves, int ways, int split_frames)
{
    return (size_t)(waves - waves / (unsigned)ways) * (size_t)split_frames * 192u * sizeof(float);
}

hipError_t rt_launch_camera_hits(const RtParams& P, bool fused, uint2* hits, unsigned long long* live, hipStream_t st)
{
    const unsigned tiles = (unsigned)P.tile_cols * (unsigned)P.tile_rows;
    if (fused)
        hipLaunchKernelGGL((rt_camera_hits_kernel<true>), dim3((tiles + 3) / 4), dim3(256), 0, st, P, hits, live);
    else
        hipLaunchKernelGGL((rt_camera_hits_kernel<false>), dim3((tiles + 3) / 4), dim3(256), 0, st, P, hits, live);
    return hipGetLastError();
}

hipError_t rt_launch_sky_frames(const RtParams& P, bool fused, hipStream_t st)
{
    const unsigned tiles = (unsigned)P.tile_cols * (unsigned)P.tile_rows;
    const unsigned blocks = std::min((tiles + 3u) / 4u, 2048u);
    if (fused)
        hipLaunchKernelGGL((rt_sky_frames_kernel<true>), dim3(blocks), dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL((rt_sky_frames_kernel<false>), dim3(blocks), dim3(256), 0, st, P);
    return hipGetLastError();
}

hipError_t rt_launch_tile_order(const uint32_t* cost, const uint32_t* chain, uint32_t* order, unsigned n, unsigned nwaves,
                                unsigned heavy_den, hipStream_t st)
{
    hipLaunchKernelGGL(rt_tile_order_kernel, dim3(1), dim3(1024), 0, st, cost, chain, order, n, nwaves, heavy_den);
    return hipGetLastError();
}

hipError_t rt_launch_tile_units(uint32_t* order, uint4* units, unsigned n, unsigned cap, int ways, int frames, int n_frames,
                                hipStream_t st)
{
    const unsigned most = n + cap * (unsigned)(ways - 1);
    hipLaunchKernelGGL(rt_tile_units_kernel, dim3((most + 255) / 256), dim3(256), 0, st, order, units, n, cap, (unsigned)ways,
                       (unsigned)frames, (unsigned)n_frames);
    return hipGetLastError();
}

hipError_t rt_launch_stream_join(const RtParams& P, hipStream_t st)
{
    // (a wave per tile that may have been split; those beyond the order kernel's count return at once)
    hipLaunchKernelGGL(rt_lite_join_kernel, dim3((P.split_tiles + 3) / 4), dim3(256), 0, st, P);
    return hipGetLastError();
}

hipError_t rt_launch_lite(const RtParams& P, bool fused, int threads, hipStream_t st)
{
    // one wave per tile, split_ways for each of the first P.split_tiles tiles of the order
    const unsigned tiles = (unsigned)P.tile_cols * (unsigned)P.tile_rows +
                           (P.split_tiles > 0 ? (unsigned)P.split_tiles * (unsigned)(P.split_ways - 1) : 0u);
    unsigned per = (unsigned)threads / 64u;
    size_t lds = rt_lite_lds_bytes();
    if (P.split_lds) {  // the runs of a tile in one workgroup, their lights beside the geometry
        per = std::max(per, (unsigned)P.split_ways);
        threads = (int)per * 64;
        lds += rt_lite_handover_bytes(per, P.split_ways, P.split_frames);
    }
    const dim3 grid((tiles + per - 1) / per), block(threads);
    const bool many = P.n_frames != 1 || P.split_tiles > 0;
    if (fused && many)
        hipLaunchKernelGGL((rt_lite_kernel<true, true>), grid, block, lds, st, P);
    else if (fused)
        hipLaunchKernelGGL((rt_lite_kernel<true, false>), grid, block, lds, st, P);
    else if (many)
        hipLaunchKernelGGL((rt_lite_kernel<false, true>), grid, block, lds, st, P);
    else
        hipLaunchKernelGGL((rt_lite_kernel<false, false>), grid, block, lds, st, P);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && P.split_tiles > 0 && !P.split_lds) {
        hipLaunchKernelGGL(rt_lite_join_kernel, dim3((P.split_tiles + 3) / 4), dim3(256), 0, st, P);
        e = hipGetLastError();
    }
    return e;
}

hipError_t rt_launch_tile(const RtParams& P, bool fused, bool record, hipStream_t st)
{
    if (fused) return record ? launch_tile<true, true>(P, st) : launch_tile<true, false>(P, st);
    return record ? launch_tile<false, true>(P, st) : launch_tile<false, false>(P, st);
}

hipError_t rt_launch_resolve(const float4* accum, uint32_t* color, int width, int x_max, int y0,
                             int y1, float divisor, hipStream_t st)
{
    if (x_max <= 0 || y1 <= y0) return hipSuccess;
    dim3 grid((x_max + 255) / 256, y1 - y0);
    hipLaunchKernelGGL(rt_resolve_kernel, grid, dim3(256), 0, st, accum, color, width, x_max, y0, y1,
                       divisor);
    return hipGetLastError();
}

hipError_t rt_launch_math_probe(int op, const float* a, const float* b, const float* c, float* out,
                                int n, hipStream_t st)
{
    hipLaunchKernelGGL(rt_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, op, a, b, c, out, n);
    return hipGetLastError();
}

hipError_t rt_launch_sieve_audit(const RtParams& P, const float* rays, int n_rays, unsigned long long* out,
                                 hipStream_t st)
{
    const size_t lds = (size_t)P.n_pad * (sizeof(float4) + (P.frags && P.n_pad <= RT_MAX_MFMA_SPHERES ? 4 * sizeof(uint4) : 0));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_sieve_audit_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rt_sieve_audit_kernel, dim3((n_rays + 255) / 256), dim3(256), lds, st, P, rays, n_rays, out);
    return hipGetLastError();
}
