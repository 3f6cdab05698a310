/*
 * rt_oracle.c -- CPU oracle for the path-trace kernel.  TEST INFRASTRUCTURE ONLY;
 * see rt_oracle.h for what it restates, how it is pinned ("parity unpinned" for
 * the floating-point path) and who may call it.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  With
 * contraction off the only fused operations are the explicit fmaf() calls of
 * the FUSED profile.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- RNG: res/shaders/def.glsl:40-56 ------------------------------------ */

uint32_t rto_wang_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

uint32_t rto_seed(uint32_t px, uint32_t py, int32_t frame, int32_t bounce)
{
    /* The row stride is the literal 1920 and the frame stride 1920*1080 whatever
       the image size is (def.glsl:52-54); all arithmetic wraps mod 2^32. */
    return px + 1920u * py + (uint32_t)frame * 1920u * 1080u +
           (uint32_t)bounce * 1920u * 1080u * 256u;
}

float rto_hash_to_float(uint32_t h)
{
    /* uint->float rounds to nearest even, so 1.0 is reachable; the division by
       2^32 is exact. */
    return (float)h / 4294967296.0f;
}

uint8_t rto_unorm8(float v)
{
    if (!(v > 0.0f)) return 0; /* also NaN */
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)floorf(v * 255.0f + 0.5f);
}

/* sin and cos of 2*pi*u for u in [0,1]: quadrant reduction + the classic single-precision
   minimax polynomials on [-pi/4, pi/4], every step an explicit fmaf, so the GPU kernels
   (which evaluate the same sequence) give the same bits.  out = {cos, sin}. */
void rto_sincos_2pi(float u, float out[2])
{
    float k = floorf(__builtin_fmaf(u, 4.0f, 0.5f));
    float f = __builtin_fmaf(k, -0.25f, u);
    float th = f * 6.283185307179586f;
    float z = th * th;
    float sp = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = __builtin_fmaf(sp * z, th, th);
    float cp = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                              4.166664568298827e-2f);
    float cn = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    switch ((int)k & 3) {
        case 0: out[0] = cn;  out[1] = sn;  break;
        case 1: out[0] = -sn; out[1] = cn;  break;
        case 2: out[0] = -cn; out[1] = -sn; break;
        default: out[0] = sn; out[1] = -cn; break;
    }
}

/* ---- the two float profiles --------------------------------------------- */

#define PFX(n) strict_##n
#define MAD(a, b, c) (((a) * (b)) + (c))
#include "rt_oracle_path.inc"
#undef PFX
#undef MAD

#define PFX(n) fused_##n
#define MAD(a, b, c) __builtin_fmaf((a), (b), (c))
#include "rt_oracle_path.inc"
#undef PFX
#undef MAD

typedef int (*shade_fn)(const rto_ubo*, const rto_sphere*, int, int, int, int, int, int32_t,
                        float*, rto_record*, const float*, float);

/* One row of 8x8 tiles, every frame of the job, for one float profile.  The
   fused copy is cloned for CPUs with an FMA unit so fmaf() is one instruction
   there and a libm call elsewhere; both give the same bits. */
#define DEFINE_ROW_FN(NAME, SHADE, ATTR)                                                        \
    ATTR static uint64_t NAME(const rto_ubo* ubo, const rto_sphere* spheres, const rto_job* job, \
                              float* accum, uint8_t* color, int tile_row)                        \
    {                                                                                            \
        const int W = job->width, H = job->height;                                               \
        const int xmax = (W / 8) * 8;                                                            \
        uint64_t rays = 0;                                                                       \
        for (int y = tile_row * 8; y < tile_row * 8 + 8; y++)                                    \
            for (int x = 0; x < xmax; x++) {                                                     \
                float* acc = accum + 4 * ((size_t)y * W + x);                                    \
                for (int j = 0; j < job->n_frames; j++) {                                        \
                    int32_t frame = (int32_t)(ubo->frameCount + (uint32_t)(j * job->frame_stride)); \
                    int clear = job->clear_mode == RTO_CLEAR_REFERENCE ? frame == 1              \
                              : job->clear_mode == RTO_CLEAR_FIRST     ? j == 0 : 0;             \
                    if (clear) acc[0] = acc[1] = acc[2] = acc[3] = 0.0f; /* shader.comp:79-81 */ \
                    float light[3];                                                              \
                    rays += (uint64_t)SHADE(ubo, spheres, W, H, job->max_depth, x, y, frame,    \
                                            light, NULL, job->materials, job->const_roughness); \
                    /* shader.comp:126-128: vec4(light, 1.0) + accumulated */                    \
                    acc[0] = light[0] + acc[0];                                                  \
                    acc[1] = light[1] + acc[1];                                                  \
                    acc[2] = light[2] + acc[2];                                                  \
                    acc[3] = 1.0f + acc[3];                                                      \
                    if (color && j == job->n_frames - 1) {                                       \
                        float div = (float)(frame + 1); /* shader.comp:129 */                    \
                        uint8_t* px = color + 4 * ((size_t)y * W + x);                           \
                        px[0] = rto_unorm8(acc[0] / div);                                        \
                        px[1] = rto_unorm8(acc[1] / div);                                        \
                        px[2] = rto_unorm8(acc[2] / div);                                        \
                        px[3] = 255;                                                             \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
        return rays;                                                                             \
    }

DEFINE_ROW_FN(row_strict, strict_shade_pixel, )
DEFINE_ROW_FN(row_fused, fused_shade_pixel, __attribute__((target_clones("fma", "default"))))

typedef struct worker {
    const rto_ubo* ubo;
    const rto_sphere* spheres;
    const rto_job* job;
    float* accum;
    uint8_t* color;
    int row0, row1;
    int* next_row; /* shared, advanced with an atomic add */
    uint64_t rays;
} worker;

static void* worker_main(void* arg)
{
    worker* w = (worker*)arg;
    for (;;) {
        int row = __atomic_fetch_add(w->next_row, 1, __ATOMIC_RELAXED);
        if (row >= w->row1) break;
        w->rays += w->job->profile == RTO_PROFILE_FUSED
                       ? row_fused(w->ubo, w->spheres, w->job, w->accum, w->color, row)
                       : row_strict(w->ubo, w->spheres, w->job, w->accum, w->color, row);
    }
    return NULL;
}

static int check_job(const rto_ubo* ubo, int n_buffer_spheres, const rto_job* job)
{
    if (!ubo || !job) return -1;
    if (job->width <= 0 || job->height <= 0 || job->max_depth < 0 || job->n_frames < 0) return -2;
    if (ubo->sphereCount < 0 || ubo->sphereCount > n_buffer_spheres) return -3;
    if (job->profile != RTO_PROFILE_STRICT && job->profile != RTO_PROFILE_FUSED) return -4;
    return 0;
}

int rto_render(const rto_ubo* ubo, const rto_sphere* spheres, int n_buffer_spheres,
               const rto_job* job, float* accum, uint8_t* color, uint64_t* rays_out)
{
    int rc = check_job(ubo, n_buffer_spheres, job);
    if (rc) return rc;
    if (!accum) return -5;
    int rows = job->height / 8;
    int row0 = job->tile_row0 > 0 ? job->tile_row0 : 0;
    int row1 = job->tile_row1 > 0 && job->tile_row1 < rows ? job->tile_row1 : rows;
    int next = row0;
    int nt = job->n_threads > 1 ? job->n_threads : 1;
    if (nt > 256) nt = 256;
    worker ws[256];
    pthread_t th[256];
    for (int t = 0; t < nt; t++) {
        worker w = { ubo, spheres, job, accum, color, row0, row1, &next, 0 };
        ws[t] = w;
    }
    for (int t = 1; t < nt; t++)
        if (pthread_create(&th[t], NULL, worker_main, &ws[t]) != 0) return -6;
    worker_main(&ws[0]);
    uint64_t rays = ws[0].rays;
    for (int t = 1; t < nt; t++) {
        pthread_join(th[t], NULL);
        rays += ws[t].rays;
    }
    if (rays_out) *rays_out = rays;
    return 0;
}

int rto_trace_crop(const rto_ubo* ubo, const rto_sphere* spheres, int n_buffer_spheres,
                   const rto_job* job, int x0, int y0, int w, int h, rto_record* out,
                   float* light_out)
{
    int rc = check_job(ubo, n_buffer_spheres, job);
    if (rc) return rc;
    if (!out || x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > job->width || y0 + h > job->height)
        return -5;
    shade_fn shade = job->profile == RTO_PROFILE_FUSED ? fused_shade_pixel : strict_shade_pixel;
    for (int j = 0; j < job->n_frames; j++) {
        int32_t frame = (int32_t)(ubo->frameCount + (uint32_t)(j * job->frame_stride));
        for (int y = y0; y < y0 + h; y++)
            for (int x = x0; x < x0 + w; x++) {
                size_t p = ((size_t)j * h + (size_t)(y - y0)) * w + (size_t)(x - x0);
                float light[3];
                shade(ubo, spheres, job->width, job->height, job->max_depth, x, y, frame, light,
                      out + p * (size_t)job->max_depth, job->materials, job->const_roughness);
                if (light_out) memcpy(light_out + 3 * p, light, sizeof light);
            }
    }
    return 0;
}

/* ---- host-side camera code ---------------------------------------------- */

static void host_normalize(float v[3])
{
    /* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
    float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

static void host_cross(const float x[3], const float y[3], float out[3])
{
    out[0] = x[1] * y[2] - y[1] * x[2];
    out[1] = x[2] * y[0] - y[2] * x[0];
    out[2] = x[0] * y[1] - y[0] * x[1];
}

void rto_camera_basis(float yaw_deg, float pitch_deg, float fwd[3], float right[3], float up[3])
{
    /* glm::radians(d) = d * 0.01745329251994329576923690768489f */
    const float k = 0.01745329251994329576923690768489f;
    float yaw = yaw_deg * k, pitch = pitch_deg * k;
    fwd[0] = cosf(yaw) * cosf(pitch);
    fwd[1] = sinf(pitch);
    fwd[2] = sinf(yaw) * cosf(pitch);
    host_normalize(fwd);
    const float world_up[3] = { 0.0f, 1.0f, 0.0f };
    host_cross(fwd, world_up, right);
    host_normalize(right);
    host_cross(right, fwd, up);
    host_normalize(up);
}

void rto_scene_update(rto_ubo* ubo, float accel[3], float vel[3], float dt)
{
    ubo->frameCount++;
    float old[3] = { ubo->camera_position[0], ubo->camera_position[1], ubo->camera_position[2] };
    float diff[3];
    for (int k = 0; k < 3; k++) {
        vel[k] += accel[k] * dt;
        ubo->camera_position[k] += vel[k] * dt;
        vel[k] = vel[k] * 0.98f;
        accel[k] = accel[k] * 0.98f;
        diff[k] = old[k] - ubo->camera_position[k];
    }
    float len = sqrtf(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    if (len > 0.01f) ubo->frameCount = 1;
}
