/*
 * rt_oracle.h -- CPU oracle for the path-trace kernel.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of the reference compute shader
 * (res/shaders/shader.comp:20-63,72-133 and res/shaders/def.glsl:1-62) plus
 * the host-side camera code that feeds it (includes/scene.hpp:106-116,
 * src/application.cpp:10-23).  Nothing under raytracing_amd/ may link, load or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker or the reported baseline.
 *
 * PINNING STATUS.  The reference ships no tests or fixtures -- one screenshot is all the
 * output it holds -- and cannot be built or run in this environment (Vulkan + MSVC only), so:
 *   - the integer RNG stream (wang_hash / seed) is pinned by hand-evaluated
 *     known-answer vectors (tests/golden/rng_kat.json);
 *   - the host vec3 helpers are pinned against the glm 1.0.0 headers vendored
 *     in the reference (oracle/glm_probe.cpp -> oracle/_ref/glm_probe);
 *   - the one rendered output the reference ships, thumbnail.png (a screenshot at
 *     an unrecorded camera pose), pins what in it does not depend on the pose: the
 *     converged sky pixel and two sphere surfaces that mirror the sky
 *     (tests/golden/thumbnail_sky.json), i.e. sky * 0.15, contribution *= albedo,
 *     the /(frameCount+1) resolve and the rounding of the rgba8 store;
 *   - at the camera pose recovered for that screenshot (tests/golden/thumbnail_pose.json,
 *     tools/recover_thumbnail_pose.py) this oracle reproduces it to PSNR 27.2 dB per
 *     pixel, 33.5 dB after 8x box filtering, 77 % of the pixels equal: image tolerance;
 *   - at the bit level the floating-point path of the shader stays PARITY UNPINNED: no
 *     output of the reference fixes the rounding.  GLSL leaves FMA contraction,
 *     inversesqrt precision and uint->float rounding open; this file fixes
 *     them in writing (see "float profiles" below).
 *
 * FLOAT PROFILES.  Every `a*b + c` that appears in the shader source is written
 * MAD(a,b,c) here.  Two profiles give it a meaning:
 *   RTO_PROFILE_STRICT : MAD(a,b,c) = round(round(a*b) + c)   (no contraction)
 *   RTO_PROFILE_FUSED  : MAD(a,b,c) = fmaf(a,b,c)             (every mul+add contracted)
 * Both are legal GLSL evaluations of the same source.  Everything else is
 * identical in both: dot() sums x->y->z left to right, normalize(v) =
 * v * (1.0f / sqrtf(dot(v,v))), reflect(I,N) = I - (2*dot(N,I))*N, IEEE-correct
 * sqrt and divide, uint->float round-to-nearest-even, subnormals kept.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* includes/scene.hpp:31-36 (std430 Sphere, def.glsl:1-5): 32 bytes. */
typedef struct rto_sphere {
    float center[3];
    float radius;
    float color[3];
    float padding;
} rto_sphere;

/* includes/scene.hpp:22-29 (std140 UBO, shader.comp:11-18): 80 bytes. */
typedef struct rto_ubo {
    float camera_forward[3];  float _pad0;
    float camera_right[3];    float _pad1;
    float camera_up[3];       float _pad2;
    float camera_position[3];
    int32_t  sphereCount;     /* offset 60 */
    uint32_t frameCount;      /* offset 64 */
    uint32_t _pad3[3];
} rto_ubo;

enum { RTO_PROFILE_STRICT = 0, RTO_PROFILE_FUSED = 1 };

/* When the accumulation texel is zeroed before a frame is added. */
enum {
    RTO_CLEAR_REFERENCE = 0, /* iff frameCount == 1 (shader.comp:79-81)          */
    RTO_CLEAR_FIRST     = 1, /* before the first frame of this call              */
    RTO_CLEAR_NEVER     = 2
};

/* One record per (frame, pixel, bounce) of a traced crop. */
typedef struct rto_record {
    int32_t  hit;      /* sphere index; -1 = miss (path ends); -2 = bounce not reached */
    uint32_t hash[3];  /* wang_hash of seeds s, s+1, s+1920 (0 when hit < 0)           */
} rto_record;

typedef struct rto_job {
    int width, height;        /* image size; only floor(W/8)*8 x floor(H/8)*8 is rendered
                                 (compute_pipeline.cpp:429-432)                          */
    int tile_row0, tile_row1; /* rows of 8x8 tiles [row0,row1) to render; row1<=0 = all  */
    int max_depth;            /* shader.comp:95 hard-codes 50                            */
    int profile;              /* RTO_PROFILE_*                                           */
    int n_frames;             /* frames rendered by this call                            */
    int frame_stride;         /* frame j uses frameCount = ubo.frameCount + j*stride     */
    int clear_mode;           /* RTO_CLEAR_*                                             */
    int n_threads;            /* host threads (<=1: run on the calling thread)          */
    const float* materials;   /* material extension: {kind, param} per sphere, or NULL    */
    float const_roughness;    /* > 0: the constant-roughness variant the shader keeps commented
                                 out (shader.comp:101: Material(sphere.color, <constant>)), used
                                 only to explain the reference's screenshot; 0 = the shipped rule
                                 (shader.comp:99-100, roughness = rand * 0.02)                  */
} rto_job;

/* def.glsl:40-48 */
uint32_t rto_wang_hash(uint32_t seed);
/* def.glsl:52-54, the seed before hashing (uint32 wrap-around arithmetic). */
uint32_t rto_seed(uint32_t px, uint32_t py, int32_t frame, int32_t bounce);
/* def.glsl:55: float(hash) / 4294967296.0 */
float rto_hash_to_float(uint32_t h);

/*
 * Run job->n_frames dispatches of shader.comp main() (shader.comp:72-133).
 * accum  : width*height float4, read-modify-written (binding 1).
 * color  : width*height rgba8, written with the last frame's resolve (binding 0);
 *          may be NULL.
 * n_buffer_spheres is the length of `spheres` (Scene::spheres().size()); the
 * loop bound is ubo->sphereCount (shader.comp:42).
 * Returns 0, or a negative value on bad arguments.  *rays_out (may be NULL)
 * receives the number of Trace() calls made.
 */
int rto_render(const rto_ubo* ubo, const rto_sphere* spheres, int n_buffer_spheres,
               const rto_job* job, float* accum, uint8_t* color, uint64_t* rays_out);

/*
 * Trace a crop [x0,x0+w) x [y0,y0+h) for job->n_frames frames and record, per
 * (frame j, pixel, bounce), the hit index and the three RNG hashes.  out holds
 * n_frames*h*w*max_depth records, index ((j*h + (y-y0))*w + (x-x0))*max_depth + bounce.
 * light_out (may be NULL) receives n_frames*h*w*3 floats: the per-frame `light`.
 */
int rto_trace_crop(const rto_ubo* ubo, const rto_sphere* spheres, int n_buffer_spheres,
                   const rto_job* job, int x0, int y0, int w, int h,
                   rto_record* out, float* light_out);

/* {cos, sin} of 2*pi*u, u in [0,1], by a fixed fmaf sequence (material extension). */
void rto_sincos_2pi(float u, float out[2]);

/* rgba8 store of shader.comp:132: clamp to [0,1] (NaN -> 0), floor(v*255 + 0.5). */
uint8_t rto_unorm8(float v);

/* src/application.cpp:10-23 -- forward/right/up from yaw, pitch in degrees. */
void rto_camera_basis(float yaw_deg, float pitch_deg, float fwd[3], float right[3], float up[3]);

/*
 * includes/scene.hpp:106-116 -- one Scene::update(dt).  state = {acceleration[3],
 * velocity[3]}; updates ubo->camera_position and ubo->frameCount in place.
 */
void rto_scene_update(rto_ubo* ubo, float accel[3], float vel[3], float dt);

#ifdef __cplusplus
}
#endif
#endif
