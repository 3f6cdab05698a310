/*
 * rt_hip.h -- C ABI of libraytracing_amd.so: the HIP (gfx950) backend for the
 * path-trace compute kernel and the scene model that feeds it.
 *
 * The reference has no FFI; its seam for this path is the C++ class boundary
 * Engine -> ComputePipeline plus the shader's four descriptor bindings.  Each
 * entry point below names the reference interface it stands in for
 * (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success and a negative RT_E_* code on failure;
 *     rt_last_error() returns the message of the calling thread's last failure
 *     (the reference throws std::runtime_error(msg) instead:
 *     src/engine/src/compute_pipeline.cpp:88-104, main.cpp:5-11);
 *   - one rt_context per GPU, one in-order HIP stream per context, a context is
 *     not thread-safe; host buffers passed in are copied before the call returns;
 *   - there is no CPU fallback: without a HIP device rt_create fails.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

/* Raised whenever a struct below grows, an array argument changes its length or an entry point
   changes its meaning; rt_abi_version() returns the value the library was built with, so that a
   binding compiled against another header can refuse to go on.  No reference counterpart.
   3: rt_counters has four words, rt_debug_cull_audit writes seven, rt_set_option /
      rt_wait_stream / rt_stream_wait exist, kernel value 7 is gone. */
#define RT_ABI_VERSION 3
int rt_abi_version(void);

enum {
    RT_OK = 0,
    RT_E_INVALID = -1,  /* bad argument                              */
    RT_E_HIP = -2,      /* a HIP runtime call failed                 */
    RT_E_STATE = -3,    /* call order (e.g. dispatch before upload)  */
    RT_E_IO = -4,       /* file could not be read or written         */
    RT_E_PARSE = -5,    /* scene file does not match the schema      */
    RT_E_NOMEM = -6
};

/* Meaning of every `a*b + c` in the shader source (GLSL leaves it open). */
enum {
    RT_PROFILE_STRICT = 0, /* multiply, round, add, round            */
    RT_PROFILE_FUSED = 1   /* one fused multiply-add                 */
};

/* Which kernel runs a dispatch.  All give bit-identical images.  RT_KERNEL_STREAM is the product: it
   picks by scene size.  RT_KERNEL_STREAM_CULLED and RT_KERNEL_BVH are the opt-in spatial hierarchy.  The
   others force one of the kernels that RT_KERNEL_STREAM chooses among, or an older formulation of
   it, and exist as cross-checks for the parity tests (TILE: the shader's own mapping, no sieve; STREAM_SGPR,
   STREAM_LDS: the vector-ALU sieve, the latter being what rt_dispatch_static runs; STREAM_MFMA and
   LITE: the default's choices forced outside their range). */
enum {
    RT_KERNEL_STREAM = 0, /* the default: brute force, the kernel chosen by sphere count -- up to 64
                             traced spheres LITE, from 65 to 65536 persistent waves (a lane
                             whose path ends takes the next pixel) with the sieve on the matrix
                             cores, beyond that with the vector-ALU sieve; camera rays are
                             traced against a per-tile list of the spheres they can meet
                             (128 B of device memory per 8x8 tile, rt_counters.list_rays)      */
    RT_KERNEL_TILE = 1,   /* one wave per 8x8 tile, one lane per pixel (the shader's mapping) */
    RT_KERNEL_STREAM_SGPR = 2, /* STREAM, sieve geometry read through the scalar cache (SGPRs) */
    RT_KERNEL_STREAM_LDS = 3,  /* STREAM, sieve geometry read from the LDS copy                */
    RT_KERNEL_STREAM_CULLED = 4, /* STREAM over a spatially sorted copy of the sphere list: groups of 8
                                  spheres whose bounding sphere no ray of the wave can touch are
                                  skipped.  Not brute force any more (the step the reference's
                                  readme.md:38 names, a spatial hierarchy); same bits.  Scenes of fewer
                                  than 97 or more than 4096 traced spheres run RT_KERNEL_STREAM.    */
    RT_KERNEL_STREAM_MFMA = 5, /* STREAM with the sieve's two dot products (rays x spheres, K = 4) evaluated by the
                                  matrix cores on exact three-term bf16 splits of the f32 operands; the vector
                                  ALU keeps 2 instructions per test.  RT_KERNEL_STREAM picks it by itself from
                                  65 to 65536 traced spheres; this value forces it from 1 sphere on. */
    RT_KERNEL_LITE = 6,        /* at most 64 traced spheres (else STREAM): one wave per 8x8 tile as in TILE, no work
                                  queues; the camera rays of a tile traced as a bundle, bounce rays with STREAM's
                                  sieve and candidate walk; from the second dispatch of a region on the heaviest
                                  tiles first, and with 16 and more frames per dispatch the frames of the heavier
                                  tiles shared among 2 to 8 waves and joined in frame order (up to 1 GiB of device
                                  memory, allocated on first use).  RT_KERNEL_STREAM picks it for such scenes. */
    /* 7: a kernel of an earlier version (path-by-path work distribution for small scenes), removed when LITE
          learnt to share a tile's frames among waves; the value stays unused */
    RT_KERNEL_BVH = 8          /* STREAM_CULLED's sorted sphere list walked as a hierarchy by every lane on its
                                  own: blocks of 64 slots and their groups of 8 behind axis-aligned boxes, the
                                  spheres of a group behind the sieve (readme.md:38, "BVH implementation");
                                  same bits.  97 to 65536 traced spheres (else STREAM); above 4096 the nodes are
                                  read from global memory and a level of 512-slot superblocks comes on top. */
};

/* When the accumulation texel is zeroed. */
enum {
    RT_CLEAR_REFERENCE = 0, /* iff frameCount == 1 (res/shaders/shader.comp:79-81) */
    RT_CLEAR_FIRST = 1,     /* before the first frame of a dispatch                */
    RT_CLEAR_NEVER = 2
};

/* Layout of the records written by rt_trace_crop. */
typedef struct rt_record {
    int32_t hit;      /* sphere index, -1 = miss, -2 = bounce not reached */
    uint32_t hash[3]; /* wang_hash(s), (s+1), (s+1920); 0 when hit < 0     */
} rt_record;

typedef struct rt_counters {
    uint64_t rays;         /* Trace() calls (res/shaders/shader.comp:96) since the last reset */
    uint64_t sphere_tests; /* sum over dispatches of rays x that dispatch's sphereCount         */
    uint64_t paths;        /* pixel-frames rendered                                             */
    uint64_t list_rays;    /* of `rays`: camera rays that were not traced against the whole sphere list --
                              traced against the candidate list of their 8x8 tile (the spheres a camera
                              ray of the tile can meet), or served by the hit their pixel's camera ray
                              found in an earlier frame of the dispatch (RT_OPTION_CAMERA_SHORTCUTS);
                              their share of `sphere_tests` was never evaluated                      */
} rt_counters;

/* ------------------------------------------------------------------ backend */

/* ComputePipeline::ComputePipeline (src/engine/src/compute_pipeline.cpp:11-22):
   selects the device, creates the stream and the W x H accumulation (rgba32f) and
   colour (rgba8) images of createAccumulationImage (:178-269). */
int rt_create(int device, int width, int height, rt_context** out);

/* ComputePipeline::~ComputePipeline (compute_pipeline.cpp:24-50). */
int rt_destroy(rt_context* ctx);

/* ComputePipeline::windowResized (compute_pipeline.cpp:52-65): new accumulation
   image of the new size, contents undefined until a clearing frame.  The caller
   resets its frame counter, as the reference does at :64.  Buffers bound with
   rt_bind_buffers are unbound. */
int rt_resize(rt_context* ctx, int width, int height);

/* Sphere half of ComputePipeline::updateScene (compute_pipeline.cpp:494-495):
   `count` 32-byte AoS spheres (includes/scene.hpp:31-36) are copied and repacked
   into the device geometry / albedo streams.  Call again only when spheres change. */
int rt_upload_spheres(rt_context* ctx, const void* spheres_aos32, int count);

/* Material extension.  No reference counterpart: the shader has one surface response
   (res/shaders/shader.comp:105-118) and its Sphere has no material field; the sampling
   primitive is the shader's unused random_unit_vector (res/shaders/def.glsl:65-73).
   kind_param holds {kind, param} per uploaded sphere: kind 0 = the reference response,
   1 = lambertian, 2 = metal (param = fuzz), 3 = dielectric (param = index of refraction).
   count must equal the uploaded sphere count; count 0 (or NULL) switches the extension
   off.  rt_upload_spheres also switches it off. */
int rt_upload_materials(rt_context* ctx, const float* kind_param, int count);

/* UBO half of updateScene (compute_pipeline.cpp:492-493): the 80-byte
   UniformBufferObject (includes/scene.hpp:22-29; sphereCount at byte 60,
   frameCount at byte 64). */
int rt_set_uniforms(rt_context* ctx, const void* ubo80);

/* The literal 50 of res/shaders/shader.comp:95 as a parameter (default 50). */
int rt_set_max_depth(rt_context* ctx, int max_depth);

/* Float profile (default RT_PROFILE_STRICT) and kernel (default RT_KERNEL_STREAM).  No
   reference counterpart: the shader source leaves the first open and has one kernel. */
int rt_set_profile(rt_context* ctx, int profile);
int rt_set_kernel(rt_context* ctx, int kernel);

/* Options of a context.  No reference counterpart: the shader's Trace() always walks the whole
   sphere list (res/shaders/shader.comp:42).
     RT_OPTION_CAMERA_SHORTCUTS (default 1): a pixel's camera ray is the same ray in every frame
       (origin and direction depend on camera and pixel only, shader.comp:73-90), and the camera rays
       of an 8x8 tile leave one point within half a degree of one another.  With 1 the default kernels
       (RT_KERNEL_STREAM and what it picks) trace camera rays against a per-tile list of the spheres
       that any of them can meet, and find a pixel's camera hit once per dispatch for all its frames;
       rt_counters.list_rays counts the Trace() calls so served.  With 0 every Trace() call of every
       frame evaluates the whole sphere list.  Same bits either way. */
enum { RT_OPTION_CAMERA_SHORTCUTS = 0 };
int rt_set_option(rt_context* ctx, int option, int value);

/* Rows [row0,row1) of 8x8 tiles this context renders (screen-tile split across
   GPUs); row1 <= 0 restores the whole image.  The reference always dispatches
   floor(W/8) x floor(H/8) groups (compute_pipeline.cpp:429-432). */
int rt_set_tile_rows(rt_context* ctx, int row0, int row1);

/* Frame schedule of a multi-frame dispatch: frame j uses
   frameCount = ubo.frameCount + j*stride (sample split across GPUs).  No reference
   counterpart: Scene::update advances frameCount by one (includes/scene.hpp:107). */
int rt_set_frame_schedule(rt_context* ctx, int frame_stride, int clear_mode);

/* vkCmdDispatch of shader.comp main() (compute_pipeline.cpp:429-432), asynchronous
   on the context's stream.  rt_dispatch_frames(n) gives the same bits as n
   dispatches with frameCount, frameCount+stride, ... but keeps each texel in
   registers across the n frames. */
int rt_dispatch(rt_context* ctx);
int rt_dispatch_frames(rt_context* ctx, int n_frames);

/* vkWaitForFences of Engine::render (src/engine/engine.cpp:48). */
int rt_sync(rt_context* ctx);

/* Copies of binding 0 (rgba8, W*H*4 bytes) and binding 1 (rgba32f, W*H*16 bytes)
   (res/shaders/shader.comp:6-7); both synchronise the stream first. */
int rt_read_color(rt_context* ctx, uint8_t* rgba8);
int rt_read_accum(rt_context* ctx, float* rgba32f);
int rt_write_accum(rt_context* ctx, const float* rgba32f);

/* Device addresses of the two images, for collectives that reduce or gather them
   in place (no reference counterpart: the reference is single-GPU). */
int rt_device_buffers(rt_context* ctx, void** accum_dev, void** color_dev);

/* Use caller-owned device memory (W*H*16 / W*H*4 bytes) for the images; NULL
   returns to the context's own allocations.  The nearest reference counterpart is
   the per-frame rebinding of the colour image to the acquired swap-chain image
   (compute_pipeline.cpp:589-655). */
int rt_bind_buffers(rt_context* ctx, void* accum_dev, void* color_dev);
/* The same without waiting for the stream: the images take effect with the NEXT dispatch, a
   dispatch in flight keeps the images it was launched with.  For two colour images used in turn,
   as the reference's two frames in flight use their two descriptor sets
   (src/engine/includes/config.hpp:12, compute_pipeline.cpp:499-513). */
int rt_bind_buffers_next(rt_context* ctx, void* accum_dev, void* color_dev);

/* Waits until the dispatch issued `back` dispatches ago has finished: 0 = the last one, 1 = the
   one before it (then the last one may still be running).  back = 1 is the reference's
   vkWaitForFences on the fence of the frame slot it is about to reuse, two frames in flight
   (src/engine/engine.cpp:48,132; config.hpp:12).  Returns at once when there is no such dispatch. */
int rt_sync_dispatch(rt_context* ctx, int back);

/* Ordering against another HIP stream of the same device without blocking the host -- the
   semaphores between the reference's queue submissions (src/engine/engine.cpp:83-84,95-96,104-105)
   and the per-slot fence of its frames in flight (engine.cpp:48,132), for callers that read or
   write the images with work of their own (a copy, an RCCL send or reduce on the collective's stream).
   rt_wait_stream: whatever is dispatched on the context AFTER this call starts only when everything
     queued on hip_stream BEFORE this call has finished (the image a transfer still reads is not
     overwritten by the next frame);
   rt_stream_wait: whatever is queued on hip_stream AFTER this call starts only when everything
     dispatched on the context BEFORE this call has finished (a transfer does not leave before its
     frame is complete).
   hip_stream is a hipStream_t; NULL is the default stream. */
int rt_wait_stream(rt_context* ctx, void* hip_stream);
int rt_stream_wait(rt_context* ctx, void* hip_stream);

/* Device memory that the other processes of a node can map, for the screen-tile split: every GPU's
   kernel writes its band of rows straight into the display GPU's colour image over xGMI instead of
   sending it there afterwards.  No reference counterpart (the reference is one process, one GPU);
   the nearest thing is the swap-chain image every frame is written into (compute_pipeline.cpp:589-655).
     rt_ipc_alloc  allocates `bytes` on `device` (zeroed) and fills handle64 with the 64 bytes that
                   another process passes to rt_ipc_open;
     rt_ipc_open   maps the allocation behind handle64 into this process for kernels on `device`
                   (the same device as the owner's, or a peer of it); the pointer is good for
                   rt_bind_buffers / rt_bind_buffers_next;
     rt_ipc_close  unmaps it (every opener, before the owner frees); rt_ipc_free frees the owner's.
   What the runtime guarantees: a kernel's writes into a mapped peer allocation are in the
   owner's memory once the writer's host has seen that kernel complete (rt_sync, rt_sync_dispatch). */
#define RT_IPC_HANDLE_BYTES 64
int rt_ipc_alloc(int device, size_t bytes, void** dev_ptr, void* handle64);
int rt_ipc_open(int device, const void* handle64, void** dev_ptr);
int rt_ipc_close(int device, void* dev_ptr);
int rt_ipc_free(int device, void* dev_ptr);

/* Re-resolve the colour image from the accumulation image with an explicit
   divisor: colour = unorm8(accum.rgb / divisor)  (shader.comp:129-132), used
   after a cross-GPU sum. */
int rt_resolve(rt_context* ctx, float divisor);

/* Debug twin of rt_dispatch_frames for a crop: traces [x0,x0+w) x [y0,y0+h) and
   writes n_frames*h*w*max_depth records (index ((j*h+y)*w+x)*max_depth+bounce) and,
   if light is non-NULL, n_frames*h*w*3 floats of per-frame `light`.  Does not
   touch the images.  No reference counterpart (the reference has no test hooks). */
int rt_trace_crop(rt_context* ctx, int n_frames, int x0, int y0, int w, int h,
                  rt_record* records, float* light);

/* Ray counters, summed in the kernels (no reference counterpart). */
int rt_get_counters(rt_context* ctx, rt_counters* out);
int rt_reset_counters(rt_context* ctx);

/* HIP-event timer on the context's stream: begin/end bracket any number of
   dispatches; rt_timer_end synchronises and returns the elapsed milliseconds
   (no reference counterpart; its FPS readout is commented out, src/application.cpp:124-133). */
int rt_timer_begin(rt_context* ctx);
int rt_timer_end(rt_context* ctx, float* ms);

/* rt_dispatch_frames bracketed by a HIP event pair of its own, so that the device time of every
   dispatch can be had without a host synchronisation between them; rt_timed_total synchronises
   the stream and returns the summed device time (ms) and the number of timed dispatches since
   it was last called.  No reference counterpart. */
int rt_dispatch_frames_timed(rt_context* ctx, int n_frames);
int rt_timed_total(rt_context* ctx, float* ms, int* dispatches);

/* Launch geometry of the stream kernel (0 = choose from the device; threads_per_block a multiple of
   64 up to 1024, of which the kernel uses at most 512).  No reference counterpart: the shader's is
   fixed at local_size 8x8 (res/shaders/shader.comp:2). */
int rt_set_launch(rt_context* ctx, int blocks, int threads_per_block);
int rt_get_info(rt_context* ctx, int* device, int* cu_count, int* blocks, int* threads);
/* Where the last stream-kernel dispatch evaluated its sieve: 0 vector ALU with geometry from LDS,
   1 vector ALU with geometry through the scalar cache, 2 the culled variant, 3 the matrix cores,
   4 the lite kernel (vector ALU, one wave per tile), 6 the per-lane traversal; -1 when the last dispatch ran the tile kernel or nothing has been dispatched.
   No reference counterpart. */
int rt_get_sieve_mode(rt_context* ctx, int* mode);
/* Test hook: how the last stream-kernel dispatch shared the frames of heavy tiles (image tiles whose pixels
   bounce long in every frame) among several units of work: out = {units per heavy tile, frames per unit, most
   tiles it would have split, tiles it did split}; all 0 when it did not.  Synchronises.  No reference
   counterpart (the shader's invocations are scheduled by the driver). */
int rt_debug_split_info(rt_context* ctx, int out[4]);

/* Test hook: out[i] = op(a[i], b[i], c[i]) evaluated on the device with the kernels'
   compile flags, so the parity tests can pin the GPU's float semantics against the
   CPU's.  op: 0 sqrt(a), 1 a/b, 2 fma(a,b,c), 3 a*b+c in two roundings, 4 float(bits
   of a as uint)/2^32, 5 1/sqrt(a), 6 unorm8(a) as bits, 7 wang_hash(bits of a) as bits.
   No reference counterpart. */
int rt_debug_math(rt_context* ctx, int op, const float* a, const float* b, const float* c,
                  float* out, int n);

/* Test hook: runs the stream kernel's conservative sieve on n_rays caller-supplied rays
   (6 floats each: origin xyz, direction xyz) against the uploaded spheres and compares
   it with the shader's literal discriminant.  out[0], out[1]: spheres the sieve rejected
   although the discriminant under the STRICT / FUSED profile was not negative (must be
   0); out[2]: spheres with a non-negative STRICT discriminant; out[3]: spheres the sieve
   passed.  The matrix-core sieve, when no materials are uploaded, also rejects spheres
   behind the ray's origin: for it "discriminant not negative" reads "discriminant not
   negative and b < 0" (what the shader goes on to take the root of) in out[0..2].
   No reference counterpart. */
int rt_debug_sieve_audit(rt_context* ctx, const float* rays, int n_rays, uint64_t out[4]);

/* Test hook for RT_KERNEL_STREAM_CULLED and RT_KERNEL_BVH: runs their culling tests on n_rays
   caller-supplied rays: the former's bounding spheres of the sorted list's groups of 8 slots, the
   latter's boxes of those groups and of the blocks of 64.  out[0]: (ray, sphere) pairs whose group
   or block was dropped although the sphere can be hit (bounding spheres: literal discriminant not
   negative; boxes: a positive root; either float profile) -- must be 0; out[1], out[2]: (ray, group)
   pairs culled / tested by the bounding spheres; out[3], out[4]: (ray, node) pairs dropped /
   examined by the boxes; out[5]: literal hits that RT_KERNEL_BVH's distance test would prune (the
   sphere itself, its group's or its block's box) when the nearest hit so far is that very hit --
   must be 0; out[6]: literal hits so examined.  No reference counterpart. */
int rt_debug_cull_audit(rt_context* ctx, const float* rays, int n_rays, uint64_t out[7]);

/* Test hook: the next stream-kernel launch of this context is not made; that dispatch returns
   RT_E_HIP and leaves the work-queue counters as a kernel that died half way would.  The
   dispatch after it must render correctly.  No reference counterpart. */
int rt_debug_fail_next_launch(rt_context* ctx);

/* One dispatch that allocates nothing.  No counterpart in the reference; it exists for callers
   that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc.
   (It is not made for HIP graph capture: it queries and sets the current device, sets a function
   attribute before its launch and checks hipGetLastError, none of which a capturing stream
   allows, and it takes the uniform block by value, so a replay would render the same frame.)
   Same bits as
       rt_create + rt_upload_spheres + rt_set_uniforms + rt_set_max_depth + rt_set_profile +
       rt_bind_buffers + rt_dispatch_frames(n_frames)
   with the defaults of a new context (whole image, frame stride 1, RT_CLEAR_REFERENCE), at the
   speed of RT_KERNEL_STREAM_LDS above 64 spheres (the vector-ALU sieve, no camera-ray hits kept
   between calls, spans in image order: about half the rate of a context's default kernel on a
   484-sphere scene), because the device memory a context would own (work-queue counters, ray
   counters, repacked sieve geometry) is a static device array inside the library, about 128 KiB
   per GPU, held until the process ends.  Inputs and outputs are the caller's:
     spheres_aos32_dev  DEVICE pointer to n_spheres 32-byte spheres (includes/scene.hpp:31-36),
                        n_spheres <= RT_STATIC_MAX_SPHERES (larger scenes are rejected with
                        RT_E_INVALID; use a context for them); no material extension;
     ubo80              host pointer to the 80-byte uniform block (sphereCount <= n_spheres);
     accum_dev, color_dev  DEVICE images, width*height*16 and width*height*4 bytes;
     rays_dev           optional DEVICE pointer: receives the number of Trace() calls of this call;
     hip_stream         a hipStream_t (NULL = the default stream); the call is asynchronous on it.
   It calls no hipMalloc/hipFree.  All calls on one GPU share that GPU's array: calls that
   overlap on one GPU (different streams or threads) race and must be serialised by the caller. */
#define RT_STATIC_MAX_SPHERES 4096
int rt_dispatch_static(int device, void* hip_stream, const void* spheres_aos32_dev, int n_spheres,
                       const void* ubo80, int width, int height, int n_frames, int max_depth,
                       int profile, void* accum_dev, void* color_dev, uint64_t* rays_dev);

const char* rt_last_error(void);

/* -------------------------------------------------------------- scene model */

/* Scene::Scene / Scene::reloadScene (src/scene.cpp:9-15,48-55): parse a scene
   file in the reference's YAML schema (includes/scene.hpp:38-97).  sphereCount
   comes from camera[4]; frameCount is forced to 0 (scene.hpp:75). */
int rt_scene_load(const char* path, rt_scene** out);
int rt_scene_parse(const char* text, size_t len, rt_scene** out);
/* Scene::save (src/scene.cpp:57-72): block-style YAML, camera row position first. */
int rt_scene_save(const rt_scene* scene, const char* path);
int rt_scene_free(rt_scene* scene);

/* Scene::spheres() / Scene::camera() (includes/scene.hpp:104-105): the sphere array
   (count x 32 bytes) and the 80-byte uniform block, mutable like the public members
   m_spheres / m_camera (scene.hpp:121-122). */
int rt_scene_sphere_count(const rt_scene* scene);
void* rt_scene_spheres(rt_scene* scene);
void* rt_scene_ubo(rt_scene* scene);

/* Material extension (no reference counterpart): count x {kind, param} floats, parallel to
   the sphere array, mutable.  In the scene file a sphere may carry a 4th element
   `[kind, param]`; the reference's decoder (includes/scene.hpp:89-95) accepts exactly three,
   and save() writes the 4th only for spheres whose kind is not 0. */
float* rt_scene_materials(rt_scene* scene);

/* The camera basis Application::Application installs at launch
   (src/application.cpp:10-23) from yaw/pitch in degrees (scene.hpp:127-128). */
int rt_scene_set_orientation(rt_scene* scene, float yaw_deg, float pitch_deg);

/* Scene::update(dt) (includes/scene.hpp:106-116) and the keyboard impulse of
   Application::handleInput (src/application.cpp:44-67): accel += dir * amount. */
int rt_scene_update(rt_scene* scene, float dt);
int rt_scene_add_acceleration(rt_scene* scene, float ax, float ay, float az);
int rt_scene_reset_frame_count(rt_scene* scene); /* scene.hpp:118 */
/* The public members Scene::acceleration / Scene::velocity (scene.hpp:125-126). */
int rt_scene_get_motion(const rt_scene* scene, float acceleration[3], float velocity[3]);

/* Seeded synthetic scenes in the same representation (no reference counterpart beyond the
   random spheres of src/scene.cpp:21-35).
   kind 0: "rtiow"  -- ground + 22x22 jittered small spheres + 3 large ones;
   kind 1: "cube"   -- `count` spheres with the distributions of src/scene.cpp:21-35;
   kind 2: kind 0 with the material extension (glass / matte / metal) switched on. */
int rt_scene_generate(int kind, int count, uint32_t seed, rt_scene** out);

#ifdef __cplusplus
}
#endif
#endif
