// rt_params.h -- kernel argument block shared by the host API and the HIP kernels.
#pragma once

#include <stdint.h>

#include "rt_hip.h"

// 32-byte AoS sphere exactly as the host hands it over
// (reference includes/scene.hpp:31-36, res/shaders/def.glsl:1-5).
struct RtSphereAos {
    float cx, cy, cz, radius;
    float r, g, b, pad;
};

// Spheres are tested 32 at a time (one bit per sphere in a 32-bit lane mask), and
// candidate masks of up to RT_SG_CHUNKS chunks are parked in LDS before the
// per-lane candidate walk.
constexpr int RT_CHUNK = 32;
constexpr int RT_SG_CHUNKS = 16;
// ... or of RT_SG_CHUNKS_WIDE chunks (RtParams::sg_wide; matrix-core sieve, 1025 to 4096 spheres).  A walk
// lasts until the lane with the most candidates is done, and over twice the spheres the lanes' counts are
// closer to one another: measured at 3840x2160, 4096 spheres 15.3 -> 14.8 ms, 1000 spheres 11.1 -> 10.8 ms
// (16 384 spheres, where the walk reads global memory: 26.7 -> 29.0 ms, so not there).
constexpr int RT_SG_CHUNKS_WIDE = 32;

// Largest sphere count whose geometry stream is staged in LDS (16 B per sphere).
#ifndef RT_MAX_LDS
#define RT_MAX_LDS 4096
#endif
constexpr int RT_MAX_LDS_SPHERES = RT_MAX_LDS;
// Up to this many spheres RT_KERNEL_STREAM reads the sieve geometry through the scalar cache.
constexpr int RT_MAX_SGPR_SPHERES = 2048;
// Relative safety margin of the sieve (2^-16 = 256 ulp): see sieve_chunk in rt_kernels.hip.
constexpr float RT_SIEVE_MARGIN = 1.52587890625e-05f;
// Margin of the sieve evaluated on the matrix cores (2^-15: see "sieve on the matrix cores" in
// rt_kernels.hip), and the largest sphere count whose operand fragments (64 B per sphere) are
// staged in LDS beside the rest.
constexpr float RT_SIEVE_MARGIN_MFMA = 3.0517578125e-05f;
constexpr int RT_MAX_MFMA_SPHERES = 1024;
// Beyond that the fragments are streamed from global memory, loaded RT_MFMA_AHEAD chunks ahead of
// their use (measured 1.5x the vector-ALU sieve from 2000 to 65536 spheres; the fragments of the
// largest scene are 4 MiB).
constexpr int RT_MAX_MFMA_FAR_SPHERES = 65536;
#ifndef RT_MFMA_AHEAD
#define RT_MFMA_AHEAD 2
#endif
// Largest workgroup of the stream kernel's vector-ALU instantiations: 8 waves per CU, two per SIMD,
// which leaves each wave the whole register file (the LDS-geometry sieve wants more than the 128
// registers of 1024-thread workgroups and spilled to scratch memory under that cap).
#ifndef RT_STREAM_MAX_THREADS
#define RT_STREAM_MAX_THREADS 512
#endif
// Largest workgroup of the matrix-core instantiation: its operand fragments and accumulators want
// more registers than 1024-thread workgroups (4 waves per SIMD, 128 registers) leave.
#ifndef RT_MFMA_MAX_THREADS
#define RT_MFMA_MAX_THREADS 768
#endif
// Culled kernel: spheres per culling group, and the relative margin on o.o of the group test
// (2^-9; see build_culled in rt_api.cpp for why it is larger than the sieve's).
constexpr int RT_CULL_GROUP = 8;
constexpr float RT_CULL_MARGIN_O = 0.001953125f;
// Per-lane traversal (RT_KERNEL_BVH): slots per block (8 groups), the number of block entries that
// follow the group entries in RtParams::boxes (RT_MAX_LDS_SPHERES / RT_BVH_BLOCK), and the relative
// padding of its boxes: a member sphere's radius grows by RT_BOX_PAD (|c| + r) on the host and by
// RT_BOX_PAD |o| for the ray at hand (see trace_bvh in rt_kernels.hip).
constexpr int RT_BVH_BLOCK = 64;
constexpr int RT_BVH_BLOCKS = 64;
constexpr float RT_BOX_PAD = 0.002f;
// Scenes too large for LDS are walked out of global memory with one level more: superblocks of
// RT_BVH_SUPER slots (8 blocks), at most 128 of them (a 128-bit mask per lane).
constexpr int RT_BVH_SUPER = 512;
constexpr int RT_BVH_FAR_MAX_SPHERES = 128 * RT_BVH_SUPER;
// Pixels a wave takes from a work queue at a time (one 8x8 tile).
#ifndef RT_SPAN_PIXELS
#define RT_SPAN_PIXELS 64
#endif
constexpr unsigned RT_SPAN = RT_SPAN_PIXELS;
// Work queues of the stream kernel: RT_QUEUES span counters RT_QUEUE_STRIDE words apart (one
// cache line each), followed by the count of finished queues and RT_QUEUES counts of finished
// workgroups.
#ifndef RT_QUEUES
#define RT_QUEUES 8
#endif
constexpr unsigned RT_QUEUE_STRIDE = 32;
constexpr unsigned RT_WORK_BYTES = 16384;
static_assert((2 * RT_QUEUES + 1) * RT_QUEUE_STRIDE * 4 <= RT_WORK_BYTES, "work block too small");
// Ray counters: RT_RAY_SLOTS 64-bit words RT_RAY_SLOT_STRIDE words (one cache line) apart.  A wave adds its rays to
// one slot when it ends; additions to one address complete one at a time (~11 ns), and the 32 400 waves of a 1080p
// frame of the lite kernel all end within its 46 us: with 32 slots the additions alone were 3 us of it.
constexpr unsigned RT_RAY_SLOTS = 256;
constexpr unsigned RT_RAY_SLOT_STRIDE = 16;
constexpr unsigned RT_RAY_BYTES = RT_RAY_SLOTS * RT_RAY_SLOT_STRIDE * 8;

struct RtParams {
    // camera (reference UBO, includes/scene.hpp:22-29)
    float fwd[3], right[3], up[3], pos[3];
    int sphere_count;  // loop bound of Trace(): ubo.sphereCount
    int n_pad;         // sphere_count rounded up to a multiple of RT_CHUNK
    int n_lds;         // spheres staged in LDS by the stream kernel: n_pad, or 0 when they do not fit
    int n_table;       // stream kernel: sphere records (32 B each) staged in LDS for the shading: n_pad, or 0 (read from `spheres`)
    int width, height;
    int tile_col0;     // first column of 8x8 tiles rendered
    int tile_cols;     // number of tile columns rendered
    unsigned tile_cols_magic;  // ceil(2^32 / tile_cols), 0 when tile_cols == 1:
                               // __umulhi(t, magic) == t / tile_cols while t * tile_cols < 2^32
    unsigned last_chunk_valid; // bit (31-n) set when sphere n of the last chunk is < sphere_count
    int last_chunk_count;      // spheres in the last chunk (1..32; 0 for an empty scene)
    int prefetch;              // stream kernel: waves ask for their next span one span ahead
    int tile_row0;     // first row of 8x8 tiles rendered
    int tile_rows;     // number of tile rows rendered
    unsigned total_pixels;  // tile_cols * tile_rows * 64
    int max_depth;
    unsigned frame0;   // ubo.frameCount
    int n_frames;
    int frame_stride;
    int clear_mode;    // RT_CLEAR_*
    // crop traced in record mode
    int rec_x0, rec_y0, rec_w, rec_h;

    const RtSphereAos* spheres;  // AoS copy, read for centre/albedo of the winning sphere
    const float4* geom;          // pair-packed sieve geometry: {cx0,cx1,cy0,cy1},{cz0,cz1,w0,w1}
    const float* radius;         // radii, n_pad entries (read by the literal test of the walk)
    const uint4* frags;          // matrix-core sieve only: per chunk 2 steps x 64 lanes x 8 bf16 sphere-side operands
    const float2* materials;     // material extension: {kind, param} per sphere; null = all kind 0
    // culled kernel only: geom/radius above are the spatially sorted copies, and
    const float4* bounds;        // pair-packed like geom: bounding sphere of every group of 8 slots,
                                 // padded to a multiple of 64 groups with never-touched entries
    int n_bounds;                // group entries in `bounds` (a multiple of 64)
    const float4* boxes;         // per-lane traversal: boxes of the n_bounds groups, then of n_block_boxes blocks of
                                 // 64 slots, then of the superblocks of 512 slots; centre and half-width
                                 // pair-packed per axis, 3 float4 per pair of entries
    int n_block_boxes;           // block entries in `boxes` (a multiple of 8, at least RT_BVH_BLOCKS)
    unsigned kd_flip[3];         // per-lane traversal: per axis (x, y, z) the bits 3, 4, ... of a slot number of the sorted list (as
                                 // bits 0, 1, ... here) whose two halves were split along that axis, lower half first
    const int* perm;             // sorted slot -> index in `spheres` (n_pad entries)
    int sg_wide;                 // != 0: candidate walks every RT_SG_CHUNKS_WIDE chunks instead of every RT_SG_CHUNKS
                                 // (brute-force kernels only; the masks in LDS are sized to match)
    const uint32_t* tile_order;  // lite and stream kernels: k-th tile handed out -> tile of the dispatched region
                                 // (tiles in descending order of their cost in the previous dispatch), or null:
                                 // image order
    const unsigned long long* cam_live;  // with cam_hits: per tile of the region, the pixels whose camera ray meets a sphere;
                                 // the stream kernel traces those, rt_sky_frames_kernel finishes the others
    const uint2* cam_hits;       // stream kernel: per pixel of the dispatched region, [tile][pixel of the tile], the nearest
                                 // hit of its camera ray {sphere index or -1, bits of t} (rt_camera_hits_kernel); null: none
    int cam_reuse;               // stream kernel: != 0: the nearest hit of a pixel's camera ray (the same ray in every frame) is
                                 // found once per dispatch and serves all the pixel's frames; 0 (then cam_hits == null):
                                 // every Trace() call evaluates the whole sphere list, as the shader's loop does
    int split_tiles;             // lite kernel, with tile_order: the first split_tiles tiles of the order are traced by
                                 // split_ways waves each, a run of split_frames frames per wave (0: none)
    int split_ways;              // 2, 4 or 8
    int split_frames;            // ceil(n_frames / split_ways)
    int split_lds;               // != 0: the runs of a tile are waves of one workgroup and hand their lights to the first run
                                 // through LDS, which joins them at the end of the kernel; 0: through split_lights and
                                 // rt_lite_join_kernel
    const uint32_t* split_count; // stream kernel: device word holding the number of leading tiles of tile_order whose frames
                                 // are shared among split_ways units of work (at most split_tiles of them count); null:
                                 // lite kernel (split_tiles itself) or no split
    const uint4* split_units;    // stream kernel: the units of work of the dispatch in the order they are handed out
                                 // (rt_tile_units_kernel): {tile, first frame, end frame, RT_PARK_*}; their number follows
                                 // the word at split_count
    float4* split_lights;        // the light of every path of the runs after a tile's first:
                                 // [tile of the order][run - 1][frame of the run][pixel of the tile]
    uint32_t* tile_cost;         // what each tile cost in this dispatch is left here (may be null): the lite kernel stores the
                                 // rays of the tile's longest lane, the stream kernel adds the tile's rays pixel by pixel
                                 // (zeroed before the launch)
    uint32_t* tile_chain;        // stream kernel, with tile_cost: the most rays a pixel of each tile took (zeroed before the launch)
    float4* accum;               // width*height rgba32f
    uint32_t* color;             // width*height rgba8
    unsigned* work_counter;      // RT_WORK_BYTES block of queue counters; all zero between launches
    unsigned long long* rays;    // += Trace() calls
    rt_record* records;          // record mode only
    float* rec_light;            // record mode only (may be null)
};
