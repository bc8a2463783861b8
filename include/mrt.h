/*
 * mrt.h -- C ABI of the MI355X path tracer that replaces the cpu_rt per-pixel
 * path-tracing loop (reference: cpu_rt/src/ray_tracer.rs:435-463 `trace_part`,
 * called by `trace_image` :413-418, `tracing_loop` :469-477 and the eight
 * worker threads of `threaded_render` :479-535).
 *
 * The reference has no FFI of its own; the seam is the body of `trace_part`.
 * A host (the Rust crates, or the C++/Python hosts in this repository)
 * flattens its `WorldInfo` (cpu_rt/src/ray_tracer/world.rs:56-62) into the
 * plain-old-data `mrt_scene_desc` below, creates a device scene once, and then
 * forwards every `trace_part` call to `mrt_render_accumulate`.
 *
 * Every entry point is plain C: pointers and sizes, no C++ or torch types.
 * No entry point aborts or throws across the boundary: failures return a
 * negative `mrt_status` and leave a message retrievable with `mrt_last_error`.
 */
#ifndef MRT_H
#define MRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_ABI_VERSION 2

/* ---------------------------------------------------------------- status */
typedef enum mrt_status {
    MRT_OK = 0,
    MRT_ERR_INVALID_ARGUMENT = -1, /* NULL pointer, bad index, bad size          */
    MRT_ERR_UNSUPPORTED = -2,      /* a feature the reference itself panics on   */
    MRT_ERR_DEVICE = -3,           /* HIP runtime error (message has the detail) */
    MRT_ERR_NO_DEVICE = -4,        /* no gfx950 device visible                   */
    MRT_ERR_OUT_OF_MEMORY = -5
} mrt_status;

/* --------------------------------------------------------------- textures
 * cpu_rt/src/ray_tracer/texture.rs: SolidColor :33-44, ImageTexture :191-224,
 * MultiplyTexture :12-32, CheckerTexture :46-70, DebugV :225-239,
 * Perlin :71-190. */
enum {
    MRT_TEX_SOLID = 0,    /* color                                         */
    MRT_TEX_IMAGE = 1,    /* image = index into mrt_scene_desc.images      */
    MRT_TEX_MULTIPLY = 2, /* a, b = child texture indices (must be < own)  */
    MRT_TEX_CHECKER = 3,  /* a = odd, b = even                             */
    MRT_TEX_DEBUG_V = 4,
    MRT_TEX_PERLIN = 5    /* image = index into mrt_scene_desc.perlins     */
};
typedef struct mrt_texture {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t image;
    float color[3];
    int32_t reserved;
} mrt_texture;

/* Decoded RGB image, f32 per channel in [0,1] (u8/255 as the reference stores
 * it), row 0 = top row of the file, index (x + y*width)*3 + c
 * (texture.rs:196-215, parallel_image.rs:66-71). */
typedef struct mrt_image {
    int32_t width;
    int32_t height;
    const float* rgb;
} mrt_image;

/* The random tables of one `Perlin` value (texture.rs:72-77, filled by
 * `Perlin::default()` :162-180 from the global RNG: 256 normalised random
 * vectors and three permutations of 0..255). The host draws them; they are
 * scene data. */
typedef struct mrt_perlin {
    double ran_float[256][3];
    uint32_t perm_x[256];
    uint32_t perm_y[256];
    uint32_t perm_z[256];
} mrt_perlin;

/* -------------------------------------------------------------- materials
 * cpu_rt/src/ray_tracer/material.rs: Lambertian :26-78, Metal :80-134,
 * Dielectric :135-206, DiffuseLight :208-249, Isotropic :251-295. */
enum {
    MRT_MAT_LAMBERTIAN = 0,    /* texture = albedo                     */
    MRT_MAT_METAL = 1,         /* texture = albedo, fuzz               */
    MRT_MAT_DIELECTRIC = 2,    /* index_refraction, color              */
    MRT_MAT_DIFFUSE_LIGHT = 3, /* texture = emit                       */
    MRT_MAT_ISOTROPIC = 4      /* texture = albedo                     */
};
typedef struct mrt_material {
    int32_t kind;
    int32_t texture;
    double fuzz;
    double index_refraction;
    float color[3];
    int32_t reserved;
} mrt_material;

/* ----------------------------------------------------------------- shapes
 * cpu_rt/src/ray_tracer/hittable/{sphere,rect,render_box,constant_medium}.rs.
 * Parameter slots p[]:
 *   SPHERE          p0 radius, p1..3 origin
 *   MOVING_SPHERE   p0 radius, p1..3 center_0, p4..6 center_1, p7 time_0, p8 time_1
 *   XY_RECT         p0 x0, p1 x1, p2 y0, p3 y1, p4 k(z), p5 normal_flip (+1 / -1)
 *   XZ_RECT         p0 x0, p1 x1, p2 z0, p3 z1, p4 k(y), p5 normal_flip
 *   YZ_RECT         p0 y0, p1 y1, p2 z0, p3 z1, p4 k(x), p5 normal_flip
 *   RENDER_BOX      p0..2 box_min, p3..5 box_max
 *   CONSTANT_MEDIUM p0 density; boundary = shape index (SPHERE or RENDER_BOX,
 *                   must be < own index); material = phase function
 *   OCTREE          octree = index into mrt_scene_desc.octrees
 */
enum {
    MRT_SHAPE_SPHERE = 0,
    MRT_SHAPE_MOVING_SPHERE = 1,
    MRT_SHAPE_XY_RECT = 2,
    MRT_SHAPE_XZ_RECT = 3,
    MRT_SHAPE_YZ_RECT = 4,
    MRT_SHAPE_RENDER_BOX = 5,
    MRT_SHAPE_CONSTANT_MEDIUM = 6,
    MRT_SHAPE_OCTREE = 7
};
typedef struct mrt_shape {
    int32_t kind;
    int32_t material;
    int32_t boundary;
    int32_t octree;
    double p[10];
} mrt_shape;

/* Object = shape + Transform (hittable.rs:64-122,184-201). `world_transform`
 * is the reference's Matrix4 (world -> object), column-major as cgmath stores
 * it: element (row r, column c) at [c*4 + r]. */
typedef struct mrt_object {
    int32_t shape;
    int32_t reserved;
    double world_transform[16];
} mrt_object;

/* ------------------------------------------------- linearised voxel octree
 * Query-side image of FastOctTree<Voxel> (hittable/fast_oct_tree/mod.rs,
 * arena.rs): one record per arena node. Children of a parent are stored as
 * eight node indices in the reference's child order x*4 + y*2 + z
 * (mod.rs `pos_to_index`). */
enum { MRT_OCT_EMPTY = 0, MRT_OCT_LEAF = 1, MRT_OCT_PARENT = 2 };
/* Voxel (fast_oct_tree/voxel.rs:12-45): Solid(Lambertian | Reflect) or
 * Volume { density, color, edge_effect }; VolumeEdgeEffect is None or
 * Solid { hit_probability, solid_material: Lambertian | Reflect }. */
enum { MRT_VOX_LAMBERTIAN = 0, MRT_VOX_REFLECT = 1, MRT_VOX_VOLUME = 2 };
enum { MRT_VOX_EDGE_NONE = 0, MRT_VOX_EDGE_LAMBERTIAN = 1, MRT_VOX_EDGE_REFLECT = 2 };
typedef struct mrt_oct_node {
    int32_t kind;       /* MRT_OCT_*                                         */
    uint32_t size;      /* log2 of the edge length                           */
    int32_t first_child; /* index into mrt_octree.children (8 entries)       */
    int32_t voxel;      /* index into mrt_octree.voxels for a LEAF           */
} mrt_oct_node;
typedef struct mrt_voxel {
    int32_t kind;          /* MRT_VOX_*                                       */
    float albedo[3];       /* Lambertian / Reflect albedo, Volume colour      */
    float fuzz;            /* Reflect                                         */
    int32_t edge_kind;     /* Volume: MRT_VOX_EDGE_*                          */
    float edge_probability; /* Volume edge: hit_probability                   */
    float edge_fuzz;       /* Volume edge, Reflect                            */
    float edge_albedo[3];  /* Volume edge: albedo of the solid material       */
    int32_t reserved;
    double density;        /* Volume                                          */
} mrt_voxel;
typedef struct mrt_octree {
    uint32_t root;
    uint32_t n_nodes;
    const mrt_oct_node* nodes;
    uint32_t n_children; /* multiple of 8 */
    const uint32_t* children;
    uint32_t n_voxels;
    const mrt_voxel* voxels;
} mrt_octree;

/* Host-side construction of an mrt_octree (no GPU involved). The reference
 * builds FastOctTree<Voxel> with `set` (fast_oct_tree/operations/set.rs:4-36,
 * mod.rs:43-169: split, recurse, re-collapse equal children), `combine`
 * (operations/combine.rs:4-19) and `load_vox` (operations/load.rs:74-127);
 * these entry points produce the same canonical tree from the voxel set.
 * `xyz` holds n coordinate triples (0 .. 2^21-1); later sets overwrite earlier
 * ones; `voxels` holds n records, or one when `one_voxel_for_all` is non-zero.
 * mrt_voxels_linearise fills `out` with pointers owned by the grid, valid until
 * the grid is changed or destroyed. */
typedef struct mrt_voxel_grid mrt_voxel_grid;
mrt_voxel_grid* mrt_voxels_create(void);
void mrt_voxels_destroy(mrt_voxel_grid* grid);
int mrt_voxels_set(mrt_voxel_grid* grid, const int32_t* xyz, const mrt_voxel* voxels, size_t n,
                   int one_voxel_for_all);
int mrt_voxels_combine(mrt_voxel_grid* dst, const mrt_voxel_grid* src, const int32_t offset[3]);
int mrt_voxels_load_vox(mrt_voxel_grid* grid, const uint8_t* vox_file_bytes, size_t len);
size_t mrt_voxels_count(const mrt_voxel_grid* grid);
int mrt_voxels_linearise(mrt_voxel_grid* grid, mrt_octree* out);

/* ------------------------------------------------------ camera, background
 * camera.rs:14-24 (CameraInfo). Angles in degrees as in the reference. */
typedef struct mrt_camera_info {
    double aspect_ratio;
    double fov;
    double origin[3];
    double look_at[3];
    double up_vector[3];
    double aperture;
    double focus_distance;
    double start_time;
    double end_time;
} mrt_camera_info;

/* background/{sky,sun_sky,constant_color}.rs */
enum { MRT_BG_SKY = 0, MRT_BG_SUN_SKY = 1, MRT_BG_CONSTANT = 2 };
typedef struct mrt_background {
    int32_t kind;
    int32_t reserved;
    double intensity;
    double sun_radius;
    double sun_theta;
    double sun_phi;
    double sun_brightness;
    float color[3];
    int32_t reserved2;
} mrt_background;

/* World.sun (sun.rs:10-21); has_sun = 0 is the reference's `None`. */
typedef struct mrt_sun {
    int32_t has_sun;
    int32_t reserved;
    double phi;
    double theta;
    double radius;
} mrt_sun;

/* ---------------------------------------------------------- scene (WorldInfo)
 * `lights` are separate Object values, exactly as `WorldInfo.lights` holds
 * clones (world.rs:56-62). `bvh_seed` seeds the split-axis draws that the
 * reference takes from its global RNG while building the tree (bvh.rs:215). */
typedef struct mrt_scene_desc {
    uint32_t abi_version;
    uint32_t n_textures;
    uint32_t n_images;
    uint32_t n_materials;
    uint32_t n_shapes;
    uint32_t n_objects;
    uint32_t n_lights;
    uint32_t n_octrees;
    const mrt_texture* textures;
    const mrt_image* images;
    const mrt_material* materials;
    const mrt_shape* shapes;
    const mrt_object* objects;
    const mrt_object* lights;
    const mrt_octree* octrees;
    mrt_camera_info camera;
    mrt_background background;
    mrt_sun sun;
    uint64_t bvh_seed;
    uint32_t n_perlins;
    uint32_t reserved;
    const mrt_perlin* perlins;
} mrt_scene_desc;

/* Flattened BVH node, pre-order, as exported for inspection. */
typedef struct mrt_bvh_node {
    double bmin[3];
    double bmax[3];
    int32_t object; /* >= 0: leaf (object index); -1: inner node            */
    int32_t skip;   /* index of the next node when this subtree is skipped  */
} mrt_bvh_node;

/* Counters of one render call. `rays` counts `World::nearest_hit` calls
 * (ray_tracer.rs:173), `light_queries` counts `World::nearest_light_hit`
 * calls (pdf.rs:83). */
typedef struct mrt_stats {
    uint64_t paths;
    uint64_t rays;
    uint64_t light_queries;
    uint64_t launches;
    double kernel_ms;      /* device time of the segment kernels (HIP events) */
    double total_ms;       /* device time of the whole call                  */
    double segment_kernel_avg_ms; /* kernel_ms / launches                     */
} mrt_stats;

typedef struct mrt_scene mrt_scene;

/* ------------------------------------------------------------ entry points */

/* Library / device management. */
int mrt_abi_version(void);
int mrt_device_count(void);                 /* number of visible HIP devices  */
const char* mrt_last_error(void);           /* thread-local, never NULL       */

/* World construction: the device-side image of `WorldInfo::build_world`
 * (world.rs:64-76), including `BvhTree::new` (bvh.rs:78-96,206-293).
 * `device` is the HIP device ordinal the scene lives on. Returns NULL and
 * fills `err` (if non-NULL) on failure. */
mrt_scene* mrt_scene_create(const mrt_scene_desc* desc, int device, char* err, size_t errlen);
void mrt_scene_destroy(mrt_scene* scene);

/* Copies the flattened BVH out (`*n_nodes` receives the node count; `out` may
 * be NULL to query the count). */
int mrt_scene_bvh(const mrt_scene* scene, mrt_bvh_node* out, size_t capacity, size_t* n_nodes);

/* `Camera::get_ray` (camera.rs:88-98) for pixel-sample (x, y, sample) of a
 * W x H image exactly as `trace_part` forms it (ray_tracer.rs:444-447).
 * pixels: n x 2 int32 (x, y); rays_out: n x 7 doubles (origin, direction, time). */
int mrt_camera_rays(mrt_scene* scene, uint32_t width, uint32_t height, const int32_t* pixels,
                    size_t n, uint64_t seed, uint64_t sample_index, double* rays_out);

/* `World::nearest_hit(ray, 0.001f32, f32::MAX)` (world.rs:112-114) for n host
 * rays. rays: n x 7 doubles (origin xyz, direction xyz, time). Outputs (each
 * may be NULL): obj_id n (-1 = miss), t n, normal n x 3 (world space),
 * position n x 3 (world space), front_face n. Ray i draws its random numbers
 * (ConstantMedium) from the stream of pixel index i, sample 0, segment 0. */
int mrt_trace_rays(mrt_scene* scene, const double* rays, size_t n, uint64_t seed, int32_t* obj_id,
                   double* t, double* normal, double* position, int32_t* front_face);

/* `spp` consecutive `trace_part` calls (ray_tracer.rs:435-463) over the
 * vertical strip x in [x0, x0 + strip_w) of a width x height image:
 * accum_rgb[(x + y*width)*3 + c] += radiance sample, row y = 0 at the bottom
 * (parallel_image.rs:328-331). Samples are numbered first_sample_index ..
 * first_sample_index + spp - 1; together with `seed` and the pixel index they
 * select the random stream, so disjoint sample ranges are independent and the
 * union of two calls equals one call over the joined range.
 * `accum_rgb` is HOST memory of width*height*3 floats. `stats` may be NULL. */
int mrt_render_accumulate(mrt_scene* scene, float* accum_rgb, uint32_t width, uint32_t height,
                          uint32_t x0, uint32_t strip_w, uint32_t spp, uint64_t seed,
                          uint64_t first_sample_index, mrt_stats* stats);

/* Same samples, split over several scene handles -- normally one per GPU,
 * created from the same description -- for a single-process host such as the
 * reference's (ray_tracer.rs:479-535 drives all its workers from one process).
 * Handle i renders the i-th share of the sample range
 * [first_sample_index, first_sample_index + spp) on its own device, concurrently
 * with the others, into an image that stays resident on that device between
 * calls (handles that share a device share its image). The strip of `accum_rgb`
 * (HOST memory) is uploaded once, the per-device images are summed on the
 * devices with one ncclAllReduce per device (RCCL, bound with dlopen at first
 * use), and the strip is downloaded once: the result is the one-handle result
 * up to the order of the f32 additions. A host-side sum of the downloaded
 * strips replaces the collective ONLY when RCCL is not there (the library
 * cannot be loaded or its communicators cannot be made), i.e. before any image
 * has been touched; a failure after the collectives were started leaves partly
 * reduced images and is MRT_ERR_DEVICE. With the environment variable
 * MRT_MULTI_FORCE_RCCL set, the collective also runs when all handles share one
 * device (a world of one), and a missing RCCL is an error instead of a
 * fall-back. `stats`: sums of the counters, the longest kernel time, the wall
 * time of the call. Multi-process hosts split the samples the same way and
 * all-reduce the buffers themselves (one rank per GPU). The per-device tables
 * of this call, of the no-allocation entry point and of mrt_bloom* hold device
 * ordinals 0..63. */
int mrt_render_accumulate_multi(mrt_scene* const* scenes, uint32_t n_scenes, float* accum_rgb,
                                uint32_t width, uint32_t height, uint32_t x0, uint32_t strip_w,
                                uint32_t spp, uint64_t seed, uint64_t first_sample_index,
                                mrt_stats* stats);

/* Same, with `accum_rgb_device` resident on the scene's device (e.g. a torch
 * tensor's data pointer). `stream` is a hipStream_t (NULL = default stream);
 * the call returns after the work has completed on that stream.
 * The buffer must be ordinary device memory of the scene's GPU (hipMalloc /
 * a torch CUDA tensor): samples are added with hardware f32 atomics, which
 * host-pinned, managed or fine-grained memory may silently drop; any other
 * kind of pointer is rejected with MRT_ERR_INVALID_ARGUMENT. */
int mrt_render_accumulate_device(mrt_scene* scene, float* accum_rgb_device, uint32_t width,
                                 uint32_t height, uint32_t x0, uint32_t strip_w, uint32_t spp,
                                 uint64_t seed, uint64_t first_sample_index, void* stream,
                                 mrt_stats* stats);

/* mrt_render_accumulate_device for callers that cannot ALLOCATE while they render (serving loops
 * that add a sample per frame under a fixed memory budget, harnesses that forbid hipMalloc after
 * start-up, hosts that have handed all free device memory to another library). It has no
 * counterpart in the reference.
 * Same arguments, same samples and the same image (up to the order of the f32 additions), but the
 * call allocates nothing on the device or the host: the path pool, the counters and the per-wave
 * work ranges live in a device array that is part of the library's code object
 * (MRT_NOALLOC_SCRATCH_BYTES bytes on every GPU the library runs on, held until the process
 * ends), and the events and the pinned counter copy were created with the scene.
 * It is NOT for code under HIP stream capture: like every render entry point it is a host-driven
 * loop -- it queries the buffer's pointer attributes, synchronises `stream` after every kernel
 * launch, reads the counters and the event times back and decides the next launch on the host --
 * and all of that is illegal while a stream is being captured.
 * Limits: the pool is MRT_NOALLOC_POOL_SLOTS path slots whatever "pool_slots" says, and the drain
 * phase does not pack the pool ("compaction" is ignored); neither grows with the image, so the
 * sizes that are rejected (MRT_ERR_INVALID_ARGUMENT) are those of every render entry point:
 * width * height > (2^32 - 1) / 3 pixels, strip_w * height * spp >= 2^53 paths, a strip outside
 * the image. Every call on a GPU uses that GPU's one array, which keeps the previous call's
 * contents (each call resets what it reads): calls on one GPU, from any thread and for any scene,
 * take turns on a per-GPU lock inside the library; a second PROCESS's calls are not ordered
 * against it but have their own copy of the array. */
#define MRT_NOALLOC_POOL_SLOTS 524288u
#define MRT_NOALLOC_SCRATCH_BYTES 44171776u
int mrt_render_accumulate_device_noalloc(mrt_scene* scene, float* accum_rgb_device, uint32_t width,
                                         uint32_t height, uint32_t x0, uint32_t strip_w, uint32_t spp,
                                         uint64_t seed, uint64_t first_sample_index, void* stream,
                                         mrt_stats* stats);

/* `ParallelImage / num_samples` (parallel_image.rs:268-278) followed by
 * `to_image`'s colour mapping (parallel_image.rs:35-65, rgb_color.rs:75-80):
 * rgb8_out[(x + (height-1-y)*width)*3 + c] = round(clamp(sqrt(accum/spp))*255).
 * Host buffers. */
int mrt_to_image(const float* accum_rgb, uint32_t width, uint32_t height, uint32_t num_samples,
                 uint8_t* rgb8_out);

/* Post-processing step of `RayTracer::tracing_loop` (ray_tracer.rs:473-476,
 * :465-467): out = bloom(accum / num_samples), bloom = ray_tracer/bloom.rs:75-94
 * (bright-pass, 7-level 9-tap blur pyramid with the bilinear up-add of
 * parallel_image.rs:193-260, tone map 1 - exp(-hdr*(bloom+hdr)), pow(1/2.2)).
 * Images are float[height][width][3] like the accumulation buffer; `out` may
 * not alias `accum`. The image must be at least 128x128 (below that the
 * reference panics in get_nearest; MRT_ERR_UNSUPPORTED). Device-resident
 * buffers on `device`; `stream` is a hipStream_t (NULL = default stream); the
 * call returns after the work has completed. */
int mrt_bloom_device(int device, const float* accum_rgb_device, uint32_t width, uint32_t height,
                     uint32_t num_samples, float* out_rgb_device, void* stream);
/* Same with host buffers (uploaded to `device`, processed there, downloaded). */
int mrt_bloom(int device, const float* accum_rgb, uint32_t width, uint32_t height,
              uint32_t num_samples, float* out_rgb);
/* mrt_to_image on device-resident buffers (parallel_image.rs:35-65). */
int mrt_to_image_device(int device, const float* rgb_device, uint32_t width, uint32_t height,
                        uint32_t num_samples, uint8_t* rgb8_out_device, void* stream);

/* `CurrentShader` (ray_tracer.rs:304-309), selected with
 * mrt_scene_set_option(scene, "shader", MRT_SHADER_*) as RayTracer::set_shader
 * does (ray_tracer.rs:409-411); the render entry points then evaluate
 *   RAYTRACING  RayTracingShader::ray_color (ray_tracer.rs:159-303), the default
 *   DIFFUSE     DiffuseShader::ray_color (ray_tracer.rs:126-157): emitted colour,
 *               scatter attenuation, black, or the background of the first hit
 *   LIGHT_MAP   LightMapShader::ray_color (ray_tracer.rs:83-124): for every
 *               light, a sampled point on it, a second nearest-hit query towards
 *               it, and the distance from that hit to the sampled point
 * Random numbers of LIGHT_MAP: segment 0 for the camera ray and the first
 * query, segment 1 + i for the sample on light i and the query towards it. */
enum mrt_shader { MRT_SHADER_RAYTRACING = 0, MRT_SHADER_DIFFUSE = 1, MRT_SHADER_LIGHT_MAP = 2 };

/* Tuning knobs (performance only; results change at most in the order of the
 * f32 accumulation adds, except "trace_zero_throughput"):
 *   "pool_slots"             path slots in flight (default 2^21)
 *   "segments_per_launch"    1..65536 segments per kernel launch (default 4096:
 *                            a launch runs until the work is handed out and most paths
 *                            have ended; smaller values bound the time of one launch)
 *   "compaction"             0/1 pack live paths in the drain phase (default 1)
 *   "traversal"              0 auto, 1 per-lane tree walk, 2 wave-synchronous
 *                            tree walk, 3 wave-synchronous walk over the leaf list,
 *                            4 candidate search (a spatial tree finds the leaves a
 *                            ray can meet; the tree walk then runs over those
 *                            leaves and their ancestors only; auto picks it for
 *                            trees of more than 64 nodes; scenes with media, octrees
 *                            or more than 1024 tree nodes fall back to 1)
 *   "candidate_queue"        entries of the per-wave work queue of the candidate
 *                            search, 256..2048 (default 2048). A wave whose queue
 *                            could overflow takes the tree walk for its rays (same
 *                            result); smaller values exist to test that path.
 *   "octree_suspend"         0/1 (default 1): in scenes whose first object is a solid-voxel octree, the
 *                            lanes of a wave that are still walking the octree when most others have
 *                            finished keep their walk for the next turn of the kernel's loop while the
 *                            others shade and start new paths (same steps, same results)
 *   "trace_zero_throughput"  1: keep walking paths whose throughput is exactly
 *                            zero, as the reference does (same image, counters
 *                            match the reference's)
 * and one that selects what is computed: "shader" (MRT_SHADER_*, see above).
 * Returns MRT_ERR_INVALID_ARGUMENT for unknown names or values out of range. */
int mrt_scene_set_option(mrt_scene* scene, const char* name, int64_t value);

/* ------------------------------------------------- random number contract
 * The reference draws from rand 0.8.5's thread RNG (prelude.rs:82-87), which
 * is not reproducible. Both the device path and the CPU checker use the
 * counter-based generator below instead, addressed by (path key, segment,
 * dimension) so that the order in which code consumes numbers is irrelevant.
 *
 *   mix64(z): z = (z ^ z>>30) * 0xBF58476D1CE4E5B9; z = (z ^ z>>27) * 0x94D049BB133111EB; z ^ z>>31
 *   key      = mix64( mix64(seed + 0x9E3779B97F4A7C15*(pixel_index+1)) ^ 0xD1B54A32D192ED03*(sample_index+1) )
 *   u64(key, seg, dim) = mix64(key + 0x9E3779B97F4A7C15 * (((seg << 16) | dim) + 1))
 *   f64 in [0,1) = (u64 >> 11) * 2^-53        u32 = u64 >> 32
 *
 * pixel_index = x + y*width. Segment s is the s-th `nearest_hit` of the path
 * (0..49); the camera dimensions live in segment 0. Rejection loops use the
 * k-th attempt's slots and give up (zero vector) after MRT_MAX_REJECT attempts.
 */
#define MRT_MAX_REJECT 64
enum {
    MRT_DIM_JITTER_U = 0,     /* ray_tracer.rs:444                          */
    MRT_DIM_JITTER_V = 1,     /* ray_tracer.rs:446                          */
    MRT_DIM_TIME = 2,         /* camera.rs:96                               */
    MRT_DIM_DIELECTRIC = 3,   /* material.rs:179                            */
    MRT_DIM_PDF_CHOICE = 4,   /* pdf.rs:240                                 */
    MRT_DIM_COS_R1 = 5,       /* prelude.rs:146                             */
    MRT_DIM_COS_R2 = 6,       /* prelude.rs:147                             */
    MRT_DIM_LIGHT_INDEX = 7,  /* pdf.rs:113                                 */
    MRT_DIM_LIGHT_A = 8,      /* rect.rs:110,231,356 first coordinate       */
    MRT_DIM_LIGHT_B = 9,      /* second coordinate                          */
    MRT_DIM_LIGHT_FACE = 10,  /* render_box.rs:158                          */
    MRT_DIM_SUN_R = 11,       /* pdf.rs:162                                 */
    MRT_DIM_SUN_THETA = 12,   /* pdf.rs:163                                 */
    MRT_DIM_LENS = 0x100,     /* + 2*k + {0,1}      camera.rs:101           */
    MRT_DIM_METAL_FUZZ = 0x200,   /* + 3*k + {0,1,2}  material.rs:111       */
    MRT_DIM_ISOTROPIC = 0x300,    /* + 3*k + i        material.rs:278       */
    MRT_DIM_LIGHT_SPHERE = 0x400, /* + 3*k + i        sphere.rs:73          */
    MRT_DIM_SKY = 0x500,          /* + 3*k + i        pdf.rs:167            */
    MRT_DIM_ISOTROPIC_PDF = 0x600, /* + 3*k + i       pdf.rs:77             */
    MRT_DIM_MEDIUM = 0x1000,      /* + ordinal of the ConstantMedium::hit call
                                     within this nearest_hit traversal
                                     (constant_medium.rs:60)                */
    MRT_DIM_LIGHT_QUERY_MEDIUM = 0x2000, /* same, inside nearest_light_hit  */
    MRT_DIM_OCT_VOLUME = 0x3000   /* + 2*n + {0 edge effect, 1 free path}: the
                                     n-th time a ray enters a volume voxel from
                                     outside any volume, counted over the octree
                                     walks of one nearest_hit (ray_trace.rs:159,168) */
};

#ifdef __cplusplus
}
#endif
#endif /* MRT_H */
