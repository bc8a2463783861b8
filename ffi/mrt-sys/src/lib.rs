//! Raw bindings of libmrt.so: mirrors include/mrt.h one to one (every function the header declares; checked by
//! tests/test_ffi_crate.py against the header). Written without bindgen so that the crate needs no libclang.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct MrtTexture  { pub kind: i32, pub a: i32, pub b: i32, pub image: i32, pub color: [f32; 3], pub reserved: i32 }
#[repr(C)] pub struct MrtImage    { pub width: i32, pub height: i32, pub rgb: *const f32 }
#[repr(C)] pub struct MrtMaterial { pub kind: i32, pub texture: i32, pub fuzz: f64, pub index_refraction: f64, pub color: [f32; 3], pub reserved: i32 }
#[repr(C)] pub struct MrtShape    { pub kind: i32, pub material: i32, pub boundary: i32, pub octree: i32, pub p: [f64; 10] }
#[repr(C)] pub struct MrtObject   { pub shape: i32, pub reserved: i32, pub world_transform: [f64; 16] } // cgmath column-major
#[repr(C)] pub struct MrtCameraInfo { pub aspect_ratio: f64, pub fov: f64, pub origin: [f64; 3], pub look_at: [f64; 3],
                                      pub up_vector: [f64; 3], pub aperture: f64, pub focus_distance: f64,
                                      pub start_time: f64, pub end_time: f64 }
#[repr(C)] pub struct MrtBackground { pub kind: i32, pub reserved: i32, pub intensity: f64, pub sun_radius: f64, pub sun_theta: f64,
                                      pub sun_phi: f64, pub sun_brightness: f64, pub color: [f32; 3], pub reserved2: i32 }
#[repr(C)] pub struct MrtSun      { pub has_sun: i32, pub reserved: i32, pub phi: f64, pub theta: f64, pub radius: f64 }
#[repr(C)] pub struct MrtOctNode  { pub kind: i32, pub size: u32, pub first_child: i32, pub voxel: i32 }
#[repr(C)] pub struct MrtVoxel    { pub kind: i32, pub albedo: [f32; 3], pub fuzz: f32, pub edge_kind: i32, pub edge_probability: f32,
                                    pub edge_fuzz: f32, pub edge_albedo: [f32; 3], pub reserved: i32, pub density: f64 }
#[repr(C)] pub struct MrtOctree   { pub root: u32, pub n_nodes: u32, pub nodes: *const MrtOctNode, pub n_children: u32,
                                    pub children: *const u32, pub n_voxels: u32, pub voxels: *const MrtVoxel }
#[repr(C)] pub struct MrtSceneDesc {
    pub abi_version: u32, pub n_textures: u32, pub n_images: u32, pub n_materials: u32, pub n_shapes: u32,
    pub n_objects: u32, pub n_lights: u32, pub n_octrees: u32,
    pub textures: *const MrtTexture, pub images: *const MrtImage, pub materials: *const MrtMaterial,
    pub shapes: *const MrtShape, pub objects: *const MrtObject, pub lights: *const MrtObject,
    pub octrees: *const MrtOctree,
    pub camera: MrtCameraInfo, pub background: MrtBackground, pub sun: MrtSun, pub bvh_seed: u64,
    pub n_perlins: u32, pub reserved: u32, pub perlins: *const MrtPerlin }
#[repr(C)] pub struct MrtPerlin { pub ran_float: [[f64; 3]; 256], pub perm_x: [u32; 256], pub perm_y: [u32; 256], pub perm_z: [u32; 256] }
#[repr(C)] #[derive(Default)] pub struct MrtStats { pub paths: u64, pub rays: u64, pub light_queries: u64, pub launches: u64,
                                                    pub kernel_ms: f64, pub total_ms: f64, pub segment_kernel_avg_ms: f64 }
#[repr(C)] pub struct MrtBvhNode { pub bmin: [f64; 3], pub bmax: [f64; 3], pub object: i32, pub skip: i32 }
pub enum MrtScene {}

extern "C" {
    pub fn mrt_abi_version() -> c_int;
    pub fn mrt_device_count() -> c_int;
    pub fn mrt_last_error() -> *const c_char;
    pub fn mrt_scene_create(desc: *const MrtSceneDesc, device: c_int, err: *mut c_char, errlen: usize) -> *mut MrtScene;
    pub fn mrt_scene_destroy(scene: *mut MrtScene);
    pub fn mrt_render_accumulate(scene: *mut MrtScene, accum_rgb: *mut f32, width: u32, height: u32, x0: u32,
                                 strip_w: u32, spp: u32, seed: u64, first_sample_index: u64, stats: *mut MrtStats) -> c_int;
    pub fn mrt_render_accumulate_device(scene: *mut MrtScene, accum_rgb_device: *mut f32, width: u32, height: u32, x0: u32,
                                 strip_w: u32, spp: u32, seed: u64, first_sample_index: u64, stream: *mut c_void,
                                 stats: *mut MrtStats) -> c_int;
    // same as mrt_render_accumulate_device but allocates nothing (see "Callers that cannot allocate" below)
    pub fn mrt_render_accumulate_device_noalloc(scene: *mut MrtScene, accum_rgb_device: *mut f32, width: u32, height: u32,
                                 x0: u32, strip_w: u32, spp: u32, seed: u64, first_sample_index: u64, stream: *mut c_void,
                                 stats: *mut MrtStats) -> c_int;
    pub fn mrt_render_accumulate_multi(scenes: *const *mut MrtScene, n_scenes: u32, accum_rgb: *mut f32, width: u32, height: u32,
                                 x0: u32, strip_w: u32, spp: u32, seed: u64, first_sample_index: u64, stats: *mut MrtStats) -> c_int;
    // inspection / parity probes: the flattened tree in the reference's topology, and Camera::get_ray as trace_part forms it
    pub fn mrt_scene_bvh(scene: *const MrtScene, out: *mut MrtBvhNode, capacity: usize, n_nodes: *mut usize) -> c_int;
    pub fn mrt_camera_rays(scene: *mut MrtScene, width: u32, height: u32, pixels: *const i32, n: usize, seed: u64,
                           sample_index: u64, rays_out: *mut f64) -> c_int;
    pub fn mrt_trace_rays(scene: *mut MrtScene, rays: *const f64, n: usize, seed: u64, obj_id: *mut i32, t: *mut f64,
                          normal: *mut f64, position: *mut f64, front_face: *mut i32) -> c_int;
    pub fn mrt_to_image(accum_rgb: *const f32, width: u32, height: u32, num_samples: u32, rgb8_out: *mut u8) -> c_int;
    pub fn mrt_scene_set_option(scene: *mut MrtScene, name: *const c_char, value: i64) -> c_int;
    // post-processing step of tracing_loop (ray_tracer.rs:473-476): out = bloom(accum / num_samples)
    pub fn mrt_bloom(device: c_int, accum_rgb: *const f32, width: u32, height: u32, num_samples: u32, out_rgb: *mut f32) -> c_int;
    pub fn mrt_bloom_device(device: c_int, accum_rgb_device: *const f32, width: u32, height: u32, num_samples: u32,
                            out_rgb_device: *mut f32, stream: *mut c_void) -> c_int;
    pub fn mrt_to_image_device(device: c_int, rgb_device: *const f32, width: u32, height: u32, num_samples: u32,
                               rgb8_out_device: *mut u8, stream: *mut c_void) -> c_int;
    // host-side octree construction (FastOctTree::set / combine / load_vox)
    pub fn mrt_voxels_create() -> *mut MrtVoxelGrid;
    pub fn mrt_voxels_destroy(grid: *mut MrtVoxelGrid);
    pub fn mrt_voxels_set(grid: *mut MrtVoxelGrid, xyz: *const i32, voxels: *const MrtVoxel, n: usize, one_voxel_for_all: c_int) -> c_int;
    pub fn mrt_voxels_combine(dst: *mut MrtVoxelGrid, src: *const MrtVoxelGrid, offset: *const i32) -> c_int;
    pub fn mrt_voxels_load_vox(grid: *mut MrtVoxelGrid, bytes: *const u8, len: usize) -> c_int;
    pub fn mrt_voxels_count(grid: *const MrtVoxelGrid) -> usize;
    pub fn mrt_voxels_linearise(grid: *mut MrtVoxelGrid, out: *mut MrtOctree) -> c_int;
}
pub enum MrtVoxelGrid {}
