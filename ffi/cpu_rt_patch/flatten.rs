//! Sketch of the scene flattening a maintainer adds to `cpu_rt` (not compiled in this repository: no Rust toolchain
//! here). The reference's trait objects cannot be downcast (`Hittable`, cpu_rt/src/ray_tracer/hittable.rs:36-52), so
//! every trait gains one method that writes the value's record into a builder and returns the record's index.
//! Shown for the types of the Cornell smoke scene; the remaining impls follow the same pattern with the parameter
//! slots listed next to `mrt_shape` in include/mrt.h.
use mrt_sys::*;

#[derive(Default)]
pub struct SceneBuilder {
    pub textures: Vec<MrtTexture>,
    pub materials: Vec<MrtMaterial>,
    pub shapes: Vec<MrtShape>,
    pub objects: Vec<MrtObject>,
    pub lights: Vec<MrtObject>,
}

pub trait FlattenTexture { fn flatten(&self, b: &mut SceneBuilder) -> i32; }
pub trait FlattenMaterial { fn flatten(&self, b: &mut SceneBuilder) -> i32; }
/// Added to `trait Hittable` (hittable.rs:36-52) as a required method.
pub trait FlattenHittable { fn flatten(&self, b: &mut SceneBuilder) -> i32; }

impl FlattenTexture for SolidColor {                       // texture.rs:33-44
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        b.textures.push(MrtTexture { kind: 0 /* MRT_TEX_SOLID */, a: -1, b: -1, image: -1,
                                     color: [self.color.red, self.color.green, self.color.blue], reserved: 0 });
        (b.textures.len() - 1) as i32
    }
}
impl FlattenMaterial for Lambertian {                      // material.rs:26-78
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let texture = self.albedo.flatten(b);
        b.materials.push(MrtMaterial { kind: 0 /* MRT_MAT_LAMBERTIAN */, texture, fuzz: 0.0, index_refraction: 0.0,
                                       color: [0.0; 3], reserved: 0 });
        (b.materials.len() - 1) as i32
    }
}
impl FlattenHittable for XZRect {                          // rect.rs:132-183; slots: x0 x1 z0 z1 k normal_flip
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let material = self.material.flatten(b);
        let mut p = [0.0f64; 10];
        p[..6].copy_from_slice(&[self.x0, self.x1, self.z0, self.z1, self.k, self.normal_flip]);
        b.shapes.push(MrtShape { kind: 3 /* MRT_SHAPE_XZ_RECT */, material, boundary: -1, octree: -1, p });
        (b.shapes.len() - 1) as i32
    }
}
impl FlattenHittable for ConstantMedium {                  // constant_medium.rs:10-33; the boundary comes first
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let boundary = self.boundary.flatten(b);
        let material = self.phase_function.flatten(b);
        let mut p = [0.0f64; 10];
        p[0] = -1.0 / self.neg_inv_density;                  // the record holds the density itself
        b.shapes.push(MrtShape { kind: 6 /* MRT_SHAPE_CONSTANT_MEDIUM */, material, boundary, octree: -1, p });
        (b.shapes.len() - 1) as i32
    }
}
impl Object {                                              // hittable.rs:184-201
    pub fn flatten(&self, b: &mut SceneBuilder) -> MrtObject {
        let shape = self.shape.flatten(b);
        let m: &[f64; 16] = self.transform.world_transform.as_ref();   // cgmath stores column-major, as the record wants
        MrtObject { shape, reserved: 0, world_transform: *m }
    }
}

/// `WorldInfo::build_world` (world.rs:64-76) additionally builds the device scene.
pub fn create_gpu_scene(info: &WorldInfo, device: i32) -> *mut MrtScene {
    let mut b = SceneBuilder::default();
    for o in &info.objects { let rec = o.flatten(&mut b); b.objects.push(rec); }
    for o in &info.lights { let rec = o.flatten(&mut b); b.lights.push(rec); }
    // counts and pointers of the builder's vectors; camera = CameraInfo field by field (camera.rs:14-24);
    // background / sun as their kinds and parameters; bvh_seed = any u64 (the reference draws its split axes
    // from the global generator, bvh.rs:215); abi_version = mrt_abi_version()
    let desc: MrtSceneDesc = scene_desc(&b, info);
    let mut err = [0 as std::os::raw::c_char; 256];
    let scene = unsafe { mrt_scene_create(&desc, device, err.as_mut_ptr(), err.len()) };
    assert!(!scene.is_null(), "mrt_scene_create: {:?}", unsafe { std::ffi::CStr::from_ptr(err.as_ptr()) });
    scene
}

/// `RayTracer::trace_part` (ray_tracer.rs:435-463) with a device scene: one call adds `spp` samples to the strip.
pub fn trace_part_gpu(scene: *mut MrtScene, frame: &mut [f32], total_width: u32, height: u32, x0: u32, strip_w: u32,
                      spp: u32, seed: u64, first_sample: u64) {
    let rc = unsafe { mrt_render_accumulate(scene, frame.as_mut_ptr(), total_width, height, x0, strip_w, spp, seed,
                                            first_sample, std::ptr::null_mut()) };
    assert_eq!(rc, 0, "{:?}", unsafe { std::ffi::CStr::from_ptr(mrt_last_error()) });
}
