// src/ffi.rs -- Rust declarations of the C ABI of librrt_hip.so (include/rrt.h), the MI355X path-tracing core.
//
// Drop this file into the reference's src/, add `mod ffi;` to src/main.rs and link with
// `cargo:rustc-link-lib=dylib=rrt_hip` (build.rs) or RUSTFLAGS="-L <dir> -l rrt_hip".
// Every struct is #[repr(C)] with the field names, order and types of the header; every function of
// the header is declared.  tests/test_rust_shim.py checks both against include/rrt.h (names, order,
// sizes, offsets, constants) without a Rust toolchain.
#![allow(dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const RRT_OK: c_int = 0;
pub const RRT_E_INVALID: c_int = -1;
pub const RRT_E_DEVICE: c_int = -2;
pub const RRT_E_STATE: c_int = -3;
pub const RRT_E_NOMEM: c_int = -4;
pub const RRT_E_UNSUPPORTED: c_int = -5;

pub const RRT_MAT_MATTE: u32 = 0;
pub const RRT_MAT_PLASTIC: u32 = 1;
pub const RRT_MAT_MIRROR: u32 = 2;
pub const RRT_MAT_GLASS: u32 = 3;

pub const RRT_LIGHT_POINT: u32 = 0;
pub const RRT_LIGHT_AREA: u32 = 1;
pub const RRT_LIGHT_DISTANT: u32 = 2;
pub const RRT_LIGHT_INFINITE: u32 = 3;

pub const RRT_RECT_AABB_STRICT: u32 = 0;
pub const RRT_RECT_AABB_INCLUSIVE: u32 = 1;
pub const RRT_RECT_AABB_OFF: u32 = 2;

pub const RRT_FILTER_NONE: u32 = 0;
pub const RRT_FILTER_GAUSSIAN: u32 = 1;
pub const RRT_FILTER_MITCHELL: u32 = 2;

pub const RRT_OPT_TIMING: c_int = 1;
pub const RRT_OPT_BATCH_SAMPLES: c_int = 2;
pub const RRT_OPT_MAX_SLOTS: c_int = 3;
pub const RRT_OPT_COUNT_TRAVERSAL: c_int = 4;
pub const RRT_OPT_SIDE_STREAM: c_int = 5;
pub const RRT_OPT_AOVS: c_int = 6;
pub const RRT_OPT_TAIL_PATHS: c_int = 7;
pub const RRT_OPT_DEVICE_BVH: c_int = 8;

pub const RRT_NOALLOC_MAX_RAYS: u32 = 65536;

/// Opaque context (one per GPU; driven by one host thread at a time).
#[repr(C)]
pub struct RrtCtx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtMaterial {
    pub kind: u32,
    pub _pad: u32,
    pub kd: [f64; 3],
    pub ks: [f64; 3],
    pub roughness: f64,
    pub ior: f64,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtLight {
    pub kind: u32,
    pub _pad: u32,
    pub intensity: [f64; 3],
    pub v0: [f64; 3],
    pub v1: [f64; 3],
    pub v2: [f64; 3],
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtRect {
    pub pos: [f64; 3],
    pub side_a: [f64; 3],
    pub side_b: [f64; 3],
    pub material: u32,
    pub light: i32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtPlane {
    pub pos: [f64; 3],
    pub normal: [f64; 3],
    pub material: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtSphere {
    pub center: [f64; 3],
    pub radius: f64,
    pub material: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtSceneDesc {
    pub n_tris: u32,
    pub n_rects: u32,
    pub n_planes: u32,
    pub n_spheres: u32,
    pub n_materials: u32,
    pub n_lights: u32,
    pub tri_pos: *const f32,
    pub tri_nrm: *const f32,
    pub tri_mat: *const u32,
    pub rects: *const RrtRect,
    pub planes: *const RrtPlane,
    pub spheres: *const RrtSphere,
    pub materials: *const RrtMaterial,
    pub lights: *const RrtLight,
    pub env_rgb8: *const u8,
    pub env_w: u32,
    pub env_h: u32,
    pub rect_aabb_mode: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct RrtSettings {
    pub width: u32,
    pub height: u32,
    pub crop_x0: u32,
    pub crop_y0: u32,
    pub crop_x1: u32,
    pub crop_y1: u32,
    pub depth_limit: u32,
    pub _pad: u32,
    pub clamp: f64,
    pub seed: u64,
}

#[repr(C)]
#[derive(Debug, Copy, Clone, Default)]
pub struct RrtStats {
    pub paths: u64,
    pub rays_extend: u64,
    pub rays_shadow: u64,
    pub path_bounces: u64,
    pub ms_raygen: f64,
    pub ms_extend: f64,
    pub ms_shade: f64,
    pub ms_connect: f64,
    pub ms_film: f64,
    pub launches_extend: u64,
    pub nodes_visited: u64,
    pub tris_tested: u64,
    pub bvh_nodes: u64,
    pub bvh_bytes: u64,
    pub nodes_by_kernel: [u64; 3],
    pub tris_by_kernel: [u64; 3],
    pub ms_primary: f64,
    pub launches_primary: u64,
    pub rays_primary: u64,
    pub ms_tail: f64,
    pub launches_tail: u64,
    pub rays_entered_by_kernel: [u64; 3],
}

#[link(name = "rrt_hip")]
extern "C" {
    pub fn rrt_device_count() -> c_int;
    pub fn rrt_create(device: c_int) -> *mut RrtCtx;
    pub fn rrt_destroy(ctx: *mut RrtCtx);
    pub fn rrt_set_stream(ctx: *mut RrtCtx, hip_stream: *mut c_void) -> c_int;
    pub fn rrt_upload_scene(ctx: *mut RrtCtx, scene: *const RrtSceneDesc) -> c_int;
    pub fn rrt_set_camera(
        ctx: *mut RrtCtx,
        raster_to_camera: *const f64,
        camera_to_world: *const f64,
        aperture: f64,
        focal_distance: f64,
    ) -> c_int;
    pub fn rrt_film_alloc(ctx: *mut RrtCtx, width: u32, height: u32) -> c_int;
    pub fn rrt_film_bind(ctx: *mut RrtCtx, device_ptr: *mut f32, width: u32, height: u32) -> c_int;
    pub fn rrt_film_clear(ctx: *mut RrtCtx) -> c_int;
    pub fn rrt_film_download(ctx: *mut RrtCtx, host_sum_rgbw: *mut f32) -> c_int;
    pub fn rrt_film_device_ptr(ctx: *mut RrtCtx) -> *mut c_void;
    pub fn rrt_set_filter(ctx: *mut RrtCtx, method: u32, radius: f64) -> c_int;
    pub fn rrt_film_tonemap(ctx: *mut RrtCtx, host_rgb8: *mut u8) -> c_int;
    pub fn rrt_render(ctx: *mut RrtCtx, settings: *const RrtSettings, sample_begin: u32, sample_end: u32) -> c_int;
    pub fn rrt_synchronize(ctx: *mut RrtCtx) -> c_int;
    pub fn rrt_film_reduce(ctxs: *mut *mut RrtCtx, n: c_int, root: c_int) -> c_int;
    pub fn rrt_intersect(ctx: *mut RrtCtx, org_dir: *const f32, n: u32, prim_id: *mut i32, t: *mut f32) -> c_int;
    pub fn rrt_intersect_noalloc(ctx: *mut RrtCtx, org_dir: *const f32, n: u32, prim_id: *mut i32, t: *mut f32) -> c_int;
    pub fn rrt_occluded(ctx: *mut RrtCtx, org_dir_tmax: *const f32, n: u32, hit: *mut u8) -> c_int;
    pub fn rrt_bvh_selfcheck(tri_pos: *const f32, n_tris: u32, n_nodes: *mut u32, max_depth: *mut u32) -> i64;
    pub fn rrt_bvh_check_uploaded(
        ctx: *mut RrtCtx,
        tri_pos: *const f32,
        n_tris: u32,
        n_nodes: *mut u32,
        max_depth: *mut u32,
        built_on_device: *mut c_int,
        build_seconds: *mut f64,
    ) -> i64;
    pub fn rrt_aov_download(ctx: *mut RrtCtx, host_normal_sum: *mut f32, host_albedo_sum: *mut f32) -> c_int;
    pub fn rrt_aov_device_ptr(ctx: *mut RrtCtx) -> *mut c_void;
    pub fn rrt_set_option(ctx: *mut RrtCtx, option: c_int, value: i64) -> c_int;
    pub fn rrt_get_stats(ctx: *mut RrtCtx, out: *mut RrtStats) -> c_int;
    pub fn rrt_reset_stats(ctx: *mut RrtCtx) -> c_int;
    pub fn rrt_last_error(ctx: *mut RrtCtx) -> *const c_char;
}

/// The library's error text for `ctx` (or for a failed rrt_create when `ctx` is null).
pub fn last_error(ctx: *mut RrtCtx) -> String {
    unsafe {
        let p = rrt_last_error(ctx);
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// Panics with the library's message when a call failed: what the reference does on every error
/// (its loader and renderer `unwrap()` / `expect()` throughout).
pub fn check(ctx: *mut RrtCtx, what: &str, rc: c_int) {
    if rc != RRT_OK {
        panic!("{} failed ({}): {}", what, rc, last_error(ctx));
    }
}
