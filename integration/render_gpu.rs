// src/render_gpu.rs -- replacement for the body of renderer::render (src/renderer.rs:92-182) that drives
// the MI355X path-tracing core through src/ffi.rs instead of spawning settings.thread_count CPU workers.
//
// What stays as it is in the reference: yaml loading (src/scene.rs, src/main.rs:244-302), Film, Camera,
// the ggez window and its polling of the ThreadMessage channel (src/main.rs:113-131), OIDN (src/denoise.rs).
// What changes: `renderer::render` calls `render_gpu::render` (same signature, same return value), and
// Film gains `render_window` and `merge_accumulators` (at the end of this file).
//
// Visibility changes this file needs in the reference (fields it reads that are private today):
//   objects/triangle.rs:22-27   p1, p2, n0, n1, n2             -> pub(crate)
//   materials/matte.rs:12-13    reflectance_color, roughness   -> pub(crate)
//   materials/plastic.rs:18-21  diffuse, specular, roughness, ior -> pub(crate)
//   materials/mirror.rs:11      reflectance_color              -> pub(crate)
//   materials/glass.rs:12-14    ior, reflection_color, refraction_color -> pub(crate)
//   lights/area.rs:14-15        object, intensity              -> pub(crate)
//   lights/distant.rs:15-16     direction, intensity           -> pub(crate)
//   lights/point.rs:12-13       position, intensity            -> pub(crate)
//   lights/infinite_area.rs:19  mip_map                        -> pub(crate)
//   textures/mip_map.rs:14      image                          -> pub(crate)
//   camera.rs:22,26             camera_to_world, raster_to_camera -> pub(crate)
//   film.rs:55-56,60-63         crop_start, crop_end, filter_radius, filter_method, bucket_size -> pub(crate)
//
// This file cannot be compiled in the build image (no rustc); it is written against the reference's
// types as they are at the cited lines.

use std::collections::HashMap;
use std::sync::mpsc::{self, Receiver, Sender};
use std::sync::Arc;
use std::thread::{self, JoinHandle};

use crate::camera::Camera;
use crate::ffi;
use crate::film::{Film, FilterMethod};
use crate::lights::Light;
use crate::materials::Material;
use crate::objects::{ArcObject, Object};
use crate::renderer::{Settings, ThreadMessage};
use crate::sampler::SobolSampler;
use crate::scene::Scene;

/// Samples rendered per rrt_render call: after each chunk the accumulators are merged into the Film,
/// so the window keeps updating as it does with the CPU workers' buckets.
const CHUNK_SAMPLES: u32 = 16;

/// The scene as flat arrays in the order of `Scene.objects` (src/scene.rs:64-106, 315-325); owns
/// everything the `RrtSceneDesc` it hands out points into.
pub struct FlatScene {
    tri_pos: Vec<f32>,
    tri_nrm: Vec<f32>,
    tri_mat: Vec<u32>,
    rects: Vec<ffi::RrtRect>,
    planes: Vec<ffi::RrtPlane>,
    spheres: Vec<ffi::RrtSphere>,
    materials: Vec<ffi::RrtMaterial>,
    lights: Vec<ffi::RrtLight>,
    env_rgb8: Vec<u8>,
    env_w: u32,
    env_h: u32,
}

// The raw pointers inside RrtSceneDesc are only ever built from a live &FlatScene.
unsafe impl Send for FlatScene {}
unsafe impl Sync for FlatScene {}

fn v3(v: &nalgebra::Vector3<f64>) -> [f64; 3] {
    [v.x, v.y, v.z]
}
fn p3(p: &nalgebra::Point3<f64>) -> [f64; 3] {
    [p.x, p.y, p.z]
}

fn flat_material(m: &Material) -> ffi::RrtMaterial {
    let zero = [0.0; 3];
    match m {
        // constructor argument order: materials/matte.rs:17-23, plastic.rs:24-33, mirror.rs:14-17, glass.rs:17-29
        Material::Matte(x) => ffi::RrtMaterial { kind: ffi::RRT_MAT_MATTE, _pad: 0, kd: v3(&x.reflectance_color), ks: zero, roughness: x.roughness, ior: 0.0 },
        Material::Plastic(x) => ffi::RrtMaterial { kind: ffi::RRT_MAT_PLASTIC, _pad: 0, kd: v3(&x.diffuse), ks: v3(&x.specular), roughness: x.roughness, ior: x.ior },
        Material::Mirror(x) => ffi::RrtMaterial { kind: ffi::RRT_MAT_MIRROR, _pad: 0, kd: v3(&x.reflectance_color), ks: zero, roughness: 0.0, ior: 0.0 },
        Material::Glass(x) => ffi::RrtMaterial { kind: ffi::RRT_MAT_GLASS, _pad: 0, kd: v3(&x.reflection_color), ks: v3(&x.refraction_color), roughness: 0.0, ior: x.ior },
    }
}

impl FlatScene {
    pub fn new(scene: &Scene) -> FlatScene {
        let mut flat = FlatScene {
            tri_pos: vec![], tri_nrm: vec![], tri_mat: vec![], rects: vec![], planes: vec![], spheres: vec![],
            materials: vec![], lights: vec![], env_rgb8: vec![], env_w: 0, env_h: 0,
        };
        // one table entry per distinct Arc<Material> (every triangle of a model shares one, src/scene.rs:319-323)
        let mut material_index: HashMap<*const Material, u32> = HashMap::new();
        let mut index_of = |m: &Arc<Material>, table: &mut Vec<ffi::RrtMaterial>| -> u32 {
            *material_index.entry(Arc::as_ptr(m)).or_insert_with(|| {
                table.push(flat_material(m));
                (table.len() - 1) as u32
            })
        };
        let light_index = |l: &Option<Arc<Light>>| -> i32 {
            match l {
                Some(l) => scene.lights.iter().position(|x| Arc::ptr_eq(x, l)).map(|i| i as i32).unwrap_or(-1),
                None => -1,
            }
        };

        // lights in scene.lights order: yaml order, the environment light last (src/scene.rs:72-132)
        for light in &scene.lights {
            let zero = [0.0; 3];
            flat.lights.push(match light.as_ref() {
                Light::Point(l) => ffi::RrtLight { kind: ffi::RRT_LIGHT_POINT, _pad: 0, intensity: v3(&l.intensity), v0: p3(&l.position), v1: zero, v2: zero },
                Light::Area(l) => match l.object.0.as_ref() {
                    Object::Rectangle(r) => ffi::RrtLight { kind: ffi::RRT_LIGHT_AREA, _pad: 0, intensity: v3(&l.intensity), v0: p3(&r.position), v1: v3(&r.side_a), v2: v3(&r.side_b) },
                    _ => panic!("area lights over shapes other than rectangles are not supported by the device library"),
                },
                Light::Distant(l) => ffi::RrtLight { kind: ffi::RRT_LIGHT_DISTANT, _pad: 0, intensity: v3(&l.intensity), v0: v3(&l.direction), v1: zero, v2: zero },
                Light::InfiniteArea(l) => {
                    // the texels as InfiniteAreaLight::new left them (lights/infinite_area.rs:113-139)
                    let image = &l.mip_map.image;
                    flat.env_w = image.width();
                    flat.env_h = image.height();
                    flat.env_rgb8 = image.as_raw().clone();
                    ffi::RrtLight { kind: ffi::RRT_LIGHT_INFINITE, _pad: 0, intensity: [1.0; 3], v0: zero, v1: zero, v2: zero }
                }
            });
        }

        // Object ids follow Scene.objects.  The device numbers triangles first, then rectangles, planes and
        // spheres; the loader produces exactly that order (mesh triangles, then the light rectangles).
        for ArcObject(object) in &scene.objects {
            match object.as_ref() {
                Object::Triangle(t) => {
                    assert!(flat.rects.is_empty() && flat.planes.is_empty() && flat.spheres.is_empty(),
                            "triangles must come first in Scene.objects");
                    for p in [&t.p0, &t.p1, &t.p2] {
                        flat.tri_pos.extend_from_slice(&[p.x as f32, p.y as f32, p.z as f32]);
                    }
                    for n in [&t.n0, &t.n1, &t.n2] {
                        flat.tri_nrm.extend_from_slice(&[n.x as f32, n.y as f32, n.z as f32]);
                    }
                    flat.tri_mat.push(index_of(&t.materials[0], &mut flat.materials));
                }
                Object::Rectangle(r) => flat.rects.push(ffi::RrtRect {
                    pos: p3(&r.position), side_a: v3(&r.side_a), side_b: v3(&r.side_b),
                    material: index_of(&r.materials[0], &mut flat.materials), light: light_index(&r.light),
                }),
                Object::Plane(p) => flat.planes.push(ffi::RrtPlane {
                    pos: p3(&p.position), normal: v3(&p.normal), material: index_of(&p.materials[0], &mut flat.materials), _pad: 0,
                }),
                Object::Sphere(s) => flat.spheres.push(ffi::RrtSphere {
                    center: p3(&s.position), radius: s.radius, material: index_of(&s.materials[0], &mut flat.materials), _pad: 0,
                }),
            }
        }
        flat
    }

    pub fn desc(&self) -> ffi::RrtSceneDesc {
        let env_rgb8 = if self.env_rgb8.is_empty() { std::ptr::null() } else { self.env_rgb8.as_ptr() };
        ffi::RrtSceneDesc {
            n_tris: self.tri_mat.len() as u32,
            n_rects: self.rects.len() as u32,
            n_planes: self.planes.len() as u32,
            n_spheres: self.spheres.len() as u32,
            n_materials: self.materials.len() as u32,
            n_lights: self.lights.len() as u32,
            tri_pos: self.tri_pos.as_ptr(),
            tri_nrm: self.tri_nrm.as_ptr(),
            tri_mat: self.tri_mat.as_ptr(),
            rects: self.rects.as_ptr(),
            planes: self.planes.as_ptr(),
            spheres: self.spheres.as_ptr(),
            materials: self.materials.as_ptr(),
            lights: self.lights.as_ptr(),
            env_rgb8,
            env_w: self.env_w,
            env_h: self.env_h,
            rect_aabb_mode: ffi::RRT_RECT_AABB_STRICT,
            _pad: 0,
        }
    }
}

/// nalgebra stores Matrix4 column by column; the C ABI takes rows.
fn row_major(m: &nalgebra::Matrix4<f64>) -> [f64; 16] {
    let mut out = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            out[4 * r + c] = m[(r, c)];
        }
    }
    out
}

/// Same signature and return value as renderer::render: one host thread per GPU instead of
/// settings.thread_count CPU workers, each sending one `finished` message when its share is done
/// (src/main.rs:113-131 counts them; `running_threads = threads.len()` at src/main.rs:338).
/// GPU g of G renders sample indices [g * max_samples / G, (g + 1) * max_samples / G) of every pixel.
pub fn render(
    scene: Scene,
    settings: Settings,
    _sampler: SobolSampler, // the device draws its own counter-based random numbers
    camera: Arc<Camera>,
) -> (Vec<JoinHandle<()>>, Receiver<ThreadMessage>) {
    let flat = Arc::new(FlatScene::new(&scene));
    let (sender, receiver): (Sender<ThreadMessage>, Receiver<ThreadMessage>) = mpsc::channel();
    let gpus = unsafe { ffi::rrt_device_count() };
    assert!(gpus > 0, "no HIP device: {}", ffi::last_error(std::ptr::null_mut()));
    let gpus = gpus as u32;
    let mut threads: Vec<JoinHandle<()>> = vec![];

    for gpu in 0..gpus {
        let thread_flat = flat.clone();
        let thread_camera = camera.clone();
        let thread_sender = sender.clone();

        let thread = thread::spawn(move || unsafe {
            let ctx = ffi::rrt_create(gpu as i32);
            assert!(!ctx.is_null(), "rrt_create({}): {}", gpu, ffi::last_error(std::ptr::null_mut()));
            let desc = thread_flat.desc();
            ffi::check(ctx, "rrt_upload_scene", ffi::rrt_upload_scene(ctx, &desc));

            let raster_to_camera = row_major(&thread_camera.raster_to_camera);
            let camera_to_world = row_major(&thread_camera.camera_to_world);
            ffi::check(ctx, "rrt_set_camera", ffi::rrt_set_camera(
                ctx, raster_to_camera.as_ptr(), camera_to_world.as_ptr(), thread_camera.aperture, thread_camera.focal_distance));

            let (width, height, window, filter_method, filter_radius) = {
                let film = thread_camera.film.read().unwrap();
                (film.image_size.x, film.image_size.y, film.render_window(), film.filter_method_code(), film.filter_radius)
            };
            ffi::check(ctx, "rrt_film_alloc", ffi::rrt_film_alloc(ctx, width, height));
            ffi::check(ctx, "rrt_set_filter", ffi::rrt_set_filter(ctx, filter_method, filter_radius));
            let device_settings = ffi::RrtSettings {
                width, height,
                crop_x0: window[0], crop_y0: window[1], crop_x1: window[2], crop_y1: window[3],
                depth_limit: settings.depth_limit, _pad: 0, clamp: settings.clamp, seed: 0,
            };

            let begin = (settings.max_samples as u64 * gpu as u64 / gpus as u64) as u32;
            let end = (settings.max_samples as u64 * (gpu as u64 + 1) / gpus as u64) as u32;
            let mut accumulators = vec![0f32; width as usize * height as usize * 4];
            let mut first = begin;
            while first < end {
                let last = (first + CHUNK_SAMPLES).min(end);
                ffi::check(ctx, "rrt_film_clear", ffi::rrt_film_clear(ctx));
                ffi::check(ctx, "rrt_render", ffi::rrt_render(ctx, &device_settings, first, last));
                ffi::check(ctx, "rrt_film_download", ffi::rrt_film_download(ctx, accumulators.as_mut_ptr()));
                // keep the write lock as short as possible (as src/renderer.rs:149-154 does)
                thread_camera.film.write().unwrap().merge_accumulators(&accumulators);
                first = last;
            }
            ffi::rrt_destroy(ctx);
            println!("GPU {gpu} done, samples {begin}..{end} of every pixel rendered");

            thread_sender
                .send(ThreadMessage { exit: false, finished: true })
                .unwrap();
        });

        threads.push(thread);
    }

    (threads, receiver)
}

// ---- additions to `impl Film`: this block goes INTO src/film.rs, next to merge_bucket_pixels_to_image_buffer
// (src/film.rs:221): it uses that module's private xyz_to_srgb / gamma_correct_srgb (src/film.rs:360-394).
impl Film {
    /// The pixels the bucket queue of init_buckets covers (src/film.rs:263-283): whole buckets laid
    /// from crop_start, capped at the image size: [x0, y0, x1, y1).
    pub fn render_window(&self) -> [u32; 4] {
        match (self.crop_start, self.crop_end) {
            (Some(start), Some(end)) => {
                let buckets_x = ((end.x - start.x) as f64 / self.bucket_size.x as f64).ceil() as u32;
                let buckets_y = ((end.y - start.y) as f64 / self.bucket_size.y as f64).ceil() as u32;
                [
                    start.x,
                    start.y,
                    (start.x + buckets_x * self.bucket_size.x).min(self.image_size.x),
                    (start.y + buckets_y * self.bucket_size.y).min(self.image_size.y),
                ]
            }
            _ => [0, 0, self.image_size.x, self.image_size.y],
        }
    }

    /// FilterMethod as the device library numbers it (RRT_FILTER_*).
    pub fn filter_method_code(&self) -> u32 {
        match self.filter_method {
            FilterMethod::None => ffi::RRT_FILTER_NONE,
            FilterMethod::Gaussian => ffi::RRT_FILTER_GAUSSIAN,
            FilterMethod::Mitchell => ffi::RRT_FILTER_MITCHELL,
        }
    }

    /// Adds device accumulators ([height][width][4] = sum r, g, b, sum weight) to the pixels and
    /// refreshes the 8-bit image of every pixel that received samples: what
    /// merge_bucket_pixels_to_image_buffer does for one bucket (src/film.rs:221-251).
    pub fn merge_accumulators(&mut self, accumulators: &[f32]) {
        let width = self.image_size.x;
        for (i, acc) in accumulators.chunks_exact(4).enumerate() {
            if acc[3] == 0.0 {
                continue;
            }
            let pixel = &mut self.pixels[i];
            pixel.sum_radiance += nalgebra::Vector3::new(acc[0] as f64, acc[1] as f64, acc[2] as f64);
            pixel.sum_weight += acc[3] as f64;
            let (x, y) = (i as u32 % width, i as u32 / width);
            if pixel.sum_weight < f64::EPSILON {
                self.image_buffer.put_pixel(x, y, image::Rgb([0, 0, 0]));
                continue;
            }
            let rgb = xyz_to_srgb(pixel.sum_radiance / pixel.sum_weight);
            self.image_buffer.put_pixel(
                x,
                y,
                image::Rgb([
                    (gamma_correct_srgb(rgb.x) * 255.0) as u8,
                    (gamma_correct_srgb(rgb.y) * 255.0) as u8,
                    (gamma_correct_srgb(rgb.z) * 255.0) as u8,
                ]),
            );
        }
    }
}
