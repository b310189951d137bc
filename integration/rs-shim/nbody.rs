// NOT COMPILED IN THIS REPOSITORY: the build image has no rustc/cargo. This is the
// drop-in replacement for blitzcode/rust-exp `rs-src/nbody.rs` a maintainer adds (option B);
// it keeps the crate's six `#[no_mangle] pub extern fn nb_*` exports and forwards them to the level-2 C ABI
// of libnbody_mi355x.so (include/nbody_mi355x.h). `lib.rs` keeps `pub mod nbody;` unchanged.
// rs-src/nbody.rs -- shim over libnbody_mi355x (include/nbody_mi355x.h, level 2)
use std::sync::Mutex;
use std::os::raw::{c_char, c_void};

#[repr(C)] pub struct NbxEngine { _private: [u8; 0] }

#[link(name = "nbody_mi355x")]
extern "C" {
    fn nbx_create(out: *mut *mut NbxEngine, device: i32) -> i32;
    fn nbx_last_error() -> *const c_char;
    fn nbx_num_particles(e: *const NbxEngine) -> i32;
    fn nbx_random_disk(e: *mut NbxEngine, n: i32) -> i32;
    fn nbx_stable_orbits(e: *mut NbxEngine, n: i32, rmin: f32, rmax: f32) -> i32;
    fn nbx_step_brute_force(e: *mut NbxEngine, dt: f32) -> i32;
    fn nbx_step_barnes_hut(e: *mut NbxEngine, theta: f32, dt: f32, nthreads: i32) -> i32;
    fn nbx_synchronize(e: *mut NbxEngine) -> i32;
    fn nbx_draw(e: *mut NbxEngine, w: i32, h: i32, fb: *mut u32) -> i32;
    // optional: state I/O for replay / checkpoints
    fn nbx_set_particles(e: *mut NbxEngine, n: i32, px: *const f32, py: *const f32,
                         vx: *const f32, vy: *const f32, m: *const f32) -> i32;
    fn nbx_get_particles(e: *mut NbxEngine, cap: i32, px: *mut f32, py: *mut f32,
                         vx: *mut f32, vy: *mut f32, m: *mut f32) -> i32;
}

struct Engine(*mut NbxEngine);
unsafe impl Send for Engine {}

lazy_static! {
    static ref ENGINE: Mutex<Engine> = {
        let mut e: *mut NbxEngine = std::ptr::null_mut();
        let rc = unsafe { nbx_create(&mut e, 0) };
        assert!(rc == 0, "nbx_create failed");
        Mutex::new(Engine(e))
    };
}

fn check(rc: i32) { // same failure mode as the original: panic (poisons the mutex)
    if rc < 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(nbx_last_error()) };
        panic!("nbody_mi355x: {:?}", msg);
    }
}

#[no_mangle] pub extern fn nb_num_particles() -> i32 { unsafe { nbx_num_particles(ENGINE.lock().unwrap().0) } }
#[no_mangle] pub extern fn nb_random_disk(n: i32) { check(unsafe { nbx_random_disk(ENGINE.lock().unwrap().0, n) }) }
#[no_mangle] pub extern fn nb_stable_orbits(n: i32, rmin: f32, rmax: f32) {
    check(unsafe { nbx_stable_orbits(ENGINE.lock().unwrap().0, n, rmin, rmax) })
}
#[no_mangle] pub extern fn nb_step_brute_force(dt: f32) {
    let g = ENGINE.lock().unwrap();
    check(unsafe { nbx_step_brute_force(g.0, dt) }); check(unsafe { nbx_synchronize(g.0) });
}
#[no_mangle] pub extern fn nb_step_barnes_hut(theta: f32, dt: f32, nthreads: i32) {
    let g = ENGINE.lock().unwrap();
    check(unsafe { nbx_step_barnes_hut(g.0, theta, dt, nthreads) }); check(unsafe { nbx_synchronize(g.0) });
}
#[no_mangle] pub extern fn nb_draw(w: i32, h: i32, fb: *mut u32) { check(unsafe { nbx_draw(ENGINE.lock().unwrap().0, w, h, fb) }) }
