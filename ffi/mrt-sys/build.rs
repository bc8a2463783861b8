// Links libmrt.so (built with `make lib` in the repository that holds include/mrt.h).
// MRT_LIB_DIR names the directory that contains it.
fn main() {
    let dir = std::env::var("MRT_LIB_DIR").unwrap_or_else(|_| "../../ray_tracing_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mrt");
    println!("cargo:rerun-if-env-changed=MRT_LIB_DIR");
}
