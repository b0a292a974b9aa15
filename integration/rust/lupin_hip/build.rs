// Points the linker at liblupin_hip.so (built by `make -C lupinpathtracer_amd/csrc`).
// LUPIN_HIP_LIB_DIR overrides the default location relative to this crate.
fn main() {
    let dir = std::env::var("LUPIN_HIP_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../../lupinpathtracer_amd", manifest)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lupin_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=LUPIN_HIP_LIB_DIR");
}
