// Points the linker at libawry_hip.so.  AWRY_HIP_LIB_DIR names the directory that holds it (the repository builds it as
// awry_amd/lib/libawry_hip.so with `python -m awry_amd.build`); the default is that directory relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("AWRY_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../awry_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=awry_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=AWRY_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/awry_hip.h");
}
