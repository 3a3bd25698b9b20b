// Links libstacker_amd.so (built by `make -C libstacker_rs_amd/csrc`; the library's RUNPATH finds the ROCm runtime).
// STACKER_AMD_LIB_DIR names the directory that holds it; default: the in-tree build next to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=STACKER_AMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/stacker.h");
    let dir = env::var("STACKER_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("libstacker_rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=stacker_amd");
    // so that `cargo run` / `cargo test` find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
