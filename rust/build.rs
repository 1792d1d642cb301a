// Links libmpvss_hip.so (built by `make -C mpvss_rs_amd/csrc`, hipcc --offload-arch=gfx950).
// MPVSS_HIP_LIB_DIR overrides the directory; the default is ../mpvss_rs_amd relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MPVSS_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("mpvss_rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mpvss_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MPVSS_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/mpvss_hip.h");
}
