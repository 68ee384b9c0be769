// Links libferreus_bbfmm_hip.so (built by `python -m ferreus_rbf_rs_amd.build`, hipcc --offload-arch=gfx950).
// FERREUS_BBFMM_HIP_LIB_DIR names the directory holding it (default: ../../ferreus_rbf_rs_amd relative to
// this crate when it still sits in the ferreus_rbf_rs_amd repository under integration/).
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=FERREUS_BBFMM_HIP_LIB_DIR");
    let dir = env::var("FERREUS_BBFMM_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ferreus_rbf_rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ferreus_bbfmm_hip");
    // the library itself links libamdhip64; make the loader find both at run time
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
}
