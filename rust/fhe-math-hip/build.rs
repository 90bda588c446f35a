//! Tells cargo where libfhe_hip.so lives: FHE_HIP_LIB_DIR, or `<repo>/fhe.rs_amd` next to this crate
//! (where `python -c "import __graft_entry__ as g; g.build()"` puts it).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var_os("FHE_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("../../fhe.rs_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=fhe_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=FHE_HIP_LIB_DIR");
}
