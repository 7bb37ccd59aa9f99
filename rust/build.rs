// UNVERIFIED (never compiled).  Builds libvrt.so with the repository's Makefile and links it.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = env::var("VRT_ROOT")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join(".."));
    let status = Command::new("make")
        .arg("-C")
        .arg(root.join("raytracing_amd/csrc"))
        .status()
        .expect("make not found");
    assert!(status.success(), "hipcc build of libvrt.so failed");
    println!("cargo:rustc-link-search=native={}", root.join("raytracing_amd").display());
    println!("cargo:rustc-link-lib=dylib=vrt");
    println!("cargo:rerun-if-changed={}", root.join("include/vrt.h").display());
}
