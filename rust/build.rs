// UNVERIFIED (never compiled).  Builds libvrt.so with the repository's Makefile (hipcc --offload-arch=gfx950)
// and links it.  VRT_ROOT names the checkout that holds raytracing_amd/csrc and include/vrt.h.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("VRT_ROOT").expect("set VRT_ROOT to the checkout that holds include/vrt.h"));
    let status = Command::new("make")
        .arg("-C")
        .arg(root.join("raytracing_amd/csrc"))
        .status()
        .expect("make not found");
    assert!(status.success(), "hipcc build of libvrt.so failed");
    println!("cargo:rustc-link-search=native={}", root.join("raytracing_amd").display());
    println!("cargo:rustc-link-lib=dylib=vrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", root.join("raytracing_amd").display());
    println!("cargo:rerun-if-changed={}", root.join("include/vrt.h").display());
    println!("cargo:rerun-if-env-changed=VRT_ROOT");
}
