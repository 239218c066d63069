fn main() {
    // libzebra_hip.so is built by `make -C zebra_amd/csrc` (hipcc --offload-arch=gfx950)
    let dir = std::env::var("ZEBRA_HIP_LIB_DIR").unwrap_or_else(|_| "/opt/zebra-hip/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zebra_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZEBRA_HIP_LIB_DIR");
}
