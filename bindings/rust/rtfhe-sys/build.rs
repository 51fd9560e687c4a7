// Links librtfhe_hip.so (built by `python rustfhe_amd/build.py`, hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("RTFHE_LIB_DIR").expect("set RTFHE_LIB_DIR to the directory holding librtfhe_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtfhe_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTFHE_LIB_DIR");
}
