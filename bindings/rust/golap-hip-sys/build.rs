// Links libgolap_hip.so (built by `make -C gpu_olap_engine_amd/csrc`, ahead-of-time gfx950 code objects; no JIT).
// GOLAP_LIB_DIR = directory that holds libgolap_hip.so (default: the in-tree build output).
fn main() {
    let dir = std::env::var("GOLAP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../../gpu_olap_engine_amd/lib").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=golap_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=GOLAP_LIB_DIR");
}
