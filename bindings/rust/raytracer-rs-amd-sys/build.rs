// Links librt_amd.so.  RT_AMD_LIB_DIR names the directory that holds it (the repository's raytracer_rs_amd/ after
// `make -C raytracer_rs_amd/csrc`); the library itself depends on libamdhip64 only (RCCL is loaded on demand).
fn main() {
    if let Ok(dir) = std::env::var("RT_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-env-changed=RT_AMD_LIB_DIR");
}
