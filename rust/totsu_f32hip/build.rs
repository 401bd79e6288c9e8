// AUTHORED, NOT COMPILED (no cargo here).  Links the C-ABI library built by `make -C totsu_amd/csrc`.
fn main() {
    let dir = std::env::var("TOTSU_F32HIP_LIB_DIR").unwrap_or_else(|_| "../../totsu_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=totsu_f32hip");
    println!("cargo:rerun-if-env-changed=TOTSU_F32HIP_LIB_DIR");
}
