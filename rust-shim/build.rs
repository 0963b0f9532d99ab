fn main() {
    // point SUMCHECK_HIP_LIB_DIR at the directory holding libsumcheck_hip.so (sumcheck_amd/ in this repo)
    if let Ok(dir) = std::env::var("SUMCHECK_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=sumcheck_hip");
}
