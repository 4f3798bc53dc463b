// Links libjpegenc_mi355x.so (built by jpeg-encoder_amd/csrc/build.sh: hipcc --offload-arch=gfx950).
// JPEGENC_MI355X_LIB_DIR overrides the in-tree location.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=JPEGENC_MI355X_LIB_DIR");
    let dir = match env::var_os("JPEGENC_MI355X_LIB_DIR") {
        Some(d) => PathBuf::from(d),
        None => {
            let manifest = PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR"));
            manifest.join("..").join("..").join("jpeg-encoder_amd")
        }
    };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=jpegenc_mi355x");
    // so that `cargo test` / `cargo run` find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
