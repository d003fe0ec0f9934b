// Links the crate against libmjx.so (jpeg-rust_amd/libmjx.so, built by `python jpeg-rust_amd/build.py`).
// MJX_LIB_DIR names the directory that holds it.
fn main() {
    let dir = std::env::var("MJX_LIB_DIR").unwrap_or_else(|_| "../../jpeg-rust_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mjx");
    println!("cargo:rerun-if-env-changed=MJX_LIB_DIR");
}
