// UNTESTED SOURCE.  Point SAPCA_LIB_DIR at single-algebra_amd/lib (where `make` puts libsapca.so).
fn main() {
    let dir = std::env::var("SAPCA_LIB_DIR").unwrap_or_else(|_| "../../../lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sapca");
    println!("cargo:rerun-if-env-changed=SAPCA_LIB_DIR");
}
