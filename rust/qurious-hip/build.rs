// Link against libqhip.so, built by `python -c "import __graft_entry__ as g; g.build()"` into qurious_amd/ (needs /opt/rocm).
fn main() {
    let dir = std::env::var("QHIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR");
        format!("{here}/../../qurious_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=qhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=QHIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/qhip.h");
}
