// Points rustc at the in-tree libsar_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("SAR_HIP_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../../strange_attractor_renderer_amd", env!("CARGO_MANIFEST_DIR"))
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sar_hip");
    println!("cargo:rerun-if-env-changed=SAR_HIP_LIB_DIR");
}
