// Links libws_hip.so (hipcc --offload-arch=gfx950; built by `python -c "import __graft_entry__ as g; g.build()"`).
// WS_HIP_LIB_DIR = directory that holds it; default: the in-tree location next to this crate.
fn main() {
    let default = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("../rustronomy-watershed_amd");
    let dir = std::env::var("WS_HIP_LIB_DIR").map(std::path::PathBuf::from).unwrap_or(default);
    let dir = dir.canonicalize().expect("WS_HIP_LIB_DIR does not exist");
    assert!(dir.join("libws_hip.so").exists(), "{} holds no libws_hip.so: build the HIP engine first", dir.display());
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ws_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=WS_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/ws_hip.h");
}
