// build.rs of peterbudai/redux with the `hip` module enabled: links libredux_hip.so (the C ABI of
// include/redux_hip.h).  The reference crate has no build script; this is the only build-side
// addition.  Cargo.toml gains:   build = "build.rs"   under [package].
//
// REDUX_HIP_LIB_DIR = the directory that holds libredux_hip.so (redux_amd/ after
// `python -m redux_amd.build`); ROCm's libamdhip64.so.7 must be on the loader path.
fn main() {
    let dir = std::env::var("REDUX_HIP_LIB_DIR").expect("set REDUX_HIP_LIB_DIR to the directory of libredux_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=redux_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=REDUX_HIP_LIB_DIR");
}
