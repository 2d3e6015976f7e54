// Links libposeidon252_hip.so; POSEIDON252_HIP_DIR must name the directory that holds it
// (poseidon252_amd/ after `python -m poseidon252_amd.build`).
fn main() {
    let dir = std::env::var("POSEIDON252_HIP_DIR").expect("set POSEIDON252_HIP_DIR to the directory of libposeidon252_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=poseidon252_hip");
    println!("cargo:rerun-if-env-changed=POSEIDON252_HIP_DIR");
}
