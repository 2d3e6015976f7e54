#!/bin/bash
# One command for the reference-parity run of the Rust shim (pins the tag a9 and the encryption construction
# f4).  Needs: an MI355X (gfx950) with ROCm, a Rust toolchain >= 1.85, and the dusk crates — from crates.io, or vendored
# (see "offline" below).  Neither exists in the image this repository is built in, so this script has never been run there;
# tests/test_rust_bindings.py checks that every file it names exists and that sys.rs matches the header.
#
#   bash bindings/rust/run_parity.sh            # online: cargo fetches dusk-poseidon 0.42.0-rc.0, dusk-safe 0.3, ...
#   bash bindings/rust/run_parity.sh --offline  # uses bindings/rust/vendor/ (made on a connected machine, see below)
#
# Output: bindings/rust/RUSTPARITY.diff — the run against the library's PREDICTION (RUSTPARITY.expected.json: tag-input bytes and tag
# limbs per shape, STREAM / DUPLEX verdict per length): one line per difference, so the first person to run this reads a diff, not a
# log — and bindings/rust/RUSTPARITY.json — {"ok": bool, "tests": {...}, "tag_inputs": {...}, "encryption": {"2": {"stream": bool,
# "duplex": bool}, ...}, "log": "RUSTPARITY.log"} — and the full cargo log beside it.  Paste the JSON into DESIGN.md §5.
#
# offline: on a machine WITH network access, in bindings/rust:   cargo vendor vendor > .cargo/vendor-config.toml
# then copy bindings/rust/vendor/ and .cargo/vendor-config.toml to the GPU box; --offline appends that file to the cargo
# configuration (CARGO_HOME-independent: `--config .cargo/vendor-config.toml`) and passes --offline to cargo.
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OFFLINE=()
if [ "${1:-}" = "--offline" ]; then
    [ -f "$HERE/.cargo/vendor-config.toml" ] && [ -d "$HERE/vendor" ] || { echo "run_parity.sh --offline: bindings/rust/vendor/ and .cargo/vendor-config.toml are missing (cargo vendor, see the header of this script)" >&2; exit 2; }
    OFFLINE=(--offline --config "$HERE/.cargo/vendor-config.toml")
fi
command -v cargo >/dev/null || { echo "run_parity.sh: no cargo on PATH (this is the step the build image cannot do)" >&2; exit 2; }
# 1. the HIP library
(cd "$ROOT" && python -m poseidon252_amd.build) || exit 1
export POSEIDON252_HIP_DIR="$ROOT/poseidon252_amd"
export LD_LIBRARY_PATH="$POSEIDON252_HIP_DIR:/opt/rocm/lib:${LD_LIBRARY_PATH:-}"
# 2. the parity tests against the real crates; parity.rs prints one `RUSTPARITY <key> <json>` line per fact
cd "$HERE"
cargo test --release "${OFFLINE[@]}" -- --nocapture --test-threads=1 2>&1 | tee RUSTPARITY.log
RC=${PIPESTATUS[0]}
# 3. collect
python3 - "$RC" <<'PY'
import json, re, sys
rc = int(sys.argv[1])
log = open("RUSTPARITY.log").read()
out = {"ok": rc == 0, "cargo_exit_code": rc, "tests": {}, "tag_inputs": {}, "encryption": {}, "truncated": {}, "trees": {}, "log": "RUSTPARITY.log"}
for m in re.finditer(r"^test (\S+) \.\.\. (\w+)", log, re.M):
    out["tests"][m.group(1)] = m.group(2)
for m in re.finditer(r"^RUSTPARITY (\S+) (\{.*\})$", log, re.M):
    kind, body = m.group(1), json.loads(m.group(2))
    if kind == "tag_input":
        out["tag_inputs"][body["pattern"]] = body
    elif kind == "encryption":
        out["encryption"][str(body["len"])] = body
    elif kind in ("truncated", "trees"):
        out[kind] = body
json.dump(out, open("RUSTPARITY.json", "w"), indent=1)
print("wrote", "bindings/rust/RUSTPARITY.json:", "PASS" if out["ok"] else "FAIL", out["tests"])
PY
# 4. against the library's prediction (RUSTPARITY.expected.json, tools/gen_rustparity_expected.py): one line per difference —
#    tag-input bytes, tag limbs, STREAM / DUPLEX verdict per length, test outcomes — or the statement that everything is pinned
python3 "$ROOT/tools/gen_rustparity_expected.py" --diff RUSTPARITY.json | tee RUSTPARITY.diff
exit $RC
