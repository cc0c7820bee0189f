#!/bin/bash
# Regenerates tests/golden/fullsize_digests.json's content with the Rust reference itself.
# Needs cargo + crates.io access; cannot run in the build image (see README.md).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${TMPDIR:-/tmp}/hodor_ref_recipe   # scratch OUTSIDE the repository: the crate's sources must never sit under a path gpurun ships
command -v cargo >/dev/null || { echo "no cargo: the reference is unbuildable here (expected in the build image)"; exit 3; }
rm -rf "$OUT/hodor" && mkdir -p "$OUT" && cp -r /root/reference "$OUT/hodor"
cp "$HERE/gen_fixtures.rs" "$OUT/hodor/src/gen_fixtures.rs"
printf '\n#[cfg(test)]\nmod gen_fixtures;\n' >> "$OUT/hodor/src/lib.rs"
mkdir -p "$HERE/../_ref"
OFFLINE=""; [ "${CARGO_NET_OFFLINE:-}" = "true" ] && OFFLINE="--offline"   # a vendored / mirrored registry (README.md lists the crates)
(cd "$OUT/hodor" && HODOR_FIXTURES_OUT="$HERE/../_ref/fullsize_digests_rust.json" \
    cargo test $OFFLINE --release gen_fixtures -- --nocapture --ignored)
python3 "$HERE/compare.py" "$HERE/../_ref/fullsize_digests_rust.json" "$HERE/../../tests/golden/fullsize_digests.json"
