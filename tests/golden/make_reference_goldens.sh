#!/usr/bin/env bash
# make_reference_goldens.sh -- the recipe that PINS parity tier P1: outputs of the REAL caesiumclt 1.4.0
# (libcaesium 0.20.3 -> mozjpeg-sys 2.2.1, oxipng 9.1.5, libwebp-sys 0.9.5, image 0.25.9; /root/reference/Cargo.lock) on the inputs
# this repository already commits.  It cannot run in the authoring container (no Rust toolchain, no network): run it on any machine
# with `cargo`, then commit the tree it writes, tests/golden/libcaesium/.  tests/test_reference_goldens.py activates by itself
# when that tree exists and reports, per SURVEY 8a row, whether the oracle (and on the MI355X the HIP path) reproduces the bytes.
#
#   usage: tests/golden/make_reference_goldens.sh            (from the repository root)
#
# Nothing here is product code; the reference's own tests hold no golden bytes (/root/reference/src/compressor.rs:786-831, 1026-1068),
# which is why this recipe exists.
set -euo pipefail
cd "$(dirname "$0")"
OUT=libcaesium
BIN=${CAESIUMCLT:-}
if [ -z "$BIN" ]; then
    cargo install caesiumclt --version 1.4.0 --locked --root "$PWD/.cargo-caesiumclt"
    BIN="$PWD/.cargo-caesiumclt/bin/caesiumclt"
fi
rm -rf "$OUT"; mkdir -p "$OUT"
{ echo "tool: $("$BIN" --version)"; echo "host: $(uname -srm)"; echo "date: $(date -u +%F)"; rustc --version 2>/dev/null || true; } > "$OUT/MANIFEST.txt"

JPEGS=(synth*.src.jpg reference_samples/j0.JPG reference_samples/level_1_0/j1.jpg)
PNGS=(reference_samples/p0.png reference_samples/level_1_0/level_2_0/p2.png)
WEBPS=(reference_samples/w0.webp reference_samples/level_1_1/w1.webp)

run() {   # run <recipe name> <input files...> -- <caesiumclt flags...>: one output directory per recipe, file names kept
    local name=$1; shift
    local files=()
    while [ "$1" != "--" ]; do files+=("$1"); shift; done
    shift
    mkdir -p "$OUT/$name"
    "$BIN" --quiet "$@" -o "$OUT/$name" "${files[@]}"
    echo "$name: caesiumclt $* (${#files[@]} files)" >> "$OUT/MANIFEST.txt"
}

# ---- JPEG rows (J1-J10, S1): the mozjpeg JCP_MAX_COMPRESSION profile -- table #3, trellis, deringing, scan search
run jpeg_q80            "${JPEGS[@]}" -- -q 80
run jpeg_q51            "${JPEGS[@]}" -- -q 51
run jpeg_q95            "${JPEGS[@]}" -- -q 95
run jpeg_q80_baseline   "${JPEGS[@]}" -- -q 80 --jpeg-baseline
run jpeg_q80_444        "${JPEGS[@]}" -- -q 80 --jpeg-chroma-subsampling 4:4:4
run jpeg_q80_422        "${JPEGS[@]}" -- -q 80 --jpeg-chroma-subsampling 4:2:2
run jpeg_lossless       "${JPEGS[@]}" -- --lossless
run jpeg_q80_exif       "${JPEGS[@]}" -- -q 80 -e
# ---- R1 + the real resize chain (zune-jpeg decode -> Lanczos3 -> image-rs re-encode -> mozjpeg)
run jpeg_q80_width100   "${JPEGS[@]}" -- -q 80 --width 100
run jpeg_q80_long1500   reference_samples/j0.JPG reference_samples/level_1_0/j1.jpg -- -q 80 --long-edge 1500
# ---- S2: the size walk (j0.JPG itself is a q51 product of it)
run jpeg_max200k        reference_samples/j0.JPG reference_samples/level_1_0/j1.jpg -- --max-size 200KB
# ---- PNG rows (P1-P4): oxipng presets, then the lossy path (imagequant + lodepng)
for lvl in 0 2 3 6; do run png_lossless_o$lvl "${PNGS[@]}" -- --lossless --png-opt-level $lvl; done
run png_q80             "${PNGS[@]}" -- -q 80
run png_lossless_w200   reference_samples/p0.png -- --lossless --width 200
# ---- WebP rows (W1-W3) and the conversions (S3)
run webp_q80            "${WEBPS[@]}" -- -q 80
run webp_lossless       "${WEBPS[@]}" -- --lossless
run jpeg_to_webp_q85_long1500 reference_samples/j0.JPG reference_samples/level_1_0/j1.jpg -- -q 85 --format webp --long-edge 1500
run jpeg_to_webp_q85    synth*.src.jpg -- -q 85 --format webp
run png_to_webp_q80     "${PNGS[@]}" -- -q 80 --format webp
run jpeg_to_png_lossless synth*.src.jpg -- --lossless --format png
run png_to_jpeg_q80     "${PNGS[@]}" -- -q 80 --format jpeg
run webp_to_jpeg_q80    "${WEBPS[@]}" -- -q 80 --format jpeg

( cd "$OUT" && find . -type f ! -name SHA256SUMS | sort | xargs sha256sum > SHA256SUMS )
echo "wrote $(find "$OUT" -type f | wc -l) files under tests/golden/$OUT -- commit them; tests/test_reference_goldens.py picks them up"
