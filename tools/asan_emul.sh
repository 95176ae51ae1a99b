#!/bin/bash
# The emulation build of every kernel + the host pipelines under AddressSanitizer / UBSan (shift checks off: the emulation's
# __mul24 shim shifts negative ints), driven through the C ABI by the same cases the parity tests use.  CPU only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/caesium-clt_amd/csrc; O=${TMPDIR:-/tmp}/csh_asan; mkdir -p $O
HIP="k_decode.hip k_decode_par.hip k_decode_prog.hip k_decode_refine.hip k_aclist.hip k_pixel.hip k_resize.hip k_png_resize.hip k_entropy.hip k_trellis.hip k_assemble.hip k_png_inflate.hip k_png_filter.hip k_png_deflate.hip k_png_parse.hip k_webp.hip k_vp8enc.hip k_webp_dec.hip k_vp8l_enc.hip"
CPP="pipeline.cpp jpeg_host.cpp capi.cpp png_pipeline.cpp webp_decode.cpp vp8l_encode.cpp"
(cd $C && g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -DCSH_EMUL -fsanitize=address,undefined -fno-sanitize=shift -fno-omit-frame-pointer -Wno-unknown-pragmas -Wno-attributes \
    $(for f in $HIP $CPP; do echo -x c++ $f; done) -o $O/libcaesium_emul.so -lpthread)
cat > $O/run.py <<PY
import sys
sys.path[:0] = ['$R', '$R/tools', '$R/tests']
import _util
pkg = _util.package()
api = pkg.CaesiumHip('$O/libcaesium_emul.so')
import test_png_emul as T, test_webp_emul as W, test_pipeline_emul as PE
from gen_synth import synth_jpeg
T.check_batch(api, _util.png_cases(), 3)
T.check_batch(api, [c for c in _util.png_cases() if c[0] in ("RGB_97x61", "palette_rgba_translucent", "adam7_1_37x11", "reduce_i16_narrow")], 6)
assert T.agree_with_oracle(api, T.damaged_pngs(1, 80)) == 0
T.test_indexed_images_lose_unused_depth(api)
T.test_emul_min_cost_path_parse(api); T.test_emul_zopfli_means_more_passes(api); T.test_emul_rows_longer_than_65536_bytes(api)   # round 6: the parse's workgroup form, scratch areas, k_png_scores<true>
import test_png_lossy_emul as PLY
PLY.test_median_falls_among_equal_keys(api)
PLY.check_lossy(api, [c for c in PLY.lossy_cases() if c[0] in ('RGB_97x61', 'RGBA_soft_alpha', 'RGB_tall_24x600', 'RGB_513_rows')])   # round 6: k_png_dither's bands and line buffers
W.check(api, W.webp_cases(), 85); W.check(api, W.webp_cases()[:2], 60, width=50)
srcs = [synth_jpeg(1, 160, 96, texture=10), synth_jpeg(2, 97, 61, subsampling=0, texture=5), synth_jpeg(5, 104, 72, progressive=True, texture=6), synth_jpeg(6, 120, 88, restart_rows=1, texture=9)]
for s, o in zip(srcs, api.batch_compress(srcs, pkg.default_parameters(jpeg_quality=80))): assert o == _util.oracle_lossy(s)
for s, o in zip(srcs, api.batch_compress(srcs, pkg.default_parameters(jpeg_optimize=True))): assert o == _util.oracle_lossless(s)
for s, o in zip(srcs, api.batch_compress(srcs[:2], pkg.default_parameters(jpeg_quality=70, width=60))): assert o == _util.oracle_resized(s, 60, 0, quality=70)
api.batch_compress(PE.fuzzed_blobs(3, 24, True), pkg.default_parameters(jpeg_quality=80))
class MP:
    def setenv(self, k, v): import os; os.environ[k] = v
    def delenv(self, k, raising=True): import os; os.environ.pop(k, None)
PE.test_emul_refinement_scans_parse_and_apply(api, MP())
PE.test_emul_irregular_progressions_decode_in_file_order(api)
W.test_boolean_coder_in_pieces(api, MP()); W.test_statistics_books_overflow_in_order(api)
import test_png_webp_emul as PW, test_jpeg_png_emul as JP
PW.check(api, _util.png_cases(), 85); PW.check(api, PW.extra_cases(), 60); PW.test_damaged_pngs_convert_like_the_oracle_or_fail(api)
JP.check(api, W.webp_cases(), True); JP.check(api, W.webp_cases()[:3], False, width=50); JP.test_mixed_batch_and_failures(api); JP.test_damaged_jpegs_convert_like_the_oracle_or_fail(api)
import test_png_resize_emul as PR, test_png_jpeg_emul as PJ
PJ.test_every_png_format_converts_like_the_oracle(api); PJ.test_resize_in_front(api); PJ.test_mixed_batch_and_failures(api)
PR.test_every_case_resizes_like_the_oracle_or_is_refused(api); PR.test_sizes_and_shapes(api); PR.test_mixed_batch_with_jpegs_and_damage(api)
# damaged PNGs through the composed paths: device and oracle must agree on refusal or on the bytes
blobs = T.damaged_pngs(77, 240)
def agree(outs, want):
    for b, o in zip(blobs, outs):
        try: w = want(b)
        except Exception as e:
            w = None
            if getattr(e, 'code', 0) == 10201 and not isinstance(o, Exception): continue   # transparency -> WebP: no oracle for the ALPH chunk
        assert isinstance(o, Exception) if w is None else o == w
agree(api.batch_convert(blobs, pkg.default_parameters(jpeg_quality=70), 0), lambda b: _util.oracle_png_to_jpeg(b, 70))
agree(api.batch_convert(blobs, pkg.default_parameters(webp_quality=70, width=30), 3), lambda b: _util.oracle_png_to_webp(b, 70, 30, 0))
agree(api.cs_batch_compress(blobs, pkg.default_parameters(png_optimize=True, png_optimization_level=1, height=20)), lambda b: _util.oracle_png_resized(b, True, 1, 0, 20))
agree(api.cs_batch_compress(blobs, pkg.default_parameters(png_optimization_level=1, png_quality=20)), lambda b: _util.oracle_png_lossy(b, 1, quality=20))
import test_webp_decode_emul as WD
WD.test_emul_synthetic_files_decode_like_libwebp(api); WD.test_emul_lossless_files_decode_like_libwebp(api); WD.test_emul_damaged_lossless_streams_fail_alone(api); WD.test_emul_damaged_and_unsupported_inputs_fail_alone(api)
WD.test_emul_transparent_files_decode_like_libwebp(api); WD.test_emul_damaged_transparent_files_fail_alone(api); WD.test_emul_transparent_sources_keep_their_alpha(api)
import test_webp_lossless_emul as WL
WL.test_emul_lossless_webp_round_trips_through_libwebp(api, '$R/tests/golden/reference_samples'); WL.test_emul_jpeg_to_lossless_webp_and_resize(api); WL.test_emul_lossless_webp_failures_stay_per_file(api); WL.test_emul_png_to_lossless_webp(api)
PW.test_transparency_becomes_an_alph_chunk(api)
# round 5: the chroma window's 16-byte loads around a row's first / last block (odd and tiny sizes), the statistics list that takes the trellis's levels and is
# compacted in place (several chunks per component), two DC refinement scans in one launch
PE.test_emul_two_dc_refinement_passes(api)
srcs5 = [synth_jpeg(30 + i, w, h, texture=t) for i, (w, h, t) in enumerate([(17, 9, 20), (33, 31, 60), (200, 8, 10), (8, 200, 10), (641, 363, 35), (1000, 520, 25)])]
for s, o in zip(srcs5, api.batch_compress(srcs5, pkg.default_parameters(jpeg_quality=80))): assert o == _util.oracle_lossy(s)
for s, o in zip(srcs5, api.batch_compress(srcs5, pkg.default_parameters(jpeg_quality=35))): assert o == _util.oracle_lossy(s, 35)
print('asan run: all cases equal the oracle')
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 python $O/run.py 2>&1 | grep -E "runtime error|AddressSanitizer|SUMMARY|asan run|Traceback|Error" | sort | uniq -c
