#!/bin/bash
# The emulation build under ThreadSanitizer: four host threads push JPEG, PNG (lossless, lossy) and WebP batches through the C ABI
# at once (the boundary is called from several threads: the reference's rayon workers, the CLI's two workers per device).  CPU only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/caesium-clt_amd/csrc; O=${TMPDIR:-/tmp}/csh_tsan; mkdir -p $O
SRC="k_decode.hip k_decode_par.hip k_decode_prog.hip k_decode_refine.hip k_aclist.hip k_pixel.hip k_resize.hip k_png_resize.hip k_entropy.hip k_trellis.hip k_assemble.hip k_png_inflate.hip k_png_filter.hip k_png_deflate.hip k_png_parse.hip k_webp.hip k_vp8enc.hip k_webp_dec.hip k_vp8l_enc.hip pipeline.cpp jpeg_host.cpp capi.cpp png_pipeline.cpp webp_decode.cpp vp8l_encode.cpp"
(cd $C && g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -DCSH_EMUL -fsanitize=thread -Wno-unknown-pragmas -Wno-attributes $(for f in $SRC; do echo -x c++ $f; done) -o $O/libcaesium_emul.so -lpthread)
cat > $O/run.py <<PY
import sys, threading
sys.path[:0] = ['$R', '$R/tools', '$R/tests']
import _util
pkg = _util.package(); api = pkg.CaesiumHip('$O/libcaesium_emul.so')
from gen_synth import synth_jpeg
jpegs = [synth_jpeg(i, 96 + 8 * i, 64, texture=5 * i) for i in range(3)]
pngs = [c[1] for c in _util.png_cases() if c[0] in ("RGB_97x61", "palette_rgb_few", "adam7_P_40x17")]
def w(k):
    for rep in range(2):
        wh = (k + rep) % 6
        if wh == 0: api.cs_batch_compress(jpegs, pkg.default_parameters())
        elif wh == 1: api.cs_batch_compress(pngs, pkg.default_parameters(png_optimize=True, png_optimization_level=1))
        elif wh == 2: api.cs_batch_compress(pngs, pkg.default_parameters(png_optimization_level=1))
        elif wh == 3: api.batch_convert(jpegs, pkg.default_parameters(webp_quality=75), 3)
        elif wh == 4: api.batch_convert(jpegs[:2], pkg.default_parameters(png_optimize=True, png_optimization_level=1), 1); api.batch_convert(pngs[:2], pkg.default_parameters(webp_quality=60), 3)
        else: api.cs_batch_compress(pngs, pkg.default_parameters(png_optimize=True, png_optimization_level=1, width=40))
ts = [threading.Thread(target=w, args=(k,)) for k in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
print('tsan run done')
PY
LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" python $O/run.py 2>&1 | grep -E "WARNING: ThreadSanitizer|SUMMARY|tsan run done" | sort | uniq -c
