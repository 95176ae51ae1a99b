"""The boundary is called from several host threads at once (the reference's rayon workers; the CLI's two workers per device):
six threads push JPEG, PNG (lossless, lossy, resized), WebP and conversion batches through the emulation build concurrently; every result must
still equal the oracle."""
import threading

from _util import (emul_api, oracle_jpeg_to_png, oracle_jpeg_to_webp, oracle_lossy, oracle_png, oracle_png_lossy, oracle_png_resized, oracle_png_to_webp, package,
                   png_cases)
from gen_synth import synth_jpeg


def test_six_threads_mixed_batches():
    api, pkg = emul_api(), package()
    jpegs = [synth_jpeg(i, 96 + 8 * i, 64, texture=5 * i) for i in range(4)]
    pngs = [c[1] for c in png_cases() if c[0] in ("RGB_97x61", "palette_rgb_few", "adam7_P_40x17", "greydepth_bw_RGB")]
    want_j = [oracle_lossy(s) for s in jpegs]
    want_p = [oracle_png(s, 2) for s in pngs]
    want_l = [oracle_png_lossy(s, 1) for s in pngs]
    want_w = [oracle_jpeg_to_webp(s, 75) for s in jpegs]
    want_c = [oracle_jpeg_to_png(s, True, 1) for s in jpegs[:2]] + [oracle_png_to_webp(s, 60) for s in pngs[:2]]
    want_r = [oracle_png_resized(s, True, 1, 40, 0) for s in pngs]
    errors = []

    def worker(k):
        try:
            for rep in range(3):
                which = (k + rep) % 6
                if which == 0:
                    assert api.cs_batch_compress(jpegs, pkg.default_parameters()) == want_j
                elif which == 1:
                    assert api.cs_batch_compress(pngs, pkg.default_parameters(png_optimize=True, png_optimization_level=2)) == want_p
                elif which == 2:
                    assert api.cs_batch_compress(pngs, pkg.default_parameters(png_optimization_level=1)) == want_l
                elif which == 3:
                    assert api.batch_convert(jpegs, pkg.default_parameters(webp_quality=75), 3) == want_w
                elif which == 4:
                    got = api.batch_convert(jpegs[:2], pkg.default_parameters(png_optimize=True, png_optimization_level=1), 1)
                    assert got + api.batch_convert(pngs[:2], pkg.default_parameters(webp_quality=60), 3) == want_c
                else:
                    assert api.cs_batch_compress(pngs, pkg.default_parameters(png_optimize=True, png_optimization_level=1, width=40)) == want_r
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
