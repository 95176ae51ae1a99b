"""Wall time of every conversion / PNG resize path on the device, N synthetic 1920x1080 sources per call (host buffers in, host buffers
out: PCIe and the host stages included -- these paths chain two batch objects, so there is no single hipEvent bracket).
usage: python tools/convert_bench.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]


def main():
    import _util as U
    from gen_synth import synth_jpeg, synth_png
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    api, pkg = U.product_api(), U.package()
    assert api.device_count() >= 1, "no HIP device"
    jpegs = [synth_jpeg(100 + i % 8, 1920, 1080) for i in range(8)]
    pngs = [synth_png(200 + i % 4, 1920, 1080, "RGB", texture=4.0, compress_level=1) for i in range(4)]
    J = [jpegs[i % 8] for i in range(n)]
    P = [pngs[i % 4] for i in range(n)]
    runs = [("JPEG -> WebP q85 long edge 1500", lambda: api.batch_convert(J, pkg.default_parameters(webp_quality=85, width=1500), 3)),
            ("PNG  -> WebP q85", lambda: api.batch_convert(P, pkg.default_parameters(webp_quality=85), 3)),
            ("PNG  -> WebP q85 long edge 1500", lambda: api.batch_convert(P, pkg.default_parameters(webp_quality=85, width=1500), 3)),
            ("JPEG -> PNG lossless o1", lambda: api.batch_convert(J, pkg.default_parameters(png_optimize=True, png_optimization_level=1), 1)),
            ("JPEG -> PNG quantised", lambda: api.batch_convert(J, pkg.default_parameters(png_optimization_level=1), 1)),
            ("PNG  -> JPEG q80", lambda: api.batch_convert(P, pkg.default_parameters(jpeg_quality=80), 0)),
            ("PNG  -> PNG lossless o1 width 1280", lambda: api.cs_batch_compress(P, pkg.default_parameters(png_optimize=True, png_optimization_level=1, width=1280)))]
    print(f"# {n} files per call, 1920x1080 sources; second call timed (pools warm)")
    for name, f in runs:
        f()
        t = time.time(); outs = f(); dt = time.time() - t
        bad = sum(isinstance(o, Exception) for o in outs)
        size = sum(len(o) for o in outs if not isinstance(o, Exception)) / max(1, n - bad)
        print(f"{name:38s} {dt * 1e3:9.1f} ms  {n / dt:8.1f} files/s  {n * 2.0736 / dt:9.1f} source MP/s  mean output {size / 1024:8.1f} KB  failed {bad}")


if __name__ == "__main__":
    main()
