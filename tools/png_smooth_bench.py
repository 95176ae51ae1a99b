import sys, time, zlib
sys.path[:0] = ['/root/repo', '/root/repo/tools', '/root/repo/tests']
from _util import package, product_api
from gen_synth import synth_png
api, pkg = product_api(), package()
for tex, zop in ((0.5, False), (0.5, True), (0.0, False)):
    src = [synth_png(200 + k, 3840, 2160, "RGB", texture=tex) for k in range(2)]
    blobs = [src[k % 2] for k in range(32)]
    p = pkg.default_parameters(png_optimize=True, png_optimization_level=3, png_force_zopfli=zop)
    for rep in range(2):
        b = api.png_batch(blobs, p); tm = b.run(); outs = b.fetch(); names = api.png_kernel_names(); b.close()
    print(f"texture {tex} zopfli {zop}: 32 files device {tm.total_ms:.0f} ms; in {len(src[0])} out {len(outs[0])} ({len(outs[0]) / len(src[0]):.3f}); " + ", ".join(f"{names[i]} {tm.kernel_ms[i]:.0f}" for i in range(len(names)) if names[i] and tm.kernel_ms[i] > 5))
