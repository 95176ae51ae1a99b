#!/usr/bin/env python3
"""Where configs[4]'s wall clock goes: the mixed tree of bench.py (1080p JPEG / PNG / WebP, `caesiumclt -q 80 -R -S`) once as a whole and once
per file type, each as a process of its own, with the tool's stage times (CSH_TRACE).  usage: tools/mixed_probe.py [files per type]"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    per_type = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    jpgs = bench.make_inputs(0, 8)
    pngs = bench.pool_map(bench._one_png_1080, range(200, 208))
    webps = bench.pool_map(bench._one_webp_1080, range(300, 308))
    d = bench.scratch_dir()
    try:
        sets = {"jpg": jpgs, "png": pngs, "webp": webps}
        for ext, blobs in sets.items():
            os.makedirs(os.path.join(d, "tree", ext))
            for k in range(per_type):
                with open(os.path.join(d, "tree", ext, f"m{k:04d}.{ext}"), "wb") as f:
                    f.write(blobs[k % 8])
        env = dict(os.environ, CSH_TRACE="1")
        only = sys.argv[2:] or ("jpg", "png", "webp", "", "")
        for what in only:
            out = os.path.join(d, "out_" + (what or "all"))
            secs, r = bench.run_cli(["-q", "80", "-R", "-S", "--quiet", "-o", out, os.path.join(d, "tree", what)], env=env)
            nout = sum(len(fs) for _, _, fs in os.walk(out))
            print(f"{what or 'all':5s} files_out={nout} rc={r.returncode} seconds={secs:.3f}", flush=True)
            for line in r.stderr.decode().splitlines():
                if line.startswith("[") and "relax" not in line:
                    print("     ", line[:300], flush=True)
            shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
