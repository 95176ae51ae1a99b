# usage (from the repo root, inside gpurun):  bash tools/gpu_pending.sh
# First device run of the paths that were validated on the emulation build only: the four test files that carry
# the "first device run pending" mark, with the mark overridden so that a mismatch is a failure, each under its own time limit; then a
# timing of every conversion at 64 x 1080p.  Results -> gpurun_out/pending_*.txt
R=$(pwd); mkdir -p gpurun_out
for t in test_zzz_png_webp_gpu test_zzz_jpeg_png_gpu test_zzz_png_resize_gpu test_zzzz_png_jpeg_gpu; do
    timeout 900 python -m pytest tests/$t.py -q -m gpu --runxfail -x > gpurun_out/pending_$t.txt 2>&1
    echo "$t: exit $? -- $(tail -1 gpurun_out/pending_$t.txt)"
done
timeout 600 python tools/convert_bench.py 64 > gpurun_out/pending_convert_bench.txt 2>&1; tail -8 gpurun_out/pending_convert_bench.txt
