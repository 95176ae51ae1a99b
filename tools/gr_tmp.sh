python -m pytest tests/test_pipeline_gpu.py tests/test_trellis_gpu.py -q -m gpu 2>&1 | tail -2
python tools/trellis_probe.py 1024 64 2>&1 | grep "debug=0\|debug=1"
python bench.py --steps 6 --warmup 2 --no-extras --no-pmc > gpurun_out/b7.json 2> gpurun_out/b7.err; python -c "
import json;d=json.load(open('gpurun_out/b7.json'));print('default',d['ms_per_step'],d['parity_spot_check'],{k:v for k,v in d['kernel_ms'].items() if v>0.3})"
