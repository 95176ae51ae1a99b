python -m pytest tests -m gpu -x -q 2>&1 | tail -3; python bench.py --cpu-images 0 --batch ${B:-512} > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; python -c "
import json;d=json.load(open('gpurun_out/bench_x.json'));print(d['value'],d['device_ms_per_step']);print({k:v for k,v in d['kernel_ms'].items() if v>0.25})"
