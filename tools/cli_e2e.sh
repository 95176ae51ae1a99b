#!/bin/bash
# end-to-end CLI timing on the GPU box: N copies of 16 synthetic 1080p files through caesium-clt_amd/bin/caesiumclt
N=${1:-2048}
D=/tmp/cli_e2e; rm -rf $D; mkdir -p $D/in
python - <<PY
import sys; sys.path.insert(0,'tools')
from gen_synth import synth_jpeg
u=[synth_jpeg(i) for i in range(16)]
for k in range($N): open('$D/in/f%05d.jpg'%k,'wb').write(u[k%16])
PY
for t in 1 2; do
  rm -rf $D/out; s=$(date +%s.%N)
  CSH_TRACE=1 caesium-clt_amd/bin/caesiumclt -q 80 -o $D/out --quiet $D/in
  e=$(date +%s.%N); python -c "print(\"run $t: $N files in %.3f s\" % ($e - $s))"
done
ls $D/out | wc -l
