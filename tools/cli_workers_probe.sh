#!/bin/bash
# the CLI end to end (tools/cli_e2e.sh's set: 2048 x 1080p in /tmp) for several host-thread counts per device and device-batch sizes
N=${1:-2048}
D=/dev/shm/cli_e2e; rm -rf $D; mkdir -p $D/in
python - <<PY
import sys; sys.path.insert(0,'tools')
from gen_synth import synth_jpeg
u=[synth_jpeg(i) for i in range(16)]
for k in range($N): open('$D/in/f%05d.jpg'%k,'wb').write(u[k%16])
PY
for cfg in "2 256" "2 256" "3 256" "4 256" "4 128" "6 128" "2 128" "1 256"; do
  set -- $cfg
  rm -rf $D/out; sleep 5; s=$(date +%s.%N)   # (a process right behind another one waits for the driver to take the VRAM back)
  CSH_CLI_WORKERS=$1 CSH_CLI_BATCH=$2 CSH_TRACE=1 caesium-clt_amd/bin/caesiumclt -q 80 -o $D/out --quiet $D/in 2>&1 | grep "\[cli\]"
  e=$(date +%s.%N); python -c "print(\"workers $1 batch $2: %.3f s\" % ($e - $s))"
done
rm -rf $D
