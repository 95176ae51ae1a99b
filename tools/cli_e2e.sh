#!/bin/bash
# end-to-end CLI timing on the GPU box: N copies of 16 synthetic 1080p files through caesium-clt_amd/bin/caesiumclt, files in -> files out.
# usage: tools/cli_e2e.sh [files=2048] [gpus=1] [variants]    (gpus > 1: caesiumclt --gpus G deals its device batches round-robin over G devices,
# a few host threads per device -- the reference's rayon par_iter over files, /root/reference/src/compressor.rs:81-100, as device shards;
# variants: "workers:batch[:warmup] ..." tried one after the other, default "3:256")
N=${1:-2048}; G=${2:-1}; V=${3:-3:256}
D=${CLI_E2E_DIR:-/tmp}/cli_e2e; rm -rf $D; mkdir -p $D/in
python - <<PY
import sys; sys.path.insert(0,'tools')
from gen_synth import synth_jpeg
u=[synth_jpeg(i) for i in range(16)]
for k in range($N): open('$D/in/f%05d.jpg'%k,'wb').write(u[k%16])
PY
for v in $V; do
  W=${v%%:*}; R=${v#*:}; B=${R%%:*}; U=${R##*:}; [ "$U" = "$B" ] && U=1
  for t in 1 2; do
    rm -rf $D/out; s=$(date +%s.%N)
    CSH_TRACE=1 CSH_CLI_WORKERS=$W CSH_CLI_BATCH=$B CSH_CLI_WARMUP=$U caesium-clt_amd/bin/caesiumclt -q 80 -o $D/out --quiet --gpus $G $D/in 2>&1 | grep "^\[cli\]"
    e=$(date +%s.%N); python -c "print(\"workers $W batch $B warmup $U run $t: $N files on $G device(s) in %.3f s = %.0f files/s\" % ($e - $s, $N / ($e - $s)))"
  done
done
ls $D/out | wc -l
