python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -1
D=/dev/shm/cli_e2e; rm -rf $D; mkdir -p $D/in
python - <<PY
import sys; sys.path.insert(0,'tools')
from gen_synth import synth_jpeg
u=[synth_jpeg(i) for i in range(16)]
for k in range(2048): open('$D/in/f%05d.jpg'%k,'wb').write(u[k%16])
PY
for w in 4096 512 4096 512; do rm -rf $D/out; sleep 3; s=$(date +%s.%N); CSH_CLI_WINDOW=$w CSH_TRACE=1 caesium-clt_amd/bin/caesiumclt -q 80 -o $D/out --quiet $D/in 2>&1 | grep "\[cli\]"; e=$(date +%s.%N); python -c "print('window $w: %.3f s' % ($e - $s))"; done
ls $D/out | wc -l; rm -rf $D
