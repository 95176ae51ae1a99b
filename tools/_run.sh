cd $GRAFT_REPO_ROOT
for N in 2048 10000; do
D=/dev/shm/cli_e2e; rm -rf $D; mkdir -p $D/in
python - <<PY
import sys; sys.path.insert(0,'tools')
from gen_synth import synth_jpeg
u=[synth_jpeg(i) for i in range(16)]
for k in range($N): open('$D/in/f%05d.jpg'%k,'wb').write(u[k%16])
PY
r=""
for t in 1 2 3; do
rm -rf $D/out; s=$(date +%s.%N)
CSH_TRACE=1 caesium-clt_amd/bin/caesiumclt -q 80 -o $D/out --quiet $D/in 2>&1 | grep "^\[cli\]" | cut -c1-200 > /tmp/t.txt
e=$(date +%s.%N); r="$r $(python -c "print('%.3f' % ($e - $s))")"
done
cat /tmp/t.txt; echo "N $N default:$r  written $(ls $D/out | wc -l)"
rm -rf $D
done
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "cli or batch_order" 2>&1 | tail -2
