#!/bin/bash
# HBM traffic of the pixel kernels (phase X) from the PMC counters: FETCH_SIZE and WRITE_SIZE in passes of their own (--pmc only, no trace domain), one step of B
# files under the default profile; per kernel: bytes per launch (FETCH_SIZE doubled: the gfx950 note of /opt/skills/guides/MI355X_MICROARCH.md -- wide coalesced reads
# are tallied at half their size; the counters' unit is KiB) -> gpurun_out/r05_pmc_hbm_xform.txt.   usage: tools/gpu_pmc_xform.sh [batch=1024]
B=${1:-1024}; R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rocprofv3 --pmc $C --kernel-include-regex "k_xform_direct|k_resample_fdct_420|k_idct_plane" --output-format csv -d $R/gpurun_out/pmc_x_$C -- python $R/bench.py --pmc-child --batch $B > /dev/null 2>&1; cd $R
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for fn in glob.glob("gpurun_out/pmc_x_%s/**/*counter_collection.csv"%C,recursive=True):
        for r in csv.DictReader(open(fn)):
            k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
with open("gpurun_out/r05_pmc_hbm_xform.txt","w") as f:
    f.write("files per step: $B (1080p 4:2:0: luma 4 177 920 B, chroma 2 088 960 B of coefficients one way, 1 044 480 B of chroma planes per file)\n")
    for k,v in agg.items():
        fe=v.get("FETCH_SIZE",0)*1024*2; wr=v.get("WRITE_SIZE",0)*1024
        f.write("%s  dispatches=%d  FETCH_SIZE x 2 = %.3f GB  WRITE_SIZE = %.3f GB  total %.3f GB\n"%(k,max(n[(k,c)] for c in v),fe/1e9,wr/1e9,(fe+wr)/1e9))
PY
rm -rf gpurun_out/pmc_x_FETCH_SIZE gpurun_out/pmc_x_WRITE_SIZE
cat gpurun_out/r05_pmc_hbm_xform.txt
