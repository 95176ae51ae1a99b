cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r06_pmc_sq_png.txt
bash tools/gpu_pmc_png.sh 64 2>&1 | tail -40
