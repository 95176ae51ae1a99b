cd $GRAFT_REPO_ROOT
python tools/variants/run.py k_png_scores st1024 st512 st256 2>&1 | tail -3
