cd $GRAFT_REPO_ROOT
for v in s352l10 s352l9 s320l11 s384l11 s352l11d9 s352l11d8 s352l11p224 s352l11p160; do python tools/variants/run.py $v 2>&1 | tail -1; done
