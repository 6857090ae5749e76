cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/lanem_profile.py --s 2 3 --grids 512 768 1024 > gpurun_out/r06_lanem_profile.log 2>&1; grep "^{" gpurun_out/r06_lanem_profile.log || tail -20 gpurun_out/r06_lanem_profile.log
