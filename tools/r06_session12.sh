cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in 0 32; do echo dbg=$d
PAMG_LANEM_DBG=$d timeout 900 python tools/microbench_lanem.py --levels 1 2 --s 2 3 --grids 512 768 1024 --tag r06_lanem_pf$d > gpurun_out/r06_microbench_lanem_pf$d.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"max_rel[^,]*,\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_pf$d.log | paste - - - - | head -60
done
