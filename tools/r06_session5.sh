cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in 0 1 3 5 7 9 11 15; do
PAMG_LANEM_DBG=$d timeout 600 python tools/microbench_lanem.py --levels 1 2 --s 2 3 --grids 0 --tag r06_lanem_dbg$d > gpurun_out/r06_microbench_lanem_dbg$d.log 2>&1; echo "dbg=$d"; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_dbg$d.log | paste - - - 
done
