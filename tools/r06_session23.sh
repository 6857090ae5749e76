cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for sw in forward backward; do for nr in 2 8; do echo "== $sw nreg=$nr"
PAMG_LANEM_NREG=$nr timeout 600 python tools/microbench_lanem.py --levels 2 3 --s 4 6 8 --grids 0 --sweep $sw --tag r06_lanem_l2_${sw}_$nr > gpurun_out/r06_microbench_lanem_l2_${sw}_$nr.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"hand_offs": [0-9]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_l2_${sw}_$nr.log | paste - - - - 
done; done
