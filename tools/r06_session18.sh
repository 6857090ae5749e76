cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "long_dependency or kaczmarz_fast" 2>&1 | tail -3
TAG=r06 bash tools/gpu_run.sh prof:c3 prof:c2 prof:c4x prof:c5 2>&1 | tail -80
