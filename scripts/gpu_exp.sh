cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 4 gpurun_out/pytest_gpu.log
python bench.py --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_config5.log 2>&1
