set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_config2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_config2.log
tail -n 5 gpurun_out/pytest_gpu.log
