set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --reads 8192 --db-targets 2048 --steps 1 --warmup 1 > gpurun_out/bench_config5.log 2>&1; echo "rc=$?" >> gpurun_out/bench_config5.log
timeout 300 python -m pytest tests -m gpu -x -q -k "database or cli" > gpurun_out/pytest_gpu_db.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu_db.log
