set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 8 gpurun_out/pytest_gpu.log
python bench.py --steps 1 --warmup 1 --match 1 --mismatch 3 --gap-open 5 --gap-extend 2 > gpurun_out/bench_config2_u8.log 2>&1
