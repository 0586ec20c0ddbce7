cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 8 gpurun_out/pytest_gpu.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_config2.log 2>&1
SSW_GPU_FILL_F16=0 python bench.py --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_config2_int16.log 2>&1
