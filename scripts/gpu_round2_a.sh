# round 2, call A: the GPU test-suite (new: clipping regimes, threads / pool, device pieces, wide alphabets) + "before" bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 6 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/a_config2.log 2>&1; tail -c 600 gpurun_out/a_config2.log
timeout 300 python bench.py --reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/a_config4.log 2>&1; tail -c 900 gpurun_out/a_config4.log
timeout 300 python bench.py --reads 8192 --db-targets 2048 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/a_config5.log 2>&1; tail -c 600 gpurun_out/a_config5.log
