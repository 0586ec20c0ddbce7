set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --reads 8192 --db-targets 2048 --steps 1 --warmup 1 > gpurun_out/bench_config5.log 2>&1; echo "rc=$?" >> gpurun_out/bench_config5.log
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5 -o bench -- python bench.py --reads 8192 --db-targets 2048 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/prof5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o bench -- python bench.py --reads 2048 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/prof4.log 2>&1
