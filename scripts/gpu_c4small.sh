set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export SSW_GPU_DEBUG=1
timeout 300 python bench.py --reads 512 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/bench_c4_512.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c4_512.log
timeout 300 python bench.py --reads 512 --read-len 10000 --ref-len 100000 --flag 0 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/bench_c4_512_f0.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c4_512_f0.log
timeout 600 python -m pytest tests -m gpu -x -q -k "database or layout" > gpurun_out/pytest_gpu_new.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu_new.log
