# pytest -m gpu, then the benches whose JSON lines are committed under profiles/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 5 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_config2.log 2>&1
python bench.py --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_config5.log 2>&1
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 128"
timeout 300 python bench.py $C4 > gpurun_out/bench_config4.log 2>&1; echo "rc=$?" >> gpurun_out/bench_config4.log
