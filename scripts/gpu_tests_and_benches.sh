cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_config2.log 2>&1
python bench.py --flag 2 --steps 1 --warmup 1 --cpu-sample 2000 > gpurun_out/bench_config2_flag2.log 2>&1
python bench.py --steps 1 --warmup 1 --match 1 --mismatch 3 --gap-open 5 --gap-extend 2 > gpurun_out/bench_config2_u8.log 2>&1
python bench.py --reads 20000 --ref-len 5000000 --steps 1 --warmup 1 --cpu-sample 8 > gpurun_out/bench_config3shape.log 2>&1
python bench.py --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/bench_config5.log 2>&1
python bench.py --reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 128 > gpurun_out/bench_config4.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29527 bench.py --gpus 2 --steps 1 --warmup 1 --reads 8192 --db-targets 1024 --cpu-sample 0 > gpurun_out/bench_db_2ranks.log 2>&1; echo "rc=$?" >> gpurun_out/bench_db_2ranks.log
