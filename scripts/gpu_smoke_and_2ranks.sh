set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --reads 20000 > gpurun_out/bench_2ranks.log 2>&1; echo "rc=$?" >> gpurun_out/bench_2ranks.log
SSW_BENCH_BACKEND=nccl timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 1 --warmup 1 --reads 20000 > gpurun_out/bench_2ranks_nccl.log 2>&1; echo "rc=$?" >> gpurun_out/bench_2ranks_nccl.log
tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench_2ranks.log | cut -c1-300; tail -4 gpurun_out/bench_2ranks_nccl.log | cut -c1-300
