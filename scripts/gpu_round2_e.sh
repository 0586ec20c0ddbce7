# round 2, call E: full-size parity of configs 2-5 against the committed reference fixtures, bench lines of configs 3 and 5 (stated
# sizes), and the in-library work queues on one GPU (two workers sharing the device)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python scripts/gpu_parity_full.py 2 3 4 5 > gpurun_out/parity_full.log 2>&1; echo "parity rc=$?"; cut -c1-420 gpurun_out/parity_full.log
timeout 200 python bench.py --config 3 > gpurun_out/e_config3.log 2>&1; echo "config3 rc=$?"; tail -c 1500 gpurun_out/e_config3.log | cut -c1-700
timeout 300 python bench.py --config 5 > gpurun_out/e_config5.log 2>&1; echo "config5 rc=$?"; tail -c 3000 gpurun_out/e_config5.log | cut -c1-900
timeout 200 python bench.py --pool 2 --reads 40000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/e_pool2.log 2>&1; echo "pool rc=$?"; tail -c 1200 gpurun_out/e_pool2.log
