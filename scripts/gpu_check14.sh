set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 16"
SSW_GPU_DEBUG=1 timeout 300 python bench.py $C4 > gpurun_out/c4_a.log 2> gpurun_out/c4_a.err; echo "rc=$?" >> gpurun_out/c4_a.log
timeout 300 python bench.py --reads 20000 --ref-len 5000000 --steps 1 --warmup 0 --cpu-sample 8 > gpurun_out/c3_small.log 2>&1; echo "rc=$?" >> gpurun_out/c3_small.log
timeout 300 python bench.py --reads 8192 --db-targets 2048 --steps 1 --warmup 0 --cpu-sample 64 > gpurun_out/c5.log 2>&1; echo "rc=$?" >> gpurun_out/c5.log
