set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -12 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 1 --warmup 1 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench1.log
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench1.log
