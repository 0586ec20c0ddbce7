# round 2, call B: work-queue strip kernel (k_chainq) + chain-best filter in k_filldb: correctness, then rows-per-lane sweep on config 4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 6 gpurun_out/pytest_gpu.log
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 1 --cpu-sample 64"
for xr in 12 8 10 16; do
  SSW_GPU_XR=$xr timeout 300 python bench.py $C4 > gpurun_out/b_config4_xr$xr.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_config4_xr$xr.log").read().strip().split("\n")[-1])
    print("XR=$xr", d["value"], d["phases_ms_per_step"], d.get("parity"))
except Exception as e:
    print("XR=$xr failed", e); print(open("gpurun_out/b_config4_xr$xr.log").read()[-800:])
PY
done
timeout 300 python bench.py --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/b_config5.log 2>&1; tail -c 700 gpurun_out/b_config5.log
