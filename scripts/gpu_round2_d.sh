# round 2, call D: GPU suite + rows-per-lane sweep of the queue kernel on config 4 + configs 2 / 5 (all under short limits)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 5 gpurun_out/pytest_gpu.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], d.get("parity"), d.get("roofline_valu", {}).get("frac"), d.get("roofline", {}).get("kernel"))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-600:])
PY
}
for xr in 12 8 10 16; do
  SSW_GPU_XR=$xr timeout 150 python bench.py --config 4 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/d_config4_xr$xr.log 2>&1; show gpurun_out/d_config4_xr$xr.log "config4 XR=$xr"
done
SSW_GPU_QUEUE=jobs timeout 150 python bench.py --config 4 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/d_config4_jobs.log 2>&1; show gpurun_out/d_config4_jobs.log "config4 job tickets"
timeout 150 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/d_config5_shape.log 2>&1; show gpurun_out/d_config5_shape.log "config5 8192x2048"
timeout 150 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/d_config2.log 2>&1; show gpurun_out/d_config2.log "config2"
SSW_GPU_SEG_REDUCE=0 timeout 150 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/d_config2_noseg.log 2>&1; show gpurun_out/d_config2_noseg.log "config2 without group maxima"
