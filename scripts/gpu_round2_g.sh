# round 2, call G: GPU suite, config 4 with the blocked team traceback, config 5 (full size and the 8192 x 2048 shape) with the size
# classes of a chunk side by side on four streams
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 5 gpurun_out/pytest_gpu.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k != "against" and k != "fields"}, d.get("roofline_valu", {}).get("frac"), d.get("roofline", {}).get("kernel"))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-600:])
PY
}
SSW_GPU_XR=8 timeout 150 python bench.py --config 4 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/g_config4.log 2>&1; show gpurun_out/g_config4.log "config4 XR=8 blocked trace"
SSW_GPU_XR=8 SSW_GPU_TRACE_BLOCKED=0 timeout 150 python bench.py --config 4 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/g_config4_unblocked.log 2>&1; show gpurun_out/g_config4_unblocked.log "config4 XR=8 one cell per thread"
timeout 150 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/g_config5_shape.log 2>&1; show gpurun_out/g_config5_shape.log "config5 8192x2048"
timeout 300 python bench.py --config 5 --cpu-sample 0 > gpurun_out/g_config5.log 2>&1; show gpurun_out/g_config5.log "config5 full"
