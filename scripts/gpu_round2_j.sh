# round 2, call J: strip kernel after ordering its LDS requests (ring entry first, only the words / rows that are used); config 2
# with equal launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 3 gpurun_out/pytest_gpu.log
show() { grep "^{" $1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_per_step'], d['roofline_valu']['frac'], (d.get('parity') or {}).get('mismatching_alignments'))"; }
timeout 150 python bench.py --config 4 --cpu-sample 0 > gpurun_out/j_config4.log 2>&1; show gpurun_out/j_config4.log "config4 default (fill 10 rows/lane, windows 8)"
SSW_GPU_XR=8 timeout 150 python bench.py --config 4 --cpu-sample 0 > gpurun_out/j_config4_xr8.log 2>&1; show gpurun_out/j_config4_xr8.log "config4 XR=8"
SSW_GPU_XR=12 timeout 150 python bench.py --config 4 --cpu-sample 0 > gpurun_out/j_config4_xr12.log 2>&1; show gpurun_out/j_config4_xr12.log "config4 XR=12"
timeout 150 python bench.py --cpu-sample 0 > gpurun_out/j_config2.log 2>&1; show gpurun_out/j_config2.log "config2"
