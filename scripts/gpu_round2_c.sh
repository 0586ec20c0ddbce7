# round 2, call C: the work-queue strip kernel after the polling fix -- a spin-heavy probe first (few jobs, many strips), every
# step under its own short timeout; then the GPU suite, the rows-per-lane sweep on config 4 and config 5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/probe.py <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "complete-striped-smith-waterman-library_amd")
import numpy as np, ssw_amd
from parity import compare_batch, make_reads
from sswutil import dna_matrix, random_ref
ctx = ssw_amd.Context(0)
rng = np.random.default_rng(3)
ref = random_ref(20000, 5, 4)
reads = make_reads(rng, ref, 3, [3000, 2990, 2500], 4, sub=0.02, ins=0.005, dele=0.005, frac_random=0.0)
mat = dna_matrix(2, 2)
for flag in (0, 2):
    Q = ctx.upload(reads); T = ctx.upload([ref])
    t = time.time()
    res, cig = ctx.align_batch(Q, T, mat, 5, 3, 1, flag, 0, 0, -1, 2)
    dt = time.time() - t
    bad = compare_batch(res, cig, reads, [ref], mat, 5, 3, 1, flag, 0, 0, -1, 2)
    print("probe flag %d: %.3f s, fill %.2f ms, %s" % (flag, dt, ctx.timing()["fill_ms"], "MISMATCH " + bad[0] if bad else "bit-exact"), flush=True)
    Q.free(); T.free()
PY
for w in 1 0; do
  echo "== probe, SSW_GPU_XR=3, queue waves $w (0 = device-wide)"; SSW_GPU_XR=3 SSW_GPU_QUEUE_WAVES=$w timeout 90 python /tmp/probe.py 2>&1 | tail -3; echo "rc=$?"
done
timeout 420 python -m pytest tests -m gpu -x -q --timeout=100 -k "long or strip or config4 or clipping" > gpurun_out/pytest_gpu_long.log 2>&1; echo "pytest(long) rc=$?"; tail -n 3 gpurun_out/pytest_gpu_long.log
timeout 600 python -m pytest tests -m gpu -x -q --timeout=100 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 4 gpurun_out/pytest_gpu.log
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 1 --cpu-sample 64"
for xr in 12 8 10 16; do
  SSW_GPU_XR=$xr timeout 200 python bench.py $C4 > gpurun_out/c_config4_xr$xr.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c_config4_xr$xr.log").read().strip().split("\n")[-1])
    print("XR=$xr", d["value"], d["phases_ms_per_step"], d.get("parity"))
except Exception as e:
    print("XR=$xr failed", e); print(open("gpurun_out/c_config4_xr$xr.log").read()[-600:])
PY
done
timeout 200 python bench.py --reads 8192 --db-targets 2048 --steps 2 --warmup 1 --cpu-sample 64 > gpurun_out/c_config5.log 2>&1; tail -c 700 gpurun_out/c_config5.log
