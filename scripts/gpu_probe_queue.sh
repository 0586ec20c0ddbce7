# tiny probe of the work-queue strip kernel: every variant under its own 40 s limit, progress on stderr (SSW_GPU_DEBUG)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/probe.py <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "complete-striped-smith-waterman-library_amd")
import numpy as np, ssw_amd
from parity import compare_batch, make_reads
from sswutil import dna_matrix, random_ref
print("start", flush=True)
import os
lib = ssw_amd.load(os.path.join("complete-striped-smith-waterman-library_amd", "libssw_hooks.so"))      # libssw.so ignores the SSW_GPU_* hooks (round-5 advisor)
assert lib.ssw_gpu_has_test_hooks() == 1
ctx = ssw_amd.Context(0, lib)
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
run() { echo "== $1"; env $1 SSW_GPU_DEBUG=1 SSW_GPU_XR=3 timeout 40 python -u /tmp/probe.py > gpurun_out/probe_$2.log 2>&1; echo "rc=$?"; tail -n 12 gpurun_out/probe_$2.log; }
run "SSW_GPU_XLANES=16" control16
run "SSW_GPU_QUEUE=jobs SSW_GPU_QUEUE_WAVES=1" jobs1
run "SSW_GPU_QUEUE=jobs" jobs
run "SSW_GPU_QUEUE=strips SSW_GPU_QUEUE_WAVES=1" strips1
run "SSW_GPU_QUEUE=strips" strips
