# round 2, call P: the new streamed-search tests on the GPU, and the f16-form pause at scale: 4096 proteins searched against
# themselves (every self-hit of 310+ residues saturates the f16 form), adaptive vs forced f16 vs int16 only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_search_db.py -m gpu -x -q > gpurun_out/p_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 gpurun_out/p_pytest.log
cat > /tmp/selfsearch.py <<'PY'
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, "complete-striped-smith-waterman-library_amd")
import numpy as np, ssw_amd, workloads as W
db, qs, mat = W.protein_config(0, queries=100, db_entries=4096)
ctx = ssw_amd.Context(0)
Q = ctx.upload(db); T = ctx.upload(db)
def run(label):
    tot = [0]
    def cb(t0, h):
        tot[0] += int(h["score1"].astype(np.int64).sum()); return 0
    ctx.search_db(Q, T, mat, 24, 3, 1, -1, 2, 512, cb)            # warm
    tot[0] = 0
    t = time.time(); ctx.search_db(Q, T, mat, 24, 3, 1, -1, 2, 512, cb); dt = time.time() - t
    tm = ctx.timing()
    print(label, "%.3f s" % dt, "fill %.1f ms" % tm["fill_ms"], tm["fill_kernel"], "repeated workgroups", tm["db_repeats"], "checksum", tot[0])
run("adaptive (self-search, 4096 x 4096)")
os.environ["SSW_GPU_DB_F16"] = "1"; run("f16 first in every chunk      ")
os.environ["SSW_GPU_DB_F16"] = "0"; run("int16 only                    ")
PY
timeout 200 python /tmp/selfsearch.py 2>&1 | tail -4
