#!/bin/bash
# GPU box: kernel timeline (start / end / queue of every dispatch) of one bench command, as CSV -- which launches really overlap.
#   bash scripts/gpu_trace_timeline.sh <name> <bench args...>    ->  gpurun_out/timeline_<name>.csv
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
name=$1; shift
P=$PWD/gpurun_out/tl_$name
rm -rf $P; mkdir -p $P
B=$PWD/bench.py
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $P -o bench -- python $B "$@" > $P.log 2>&1; echo "trace rc=$?")
python3 - "$P" "gpurun_out/timeline_$name.csv" <<'PY'
import glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True))
out = open(sys.argv[2], "w")
for d in dbs:
    c = sqlite3.connect(d)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    out.write("# columns of the kernels view: %s\n" % ",".join(cols))
    want = [x for x in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "workgroup_x", "lds_size") if x in cols]
    out.write(",".join(want) + "\n")
    for r in c.execute("select %s from kernels order by start" % ",".join(want)):
        out.write(",".join('"%s"' % v if isinstance(v, str) else str(v) for v in r) + "\n")
out.close()
print(open(sys.argv[2]).read()[:600])
PY
rm -rf $P
