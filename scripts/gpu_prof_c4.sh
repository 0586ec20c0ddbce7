set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_c4
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4/trace -o bench -- python bench.py $C4 > gpurun_out/prof_c4/trace.log 2>&1
find gpurun_out/prof_c4 -name "*.csv" | head
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_c4/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=open('gpurun_out/prof_c4/dispatches.txt','w')
for r in rows:
    n=r['Kernel_Name'][:60]
    if 'selftest' in n: continue
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    out.write("%-60s grid %8s wg %5s lds %7s  %10.3f ms\n"%(n,r.get('Grid_Size_X',r.get('Grid_Size','?')),r.get('Workgroup_Size_X',r.get('Workgroup_Size','?')),r.get('LDS_Block_Size','?'),d))
out.close()
PY
cat gpurun_out/prof_c4/dispatches.txt | tail -40
