# rocprofv3 evidence of round 2 (outputs under gpurun_out/prof2, summaries copied to profiles/ by scripts/summarize_round2.py):
#   kernel-trace --stats of the bench commands of configs 2, 4, 5;  PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters, one pass
#   per counter group, no tracing domains next to --pmc) on reduced batches of the same workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/prof2
rm -rf $P; mkdir -p $P
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
trace() { timeout $1 rocprofv3 --kernel-trace --stats -d $P/$2 -o bench -- python $B $3 > $P/$2.log 2>&1; echo "$2 rc=$?"; }
pmc() { timeout $1 rocprofv3 --pmc $4 -d $P/$2 -o bench -- python $B $3 > $P/$2.log 2>&1; echo "$2 rc=$?"; }
trace 200 trace_config2 "--config 2 --steps 1 --warmup 1 --cpu-sample 0"
trace 200 trace_config4 "--config 4 --steps 1 --warmup 1 --cpu-sample 0"
trace 300 trace_config5 "--config 5 --steps 1 --warmup 0 --cpu-sample 0"
S2="--config 2 --reads 16000 --steps 1 --warmup 0 --cpu-sample 0"
S4="--config 4 --reads 2000 --steps 1 --warmup 0 --cpu-sample 0"
S5="--config 5 --reads 8192 --db-targets 2048 --steps 1 --warmup 0 --cpu-sample 0"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"
pmc 150 pmc2_fetch "$S2" FETCH_SIZE; pmc 150 pmc2_write "$S2" WRITE_SIZE; pmc 150 pmc2_sq1 "$S2" "$SQ1"
pmc 150 pmc4_fetch "$S4" FETCH_SIZE; pmc 150 pmc4_write "$S4" WRITE_SIZE; pmc 150 pmc4_sq1 "$S4" "$SQ1"
pmc 150 pmc5_fetch "$S5" FETCH_SIZE; pmc 150 pmc5_write "$S5" WRITE_SIZE; pmc 150 pmc5_sq1 "$S5" "$SQ1"
cd $GRAFT_REPO_ROOT
python scripts/summarize_round2.py gpurun_out/prof2 > gpurun_out/prof2_summary.log 2>&1; tail -n 30 gpurun_out/prof2_summary.log
find $P -name "*.db" -size +30M -delete
du -sh gpurun_out
