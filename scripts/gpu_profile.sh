#!/bin/bash
# rocprofv3 evidence (outputs under gpurun_out/prof, summaries -> profiles/<round>_* by scripts/summarize_profile.py):
#   kernel-trace --stats of the bench commands of configs 2, 4, 5, 6;  PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters / GRBM, one pass
#   per counter group, no tracing domains next to --pmc) on reduced batches of configs 2 and 5 and on config 4 at full size
#   usage: bash scripts/gpu_profile.sh [round-name, default round4]
cd "${GRAFT_REPO_ROOT:-.}"
ROUND=${1:-round6}
export TMPDIR=/tmp
P=$PWD/gpurun_out/prof
rm -rf $P; mkdir -p $P
B=$PWD/bench.py
R=$PWD
cd /tmp
trace() { timeout $1 rocprofv3 --kernel-trace --stats -d $P/$2 -o bench -- python $B $3 > $P/$2.log 2>&1; echo "$2 rc=$?"; }
pmc() { timeout $1 rocprofv3 --pmc $4 -d $P/$2 -o bench -- python $B $3 > $P/$2.log 2>&1; echo "$2 rc=$?"; }
trace 200 trace_config2 "--config 2 --steps 1 --warmup 1 --cpu-sample 0 --also none --plain"
trace 200 trace_config4 "--config 4 --steps 1 --warmup 1 --cpu-sample 0 --plain"
trace 300 trace_config5 "--config 5 --steps 1 --warmup 0 --cpu-sample 0"
trace 200 trace_config6 "--config 6 --steps 1 --warmup 1 --cpu-sample 0 --plain"
S2="--config 2 --reads 16000 --steps 1 --warmup 0 --cpu-sample 0 --also none --plain"
S4="--config 4 --steps 1 --warmup 0 --cpu-sample 0 --plain"      # full size: below ~2048 jobs the queue hands out whole jobs and runs one wavefront per SIMD
S5="--config 5 --reads 8192 --db-targets 2048 --steps 1 --warmup 0 --cpu-sample 0"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY"
GR="GRBM_GUI_ACTIVE GRBM_COUNT"
# second SQ pass (round 5): LDS bank conflicts / LDS-array cycles, LDS and VMEM instruction cycles, wave-parked cycles -- what the contract's
# "LDS / VALU utilisation" asks for.  Only names this rocprofv3 knows on this device are kept (an unknown counter fails the whole pass).
AVAIL=$(rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+\b" | sort -u)
SQ2=""
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM; do
  if echo "$AVAIL" | grep -qx "$c"; then SQ2="$SQ2 $c"; fi
done
SQ2=${SQ2# }
[ -z "$SQ2" ] && SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
echo "second SQ pass: $SQ2"
pmc 150 pmc2_fetch "$S2" FETCH_SIZE; pmc 150 pmc2_write "$S2" WRITE_SIZE; pmc 150 pmc2_sq1 "$S2" "$SQ1"; pmc 150 pmc2_sq2 "$S2" "$SQ2"; pmc 150 pmc2_grbm "$S2" "$GR"
pmc 150 pmc4_fetch "$S4" FETCH_SIZE; pmc 150 pmc4_write "$S4" WRITE_SIZE; pmc 150 pmc4_sq1 "$S4" "$SQ1"; pmc 150 pmc4_sq2 "$S4" "$SQ2"; pmc 150 pmc4_grbm "$S4" "$GR"
pmc 150 pmc5_fetch "$S5" FETCH_SIZE; pmc 150 pmc5_write "$S5" WRITE_SIZE; pmc 150 pmc5_sq1 "$S5" "$SQ1"; pmc 150 pmc5_sq2 "$S5" "$SQ2"; pmc 150 pmc5_grbm "$S5" "$GR"
cd $R
python scripts/summarize_profile.py $ROUND gpurun_out/prof > gpurun_out/prof_summary.log 2>&1; tail -n 40 gpurun_out/prof_summary.log
find $P -name "*.db" -delete
du -sh gpurun_out
