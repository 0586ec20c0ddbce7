# rocprofv3 evidence for the bench workload (written under gpurun_out/, summaries copied to profiles/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
BENCH="python bench.py --steps 1 --warmup 1 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o bench -- $BENCH > gpurun_out/prof/trace.log 2>&1
SMALL="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --reads 16000"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d gpurun_out/prof/pmc_sq1 -o bench -- $SMALL > gpurun_out/prof/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/prof/pmc_sq2 -o bench -- $SMALL > gpurun_out/prof/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch -o bench -- $SMALL > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write -o bench -- $SMALL > gpurun_out/prof/pmc_write.log 2>&1
find gpurun_out/prof -name "*.csv" | head -40
du -sh gpurun_out/prof
