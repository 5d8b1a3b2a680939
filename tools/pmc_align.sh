set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pa_*
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pa_sq -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pa_sq2 -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_sq2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pa_tcc -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_tcc.log 2>&1
for d in pa_sq pa_sq2 pa_tcc; do python3 tools/pmc_summary.py $(find gpurun_out/$d -name '*counter_collection.csv' | head -1) k_gmm_f16_align; python3 tools/pmc_summary.py $(find gpurun_out/$d -name '*counter_collection.csv' | head -1) k_em_fused; done > gpurun_out/pa_summary.txt 2>&1
cat gpurun_out/pa_summary.txt
