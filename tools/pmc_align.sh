set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pa_*
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pa_fetch -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pa_tcc -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pa_tcp -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_tcp.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pa_sq -o p -- python3 tools/align_bench.py --utts 10000 --no-dense --reps 1 > gpurun_out/pa_sq.log 2>&1
echo all-done
