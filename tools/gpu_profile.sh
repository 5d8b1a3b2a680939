set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t15.log 2>&1
timeout -k 10 240 python bench.py --steps 10 --warmup 3 > gpurun_out/bench8.log 2> gpurun_out/bench8.err
rm -rf gpurun_out/prof_r01g gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01g -o r01g -- python3 bench.py --steps 30 --warmup 10 --no-cpu > gpurun_out/prof_r01g.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_mfma.log 2>&1
echo all-done
