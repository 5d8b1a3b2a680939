# Round-3 evidence run (one gpurun call): GPU tests, the default bench line, a kernel trace of the bench with the bench
# line of that same run, the PMC passes of the headline workload (separate runs, no trace options) and of the
# 10,000-utterance alignment / training workload.  Results land under gpurun_out/; the summaries are copied to profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_gputest.log 2>&1
tail -2 gpurun_out/r03_gputest.log
timeout -k 10 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
rm -rf gpurun_out/prof_r03 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o r03 -- python3 bench.py --steps 30 --warmup 10 --no-cpu > gpurun_out/r03_bench_under_rocprof.json 2> gpurun_out/prof_r03.log
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu --train-utts 0 --align-utts 0 --decode-words 0 > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu --train-utts 0 --align-utts 0 --decode-words 0 > gpurun_out/pmc_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu --train-utts 0 --align-utts 0 --decode-words 0 > gpurun_out/pmc_mfma.log 2>&1
python3 tools/pmc_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --mfma gpurun_out/pmc_mfma --frames 499674 --out gpurun_out/r03_pmc_traffic.json --tag "round 3, bench.py --steps 2 --warmup 1 --no-cpu --train-utts 0 --align-utts 0 --decode-words 0" > gpurun_out/r03_pmc_traffic.txt
bash tools/pmc_align.sh > /dev/null
timeout -k 10 300 python tools/align_bench.py --utts 100000 --no-dense > gpurun_out/r03_align_train_100k_utts.json 2> /dev/null
timeout -k 10 200 python tools/align_bench.py --utts 10000 > gpurun_out/r03_align_train_10k_utts.json 2> /dev/null
echo all-done
