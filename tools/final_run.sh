cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu.log 2>&1; grep -a 'passed\|failed' gpurun_out/r02_pytest_gpu.log | tail -2
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 300 gpurun_out/r02_bench.err
timeout 300 bash tools/pmc.sh r02_pmc_t1 tools/t1_prof_target.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r02_pmc_t1 > gpurun_out/r02_pmc_t1_summary.txt
timeout 300 bash tools/pmc.sh r02_pmc_dec tools/dec_perf.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r02_pmc_dec dec_ > gpurun_out/r02_pmc_dec_summary.txt
timeout 300 bash tools/final_stats.sh > gpurun_out/r02_final_stats.txt 2>&1
head -c 600 gpurun_out/r02_bench.json
