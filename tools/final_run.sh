cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu.log 2>&1; grep -a 'passed\|failed' gpurun_out/r02_pytest_gpu.log | tail -2
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 300 gpurun_out/r02_bench.err
mkdir -p gpurun_out/r02_ks_bench; (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_ks_bench -o k -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_prof.json 2>/dev/null)
bash tools/pmc.sh r02_pmc_t1 tools/t1_prof_target.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r02_pmc_t1 > gpurun_out/r02_pmc_t1_summary.txt
head -c 600 gpurun_out/r02_bench.json
