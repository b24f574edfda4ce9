#!/bin/bash
# Round-6 evidence run on the GPU box (everything lands in gpurun_out/r06/; summaries are copied to profiles/ afterwards):
#   the driver's bench command, the full GPU test suite, rocprofv3 kernel stats of the headline command / one T2 line / one T3 step,
#   the T1 counter passes, the smm / t2 (GMM, K = 10, Student-t) / t3 (GMM, Student-t) bench workloads, counters of the T2 kernels
#   (forward with in-kernel noise, ring backward), the kernel sequence of the graph-replayed minibatch step.
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; echo "driver cmd rc=$?"
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -a 'passed\|failed' $O/pytest_gpu.log | tail -2
  cp gpurun_out/r03_parity_errors.json $O/parity_errors.json 2>/dev/null
fi
python bench.py > $O/bench_default.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
for t in headline t2 t3; do mkdir -p $O/ks_$t; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_headline -o k -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_headline_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_t2 -o k -- python3 $R/bench.py --workload t2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_t2_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_t3 -o k -- python3 $R/tools/t3_prof_target.py 1000000 > $O/t3_prof.txt 2>/dev/null
for t in headline t2 t3; do cp $(find $O/ks_$t -name 'k_kernel_stats.csv' | head -1) $O/${t}_kernel_stats.csv; rm -rf $O/ks_$t; done
cd $R
export REPS=5
timeout 400 bash tools/pmc.sh r06/pmc_t1 tools/t1_prof_target.py > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r06/pmc_t1 pass_xdl > $O/t1_pmc_summary.txt 2>&1
rm -rf gpurun_out/r06/pmc_t1
timeout 300 bash tools/pmc.sh r06/pmc_dec tools/dec_perf.py > /dev/null 2>&1
{ echo "# decoder kernels, counters per launch (tools/dec_perf.py: 65 536 data rows = 1.05e7 decoder rows = 655 360 tiles of 16 rows; U = 50);"; echo "# SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles"; python3 tools/pmc_summary.py gpurun_out/r06/pmc_dec dec_bwd_kernel; python3 tools/pmc_summary.py gpurun_out/r06/pmc_dec dec_fwd_kernel; } > $O/decoder_pmc_summary.txt 2>&1
rm -rf gpurun_out/r06/pmc_dec
export REPS=3
timeout 400 bash tools/pmc.sh r06/pmc_t2f tools/t2_fwd_rng_prof.py > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r06/pmc_t2f svae_estep_fwd4 > $O/t2_fwd_rng_pmc_summary.txt 2>&1
rm -rf gpurun_out/r06/pmc_t2f
python bench.py --workload smm --no-extra > $O/bench_smm.json 2>> $O/bench.err
python bench.py --workload t2 --steps 20 --warmup 5 > $O/bench_t2.json 2>> $O/bench.err
python bench.py --workload t2 --k 10 --steps 20 --warmup 5 > $O/bench_t2_k10.json 2>> $O/bench.err
python bench.py --workload t2 --smm --steps 20 --warmup 5 > $O/bench_t2_smm.json 2>> $O/bench.err
python bench.py --workload t2 --smm --k 10 --steps 20 --warmup 5 > $O/bench_t2_smm_k10.json 2>> $O/bench.err
python bench.py --workload t3 --steps 5 --warmup 3 > $O/bench_t3.json 2>> $O/bench.err
python bench.py --workload t3 --smm --steps 5 --warmup 3 > $O/bench_t3_smm.json 2>> $O/bench.err
VARIANTS="16_0 10_0 16_1 10_1" bash tools/r4_t2_pmc.sh
for v in 16_0 10_0 16_1 10_1; do mv gpurun_out/r4_t2_pmc_$v.txt $O/t2_pmc_$v.txt; done
bash tools/kseq.sh r06mb enc_prep tools/r5_mb_graph.py > $O/minibatch64_direct_kernel_seq.txt 2>&1     # (the mid-round 13-node sequence: profiles/r06_t3_minibatch64_kernel_seq.txt)
python tools/ubench/hbm_rw.py > $O/hbm_rw.txt 2>&1
for i in 1 2; do python3 tools/dec_perf.py 262144 2>&1 | tail -1; done > $O/dec_perf.txt
grep "pass_xdl\|pass_kernel\|finalize" $O/headline_kernel_stats.csv | cut -c1-200
head -c 300 $O/bench_driver_cmd.json; echo
for f in t2 t2_k10 t2_smm t2_smm_k10 t3 t3_smm smm; do python3 - <<PY
import json
j=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); r=j['roofline']
print('$f', 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f ms' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'cpu x%.0f' % j.get('speedup_vs_cpu_baseline', float('nan')))
PY
done
# C2-shaped T2 line (verdict r5, item 8): N = 1e5, L = 2, K = 10 - the generic backward kernel's shape (the ring needs even L >= 4)
python bench.py --workload t2 --n 100000 --d 2 --k 10 --steps 20 --warmup 5 > $O/bench_t2_c2.json 2>> $O/bench.err
