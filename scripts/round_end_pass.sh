# One gpurun call at the end of a work block: GPU tests, smoke, kernel-trace stats, HBM and SQ counter passes, then the default
# bench (after the HBM pass, so that its roofline.traffic quotes a profile of this very build).
#   gpurun --timeout 1200 -- 'bash scripts/round_end_pass.sh r03b tests'      GPU tests + smoke (~9 min)
#   gpurun --timeout 1200 -- 'bash scripts/round_end_pass.sh r03b profiles'   traces, counter passes, bench (~10 min)
# (one call for both no longer fits gpurun's 20-minute limit)
# Every step's exit status is recorded in gpurun_out/<tag>_steps.txt and the script exits non-zero if any step failed
# (a failed pytest must not leave a "complete" profile set behind).
set -u
T=$1
WHAT=${2:-all}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
FAIL=0
: > gpurun_out/${T}_steps_${WHAT}.txt
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" >> gpurun_out/${T}_steps_${WHAT}.txt; [ $rc -ne 0 ] && FAIL=1; return 0; }
if [ "$WHAT" != profiles ]; then
step gpu_tests bash -c "timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=500 -p no:cacheprovider > gpurun_out/${T}_gpu_tests.log 2>&1"; tail -3 gpurun_out/${T}_gpu_tests.log
# (smoke() on the prebuilt library, as the driver runs it: `make` on the box would rebuild everything -- file times do not survive the snapshot)
step smoke bash -c "timeout -k 10 300 python -c \"import __graft_entry__ as g; g.smoke()\" > gpurun_out/${T}_smoke.log 2>&1"; tail -1 gpurun_out/${T}_smoke.log
fi
if [ "$WHAT" != tests ]; then
Q="--steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io"
step kernel_trace bash -c "cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_kt -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/gpurun_out/${T}_cdu_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_kt.err"
find gpurun_out/${T}_kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_cdu_b100000_kernel_stats.csv \;
find gpurun_out/${T}_kt -name '*kernel_trace.csv' -exec cp {} gpurun_out/${T}_cdu_b100000_kernel_trace.csv \;
python3 scripts/timeline.py gpurun_out/${T}_cdu_b100000_kernel_trace.csv > gpurun_out/${T}_cdu_b100000_timeline.txt 2>/dev/null
step pmc_hbm bash -c "timeout -k 10 400 bash scripts/pmc_hbm.sh cdu_b100000 --steps 1 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io"; cp profiles/pmc_hbm_cdu_b100000.json gpurun_out/
step pmc_hbm_nn bash -c "timeout -k 10 300 bash scripts/pmc_hbm.sh nn_b1048576 --workload nn --steps 1 --warmup 1"; cp profiles/pmc_hbm_nn_b1048576.json gpurun_out/ 2>/dev/null
step pmc_hbm_cstrs bash -c "timeout -k 10 300 bash scripts/pmc_hbm.sh cstrs_b10000 --workload cstrs --batch 10000 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io"; cp profiles/pmc_hbm_cstrs_b10000.json gpurun_out/ 2>/dev/null
step pmc_sq bash -c "timeout -k 10 400 bash scripts/pmc_sq.sh ${T} --steps 1 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io"; cp profiles/${T}_pmc_sq.json gpurun_out/
step bench bash -c "NNMPC_BENCH_DETAIL=gpurun_out/${T}_bench_detail.json timeout -k 10 700 python bench.py > gpurun_out/${T}_cdu_b100000_bench.json 2> gpurun_out/${T}_bench.err"; tail -c 300 gpurun_out/${T}_bench.err
fi
rm -rf gpurun_out/${T}_kt gpurun_out/pmc_hbm_cdu_b100000 gpurun_out/pmc_hbm_nn_b1048576 gpurun_out/pmc_hbm_cstrs_b10000 gpurun_out/pmc_${T} gpurun_out/${T}_cdu_b100000_kernel_trace.csv
cat gpurun_out/${T}_steps_${WHAT}.txt
exit $FAIL
