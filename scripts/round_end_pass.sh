set -u
T=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout -k 10 500 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider > gpurun_out/${T}_gpu_tests.log 2>&1); tail -3 gpurun_out/${T}_gpu_tests.log
(timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1); tail -1 gpurun_out/${T}_smoke.log
(timeout -k 10 600 python bench.py > gpurun_out/${T}_cdu_b100000_bench.json 2> gpurun_out/${T}_bench.err); tail -c 300 gpurun_out/${T}_bench.err
Q="--steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_kt -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/gpurun_out/${T}_cdu_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_kt.err); 
find gpurun_out/${T}_kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_cdu_b100000_kernel_stats.csv \;
timeout -k 10 400 bash scripts/pmc_hbm.sh cdu_b100000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io; cp profiles/pmc_hbm_cdu_b100000.json gpurun_out/
timeout -k 10 400 bash scripts/pmc_sq.sh ${T} --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-parity --no-pdip --no-host-io; cp profiles/${T}_pmc_sq.json gpurun_out/
rm -rf gpurun_out/${T}_kt gpurun_out/pmc_hbm_cdu_b100000 gpurun_out/pmc_${T}
ls -la gpurun_out | tail -12
