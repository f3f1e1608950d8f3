#!/bin/bash
# HBM traffic per kernel of one bench.py command (run ON the GPU box, from the repo root):
#   scripts/pmc_hbm.sh <name> <bench.py args...>      ->  profiles/pmc_hbm_<name>.json
# Two rocprofv3 passes as MI355X_MICROARCH.md prescribes: --pmc FETCH_SIZE, then --pmc WRITE_SIZE, each with
# --kernel-trace only (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2: they do not fit one pass).
set -u
name=$1; shift
out=$PWD/gpurun_out/pmc_hbm_$name
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py "$@" > "$out/fetch.log" 2>&1 || { echo "FETCH pass failed"; tail -5 "$out/fetch.log"; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py "$@" > "$out/write.log" 2>&1 || { echo "WRITE pass failed"; tail -5 "$out/write.log"; }
python3 scripts/pmc_hbm.py "$out" "profiles/pmc_hbm_$name.json" "bench.py $*"
