#!/bin/bash
# SQ / GRBM PMC passes of one bench.py command (run ON the GPU box, from the repo root):
#   scripts/pmc_sq.sh <tag> <bench.py args...>
# Each pass: rocprofv3 --kernel-trace --pmc <<= 8 SQ counters + GRBM_GUI_ACTIVE> (no other trace domain).
# Counter names are filtered by `rocprofv3 -L` so an unknown name cannot fail a pass.
set -u
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_list.txt" 2>&1 || true
pick() { local r=""; for c in "$@"; do if grep -qw "$c" "$out/counters_list.txt"; then r="$r $c"; fi; done; echo $r; }
P1=$(pick GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES)
P2=$(pick GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS)
P3=$(pick GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM)
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  echo "pass $i: $P"
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$out/p$i" -- python3 bench.py "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; }
done
python3 scripts/pmc_sq.py "$out" "profiles/${tag}_pmc_sq.json" "bench.py $*"
