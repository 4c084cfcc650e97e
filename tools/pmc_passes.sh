#!/bin/bash
# Runs on the GPU box (gpurun): hardware counters of the headline step's kernels, one rocprofv3 --pmc pass per
# counter group (SQ: 8 slots, TCC: 4; never combined with trace domains other than --kernel-trace).
# Summarised by tools/pmc_summary.py into the table committed under profiles/.
#   tools/pmc_passes.sh [outdir] [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/r02/pmc}
shift || true
ARGS=${*:---steps 6 --warmup 3 --no-cpu-baseline}
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  echo "== pass $name: $*"
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.bench.json" 2> "$OUT/$name.err" || echo "pass $name failed (rc $?)"
  find "$OUT/$name" -name "*.csv" -size +30M -delete
}
pass sq_time   SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass sq_insts  SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU
pass sq_vmem   SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT
pass grbm      GRBM_GUI_ACTIVE GRBM_COUNT
pass tcc_wr    TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL TCC_BUSY
pass tcc_rd    TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL
pass tcc_req   TCC_REQ TCC_HIT TCC_MISS TCC_CYCLE
pass tcp       TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ
pass ta        TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY TD_TD_BUSY
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err" || true
ls "$OUT"
