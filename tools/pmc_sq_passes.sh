set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_sq}
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OUT/$name.bench.json" 2> "$OUT/$name.err" || echo "pass $name failed (rc $?)"
  find "$OUT/$name" -name "*.csv" -size +30M -delete
}
pass sq_time   SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass sq_insts  SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU
pass grbm      GRBM_GUI_ACTIVE GRBM_COUNT
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err" || true
cat "$OUT/summary.md"
