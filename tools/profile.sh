#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace statistics and, in separate passes, the HBM
# traffic counters of the headline bench.  Summaries are written by tools/summarize_profile.py.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m "${1:-$ROOT/gpurun_out/prof}")
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 200 --warmup 50 --no-cpu-baseline > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err" || true
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -50
