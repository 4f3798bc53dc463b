#!/bin/bash
# Collect the judged measurement artefacts of one round on the MI355X box (run through gpurun from the
# repo root):   tools/profile_round.sh r01_f
# Writes under gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (see profiles/README.md).
# --pmc passes are separate runs and never combined with a trace option.
set -u
tag=${1:-rXX}    # (a tag used before: remove its gpurun_out/<tag>_{stats,fetch,write,sq,sq2} locally first - gpurun merges new files beside old ones)
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
# long enough that the ~50 slower launches after start-up do not move the per-kernel average
BENCH="bench.py --steps 1000 --warmup 20 --cpu-seconds 0.2 --headline-only"
PMC_BENCH="bench.py --steps 20 --warmup 5 --settle-ms 0 --cpu-seconds 0.2 --headline-only"

# (nothing of an earlier round of the same tag: reduce_pmc.py takes one counter file per pass)
rm -rf "$out/${tag}_stats" "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_sq" "$out/${tag}_sq2"
# which sources the library under test was built from (bench.py compares it with the tree it runs in)
python3 jpeg-encoder_amd/srchash.py > "$out/${tag}_srchash.txt"
python3 tools/step_series.py 400 > "$out/${tag}_step_series.txt" 2>/dev/null
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"

timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 $BENCH \
    > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.err"

timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_fetch" -- python3 $PMC_BENCH > /dev/null 2> "$out/${tag}_fetch.err"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_write" -- python3 $PMC_BENCH > /dev/null 2> "$out/${tag}_write.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$out/${tag}_sq" -- python3 $PMC_BENCH > /dev/null 2> "$out/${tag}_sq.err"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$out/${tag}_sq2" -- python3 $PMC_BENCH > /dev/null 2> "$out/${tag}_sq2.err"

# keep only the small summaries (the per-dispatch traces are large)
find "$out/${tag}_stats" -name '*kernel_trace.csv' -delete
ls -R "$out" | grep "${tag}" | head -40
cat "$out/${tag}_bench.json"
