#!/bin/bash
# round 5: (1) the layout survey of round 2 on this tree; (2) kernel trace of ONE config-5 frame: device-resident
# 4K RGB 4:4:4 progressive(4) + optimised, sequential optimised and baseline through jpegenc_encoder_encode_device
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 tools/diag/layout_survey.py > gpurun_out/r05/layout_survey.txt 2>&1
python3 tools/bench_c5_device.py > gpurun_out/r05/c5_device.jsonl 2>&1
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/c5_trace -o t -- python3 $R/tools/bench_c5_device.py > /dev/null 2>&1)
f=$(find gpurun_out/r05/c5_trace -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:30]: print(f\"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}\")" "$f" > gpurun_out/r05/c5_kernel_stats.txt
find gpurun_out/r05/c5_trace -name '*kernel_trace.csv' -delete
