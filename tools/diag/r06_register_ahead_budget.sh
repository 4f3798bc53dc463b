#!/bin/bash
# round 6: config 3's batch from pageable frames, staged (default) against register-ahead uploads (jpegenc_encoder_set_batch_upload),
# unconfined and inside 2 / 4 fixed CPUs: frames/s, fraction of the link, CPUs busy
cd "$GRAFT_REPO_ROOT" || exit 1
for mask in 0-15 0-1 0-3; do
  echo "== taskset -c $mask"
  for ra in "" "--register-ahead"; do
    timeout 300 taskset -c $mask python3 tools/diag/r06_worker_cpu.py --passes 8 --workers 0,2,4 --pinned 0 $ra 2>&1 | grep -v amdgpu.ids > /tmp/ra_rows.jsonl
    python3 tools/diag/r06_summary.py /tmp/ra_rows.jsonl | grep -v "^=="
  done
done
