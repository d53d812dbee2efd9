#!/bin/bash
# Development aid: the small-record logs under several settings of the scan's tune word (same box, one after the other)
cd "$GRAFT_REPO_ROOT"
for t in ${TUNES:-0 1}; do
  export TCRC_WAL_TUNE=$t
  echo "tune $t"
  timeout -k 10 200 python3 -u tools/ab_probe.py ${SHAPES:-s128,s512,s1k} 2>&1 | grep -v amdgpu || exit 1
done
