#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cs in 0 1 2 3 4 5; do
  export TCRC_WAL_CHUNK_SHIFT=$cs
  echo "chunk shift $cs"
  timeout -k 10 200 python3 -u tools/ab_probe.py mixed,2k 2>&1 | grep -v amdgpu || exit 1
done
