#!/bin/bash
# Development aid: runs the shape / log probes against several builds of the library in turn
# (TCRC_LIBRARY selects the build), twice, so that box-to-box and run-to-run noise cancels.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in "$@"; do
    lib=$GRAFT_REPO_ROOT/terarkdb_amd/lib/libtcrc$v.so
    echo "== build '$v' rep $rep"
    TCRC_LIBRARY=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'kernel', d['roofline']['achieved'], d['roofline']['avg_launch_ms'])" || exit 1
    TCRC_LIBRARY=$lib timeout -k 10 200 python3 tools/shape_probe.py 2>&1 | grep "lanes 4" || exit 1
    TCRC_LIBRARY=$lib MP_RECORDS=1000000 timeout -k 10 200 python3 tools/wal_probe.py 2>&1 | grep verify | tail -2 || exit 1
    TCRC_LIBRARY=$lib timeout -k 10 200 python3 tools/wal_mixed_probe.py | tail -1 || exit 1
  done
done
