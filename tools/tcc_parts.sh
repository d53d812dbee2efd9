#!/bin/bash
# Development aid: the L2 counters behind FETCH_SIZE, one by one, for one configuration of tools/measure_configs.py
# usage: CFG=log_2k NEEDLE=k_wal_ bash tools/tcc_parts.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${CFG:-log_2k}; NEEDLE=${NEEDLE:-k_wal_}
O=gpurun_out/tcc_$CFG
rm -rf $O && mkdir -p $O
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum -d $O/a -o a -- python3 tools/measure_configs.py --only $CFG --launches 5 --settle 0.2 > $O/a.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum -d $O/b -o b -- python3 tools/measure_configs.py --only $CFG --launches 5 --settle 0.2 > $O/b.log 2>&1 || exit 1
python3 - <<PY
import sqlite3
for db in ("$O/a/a_results.db", "$O/b/b_results.db"):
    c = sqlite3.connect(db)
    for name, v, n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$NEEDLE%' group by counter_name"):
        print("$CFG", name, round(v, 1), n, flush=True)
PY
rm -rf $O/a $O/b
