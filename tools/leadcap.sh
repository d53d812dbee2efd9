#!/bin/bash
# Development aid: time and FETCH_SIZE of the 2 KiB-record log scan with the placer's lead over the checksum waves capped
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/leadcap
rm -rf $O && mkdir -p $O
for cap in ${CAPS:-0 64 32 16 8}; do
  export TCRC_WAL_TUNE=$(( (cap << 16) | ${TUNE_LOW:-0} ))
  timeout -k 10 120 python3 tools/measure_configs.py --only ${CFG:-log_2k} --launches 30 --settle 1.0 2>/dev/null | grep '^{' > $O/t_$cap.json || exit 1
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/f_$cap -o fetch -- python3 tools/measure_configs.py --only ${CFG:-log_2k} --launches 5 --settle 0.2 > $O/f_$cap.log 2>&1 || exit 1
  python3 - <<PY
import json, sqlite3
t=json.loads(open("$O/t_$cap.json").read().splitlines()[0])
c=sqlite3.connect("$O/f_$cap/fetch_results.db")
kb,n=c.execute("select avg(value), count(*) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%k_wal_%'").fetchone()
rb=2.0*kb*1024.0
print("cap", $cap, "us", t["roofline"]["avg_launch_us"], "read_bytes", rb, "dispatches", n, "ratio", round(rb/t["roofline"]["algorithmic_bytes_per_launch"],4), flush=True)
PY
  rm -rf $O/f_$cap
done
