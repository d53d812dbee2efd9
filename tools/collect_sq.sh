#!/bin/bash
# One SQ-counter pass (rocprofv3 --pmc, eight counters, no tracing beside it) over the headline bench and three
# other configurations: where the waves' cycles go, and whether the lookups conflict in LDS.
# Results: gpurun_out/collect_sq/<name>_sq.json (copy into profiles/ as rNN_<name>_sq.json).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/collect_sq
rm -rf $O && mkdir -p $O
set -e
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 120 rocprofv3 --pmc $C -d $O/bench -o sq -- python3 bench.py --no-cpu-baseline --steps 5 --settle-seconds 0.3 > $O/bench.log 2>&1
python3 tools/rocprof_summary.py sq $O/bench/sq_results.db k_crc_batch $O/bench_sq.json > /dev/null
rm -rf $O/bench
for spec in spans_2k:k_crc_batch log_2k:k_wal_fused log_mixed:k_wal_fused; do
  name=${spec%%:*}; needle=${spec#*:}
  timeout -k 10 120 rocprofv3 --pmc $C -d $O/$name -o sq -- python3 tools/measure_configs.py --only $name --launches 5 --settle 0.2 > $O/$name.log 2>&1
  python3 tools/rocprof_summary.py sq $O/$name/sq_results.db $needle $O/${name}_sq.json > /dev/null
  rm -rf $O/$name
  echo "$name done"
done
cat $O/bench_sq.json
