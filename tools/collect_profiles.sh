#!/bin/bash
# Collects the measurements and rocprofv3 evidence kept under profiles/ (run on a GPU box through gpurun):
#   - the headline bench plain, under kernel tracing, and under two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs);
#   - every other configuration (tools/measure_configs.py) plain, and one at a time under kernel tracing and the
#     same two PMC passes.
# Results land in gpurun_out/collect/; tools/merge_profiles.py turns them into the files under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# The whole collection is longer than one GPU call may be: PART=head does the headline bench and the plain run of
# every configuration, PART=a / PART=b one half each of the profiled configurations (results of all three parts land
# in the same directory and are merged afterwards).
O=gpurun_out/collect
PART=${PART:-head}
mkdir -p $O
set -e
if [ "$PART" = head ]; then
timeout -k 10 300 python3 -u bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 bench.py --no-cpu-baseline --steps 5 > $O/fetch.log 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o wr -- python3 bench.py --no-cpu-baseline --steps 5 > $O/write.log 2>&1
python3 tools/rocprof_summary.py kernel $O/kt/kt_results.db $O/bench_kernel_stats.csv > /dev/null
python3 tools/rocprof_summary.py pmc $O/fetch/fetch_results.db $O/write/wr_results.db k_crc_batch $O/hbm_traffic.json > /dev/null
rm -rf $O/kt $O/fetch $O/write
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --corpus-gib 8 > $O/bench_8gib_shard.json 2> $O/bench8.err
echo "headline done"
timeout -k 10 600 python3 -u tools/measure_configs.py 2> $O/configs.err | grep '^{' > $O/configs.jsonl
echo "configs plain done"
timeout -k 10 200 python3 -u tools/latency_probe.py > $O/latency.txt 2>&1 || true
timeout -k 10 200 python3 -u tools/frame_probe.py > $O/frame.txt 2>&1 || true
timeout -k 10 200 python3 -u tools/tiny_probe.py > $O/log_small.txt 2>&1 || true
cat $O/bench.json
head -4 $O/bench_kernel_stats.csv
cat $O/hbm_traffic.json
exit 0
fi
# name:kernel-substring:extra arguments
SQC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ "$PART" = a ]; then
  SPECS="log_2k:k_wal_split::sq log_mixed:k_wal_split::sq log_mixed_recyclable:k_wal_split:: log_small_128:k_wal_split::sq log_small_512:k_wal_split::sq log_small_1k:k_wal_split::sq frame_2040:k_wal_frame::sq"
else
  SPECS="blocks4k_var_verify:k_crc_batch:--phase=verify: blocks4k_var_stamp:k_crc_batch:--phase=stamp: blocks8k_var_verify:k_crc_batch:--phase=verify: blocks8k_var_stamp:k_crc_batch:--phase=stamp: spans_2k:k_crc_batch::sq long_64x4m:k_crc_batch:: long_256m:k_crc_batch:: long_1g:k_crc_batch:: long_single:k_crc_batch::"
fi
if [ "$PART" = b ]; then
timeout -k 10 120 rocprofv3 --pmc $SQC -d $O/sq_bench -o sq -- python3 bench.py --no-cpu-baseline --steps 5 --settle-seconds 0.3 > $O/sq_bench.log 2>&1
python3 tools/rocprof_summary.py sq $O/sq_bench/sq_results.db k_crc_batch $O/bench_sq.json > /dev/null
rm -rf $O/sq_bench
fi
# name:kernel-substring:extra arguments:sq (a pass of eight SQ counters as well)
for spec in $SPECS; do
  IFS=: read -r name needle extra sq <<< "$spec"
  only=${name%_verify}; only=${only%_stamp}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- python3 tools/measure_configs.py --only $only $extra --launches 30 --settle 1.0 > $O/kt_$name.log 2>&1
  python3 tools/rocprof_summary.py kernel $O/kt_$name/kt_results.db $O/${name}_kernel_stats.csv > /dev/null
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/f_$name -o fetch -- python3 tools/measure_configs.py --only $only $extra --launches 5 --settle 0.2 > $O/f_$name.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $O/w_$name -o wr -- python3 tools/measure_configs.py --only $only $extra --launches 5 --settle 0.2 > $O/w_$name.log 2>&1
  python3 tools/rocprof_summary.py pmc $O/f_$name/fetch_results.db $O/w_$name/wr_results.db $needle $O/${name}_hbm_traffic.json > /dev/null
  if [ -n "$sq" ]; then
    timeout -k 10 120 rocprofv3 --pmc $SQC -d $O/sq_$name -o sq -- python3 tools/measure_configs.py --only $only $extra --launches 5 --settle 0.2 > $O/sq_$name.log 2>&1
    python3 tools/rocprof_summary.py sq $O/sq_$name/sq_results.db $needle $O/${name}_sq.json > /dev/null
  fi
  rm -rf $O/kt_$name $O/f_$name $O/w_$name $O/sq_$name
  echo "$name done"
done
echo "part $PART done"
