#!/bin/bash
# Collects the measurements and rocprofv3 evidence kept under profiles/ (run on a GPU box through gpurun):
#   - the headline bench plain, under kernel tracing, and under two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs);
#   - every other configuration (tools/measure_configs.py) plain, and one at a time under kernel tracing and the
#     same two PMC passes.
# Results land in gpurun_out/collect/; tools/merge_profiles.py turns them into the files under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/collect
rm -rf $O && mkdir -p $O
set -e
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
# name:kernel-substring:extra arguments
for spec in log_2k:k_wal_fused: log_mixed:k_wal_fused: log_mixed_recyclable:k_wal_fused: \
            blocks4k_var_verify:k_crc_batch:--phase=verify blocks4k_var_stamp:k_crc_batch:--phase=stamp \
            blocks8k_var_verify:k_crc_batch:--phase=verify blocks8k_var_stamp:k_crc_batch:--phase=stamp spans_2k:k_crc_batch: long_64x4m:k_crc_batch: long_single:k_crc_batch: ; do
  name=${spec%%:*}; rest=${spec#*:}; needle=${rest%%:*}; extra=${rest#*:}
  only=${name%_verify}; only=${only%_stamp}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- python3 tools/measure_configs.py --only $only $extra --launches 30 --settle 1.0 > $O/kt_$name.log 2>&1
  python3 tools/rocprof_summary.py kernel $O/kt_$name/kt_results.db $O/${name}_kernel_stats.csv > /dev/null
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/f_$name -o fetch -- python3 tools/measure_configs.py --only $only $extra --launches 5 --settle 0.2 > $O/f_$name.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $O/w_$name -o wr -- python3 tools/measure_configs.py --only $only $extra --launches 5 --settle 0.2 > $O/w_$name.log 2>&1
  python3 tools/rocprof_summary.py pmc $O/f_$name/fetch_results.db $O/w_$name/wr_results.db $needle $O/${name}_hbm_traffic.json > /dev/null
  rm -rf $O/kt_$name $O/f_$name $O/w_$name
  echo "$name done"
done
timeout -k 10 200 python3 -u tools/latency_probe.py > $O/latency.txt 2>&1 || true
cat $O/bench.json
head -4 $O/bench_kernel_stats.csv
cat $O/hbm_traffic.json
