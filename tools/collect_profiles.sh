#!/bin/bash
# Runs the headline bench plain, under rocprofv3 kernel tracing, and under two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs), then the other configurations.  Results land in
# gpurun_out/collect/; the summaries are copied to profiles/ afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/collect
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 -u bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 bench.py --no-cpu-baseline --steps 5 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o wr -- python3 bench.py --no-cpu-baseline --steps 5 > $O/write.log 2>&1
python3 tools/rocprof_summary.py kernel $O/kt/kt_results.db $O/bench_kernel_stats.csv > /dev/null
python3 tools/rocprof_summary.py pmc $O/fetch/fetch_results.db $O/write/wr_results.db k_crc_batch $O/hbm_traffic.json > /dev/null
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --corpus-gib 8 > $O/bench_8gib_shard.json 2> $O/bench8.err
timeout -k 10 200 python3 -u tools/latency_probe.py > $O/latency.txt 2>&1
timeout -k 10 900 python3 -u tools/measure_paths.py $O/paths.json > $O/paths.log 2>&1
timeout -k 10 600 python3 -u tools/measure_files.py $O/files.json > $O/files.log 2>&1
rm -rf $O/kt $O/fetch $O/write
cat $O/bench.json
head -4 $O/bench_kernel_stats.csv
cat $O/hbm_traffic.json
