#!/bin/bash
# Development aid: SQ counters of the log-record CRC kernel (two passes of at most 8 SQ counters).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/walpmc
rm -rf $O && mkdir -p $O
export MP_RECORDS=1000000
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/a -o a -- python3 tools/wal_probe.py > $O/a.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/b -o b -- python3 tools/wal_probe.py > $O/b.log 2>&1 || exit 1
python3 tools/pmc_table.py $O/a/a_results.db "k_crc_batch<4" > $O/table.txt
python3 tools/pmc_table.py $O/b/b_results.db "k_crc_batch<4" >> $O/table.txt
rm -rf $O/a $O/b
cat $O/table.txt
