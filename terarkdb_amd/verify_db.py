"""Checksum verification of a database directory on all GPUs of a node.

    python -m terarkdb_amd.verify_db DIR                                   # one GPU
    python -m torch.distributed.run --nproc-per-node N -m terarkdb_amd.verify_db DIR

One process per GPU; the table, log and MANIFEST files are dealt to the ranks by size (largest
first, to the rank with the fewest bytes so far), every rank
verifies its share (DBImpl::VerifyChecksum's loop, db/db_impl.cc:4657-4690), and the ranks exchange their
eight counters and their first error with one all-gather each (RCCL).  Rank 0 prints the totals and exits
with 1 if any table file is corrupt or any log record fails its checksum.
"""
import os
import sys


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print(__doc__)
        return 2
    import torch
    import torch.distributed as dist
    from . import capi, files

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if "TORCHELASTIC_RUN_ID" in os.environ or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    eng = capi.Engine(local_rank)
    res = files.verify_db_directory(eng, argv[0], rank, world)
    counters = [res.table_files, res.table_files_declined, res.table_blocks, res.table_bytes_covered, res.log_files,
                res.log_bytes, res.log_records, res.log_bad_records, 0 if res.ok else 1]
    problems = [None] * world
    if dist.is_initialized():
        mine = torch.tensor(counters, dtype=torch.int64, device="cuda")
        everyone = torch.empty(len(counters) * world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(everyone, mine)
        rows = everyone.view(world, len(counters)).sum(dim=0).tolist()
        dist.all_gather_object(problems, None if res.ok else (res.bad_file, res.status))
    else:
        rows = counters
        problems = [None if res.ok else (res.bad_file, res.status)]
    rc = 0
    if rank == 0:
        print(f"table files: {rows[0]} verified ({rows[2]} blocks, {rows[3]} bytes covered), {rows[1]} declined")
        print(f"log files: {rows[4]} ({rows[5]} bytes, {rows[6]} records, {rows[7]} with a checksum mismatch)")
        for p in problems:
            if p:
                print(f"{p[0]} is corrupted: {p[1]}")
        rc = 1 if rows[8] or rows[7] else 0
        print("OK" if rc == 0 else "FAILED")
    eng.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
