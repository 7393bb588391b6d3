"""The N>1 path on CPU: world_size-2 `gloo` processes (no GPU here).  The data path of the
multi-GPU design has no collective (rows of C are split, B replicated); what runs across ranks is
the row partition, the barrier and the max-over-ranks timing of bench.py.  Each rank multiplies
its own slab -- with the CPU oracle standing in for the device, which is fine for a test of the
partition -- and rank 0 checks that the stacked slabs equal the unsplit product bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gemm_hls_amd.partition import row_slab  # noqa: E402


@pytest.mark.parametrize("n,world", [(16384, 8), (65536, 8), (513, 2), (5, 8), (1, 4), (256, 3), (6000, 2), (1000, 8), (20000, 3)])
def test_row_slabs_tile_the_rows(n, world):
    covered = []
    for r in range(world):
        row0, rows = row_slab(n, world, r, 128)
        assert 0 <= rows and row0 + rows <= n
        covered.extend(range(row0, row0 + rows))
    assert covered == list(range(n))
    # slabs are ceil(N/G) rounded up to whole 128-row macro-tiles: every slab start is tile-aligned
    slab = min(n, -(-(-(-n // world)) // 128) * 128)
    assert max(row_slab(n, world, r, 128)[1] for r in range(world)) == slab
    assert all(row_slab(n, world, r, 128)[0] % 128 == 0 or row_slab(n, world, r, 128)[1] == 0 for r in range(world))
    with pytest.raises(TypeError):
        row_slab(n, world, 0)        # the tile height is the caller's to state (or row_slab_for's to ask the library)


@pytest.mark.parametrize("dtype", ["float", "half", "double", "int", "uint8_t"])
@pytest.mark.parametrize("n,k,m,world", [(16384, 16384, 16384, 8), (65536, 16384, 16384, 8), (64512, 16384, 16384, 8),
                                         (33792, 16384, 16384, 8), (513, 528, 528, 2), (5, 16, 16, 8), (1, 16, 16, 4),
                                         (6000, 2048, 2048, 2), (1000, 512, 272, 8), (20000, 4096, 4096, 3), (0, 16, 16, 2)])
def test_library_row_slab_is_the_python_rule_with_the_running_kernels_tile(dtype, n, k, m, world):
    """mm_row_slab (the arithmetic mm_gemm_multi_device and bench.py share; pure arithmetic, runs without a device):
    slabs tile the rows, and are row_slab() with tile_rows = the tile height of the kernel that serves a slab."""
    import gemm_hls_amd as g
    from gemm_hls_amd.partition import row_slab_for
    cfg = g.make_config(dtype)
    got = [row_slab_for(cfg, n, k, m, world, r) for r in range(world)]
    covered = [x for row0, rows in got for x in range(row0, row0 + rows)] if n <= 70000 else None
    assert covered == list(range(n))
    if n == 0:
        assert got == [(0, 0)] * world
        return
    cand = min(n, -(-(-(-n // world)) // 128) * 128)
    tile = g.kernel_info(cfg, cand, k, m).tile_n if cand == n else None
    if tile is None:   # a slab of a bigger job: the tile of the geometry the library reports for the slab it finally hands out
        tile = next(t for t in (64, 128, 256) if got == [row_slab(n, world, r, t) for r in range(world)])
        assert tile in (64, 128, 256)
    assert got == [row_slab(n, world, r, tile) for r in range(world)]
    # no busy rank but the last busy one owns a ragged tile row
    busy = [rows for _, rows in got if rows]
    assert all(rows % tile == 0 for rows in busy[:-1])


def test_row_slab_of_the_baseline_job_and_of_a_half_tile_remainder():
    import gemm_hls_amd as g
    cfg = g.make_config("float")
    assert [g.row_slab(cfg, 65536, 16384, 16384, 8, r) for r in range(8)] == [(8192 * r, 8192) for r in range(8)]
    # 64512 / 8 = 8064 rows = 31.5 tiles of the 256-row kernel that serves such a slab: rounded to whole tile rows
    assert [g.row_slab(cfg, 64512, 16384, 16384, 8, r)[1] for r in range(8)] == [8192] * 7 + [7168]
    with pytest.raises(g.MMError):
        g.row_slab(cfg, 100, 16, 16, 2, 2)
    with pytest.raises(g.MMError):
        g.row_slab(cfg, 100, 16, 16, 0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dtype, ops, shape, out_dir):
    import torch
    import torch.distributed as dist
    import _oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)          # every rank regenerates the seeded inputs
    row0, rows = row_slab(n, world, rank, 128)
    c_slab = _oracle.naive(dtype, ops[0], ops[1], a[row0:row0 + rows], b, threads=1) if rows else \
        np.empty((0, m), a.dtype)
    # bench.py's timing protocol: barrier, local time, MAX over ranks
    dist.barrier()
    t = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 0.25 + world - 1
    gathered = [None] * world
    dist.all_gather_object(gathered, (row0, rows, c_slab.tobytes()))
    if rank == 0:
        full = _oracle.naive(dtype, ops[0], ops[1], a, b, threads=1)
        stacked = b"".join(g[2] for g in sorted(gathered))
        ok = stacked == full.tobytes() and sum(g[1] for g in gathered) == n
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,ops,shape", [("float", ("Multiply", "Add"), (37, 32, 48)),
                                             ("int", ("Multiply", "Add"), (5, 16, 16)),
                                             ("float", ("Add", "Min"), (64, 16, 32))])
def test_world_size_2_row_split_reproduces_unsplit_product(tmp_path, dtype, ops, shape):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, dtype, ops, shape, str(tmp_path)), nprocs=2, join=True,
                       start_method="spawn")
    assert open(tmp_path / "ok").read() == "1"
