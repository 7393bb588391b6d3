"""Row-slab partition of C = A.B over the GPUs of one node: every outer tile of C
is independent (kernel/Compute.cpp:53-60, kernel/Memory.cpp:114-127), so device g owns the
contiguous rows [row0, row0+rows) of A and C and a replica of B; no collective is needed.
The same arithmetic lives in mm_gemm_multi_device (csrc/mm_capi.hip)."""


def row_slab(size_n, world_size, rank):
    """(row0, rows) of `rank`'s slab: ceil(N/G) rows each, the last slabs possibly short/empty."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    slab = (size_n + world_size - 1) // world_size
    row0 = min(rank * slab, size_n)
    return row0, min(slab, size_n - row0)
