"""Row-slab partition of C = A.B over the GPUs of one node: every outer tile of C
is independent (kernel/Compute.cpp:53-60, kernel/Memory.cpp:114-127), so device g owns the
contiguous rows [row0, row0+rows) of A and C and a replica of B; no collective is needed.
The same arithmetic lives in mm_gemm_multi_device (csrc/mm_capi.hip)."""


TILE_ROWS = 128  # rows of the default fp32 macro-tile (csrc/mm_mfma_f32.hip, 128 x 256); 256-row tiles divide evenly too


def row_slab(size_n, world_size, rank, tile_rows=TILE_ROWS):
    """(row0, rows) of `rank`'s slab: ceil(N/G) rows rounded UP to a whole macro-tile, so that no
    rank but the last one owns a ragged tile row ("contiguous slabs aligned to the
    kernel's N macro-tile"); trailing slabs may be short or empty."""
    if world_size < 1 or not (0 <= rank < world_size) or tile_rows < 1:
        raise ValueError("bad world_size/rank/tile_rows")
    slab = (size_n + world_size - 1) // world_size
    slab = (slab + tile_rows - 1) // tile_rows * tile_rows
    row0 = min(rank * slab, size_n)
    return row0, min(slab, size_n - row0)
