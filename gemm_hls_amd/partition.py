"""Row-slab partition of C = A.B over the GPUs of one node: every outer tile of C
is independent (kernel/Compute.cpp:53-60, kernel/Memory.cpp:114-127), so device g owns the
contiguous rows [row0, row0+rows) of A and C and a replica of B; no collective is needed.

The arithmetic lives in ONE place, the library (mm_row_slab in csrc/mm_capi.hip, which
mm_gemm_multi_device uses itself): slabs are ceil(N/G) rows rounded up to whole tile rows of the
kernel that will run on them.  `row_slab_for` asks the library -- the partition every caller should use;
`row_slab` is the same rule with an EXPLICIT tile height (no default: 64512 rows over 8 ranks are 8064-row slabs
with 128-row tiles but 8192-row slabs with the 256-row tiles the large fp32 / half / int8 kernels keep),
for tests and callers that already know the tile."""


def row_slab(size_n, world_size, rank, tile_rows):
    """(row0, rows) of `rank`'s slab: ceil(N/G) rows rounded UP to a whole macro-tile, so that no
    rank but the last busy one owns a ragged tile row ("contiguous slabs aligned to the
    kernel's N macro-tile"); trailing slabs may be short or empty."""
    if world_size < 1 or not (0 <= rank < world_size) or tile_rows < 1:
        raise ValueError("bad world_size/rank/tile_rows")
    slab = (size_n + world_size - 1) // world_size
    slab = min(size_n, (slab + tile_rows - 1) // tile_rows * tile_rows)
    row0 = min(rank * slab, size_n)
    return row0, min(slab, size_n - row0)


def row_slab_for(cfg, size_n, size_k, size_m, world_size, rank):
    """The slab the library itself would give `rank` for this configuration and shape (tile height of the
    kernel that will run): what bench.py's ranks own, identical to mm_gemm_multi_device's split."""
    import gemm_hls_amd as g
    return g.row_slab(cfg, size_n, size_k, size_m, world_size, rank)
