"""Column decomposition across GPUs (one process per GPU, no halo, no collective on the physics path).

Restates the block split of the reference's create_domain_decomposition_1D (src/utils/utils.cc:27-44):
n_small = N // P, the first N % P ranks own one extra column, blocks are contiguous and ordered by rank.
"""


def block_range(ncols, world_size, rank):
    """-> (start, count) of the columns owned by `rank`."""
    if world_size <= 0 or not (0 <= rank < world_size) or ncols < 0:
        raise ValueError("bad decomposition arguments")
    n_small = ncols // world_size
    n_big = ncols % world_size
    if rank < n_big:
        return rank * (n_small + 1), n_small + 1
    return n_big * (n_small + 1) + (rank - n_big) * n_small, n_small


def all_ranges(ncols, world_size):
    return [block_range(ncols, world_size, r) for r in range(world_size)]
