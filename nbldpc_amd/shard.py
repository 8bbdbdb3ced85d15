"""Static partition of a batch of independent codewords ("lanes", main.cpp:46) over ranks / GPUs.

Frames are independent, so the multi-GPU path is a contiguous split of the batch index with no data-path
collective: rank r of W gets [r*B/W, (r+1)*B/W).  Error counters are summed in lane order afterwards.
"""


def shard_range(B, rank, world):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (B * rank) // world, (B * (rank + 1)) // world


def shard_sizes(B, world):
    return [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
