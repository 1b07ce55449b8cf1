"""Image-space tile ownership for multi-GPU rendering.

The reference shards a frame over worker threads as RenderTiles handed out in Z-order
(src/core/renderer/RenderTileMap.cpp:26-134) and over processes as `--itx/--ity` image tiles
(src/core/renderer/RenderFactory.cpp:16-42).  Pixels are the independent unit (per-pixel RNG stream,
RenderRandomMap), so ranks own disjoint sets of fixed-size tiles dealt round-robin along the Z-order
curve -- interleaving balances the load when geometry density varies over the image.
"""


def _morton(x, y):
    def spread(v):
        v &= 0xFFFF
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def all_tiles(width, height, tile=64):
    """Tiles (x0, y0, x1, y1) covering the film, sorted along the Z-order curve."""
    tiles = []
    for ty in range(0, height, tile):
        for tx in range(0, width, tile):
            tiles.append((_morton(tx // tile, ty // tile), (tx, ty, min(width, tx + tile), min(height, ty + tile))))
    tiles.sort()
    return [t for _, t in tiles]


def tiles_for_rank(width, height, rank, world_size, tile=64):
    """Tile k of the Z-ordered list belongs to rank k mod world_size."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    return [t for k, t in enumerate(all_tiles(width, height, tile)) if k % world_size == rank]


def owned_pixel_count(tiles):
    return sum((x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles)
