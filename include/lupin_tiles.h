/*
 * lupin_tiles.h -- which rank owns which tile of a tile-sharded frame (lupin_hip_pathtrace_scene_tiles, pack / unpack).
 * Tiles are numbered row-major like the reference's TileParams (renderer.rs:816-817): t = ty * tiles_x + tx.
 *
 * Round-robin: rank r owns tiles r, r + world, r + 2 world, ...  When a row holds a multiple of `world` tiles that rule
 * would give every rank fixed columns (stripes), so in that case each row is rotated by one more: owner = (tx + ty) % world.
 * Either way a rank's tiles are enumerated row-major: lupin_owned_tile(j, ...) is its j-th tile.
 * Shared by the kernels, the host library and (restated) lupinpathtracer_amd/distributed.py.
 */
#ifndef LUPIN_TILES_H
#define LUPIN_TILES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LUPIN_TILES_FN __host__ __device__ static inline
#else
#define LUPIN_TILES_FN static inline
#endif

LUPIN_TILES_FN uint32_t lupin_tile_owner(uint32_t t, uint32_t tiles_x, uint32_t world)
{
    if (world > 1u && tiles_x % world == 0u) return (t % tiles_x + t / tiles_x) % world;
    return t % world;
}

LUPIN_TILES_FN uint32_t lupin_owned_tile_count(uint32_t total_tiles, uint32_t rank, uint32_t world)
{
    return total_tiles > rank ? (total_tiles - rank + world - 1u) / world : 0u;   /* = total / world in the rotated case */
}

LUPIN_TILES_FN uint32_t lupin_owned_tile(uint32_t j, uint32_t rank, uint32_t world, uint32_t tiles_x)
{
    if (world > 1u && tiles_x % world == 0u)
    {
        const uint32_t per_row = tiles_x / world;
        const uint32_t ty = j / per_row, i = j % per_row;
        const uint32_t first = (rank + world - ty % world) % world;   /* (rank - ty) mod world */
        return ty * tiles_x + first + i * world;
    }
    return rank + j * world;
}

/* inverse of lupin_owned_tile: which of its owner's tiles (row-major enumeration) is tile t */
LUPIN_TILES_FN uint32_t lupin_owned_index(uint32_t t, uint32_t world, uint32_t tiles_x)
{
    if (world > 1u && tiles_x % world == 0u)
    {
        const uint32_t per_row = tiles_x / world;
        return (t / tiles_x) * per_row + (t % tiles_x) / world;   /* the row's owned columns are first, first + world, ... */
    }
    return t / world;
}

#endif /* LUPIN_TILES_H */
