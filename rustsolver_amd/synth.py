"""Host mirror of the device-side synthetic fills (csrc/rs_kernels.hip k_fill_random / k_fill_uniform):
pure functions of (seed, cell index), so tests can reproduce on the host what the GPU generated."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + _GOLD
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _hash(seed, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) ^ (idx * _GOLD))


def fill_values(seed, idx, lo, hi):
    """value of cell `idx` (global element index within the table array) for k_fill_random"""
    span = np.uint64(hi - lo + 1)
    return (np.int64(lo) + (_hash(seed, idx) % span).astype(np.int64))


def table_node_values(table, node, seed, lo, hi, ssum=False):
    """[A][lanes] int64 values rs_table_fill_random wrote for `node` (regrets, or strategy_sum with ssum=True)"""
    d = table.node_desc(node)
    pitch, off, lanes = table.pitch(node), table.cell_offset(node), table.lanes(node)
    s = (seed ^ 0x5353554D) if ssum else seed
    T = table.tile_lanes(node)   # block layout [pitch / T][A][T] (T == pitch: plain rows)
    lane = np.arange(lanes, dtype=np.uint64)[None, :]
    a = np.arange(d.n_actions, dtype=np.uint64)[:, None]
    idx = off + ((lane // np.uint64(T)) * np.uint64(d.n_actions) + a) * np.uint64(T) + lane % np.uint64(T)
    return fill_values(s, idx, lo, hi)


def table_lane_values(table, node, seed, lo, hi, lanes, ssum=False):
    """[A][len(lanes)] int64 values rs_table_fill_random wrote for the given lanes of `node`: table_node_values for a lane subset, for tables far too
    big to mirror as a whole (config 3: 11.76 M lanes per river node)"""
    d = table.node_desc(node)
    off = table.cell_offset(node)
    s = (seed ^ 0x5353554D) if ssum else seed
    T = np.uint64(table.tile_lanes(node))
    lane = np.asarray(lanes, dtype=np.uint64)[None, :]
    a = np.arange(d.n_actions, dtype=np.uint64)[:, None]
    idx = np.uint64(off) + ((lane // T) * np.uint64(d.n_actions) + a) * T + lane % T
    return fill_values(s, idx, lo, hi)


def uniform_f32(seed, n, lo, hi):
    """k_fill_uniform: lo + (hi-lo) * u, u = top 24 bits of the hash * 2^-24, f32 mul then add"""
    h = _hash(seed, np.arange(n, dtype=np.uint64))
    u = (h >> np.uint64(40)).astype(np.uint32).astype(np.float32) * np.float32(5.9604644775390625e-08)
    width = np.float32(np.float32(hi) - np.float32(lo))
    return (np.float32(lo) + (width * u).astype(np.float32)).astype(np.float32)
