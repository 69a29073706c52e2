"""Tile schedule of the chained in-place passes -- numpy reference of the builder in csrc/cwr_engine.hip (build_chain_schedule).

The tiled J^2 pass is a persistent grid: every block walks a list of 64-row tiles.  Walking them in the engine's default order
and ping-ponging between two vectors is a block-Jacobi iteration between tiles.  Here the tiles are linked into CHAINS along
the flow of one time level -- tile t -> the neighbour tile u that takes most of t's outflow, kept when u's largest inflow
comes from t -- and every block walks chains, in place: a tile then reads what its upstream neighbour of the same chain has
just written (block Gauss-Seidel along the flow; tests/models/chain_gs_probe.py: 24 -> 17 passes at CFL 2.5, 123 -> 47 at
CFL 25 with four tile-local applications).  No block ever waits for another: the order only decides how fresh the values a
tile reads are, so a schedule built for another flow direction costs passes, never correctness.

The product path builds the same schedule in C++; this module is the specification the tests compare it with."""
from __future__ import annotations

import numpy as np

N_XCD = 8


def tile_links(face1, face2, flow, n_rows: int, tile_rows: int, ntiles: int):
    """(src tile, dst tile, flux) of every ordered pair of distinct tiles with flow between them: flux = sum of |face_flow|
    over the faces whose flow leaves a cell of src for a cell of dst (flow > 0: face1 -> face2)."""
    f1 = np.asarray(face1, dtype=np.int64)
    f2 = np.asarray(face2, dtype=np.int64)
    a = np.asarray(flow, dtype=np.float64)
    ok = (f1 < n_rows) & (f2 < n_rows) & (a != 0)
    src = np.where(a > 0, f1, f2)[ok] // tile_rows
    dst = np.where(a > 0, f2, f1)[ok] // tile_rows
    w = np.abs(a[ok])
    m = src != dst
    key = src[m] * ntiles + dst[m]
    uk, inv = np.unique(key, return_inverse=True)
    # (float32, as the engine's k_link_flux stores them: the comparisons below then see the same numbers)
    return uk // ntiles, uk % ntiles, np.bincount(inv, weights=w[m], minlength=len(uk)).astype(np.float32).astype(np.float64)


def chains(us, ud, w, ntiles: int):
    """Tiles linked t -> next[t] where next[t] takes most of t's outflow AND gets most of its inflow from t; returns the
    chains (lists of tile ids) ordered by their first tile.  Ties go to the smaller tile id (deterministic)."""
    best_dn = np.full(ntiles, -1, dtype=np.int64)
    best_up = np.full(ntiles, -1, dtype=np.int64)
    o = np.lexsort((ud, -w, us))                       # per source: largest flux first, then the smaller destination
    first = np.ones(len(o), bool); first[1:] = us[o][1:] != us[o][:-1]
    best_dn[us[o][first]] = ud[o][first]
    o = np.lexsort((us, -w, ud))
    first = np.ones(len(o), bool); first[1:] = ud[o][1:] != ud[o][:-1]
    best_up[ud[o][first]] = us[o][first]
    nxt = np.full(ntiles, -1, dtype=np.int64)
    t = np.arange(ntiles)
    has = best_dn >= 0
    mutual = has & (best_up[np.maximum(best_dn, 0)] == t)
    nxt[mutual] = best_dn[mutual]
    has_prev = np.zeros(ntiles, bool)
    has_prev[nxt[nxt >= 0]] = True
    out, seen = [], np.zeros(ntiles, bool)
    for start in list(np.nonzero(~has_prev)[0]) + list(range(ntiles)):       # heads first, then whatever sits on a cycle
        if seen[start]:
            continue
        ch, c = [], int(start)
        while c >= 0 and not seen[c]:
            seen[c] = True
            ch.append(c)
            c = int(nxt[c])
        out.append(ch)
    out.sort(key=lambda ch: ch[0])
    return out


def schedule(chain_list, ntiles: int, grid: int, streams_per_block: int = 2) -> np.ndarray:
    """(depth, grid) int32, -1 padded.  The chains, in the order of their first tile (= along the cell curve: an XCD keeps a
    compact region), are concatenated and cut into grid * streams_per_block consecutive STREAMS of equal length (+-1); block
    b = lidx * 8 + xcd walks the streams (xcd * (grid / 8) + lidx) * spb .. + spb - 1 INTERLEAVED (A1 B1 A2 B2 ...): the
    kernel prefetches a tile's x rows one tile ahead, so a tile's chain successor must come two slots later to see its
    results.  With the engine's column reuse (the default) a successor takes the rows it shares with its predecessor from LDS and
    simply comes next: streams_per_block = 1."""
    seq = np.concatenate([np.asarray(c, dtype=np.int64) for c in chain_list]) if chain_list else np.zeros(0, np.int64)
    assert len(seq) == ntiles
    ns = grid * streams_per_block
    bounds = (np.arange(ns + 1, dtype=np.int64) * ntiles) // ns
    longest = int(np.max(np.diff(bounds))) if ns else 0
    depth = longest * streams_per_block
    bpx = grid // N_XCD
    out = np.full((depth, grid), -1, dtype=np.int32)
    for b in range(grid):
        xcd, lidx = b % N_XCD, b // N_XCD
        s0 = (xcd * bpx + lidx) * streams_per_block
        parts = [seq[bounds[s0 + q]:bounds[s0 + q + 1]] for q in range(streams_per_block)]
        lst = []
        for i in range(longest):
            for p in parts:
                if i < len(p):
                    lst.append(int(p[i]))
        out[:len(lst), b] = lst
    return out


def chain_schedule(face1, face2, flow, n_rows: int, tile_rows: int, ntiles: int, grid: int, streams_per_block: int = 2):
    us, ud, w = tile_links(face1, face2, flow, n_rows, tile_rows, ntiles)
    return schedule(chains(us, ud, w, ntiles), ntiles, grid, streams_per_block)
