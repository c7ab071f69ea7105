"""Prototype (numpy / Python ints, CPU) of the mask-parallel legal-placement search that csrc/trl_device.h's
movegen_closure_wave implements on the device, checked against the reference-generated fixtures.

The reference's 'convolutional' generator (move_generation.py:325-488) is a FIFO walk, but for every piece other than T its
RESULT does not depend on the walk's order: the visited set is the closure of the spawn cell under {left, right, down} moves
inside a rotation state and under kick edges from edge cells (a static graph), and the placements are its cells that cannot
move down.  For T the order only matters for cells that receive flagged arrivals BOTH through a last kick and through another
kick ("mixed" cells: the last flagged writer decides the sub-plane, move_generation.py:671-677); a T search without such a
cell is order-free too.  This script states the closure on 64-bit COLUMN masks (bit = map row), one (rotation, map column)
per lane as on the device, and reports (1) equality with the fixtures' legal lists, (2) how often a T search needs the
ordered walk.

usage: python scripts/bfs_closure_proto.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

PZ, PL, PO, PS, PI, PJ, PT, P_NONE = range(8)
MINO = {  # (type, rot) -> 4 x (cx, cy)   (trl_tables.h C_MINO)
    PZ: [[(0,0),(1,0),(1,1),(2,1)], [(1,1),(1,2),(2,0),(2,1)], [(0,1),(1,1),(1,2),(2,2)], [(0,1),(0,2),(1,0),(1,1)]],
    PL: [[(0,1),(1,1),(2,0),(2,1)], [(1,0),(1,1),(1,2),(2,2)], [(0,1),(0,2),(1,1),(2,1)], [(0,0),(1,0),(1,1),(1,2)]],
    PO: [[(0,0),(0,1),(1,0),(1,1)]] * 4,
    PS: [[(0,1),(1,0),(1,1),(2,0)], [(1,0),(1,1),(2,1),(2,2)], [(0,2),(1,1),(1,2),(2,1)], [(0,0),(0,1),(1,1),(1,2)]],
    PI: [[(0,1),(1,1),(2,1),(3,1)], [(2,0),(2,1),(2,2),(2,3)], [(0,2),(1,2),(2,2),(3,2)], [(1,0),(1,1),(1,2),(1,3)]],
    PJ: [[(0,0),(0,1),(1,1),(2,1)], [(1,0),(1,1),(1,2),(2,0)], [(0,1),(1,1),(2,1),(2,2)], [(0,2),(1,0),(1,1),(1,2)]],
    PT: [[(0,1),(1,0),(1,1),(2,1)], [(1,0),(1,1),(1,2),(2,1)], [(0,1),(1,1),(1,2),(2,1)], [(0,1),(1,0),(1,1),(1,2)]],
}
KICK = [  # [(from * 3 + dir - 1)] -> list of (kx, ky)     (trl_tables.h C_KICK)
    [(0,0),(-1,0),(-1,1),(0,-2),(-1,-2)], [(0,0),(0,1),(1,1),(-1,1),(1,0),(-1,0)], [(0,0),(1,0),(1,1),(0,-2),(1,-2)],
    [(0,0),(1,0),(1,-1),(0,2),(1,2)], [(0,0),(1,0),(1,2),(1,1),(0,2),(0,1)], [(0,0),(1,0),(1,-1),(0,2),(1,2)],
    [(0,0),(1,0),(1,1),(0,-2),(1,-2)], [(0,0),(0,-1),(-1,-1),(1,-1),(-1,0),(1,0)], [(0,0),(-1,0),(-1,1),(0,-2),(-1,-2)],
    [(0,0),(-1,0),(-1,-1),(0,2),(-1,2)], [(0,0),(-1,0),(-1,2),(-1,1),(0,2),(0,1)], [(0,0),(-1,0),(-1,-1),(0,2),(-1,2)],
]
IKICK = [
    [(0,0),(-2,0),(1,0),(-2,-1),(1,2)], [(0,0),(0,1)], [(0,0),(-1,0),(2,0),(-1,2),(2,-1)],
    [(0,0),(-1,0),(2,0),(-1,2),(2,-1)], [(0,0),(1,0)], [(0,0),(2,0),(-1,0),(2,1),(-1,-2)],
    [(0,0),(2,0),(-1,0),(2,1),(-1,-2)], [(0,0),(0,-1)], [(0,0),(1,0),(-2,0),(1,-2),(-2,1)],
    [(0,0),(1,0),(-2,0),(1,-2),(-2,1)], [(0,0),(-1,0)], [(0,0),(-2,0),(1,0),(-2,-1),(1,2)],
]
PLANE_BASE = [1, 7, 0, 3, 5, 11, 15]
NROT = [2, 4, 1, 2, 2, 4, 4]
BOX = [3, 3, 2, 3, 4, 3, 3]
MAP_H, MAP_W, SPAWN_Y = 44, 14, 17
M44 = (1 << MAP_H) - 1


def down_fill(m, seed):
    """cells of m reachable from seed by moving to higher bits (= down the board) through m; seed is a subset of m"""
    return ((((m + seed) ^ m) & m) | seed) & M44


def closure_search(rows, ptype, need_spawn):
    """-> (present, flagp, ulp, mixed): [4][MAP_H] row masks as MgScratch holds them, and whether a T cell got flagged
    arrivals of both kinds (the ordered walk is needed then); None if the piece cannot spawn"""
    free = [0] * 10
    for r in range(40):
        for x in range(10):
            if not (rows[r] >> x) & 1:
                free[x] |= 1 << (r + 2)
    fcol = lambda x: free[x] if 0 <= x < 10 else 0
    valid = [[0] * MAP_W for _ in range(4)]                      # valid[rot][mx]: column mask over my
    for rot in range(4):
        for mx in range(MAP_W):
            v = M44
            for cx, cy in MINO[ptype][rot]:
                v &= fcol(mx - 2 + cx) >> cy
            valid[rot][mx] = v & M44
    vget = lambda rot, mx: valid[rot][mx] if 0 <= mx < MAP_W else 0
    occ = [r for r in range(40) if rows[r]]
    highest = occ[0] if occ else 40
    sy = max(highest - BOX[ptype], SPAWN_Y)
    sx = 3 + (1 if ptype == PO else 0)
    if not (valid[0][sx + 2] >> (sy + 2)) & 1:
        return None
    if need_spawn and not (valid[0][sx + 2] >> (SPAWN_Y + 2)) & 1:
        return None
    stuck = [[valid[r][x] & ~(valid[r][x] >> 1) & M44 for x in range(MAP_W)] for r in range(4)]
    edge = [[valid[r][x] & (~(valid[r][x] >> 1) | ~vget(r, x - 1) | ~vget(r, x + 1)) & M44 for x in range(MAP_W)] for r in range(4)]
    ktab = IKICK if ptype == PI else KICK
    # static kick selection: sel[rot][mx][d][ki] = source cells (column mask) whose first fitting kick in direction d is ki
    sel = [[[[0] * 6 for _ in range(3)] for _ in range(MAP_W)] for _ in range(4)]
    if ptype != PO:
        for rot in range(4):
            for mx in range(MAP_W):
                for d in range(3):
                    nrot = (rot + d + 1) & 3
                    remaining = edge[rot][mx]
                    for ki, (kx, ky) in enumerate(ktab[rot * 3 + d]):
                        vt = vget(nrot, mx + kx)
                        fit = (vt << ky) & M44 if ky >= 0 else vt >> (-ky)      # source my <-> target my - ky
                        s = remaining & fit
                        remaining &= ~fit
                        if ky > 0:
                            s &= ~((1 << (ky + 2)) - 1)                         # target my' = my - ky < 2: discarded (new_y < 0)
                        sel[rot][mx][d][ki] = s
    reach = [[0] * MAP_W for _ in range(4)]
    reach[0][sx + 2] = 1 << (sy + 2)
    a_last = [[0] * MAP_W for _ in range(4)]       # flagged arrivals through the last kick of a list (dir != 180)
    a_other = [[0] * MAP_W for _ in range(4)]
    done_src = [[0] * MAP_W for _ in range(4)]
    rounds = iters = 0
    snaps = []                                      # reach after the fixed point of round 0, 1 (generation <= 0, <= 1 cells)
    while True:
        # closure under down / left / right inside each rotation state
        while True:
            changed = False
            iters += 1
            for rot in range(4):
                for mx in range(MAP_W):
                    r = reach[rot][mx]
                    r = down_fill(valid[rot][mx], r)
                    side = ((reach[rot][mx - 1] if mx > 0 else 0) | (reach[rot][mx + 1] if mx + 1 < MAP_W else 0)) & valid[rot][mx]
                    r |= side
                    if r != reach[rot][mx]:
                        changed = True
                    reach[rot][mx] = r
            if not changed:
                break
        if len(snaps) < 2:
            snaps.append([row[:] for row in reach])
        if ptype == PO:
            break
        # kicks from the edge cells that have not sent theirs yet
        rounds += 1
        new_any = False
        add = [[0] * MAP_W for _ in range(4)]
        for rot in range(4):
            for mx in range(MAP_W):
                src = reach[rot][mx] & edge[rot][mx] & ~done_src[rot][mx]
                done_src[rot][mx] |= src
                if not src:
                    continue
                for d in range(3):
                    nrot = (rot + d + 1) & 3
                    kl = ktab[rot * 3 + d]
                    for ki, (kx, ky) in enumerate(kl):
                        moved = src & sel[rot][mx][d][ki]
                        if not moved:
                            continue
                        dest = moved >> ky if ky >= 0 else (moved << (-ky)) & M44
                        tx = mx + kx
                        assert 0 <= tx < MAP_W and not (dest & ~valid[nrot][tx])
                        add[nrot][tx] |= dest
                        if ptype == PT and d != 1 and ki == len(kl) - 1:
                            a_last[nrot][tx] |= dest
                        else:
                            a_other[nrot][tx] |= dest
        for rot in range(4):
            for mx in range(MAP_W):
                n = add[rot][mx] & ~reach[rot][mx]
                if n:
                    new_any = True
                    reach[rot][mx] |= n
        if not new_any:
            break
    resolved = 0
    if ptype == PT and RESOLVE_MIXED:
        resolved = resolve_mixed(valid, edge, sel, ktab, reach, snaps, a_last, a_other, stuck)
    # column form -> the row planes of MgScratch
    present = [[0] * MAP_H for _ in range(4)]
    flagp = [[0] * MAP_H for _ in range(4)]
    ulp = [[0] * MAP_H for _ in range(4)]
    mixed = False
    for rot in range(4):
        for mx in range(MAP_W):
            p = reach[rot][mx] & stuck[rot][mx]
            f1 = a_last[rot][mx] & stuck[rot][mx]
            f0 = a_other[rot][mx] & stuck[rot][mx]
            if f1 & f0:
                mixed = True
            for my in range(MAP_H):
                if (p >> my) & 1:
                    present[rot][my] |= 1 << mx
                if ((f1 | f0) >> my) & 1:
                    flagp[rot][my] |= 1 << mx
                if (f1 >> my) & 1:
                    ulp[rot][my] |= 1 << mx
    return present, flagp, ulp, mixed, rounds, iters, resolved


RESOLVE_MIXED = True


def resolve_mixed(valid, edge, sel, ktab, reach, snaps, a_last, a_other, stuck):
    """Mixed T cells whose winner follows from the first two generations of the walk, without walking: the walk's FIFO order
    is the order of the pushes, and as long as every rotation state's generation-1 cells are flooded by ONE queue item (the
    first one pushed into that state from the spawn region) the time of every flagged write is a tuple of map coordinates:
      * arrival from a generation-0 source (spawn region, rotation 0) on a cell first visited in generation 1: written when
        the pushed item is popped: (1, key0(source), 0)                     key0 = (row, column, direction) of the source
      * arrival from a generation-1 source s on a cell that has been visited by then (cells of the spawn region, or of a
        rotation state whose item was popped before s's): written while s's state is flooded: (1, seedkey(rot s), 1, key(s))
      * ... on a cell not visited yet: pushed, written in generation 2: (2, seedkey(rot s), key(s))
    The last write wins (move_generation.py:671-677).  Clears the losing class's bit of every cell it settles; returns the
    number of cells settled, -1 if some mixed cell is outside these cases (deeper generations, several floods per state):
    the ordered walk is needed then."""
    r0 = snaps[0]
    r1 = snaps[1] if len(snaps) > 1 else reach
    mixed_cells = [(rot, mx, my) for rot in range(4) for mx in range(MAP_W) for my in range(MAP_H)
                   if ((a_last[rot][mx] & a_other[rot][mx] & stuck[rot][mx]) >> my) & 1]
    if not mixed_cells:
        return 0
    # the first push into each rotation state from the spawn region (rotation 0): sources in (row, column, direction) order
    seedkey, seedcell = {}, {}
    for d in range(3):
        nrot = d + 1
        best = None
        for mx in range(MAP_W):
            src = r0[0][mx] & edge[0][mx]
            for ki, (kx, ky) in enumerate(ktab[d]):
                moved = src & sel[0][mx][d][ki]
                if moved:
                    my = (moved & -moved).bit_length() - 1
                    cand = ((my << 6) | (mx << 2) | d, (nrot, mx + kx, my - ky))
                    if best is None or cand[0] < best[0]:
                        best = cand
        if best is not None:
            seedkey[nrot], seedcell[nrot] = best
    # one flood per rotation state in generation 1?
    dm = [[0] * MAP_W for _ in range(4)]
    for nrot, (_, tx, ty) in seedcell.items():
        dm[nrot][tx] = 1 << ty
    while True:
        changed = False
        for rot in range(1, 4):
            for mx in range(MAP_W):
                r = down_fill(valid[rot][mx], dm[rot][mx])
                r |= ((dm[rot][mx - 1] if mx > 0 else 0) | (dm[rot][mx + 1] if mx + 1 < MAP_W else 0)) & valid[rot][mx]
                if r != dm[rot][mx]:
                    changed = True
                dm[rot][mx] = r
        if not changed:
            break
    # (a state's generation-1 cells may have more pieces than this first flood -- pockets that later items reach; the rule
    # below needs only the cells it looks at, the mixed cell and its sources, to lie in the first flood)
    settled = 0
    for rK, xK, yK in mixed_cells:
        inK = ((r0[0][xK] if rK == 0 else dm[rK][xK]) >> yK) & 1
        if not inK:
            return -1
        best = None
        for rs in range(4):
            if rs == rK:
                continue
            d = (rK - rs - 1) & 3
            kl = ktab[rs * 3 + d]
            for ki, (kx, ky) in enumerate(kl):
                sx, sy = xK - kx, yK + ky
                if not (0 <= sx < MAP_W and 0 <= sy < MAP_H):
                    continue
                if not ((reach[rs][sx] & edge[rs][sx] & sel[rs][sx][d][ki]) >> sy) & 1:
                    continue
                ulk = d != 1 and ki == len(kl) - 1
                skey = (sy << 6) | (sx << 2) | d
                if rs == 0:
                    if not (r0[0][sx] >> sy) & 1:
                        return -1
                    t = (1 << 40) | (skey << 21)
                else:
                    if not (dm[rs][sx] >> sy) & 1:
                        return -1
                    visited = rK == 0 or seedkey[rK] < seedkey[rs]
                    t = ((1 << 40) | (seedkey[rs] << 21) | (1 << 20) | skey) if visited else ((2 << 40) | (seedkey[rs] << 21) | skey)
                if best is None or t > best[0]:
                    best = (t, ulk)
        assert best is not None
        if best[1]:
            a_other[rK][xK] &= ~(1 << yK)
        else:
            a_last[rK][xK] &= ~(1 << yK)
        settled += 1
    return settled


def placements_to_moves(present, flagp, ulp, ptype):
    """csrc/trl_device.h placements_to_moves"""
    out = []
    isT = ptype == PT
    zsi = ptype in (PZ, PS, PI)
    for ts in range(3 if isT else 1):
        for ri in range(NROT[ptype]):
            plane = PLANE_BASE[ptype] + ri + 4 * ts
            for row in range(39):
                def src(o, my):
                    p = present[o][my]
                    if not isT:
                        return p
                    f, u = flagp[o][my], ulp[o][my]
                    return (p & ~f) if ts == 0 else ((f & ~u) if ts == 1 else (f & u))
                m = src(ri, row + 2)
                if zsi:
                    if ri == 0:
                        if row >= 1:
                            m |= src(2, row + 1)
                    else:
                        m |= src(3, row + 2) >> 1
                m &= 0x7FF
                for c in range(11):
                    if (m >> c) & 1:
                        out.append((plane << 10) | (row << 4) | c)
    return out


def legal_for_state(st):
    turn = int(st[312])
    p = st[144 * turn:144 * (turn + 1)]
    rows = np.frombuffer(bytes(p[:80]), dtype="<u2").astype(int).tolist()
    piece, held, qlen = int(p[137]), int(p[138]), int(p[139])
    q0 = int(p[112]) & 15
    p2 = held if held != P_NONE else (q0 if qlen > 0 else P_NONE)
    types = [(piece, False)] + ([(p2, True)] if p2 != piece else [])
    moves, info = [], []
    for t, need_spawn in types:
        if t == P_NONE:
            continue
        r = closure_search(rows, t, need_spawn)
        if r is None:
            continue
        present, flagp, ulp, mixed, rounds, iters, resolved = r
        info.append((t, mixed, rounds, iters, resolved))
        moves += placements_to_moves(present, flagp, ulp, t)
    return sorted(moves), info


def main():
    d = np.load(os.path.join(GOLD, "env_traces.npz"))
    off = np.concatenate([[0], np.cumsum(d["legal_counts"])])
    n = len(d["states"])
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    bad = t_searches = t_mixed = searches = t_resolved = 0
    rounds_hist, iters_hist = [], []
    for i in range(0, n, step):
        st = d["states"][i]
        if st[141] or st[144 + 141]:
            continue
        want = d["legal_flat"][off[i]:off[i + 1]].tolist()
        got, info = legal_for_state(st)
        has_mixed = any(m for _, m, _, _, _ in info)
        for t, m, r, it, rs in info:
            t_resolved += int(rs > 0 and not m)
            searches += 1
            rounds_hist.append(r); iters_hist.append(it)
            if t == PT:
                t_searches += 1
                t_mixed += int(m)
        if got != want and not has_mixed:
            bad += 1
            print("MISMATCH at state", i, len(got), len(want))
    print("env_traces: %d states checked (every %d-th), %d searches, %d mismatches outside mixed T searches; "
          "T searches %d, of them left to the ordered walk %d (%.1f %%), settled by resolve_mixed %d" %
          (len(range(0, n, step)), step, searches, bad, t_searches, t_mixed, 100.0 * t_mixed / max(t_searches, 1), t_resolved))
    print("kick rounds per search: mean %.2f max %d; closure iterations per search: mean %.1f max %d" %
          (np.mean(rounds_hist), max(rounds_hist), np.mean(iters_hist), max(iters_hist)))
    # the T fixtures: positions on which the reference's two generators disagree (complex T boards) and 120 where they agree
    d = np.load(os.path.join(GOLD, "movegen_t.npz"))
    offc = np.concatenate([[0], np.cumsum(d["conv_counts"])])
    bad = mixed = settled = 0
    for i in range(len(d["rows"])):
        rows = d["rows"][i].astype(int).tolist()
        piece, held, q0 = int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i])
        p2 = held if held != P_NONE else q0
        moves, anym = [], False
        for t, need in [(piece, False)] + ([(p2, True)] if p2 != piece else []):
            r = closure_search(rows, t, need)
            if r is None:
                continue
            anym |= r[3]
            settled += int(r[6] > 0 and not r[3])
            moves += placements_to_moves(r[0], r[1], r[2], t)
        mixed += int(anym)
        if sorted(moves) != d["conv_flat"][offc[i]:offc[i + 1]].tolist() and not anym:
            bad += 1
            print("MISMATCH in movegen_t case", i)
    print("movegen_t: %d cases, %d mismatches outside the searches left to the ordered walk, %d left to it (%.1f %%), %d searches "
          "settled by resolve_mixed" % (len(d["rows"]), bad, mixed, 100.0 * mixed / len(d["rows"]), settled))


if __name__ == "__main__":
    main()
