#!/usr/bin/env python3
"""Golden fixture for SURVEY.md H1 at the depth section 8(c) item 2 asks for: >= 200 T-piece positions on which the
reference's 'convolutional' and 'brute-force' generators (move_generation.py:325-488 / 204-243) DISAGREE (the T-spin
sub-plane of a placement depends on the order in which rotation-flagged arrivals reach it), plus positions where they
agree, each with both legal lists.  Positions: random cave-like stacks (where kicks reach into overhangs) with T as the
active piece or as the piece a hold would bring.

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_movegen_t.py  ->  tests/golden/movegen_t.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (imports the reference; runs no generation)
import ref_harness as rh  # noqa: E402

WANT_DIFF, WANT_SAME, MAX_TRY = 240, 120, 200000


def main():
    rs = np.random.RandomState(20261004)
    rows, piece, held, q0, conv, brute, differ = [], [], [], [], [], [], []
    n_diff = n_same = 0
    for t in range(MAX_TRY):
        if n_diff >= WANT_DIFF and n_same >= WANT_SAME:
            break
        g, rng = gg.new_game("s2" if t % 2 else "s1", 7000 + t)
        pl = g.players[0]
        h = rs.randint(3, 14)
        dens = rs.uniform(0.5, 0.92)
        for r in range(40 - h, 40):
            row = rs.rand(10) < dens
            if row.all():
                row[rs.randint(10)] = False
            for c in range(10):
                pl.board.grid[r][c] = 1 if row[c] else 0
        other = gg.MINOS[rs.randint(6)]                       # any letter but T ("ZLOSIJ")
        if t % 3 == 0:                                        # T comes out of hold: the generator's second piece type
            pl.piece = gg.Piece(type=other); pl.piece.move_to_spawn()
            pl.held_piece = "T"
        else:
            pl.piece = gg.Piece(type="T"); pl.piece.move_to_spawn()
            pl.held_piece = other if t % 3 == 1 else None     # None: a hold would bring the next piece
        a = rh.mask_to_moves(gg.get_move_matrix(pl, "convolutional"))
        if not a:
            continue
        b = rh.mask_to_moves(gg.get_move_matrix(pl, "brute-force"))
        d = a != b
        if (d and n_diff >= WANT_DIFF) or (not d and n_same >= WANT_SAME):
            continue
        n_diff += d; n_same += (not d)
        rows.append(gg.rows_of(pl.board.grid))
        piece.append(gg.MINOS.index(pl.piece.type))
        held.append(gg.MINOS.index(pl.held_piece) if pl.held_piece else rh.NO_PIECE)
        q0.append(gg.MINOS.index(pl.queue.pieces[0]) if pl.queue.pieces else rh.NO_PIECE)
        conv.append(a); brute.append(b); differ.append(int(d))
        if d and n_diff % 40 == 0:
            print("tried %d: %d differing, %d agreeing" % (t + 1, n_diff, n_same), flush=True)
    cc, cf = gg.ragged(conv)
    bc, bf = gg.ragged(brute)
    np.savez_compressed(os.path.join(gg.OUT, "movegen_t.npz"), rows=np.array(rows, dtype=np.uint16),
                        piece=np.array(piece, dtype=np.uint8), held=np.array(held, dtype=np.uint8), q0=np.array(q0, dtype=np.uint8),
                        conv_counts=cc, conv_flat=cf, brute_counts=bc, brute_flat=bf, differ=np.array(differ, dtype=np.uint8))
    print("movegen_t: %d cases, conv != brute on %d" % (len(conv), n_diff))


if __name__ == "__main__":
    main()
