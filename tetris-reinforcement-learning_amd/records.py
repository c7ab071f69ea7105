"""Training records in the reference's on-disk layout (host side, SURVEY.md section 8 rows D1, N1 and
"next" row 1).

A record is the 13-field list play_game produces (ai.py:1613-1666, 1675-1695):
    [grid_a, pieces_a, b2b_a, combo_a, garbage_a, grid_o, pieces_o, b2b_o, combo_o, garbage_o,
     color, z, pi]
with the active player first (game_to_X, ai.py:1399-1413), pi = search_statistics
(ai.py:1330-1361) and, when Config.augment_data, four mirror variants per saved move
(ai.py:1423-1516).  The engine hands over a saved move as (canonical 320-byte root state, root
children's encoded moves, their visit counts); everything here is plain Python on those.
Element types (float 0.0/1.0 vs int 0/1) follow the reference so json.dumps output is identical.
"""
import struct

import numpy as np

MINOS = "ZLOSIJT"
ROWS, COLS, PREVIEWS = 40, 10, 5
POLICY_SHAPE = (27, 39, 11)
NO_PIECE = 7
_PLAYER_FMT = "<40H4QQiiihhBBBBBB2x"

# plane -> (piece letter, rotation, t-spin index)   (const.py:82-118)
_PLANE_PIECE = ([("O", 0, 0)] + [("Z", r, 0) for r in range(2)] + [("S", r, 0) for r in range(2)] +
                [("I", r, 0) for r in range(2)] + [("L", r, 0) for r in range(4)] + [("J", r, 0) for r in range(4)] +
                [("T", r, t) for t in range(3) for r in range(4)])
_PLANE_BASE = {"O": 0, "Z": 1, "S": 3, "I": 5, "L": 7, "J": 11, "T": 15}
_BOX = {"Z": 3, "L": 3, "O": 2, "S": 3, "I": 4, "J": 3, "T": 3}            # len(piece_dict[piece])


def _plane_of(piece, rotation, tspin):
    return _PLANE_BASE[piece] + rotation + (4 * tspin if piece == "T" else 0)


def unpack_player(buf):
    f = struct.unpack(_PLAYER_FMT, bytes(buf))
    rows, garb, q = f[0:40], f[40:44], f[44]
    pieces, lsent, lcleared, b2b, combo, lvl, piece, held, ql, gl, over = f[45:56]
    return dict(rows=rows, garbage=[(garb[k >> 4] >> (4 * (k & 15))) & 15 for k in range(gl)],
                queue=[(q >> (4 * k)) & 15 for k in range(ql)], pieces=pieces, lines_sent=lsent,
                lines_cleared=lcleared, b2b=b2b, combo=combo, b2b_level=lvl, piece=piece, held=held,
                game_over=bool(over))


def unpack_state(state):
    state = bytes(np.asarray(state, dtype=np.uint8).tobytes())
    tail = struct.unpack("<IIIIIIBBB5x", state[288:320])
    return dict(players=[unpack_player(state[0:144]), unpack_player(state[144:288])], stream=tail[0],
                bag_ctr=tail[1], garb_ctr=tail[2], ply=tail[3], turn=tail[6], flags=tail[7], ruleset=tail[8])


def _grid(player):
    """simplify_grid (ai.py:1364-1366): float32 0/1 array"""
    g = np.zeros((ROWS, COLS), dtype=np.float32)
    for r in range(ROWS):
        bits = player["rows"][r]
        for c in range(COLS):
            if (bits >> c) & 1:
                g[r, c] = 1
    return g


def _pieces(player):
    """get_pieces (ai.py:1379-1389): row 0 active, row 1 held, rows 2..6 previews; one-hot over ZLOSIJT"""
    t = np.zeros((2 + PREVIEWS, len(MINOS)), dtype=np.float32)
    if player["piece"] != NO_PIECE:
        t[0][player["piece"]] = 1
    if player["held"] != NO_PIECE:
        t[1][player["held"]] = 1
    for j, p in enumerate(player["queue"][:PREVIEWS]):
        t[j + 2][p] = 1
    return t


def game_to_x(state):
    """game_to_X (ai.py:1399-1413) from a canonical state: active player's features first"""
    s = unpack_state(state) if not isinstance(state, dict) else state
    a, o = s["players"][s["turn"]], s["players"][1 - s["turn"]]
    return (_grid(a), _pieces(a), a["b2b"], a["combo"], len(a["garbage"]),
            _grid(o), _pieces(o), o["b2b"], o["combo"], len(o["garbage"]),
            s["turn"])                       # color of the side to move == its index (player.py:253-261)


def search_statistics(child_moves, child_visits):
    """pi target (ai.py:1330-1361): round(N_child / sum N, 4) scattered into [27][39][11]"""
    pi = np.zeros(POLICY_SHAPE, dtype=int).tolist()
    total = int(sum(int(v) for v in child_visits))
    assert total != 0
    for m, n in zip(child_moves, child_visits):
        n = int(n)
        if n != 0:
            m = int(m)
            pi[(m >> 10) & 31][(m >> 4) & 63][m & 15] = round(n / total, 4)
    return pi


# ---- mirror augmentation (ai.py:1423-1516) ----------------------------------------------------
def reflect_grid(grid):
    if isinstance(grid, np.ndarray):
        return np.fliplr(grid).tolist()
    return [row[::-1] for row in grid]


_PIECE_MIRROR = {0: 3, 3: 0, 1: 5, 5: 1}          # Z<->S, L<->J (indices into ZLOSIJT)


def reflect_pieces(piece_table):
    out = np.zeros((2 + PREVIEWS, len(MINOS)), dtype=int)
    for i, row in enumerate(piece_table):
        if 1 in row:
            idx = row.tolist().index(1)
            out[i][_PIECE_MIRROR.get(idx, idx)] = 1
    return out


def reflect_policy(pi):
    out = np.zeros(POLICY_SHAPE, dtype=int).tolist()
    letter_mirror = {"Z": "S", "S": "Z", "L": "J", "J": "L"}
    for plane in range(POLICY_SHAPE[0]):
        piece, rotation, tspin = _PLANE_PIECE[plane]
        size = _BOX[piece]
        new_piece = letter_mirror.get(piece, piece)
        new_rot = {1: 3, 3: 1}.get(rotation, rotation)
        adjust = 0
        if new_piece in ("Z", "S", "I") and new_rot == 3:
            adjust, new_rot = -1, 1               # rotation 3 is stored as rotation 1 shifted one column
        new_plane = _plane_of(new_piece, new_rot, tspin)
        for col in range(POLICY_SHAPE[2]):
            for row in range(POLICY_SHAPE[1]):
                v = pi[plane][row][col]
                if v > 0:
                    new_col = (10 - (col - 2) - size) + 2 + adjust
                    out[new_plane][row][new_col] = v
    return out


def records_for_move(state, child_moves, child_visits, augment=True):
    """The record(s) play_game appends for one saved move, without z (ai.py:1613-1666)."""
    pi = search_statistics(child_moves, child_visits)
    move_data = [*game_to_x(state)]
    out = []
    if augment:
        rpi = reflect_policy(pi)
        for active in range(2):
            for other in range(2):
                rec = []
                for f in move_data:
                    rec.append(f.copy() if isinstance(f, np.ndarray) else f)
                if active == 1:
                    rec[0] = reflect_grid(rec[0]); rec[1] = reflect_pieces(rec[1])
                if other == 1:
                    rec[5] = reflect_grid(rec[5]); rec[6] = reflect_pieces(rec[6])
                rec = [x.tolist() if isinstance(x, np.ndarray) else x for x in rec]
                rec.append(pi if active == 0 else rpi)
                out.append(rec)
    else:
        rec = [x.tolist() if isinstance(x, np.ndarray) else x for x in move_data]
        rec.append(pi)
        out.append(rec)
    return out


def finish_game(game_data, winner, value_min=0, value_mid=0.5, value_max=1):
    """z insertion + flattening at the end of play_game (ai.py:1675-1695).
    game_data = [records of player 0, records of player 1]; returns player 0's then player 1's."""
    for player in range(2):
        z = value_mid if winner == -1 else (value_max if winner == player else value_min)
        for rec in game_data[player]:
            rec.insert(-1, z)
    return game_data[0] + game_data[1]
