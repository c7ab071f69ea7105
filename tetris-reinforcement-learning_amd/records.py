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
# Everything is a permutation: of the board columns, of the piece letters (Z<->S, L<->J), and of the 11 583 policy
# cells (mirror_index(), one precomputed table that the text writer and the packed sets use as well).
_PIECE_MIRROR = {0: 3, 3: 0, 1: 5, 5: 1}          # Z<->S, L<->J (indices into ZLOSIJT)
_LETTER_PERM = np.array([_PIECE_MIRROR.get(i, i) for i in range(len(MINOS))])
_mirror_table = None


def mirror_index():
    """flat policy index -> flat index of its mirror image, -1 where the image leaves the table.  A placement of piece P
    in rotation r whose box (side b) starts at board column c becomes the mirrored letter in the mirrored rotation
    (1 <-> 3) at column 10 - c - b; Z / S / I store rotation 3 as rotation 1 one column to the left (const.py:82-118,
    move_generation.py:698-704), hence the -1 for those."""
    global _mirror_table
    if _mirror_table is None:
        n_plane, n_row, n_col = POLICY_SHAPE
        table = np.full((n_plane, n_row, n_col), -1, dtype=np.int64)
        cols = np.arange(n_col)
        for plane, (piece, rotation, tspin) in enumerate(_PLANE_PIECE):
            image = MINOS[_PIECE_MIRROR.get(MINOS.index(piece), MINOS.index(piece))]
            rot = (4 - rotation) % 4                                  # 0, 3, 2, 1
            shift = 0
            if image in "ZSI" and rot == 3:
                rot, shift = 1, -1
            new_cols = (COLS - (cols - 2) - _BOX[piece]) + 2 + shift
            ok = (new_cols >= 0) & (new_cols < n_col)
            base = _plane_of(image, rot, tspin) * n_row
            for row in range(n_row):
                table[plane, row, cols[ok]] = (base + row) * n_col + new_cols[ok]
        _mirror_table = table.reshape(-1)
    return _mirror_table


def reflect_grid(grid):
    """columns reversed; element types kept (a float32 array gives floats, a list its own elements)"""
    if isinstance(grid, np.ndarray):
        return grid[:, ::-1].tolist()
    return [list(reversed(row)) for row in grid]


def reflect_pieces(piece_table):
    """one-hot rows over ZLOSIJT with the letters swapped; integer typed like the reference's (np.zeros(dtype=int))"""
    return (np.asarray(piece_table)[:, _LETTER_PERM] != 0).astype(int)


def reflect_policy(pi):
    """every positive cell of the [27][39][11] table moved to its mirror image; integer zeros elsewhere"""
    n_plane, n_row, n_col = POLICY_SHAPE
    flat = [v for plane in pi for row in plane for v in row]
    out = [0] * len(flat)
    table = mirror_index()
    for k, v in enumerate(flat):
        if v > 0:
            if table[k] < 0:
                raise IndexError("mirror image of a policy cell leaves the table")
            out[table[k]] = v
    return [[out[(pl * n_row + r) * n_col:(pl * n_row + r + 1) * n_col] for r in range(n_row)] for pl in range(n_plane)]


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


# ---- fast path: the same records as JSON text, without building Python lists --------------------
# A record of the reference's layout is ~12 400 numbers, almost all of them zeros of the policy target; building it
# as nested lists and running json.dumps over them costs ~1 ms per record on the host -- six times the device time
# of the search that produced it.  The writer below patches precomputed text templates instead (same bytes as
# json.dumps of records_for_move(...) + z, pinned by tests/test_records.py against the reference's sha256 fixtures).
class _TextTemplates:
    def __init__(self):
        import json

        def template(arr):
            """text of json.dumps(arr.tolist()) as a uint8 array + the offset of every element's first character"""
            txt = json.dumps(arr.tolist()).encode()
            buf = np.frombuffer(txt, dtype=np.uint8)
            is_num = (buf == ord("0"))
            starts = np.nonzero(is_num & ~np.concatenate([[False], (buf[:-1] == ord(".")) | (buf[:-1] == ord("0"))]))[0]
            assert starts.size == arr.size
            return txt, buf, starts

        self.grid_txt, self.grid_buf, self.grid_pos = template(np.zeros((ROWS, COLS), dtype=np.float32))
        self.pf_txt, self.pf_buf, self.pf_pos = template(np.zeros((2 + PREVIEWS, len(MINOS)), dtype=np.float32))
        self.pi_txt_i, self.pi_buf_i, self.pi_pos_i = template(np.zeros((2 + PREVIEWS, len(MINOS)), dtype=int))
        self.pol_txt, _, self.pol_pos = template(np.zeros(POLICY_SHAPE, dtype=int))
        self.mirror = mirror_index()


_templates = None


def _tt():
    global _templates
    if _templates is None:
        _templates = _TextTemplates()
    return _templates


def _grid_text(rows, mirrored):
    t = _tt()
    bits = (np.asarray(rows, dtype=np.uint32)[:, None] >> np.arange(COLS, dtype=np.uint32)) & 1
    if mirrored:
        bits = bits[:, ::-1]
    buf = t.grid_buf.copy()
    buf[t.grid_pos[bits.reshape(-1).astype(bool)]] = ord("1")
    return buf.tobytes()


def _pieces_text(player, mirrored):
    t = _tt()
    idx = [player["piece"], player["held"]] + list(player["queue"][:PREVIEWS])
    idx += [NO_PIECE] * (2 + PREVIEWS - len(idx))
    buf = (t.pi_buf_i if mirrored else t.pf_buf).copy()
    pos = t.pi_pos_i if mirrored else t.pf_pos
    for i, p in enumerate(idx):
        if p != NO_PIECE:
            buf[pos[i * len(MINOS) + (_PIECE_MIRROR.get(p, p) if mirrored else p)]] = ord("1")
    return buf.tobytes()


def _policy_text(cells):
    """cells: {flat index: float} -> text of the [27][39][11] table with int zeros elsewhere"""
    t = _tt()
    out, prev = [], 0
    for k in sorted(cells):
        p = int(t.pol_pos[k])
        out.append(t.pol_txt[prev:p])
        out.append(repr(cells[k]).encode())
        prev = p + 1
    out.append(t.pol_txt[prev:])
    return b"".join(out)


def record_texts_for_move(state, child_moves, child_visits, augment=True):
    """records_for_move as JSON text: a list of (head, tail) byte strings, the record being
    b"[" + head + b", " + json(z) + b", " + tail + b"]" once the game's outcome is known."""
    t = _tt()
    s = unpack_state(state)
    a, o = s["players"][s["turn"]], s["players"][1 - s["turn"]]
    total = int(sum(int(v) for v in child_visits))
    assert total != 0
    n_row, n_col = POLICY_SHAPE[1], POLICY_SHAPE[2]
    cells = {}
    for m, n in zip(child_moves, child_visits):
        n = int(n)
        if n != 0:
            m = int(m)
            cells[(((m >> 10) & 31) * n_row + ((m >> 4) & 63)) * n_col + (m & 15)] = round(n / total, 4)
    pi = _policy_text(cells)

    def scalars(p):
        return b"%d, %d, %d" % (p["b2b"], p["combo"], len(p["garbage"]))

    def head(ma, mo):
        return b", ".join([_grid_text(a["rows"], ma), _pieces_text(a, ma), scalars(a),
                           _grid_text(o["rows"], mo), _pieces_text(o, mo), scalars(o), b"%d" % s["turn"]])

    if not augment:
        return [(head(False, False), pi)]
    mirrored = {}
    for k in sorted(cells):                      # reflect_policy visits (plane, col, row); a later writer would win,
        if cells[k] > 0:                         # but the mirror is injective, so the order does not matter
            mk = int(t.mirror[k])
            if mk < 0:
                raise IndexError("mirror image of a policy cell leaves the table")
            mirrored[mk] = cells[k]
    rpi = _policy_text(mirrored)
    return [(head(bool(active), bool(other)), pi if active == 0 else rpi) for active in range(2) for other in range(2)]


def finish_game_texts(game_texts, winner, value_min=0, value_mid=0.5, value_max=1):
    """finish_game for record_texts_for_move output -> list of complete record texts (bytes)"""
    import json
    out = []
    for player in range(2):
        z = value_mid if winner == -1 else (value_max if winner == player else value_min)
        zt = json.dumps(z).encode()
        for h, tl in game_texts[player]:
            out.append(b"[" + h + b", " + zt + b", " + tl + b"]")
    return out


# ---- packed sets: the saved moves of a set file, for the learner ---------------------------------------------------
# The reference's learner parses {n}.txt (hundreds of MB of JSON) and feeds 13 nested Python lists per record to
# torch.tensor.  make_training_set also writes the same saved moves as a small .npz next to the data directory
# (320-byte position, root children's moves and visit counts, z); PackedSet.features() expands them to the very same
# 13 feature arrays, mirror variants included, with numpy -- tests/test_records.py checks them element for element.
class PackedSet:
    def __init__(self, states, nchild, moves, visits, z, augment):
        self.states = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, 320)
        self.nchild = np.asarray(nchild, dtype=np.int64)
        self.moves = np.asarray(moves, dtype=np.uint16).reshape(len(self.nchild), -1)
        self.visits = np.asarray(visits, dtype=np.uint16).reshape(len(self.nchild), -1)
        self.z = np.asarray(z, dtype=np.float64)
        self.augment = bool(augment)
        self._cache = None

    @classmethod
    def from_moves(cls, packed_moves, augment):
        """packed_moves: [(state[320], moves[n], visits[n], z)] in the set file's record order"""
        m = len(packed_moves)
        width = max([len(p[1]) for p in packed_moves] + [1])
        moves, visits = np.zeros((m, width), dtype=np.uint16), np.zeros((m, width), dtype=np.uint16)
        for i, (_, mv, vis, _) in enumerate(packed_moves):
            moves[i, :len(mv)] = mv
            visits[i, :len(vis)] = vis
        states = np.stack([np.asarray(p[0], dtype=np.uint8) for p in packed_moves]) if m else np.zeros((0, 320), np.uint8)
        return cls(states, [len(p[1]) for p in packed_moves], moves, visits, [p[3] for p in packed_moves], augment)

    def save(self, path):
        np.savez_compressed(path, states=self.states, nchild=self.nchild, moves=self.moves, visits=self.visits, z=self.z,
                            augment=np.array(self.augment))

    @classmethod
    def load(cls, path):
        d = np.load(path)
        return cls(d["states"], d["nchild"], d["moves"], d["visits"], d["z"], bool(d["augment"]))

    def __len__(self):
        return len(self.nchild) * (4 if self.augment else 1)

    def _per_move(self):
        """features of every saved move by absolute player: grids [M,2,40,10], pieces [M,2,7,7], scalars [M,2,3], turn"""
        if self._cache is not None:
            return self._cache
        st = self.states
        m = st.shape[0]
        grids = np.zeros((m, 2, ROWS, COLS), dtype=np.float32)
        pieces = np.zeros((m, 2, 2 + PREVIEWS, len(MINOS)), dtype=np.float32)
        scal = np.zeros((m, 2, 3), dtype=np.int64)
        ar = np.arange(m)
        for p, base in enumerate((0, 144)):
            rows = st[:, base:base + 80].copy().view("<u2").astype(np.uint32)                    # [M, 40]
            grids[:, p] = ((rows[:, :, None] >> np.arange(COLS, dtype=np.uint32)) & 1).astype(np.float32)
            queue = st[:, base + 112:base + 120].copy().view("<u8")[:, 0]
            qlen = np.minimum(st[:, base + 139].astype(np.int64), PREVIEWS)
            slots = [st[:, base + 137].astype(np.int64), st[:, base + 138].astype(np.int64)]
            for k in range(PREVIEWS):
                pc = ((queue >> np.uint64(4 * k)) & np.uint64(15)).astype(np.int64)
                slots.append(np.where(k < qlen, pc, NO_PIECE))
            for row, pc in enumerate(slots):
                ok = pc != NO_PIECE
                pieces[ar[ok], p, row, pc[ok]] = 1.0
            scal[:, p, 0] = st[:, base + 132:base + 134].copy().view("<i2")[:, 0]
            scal[:, p, 1] = st[:, base + 134:base + 136].copy().view("<i2")[:, 0]
            scal[:, p, 2] = st[:, base + 140]
        self._cache = (grids, pieces, scal, st[:, 312].astype(np.int64))
        return self._cache

    def features(self, idx=None):
        """the 13 feature arrays of records idx (default: all, in file order), as train_network_pytorch builds them
        from the JSON records: grids float32 [K,40,10], pieces float32 [K,7,7], counters int64 [K], colour int64,
        z float64, policy float32 [K,27,39,11]"""
        grids, pieces, scal, turn = self._per_move()
        n = len(self)
        idx = np.arange(n) if idx is None else np.asarray(idx, dtype=np.int64)
        if self.augment:
            mv, ma, mo = idx // 4, ((idx % 4) // 2).astype(bool), (idx % 2).astype(bool)
        else:
            mv, ma, mo = idx, np.zeros(len(idx), bool), np.zeros(len(idx), bool)
        t = turn[mv]
        k = np.arange(len(idx))
        perm = np.arange(len(MINOS))
        for a_, b_ in _PIECE_MIRROR.items():
            perm[a_] = b_

        def side(player, mirrored):
            g = grids[mv, player].copy()
            g[mirrored] = g[mirrored][:, :, ::-1]
            p = pieces[mv, player].copy()
            p[mirrored] = p[mirrored][:, :, perm]            # one-hot column c moves to column mirror(c) (an involution)
            return g, p, scal[mv, player]

        ag, ap, asc = side(t, ma)
        og, op, osc = side(1 - t, mo)
        tt = _tt()
        pol = np.zeros((len(idx), POLICY_SHAPE[0] * POLICY_SHAPE[1] * POLICY_SHAPE[2]), dtype=np.float32)
        n_row, n_col = POLICY_SHAPE[1], POLICY_SHAPE[2]
        cells_of = {}
        for j in range(len(idx)):
            i = int(mv[j])
            if i not in cells_of:
                nc = int(self.nchild[i])
                total = int(self.visits[i, :nc].astype(np.int64).sum())
                cells = {}
                for m_, v_ in zip(self.moves[i, :nc].tolist(), self.visits[i, :nc].tolist()):
                    if v_ != 0:
                        cells[(((m_ >> 10) & 31) * n_row + ((m_ >> 4) & 63)) * n_col + (m_ & 15)] = round(v_ / total, 4)
                keys = np.fromiter(cells.keys(), dtype=np.int64, count=len(cells))
                vals = np.fromiter(cells.values(), dtype=np.float64, count=len(cells))
                pos = vals > 0
                cells_of[i] = (keys, vals, tt.mirror[keys[pos]], vals[pos])
            keys, vals, mkeys, mvals = cells_of[i]
            if ma[j]:
                pol[j, mkeys] = mvals
            else:
                pol[j, keys] = vals
        return [ag, ap, asc[:, 0], asc[:, 1], asc[:, 2], og, op, osc[:, 0], osc[:, 1], osc[:, 2], t.copy(),
                self.z[mv], pol.reshape((len(idx),) + POLICY_SHAPE)]
