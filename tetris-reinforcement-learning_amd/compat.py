"""Single-game mirror of the reference's object API (seams S5-S8 of SURVEY.md section 8b):
Game / Player objects, get_move_matrix, evaluate and MCTS with the reference's signatures, each
call executed by the HIP engine on a one-slot batch.  Meant for interactive use, analysis code
and tests that walk `tree.get_node("root").successors()`; throughput work goes through
SelfPlay / make_training_set, which keep thousands of games on the device.
"""
import struct

import numpy as np

from . import engine as _eng
from . import records, selfplay

MINOS = "ZLOSIJT"
POLICY_SHAPE = (27, 39, 11)
_PLAYER_FMT = "<40H4QQiiihhBBBBBB2x"
_TAIL_FMT = "<IIIIIIBBB5x"
_engines = {}


def _engine(ruleset, seed):
    key = (ruleset, seed)
    if key not in _engines:
        _engines[key] = _eng.Engine(1, ruleset=ruleset, seed=seed)
    return _engines[key]


class Stats:
    def __init__(self, ruleset):
        self.pieces = self.lines_cleared = self.b2b_level = self.combo = self.lines_sent = 0
        self.b2b = -1
        self.ruleset = ruleset


class Queue:
    def __init__(self, pieces=None):
        self.pieces = pieces if pieces is not None else []


class Piece:
    def __init__(self, type=None):
        self.type = type


class Board:
    def __init__(self):
        self.grid = np.zeros((40, 10), dtype=object)


class Player:
    """player.py:10-27 attribute names"""

    def __init__(self, ruleset, color):
        self.board, self.queue, self.stats = Board(), Queue(), Stats(ruleset)
        self.game_over = False
        self.garbage_to_receive = []
        self.color = color
        self.piece = None
        self.held_piece = None
        self.ruleset = ruleset

    def _pack(self):
        rows = [sum(1 << c for c in range(10) if self.board.grid[r][c] != 0) for r in range(40)]
        garb = [0, 0, 0, 0]
        if len(self.garbage_to_receive) > 64 or len(self.queue.pieces) > 16:
            raise OverflowError("queue / garbage longer than the engine's caps")
        for k, col in enumerate(self.garbage_to_receive):
            garb[k >> 4] |= col << (4 * (k & 15))
        q = 0
        for k, p in enumerate(self.queue.pieces):
            q |= MINOS.index(p) << (4 * k)
        st = self.stats
        return struct.pack(_PLAYER_FMT, *rows, *garb, q, st.pieces, st.lines_sent, st.lines_cleared, st.b2b, st.combo,
                           st.b2b_level, MINOS.index(self.piece.type) if self.piece else 7,
                           MINOS.index(self.held_piece) if self.held_piece else 7, len(self.queue.pieces),
                           len(self.garbage_to_receive), 1 if self.game_over else 0)

    def _unpack(self, buf):
        d = records.unpack_player(buf)
        for r in range(40):
            for c in range(10):
                self.board.grid[r][c] = 1 if (d["rows"][r] >> c) & 1 else 0
        self.garbage_to_receive = list(d["garbage"])
        self.queue.pieces = [MINOS[p] for p in d["queue"]]
        st = self.stats
        st.pieces, st.lines_sent, st.lines_cleared = d["pieces"], d["lines_sent"], d["lines_cleared"]
        st.b2b, st.combo, st.b2b_level = d["b2b"], d["combo"], d["b2b_level"]
        self.piece = Piece(MINOS[d["piece"]]) if d["piece"] != 7 else None
        self.held_piece = MINOS[d["held"]] if d["held"] != 7 else None
        self.game_over = d["game_over"]


class Game:
    """game.py:6-232 without history / rendering; `seed`/`stream` name the engine RNG stream that
    replaces the reference's global `random` (bags, garbage columns)."""

    def __init__(self, ruleset, seed=0, stream=0):
        if ruleset not in ("s1", "s2"):
            raise ValueError("Unknown ruleset: %r" % (ruleset,))
        self.ruleset, self.seed, self.stream = ruleset, seed, stream
        self.players = [Player(ruleset, 0), Player(ruleset, 1)]
        self.human_player, self.ai_player = self.players
        self.turn = 0
        self._bag_ctr = self._garb_ctr = self._ply = 0

    # canonical state <-> objects
    def _pack(self):
        tail = struct.pack(_TAIL_FMT, self.stream, self._bag_ctr, self._garb_ctr, self._ply, 0, 0, self.turn, 0,
                           1 if self.ruleset == "s1" else 2)
        return np.frombuffer(self.players[0]._pack() + self.players[1]._pack() + tail, dtype=np.uint8).copy()

    def _unpack(self, st):
        st = bytes(np.asarray(st, dtype=np.uint8).tobytes())
        self.players[0]._unpack(st[0:144])
        self.players[1]._unpack(st[144:288])
        t = struct.unpack(_TAIL_FMT, st[288:320])
        self.stream, self._bag_ctr, self._garb_ctr, self._ply, self.turn = t[0], t[1], t[2], t[3], t[6]
        if t[7] & 2:
            raise ValueError("illegal move for this position")
        if t[7] & 1:
            raise OverflowError("an engine cap (queue / garbage / legal moves) overflowed")

    def setup(self):
        e = _engine(self.ruleset, self.seed)
        e.reset(np.array([self.stream], dtype=np.uint32))
        self._unpack(e.get_state()[0])

    def make_move(self, move, add_bag=True, add_history=True):
        e = _engine(self.ruleset, self.seed)
        e.set_state(self._pack()[None])
        e.step(np.array([_eng.encode_move(*move)], dtype=np.uint16), add_bag=add_bag, check_legal=True)
        self._unpack(e.get_state()[0])

    def copy(self):
        g = Game(self.ruleset, self.seed, self.stream)
        g._unpack(self._pack())
        return g

    @property
    def is_terminal(self):
        return any(p.game_over for p in self.players)

    @property
    def no_move(self):
        p = self.players[self.turn]
        return p.piece is None and p.held_piece is None

    @property
    def winner(self):
        return 1 if self.players[0].game_over else (0 if self.players[1].game_over else -1)


def get_move_matrix(player, algo="brute-force"):
    """move_generation.py:752-790 with the reference's default argument: 'brute-force' (move_generation.py:204-243)
    and 'convolutional' (:325-488, Config.move_algorithm's default, what the search uses) both run on the device.
    The lossy 'faster-but-loss' / 'harddrop' generators are not built and raise ValueError like an unknown name."""
    if algo not in _eng.MOVE_ALGORITHMS:
        raise ValueError("Unknown algorithm: %s" % algo)
    g = Game(player.ruleset)
    g.players[0] = player
    st = g._pack()
    e = _engine(player.ruleset, 0)
    e.set_state(st[None])
    counts, moves = e.legal_moves(algo)
    mask = np.zeros(POLICY_SHAPE, dtype=bool)
    for m in moves[0, :counts[0]]:
        m = int(m)
        mask[(m >> 10) & 31, (m >> 4) & 63, m & 15] = True
    return mask


def get_move_list(move_matrix, policy_matrix):
    """ai.py:1016-1024"""
    return [(policy_matrix[pi, ri, ci], (int(pi), int(ci) - 2, int(ri))) for pi, ri, ci in np.argwhere(move_matrix)]


def evaluate(config, game, network):
    """ai.py:1222-1319 (PyTorch path): (value float, softmaxed policy ndarray [27, 39, 11])"""
    import torch
    x = records.game_to_x(game._pack())
    grids = np.stack([x[0].reshape(-1), x[5].reshape(-1)])[:, None, :]
    vec = np.concatenate([x[1].reshape(-1), [x[2], x[3], x[4]], x[6].reshape(-1), [x[7], x[8], x[9]], [x[10]]])[None].astype(np.float32)
    dev = next(network.parameters()).device
    v, p = network.evaluate_packed(torch.from_numpy(grids).to(dev), torch.from_numpy(vec).to(dev))
    return float(v[0]), p[0].cpu().numpy().reshape(POLICY_SHAPE)


class NodeState:
    def __init__(self, move=None, visit_count=0, value_avg=0.0, policy=0.0, game=None):
        self.game, self.move = game, move
        self.visit_count, self.value_avg, self.policy = visit_count, value_avg, policy
        self.value_sum = value_avg * visit_count


class MCTSNode:
    def __init__(self, identifier, data, parent_id=None):
        self.identifier, self.data, self._parent_id, self._children = identifier, data, parent_id, []

    def is_root(self):
        return self._parent_id is None

    def is_leaf(self):
        return not self._children

    def successors(self, _tree_id=None):
        return self._children

    def predecessor(self, _tree_id=None):
        return self._parent_id


class MCTSTree:
    """root and its children (what play_game / search_statistics / the analysis code read, ai.py:237-279)"""

    def __init__(self):
        self._nodes, self.identifier = {}, 0

    def get_node(self, identifier):
        return self._nodes[identifier]


def MCTS(config, game, interference_network, policy_head="full"):
    """ai.py:299-659: one search for the side to move -> (move, tree, save_move); `game` is not modified."""
    import torch
    net = interference_network
    dev = next(net.parameters()).device
    e = _eng.Engine(1, ruleset=game.ruleset, seed=game.seed, device=dev.index or 0,
                    stream=torch.cuda.current_stream(dev).cuda_stream)
    e.set_state(game._pack()[None])
    e.mcts_create(selfplay.search_options(config, auto_reset=0, random_start=0))    # random openings are play_game's (ai.py:1588-1608)
    grids = torch.zeros((2, 1, _eng.GRID_FLOATS), dtype=torch.float32, device=dev)
    vec = torch.zeros((1, _eng.VEC_FLOATS), dtype=torch.float32, device=dev)
    rec, sims = None, 0
    for _ in range(100000):
        e.mcts_select(grids.data_ptr(), vec.data_ptr())
        v, p = net.evaluate_packed(grids, vec)
        e.mcts_expand(v.data_ptr(), p.data_ptr())
        torch.cuda.current_stream(dev).synchronize()
        _, info = e.mcts_leaf_states()
        sims = int(info[0, 0])
        if info[0, 2] == 3 or info[0, 0] >= info[0, 1]:         # search finished on this tick
            r = e.mcts_drain_records()
            r = r[r["kind"] == _eng.REC_MOVE]
            if len(r):
                rec = r[0]
                break
    if rec is None:
        raise RuntimeError("search did not finish")
    counts, moves, visits, q, p = e.mcts_root_children()
    tree = MCTSTree()
    root = MCTSNode("root", NodeState(game=game.copy(), visit_count=sims))
    tree._nodes["root"] = root
    for k in range(counts[0]):
        node = MCTSNode(k + 1, NodeState(move=_eng.decode_move(moves[0, k]), visit_count=int(visits[0, k]),
                                         value_avg=float(q[0, k]), policy=float(p[0, k])), "root")
        tree._nodes[k + 1] = node
        root._children.append(k + 1)
    e.close()
    return _eng.decode_move(rec["chosen"]), tree, bool(rec["saved"])


def search_statistics(tree):
    """ai.py:1330-1361 on a tree returned by MCTS()"""
    kids = [tree.get_node(c).data for c in tree.get_node("root").successors()]
    return records.search_statistics([_eng.encode_move(*k.move) for k in kids], [k.visit_count for k in kids])
