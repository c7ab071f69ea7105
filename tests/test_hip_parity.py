"""Parity of the HIP engine (through the C ABI) with the golden fixtures the Python reference
produced and with the CPU oracle on seeded inputs.  Bit-exact everywhere: the path is integer
work.  Needs a real MI355X: run with  python -m pytest tests -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_py as op
import tetris_rl_amd as trl

pytestmark = pytest.mark.gpu

SEED = 20260630


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _state_for_case(rows, piece, held, q0):
    st = np.zeros(320, dtype=np.uint8)
    st[0:80] = np.asarray(rows, dtype="<u2").view(np.uint8)
    st[112] = q0 & 15
    st[132:134] = np.array([-1], dtype="<i2").view(np.uint8)
    st[137], st[138] = piece, held
    st[139] = 1 if q0 != 7 else 0
    st[144 + 132:144 + 134] = np.array([-1], dtype="<i2").view(np.uint8)
    st[144 + 137] = st[144 + 138] = 7
    st[314] = 1
    return st


def _lists(counts, moves):
    return [moves[i, :counts[i]] for i in range(counts.shape[0])]


def test_attack_formulas_exhaustive(golden_dir):
    """integer restatement of Stats.get_attack on the device == the reference's floats,
    all 604 800 combinations (stats.py:45-157)."""
    t = np.ascontiguousarray(_load(golden_dir, "attack.npz")["table"])
    L = trl.load_library()
    rows = np.ascontiguousarray(t[:, :8])
    out = np.zeros((rows.shape[0], 4), dtype=np.int16)
    L.trl_debug_attack_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert L.trl_debug_attack_table(rows.ctypes.data, rows.shape[0], out.ctypes.data) == 0
    bad = np.nonzero((out != t[:, 8:12]).any(axis=1))[0]
    assert bad.size == 0, "first mismatch: in=%s ref=%s hip=%s" % (t[bad[0], :8], t[bad[0], 8:], out[bad[0]])


def test_env_traces_lockstep(golden_dir):
    """80 reference games replayed in lockstep: at every ply the legal list (argwhere order)
    and the complete two-player state equal the reference's."""
    d = _load(golden_dir, "env_traces.npz")
    meta = d["meta"]
    counts_all = d["legal_counts"]
    offs = np.concatenate([[0], np.cumsum(counts_all)])
    n = meta.shape[0]
    moves_off = np.concatenate([[0], np.cumsum(meta[:, 3])])
    e = trl.Engine(n, ruleset="s2", seed=int(d["seed"]))
    e.set_state(d["states"][meta[:, 4]])         # ply-0 states (both rulesets: the state carries it)
    max_plies = int(meta[:, 3].max())
    checked = 0
    for k in range(max_plies + 1):
        st = e.get_state()
        cnt, mv = e.legal_moves()
        step = np.full(n, trl.MOVE_NONE, dtype=np.uint16)
        for g in range(n):
            npl, s_off, l_off = int(meta[g, 3]), int(meta[g, 4]), int(meta[g, 5])
            if k > npl:
                continue
            a, b = st[g].copy(), d["states"][s_off + k].copy()
            a[304:308] = 0; b[304:308] = 0       # pol_ctr: moves are replayed, not drawn
            assert np.array_equal(a, b), "state mismatch game %d ply %d" % (g, k)
            ref = d["legal_flat"][offs[l_off + k]:offs[l_off + k + 1]]
            assert np.array_equal(mv[g, :cnt[g]], ref), "legal list mismatch game %d ply %d" % (g, k)
            checked += 1
            if k < npl:
                step[g] = d["moves"][moves_off[g] + k]
        e.step(step, add_bag=True, check_legal=True)
    assert checked == d["states"].shape[0]
    assert not (e.get_state()[:, 313] & 3).any(), "overflow / bad-move flag set"


def test_step_cases(golden_dir):
    """8092 (state, move) -> state cases of crafted / cave / sampled positions in one batch."""
    d = _load(golden_dir, "step_cases.npz")
    sin, cs, cm, sout = d["states_in"], d["case_state"], d["case_move"], d["states_out"]
    e = trl.Engine(cm.shape[0], ruleset="s2", seed=int(d["seed"]))
    e.set_state(sin[cs])
    cnt, mv = e.legal_moves()
    # the legal list of every input state is exactly the set of moves the reference enumerated
    for sid in range(sin.shape[0]):
        idx = np.nonzero(cs == sid)[0]
        if idx.size:
            assert np.array_equal(mv[idx[0], :cnt[idx[0]]], cm[idx]), "legal list of state %d" % sid
    e.step(cm, add_bag=True, check_legal=True)
    got = e.get_state()
    bad = np.nonzero((got != sout).any(axis=1))[0]
    assert bad.size == 0, "case %d (state %d move %d) differs at bytes %s" % (
        bad[0], cs[bad[0]], cm[bad[0]], np.nonzero(got[bad[0]] != sout[bad[0]])[0][:16])
    # states without any legal move (topped-out crafted boards) must enumerate nothing
    empty = [sid for sid in range(sin.shape[0]) if not (cs == sid).any()]
    if empty:
        e2 = trl.Engine(len(empty), seed=int(d["seed"]))
        e2.set_state(sin[empty])
        c2, _ = e2.legal_moves()
        assert not c2.any()


def test_movegen_cases(golden_dir):
    """reference test boards (util.py:34-37) x 7 pieces x {no hold, hold == piece}, the empty
    board (tests.py:23-32: O -> 9 placements) and 592 sampled positions."""
    d = _load(golden_dir, "movegen.npz")
    n = d["rows"].shape[0]
    states = np.stack([_state_for_case(d["rows"][i], int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i]))
                       for i in range(n)])
    e = trl.Engine(n, seed=SEED)
    e.set_state(states)
    cnt, mv = e.legal_moves()
    cc = np.concatenate([[0], np.cumsum(d["conv_counts"])])
    for i in range(n):
        assert np.array_equal(mv[i, :cnt[i]], d["conv_flat"][cc[i]:cc[i + 1]]), "case %d" % i
    # known answer of the reference's own test
    st = _state_for_case(np.zeros(40, dtype=np.uint16), 2, 2, 0)
    e1 = trl.Engine(1, seed=SEED)
    e1.set_state(st[None])
    c1, _ = e1.legal_moves()
    assert int(c1[0]) == 9


def test_bruteforce_generator_and_t_piece_disagreements(golden_dir):
    """Row M6: the device's 'brute-force' generator (trl_env_legal_moves_algo, the reference's default argument of
    get_move_matrix) against the reference's brute-force lists -- movegen.npz where the fixture has them (incl. the
    reference's own test boards) and all of movegen_t.npz: 240 T-piece positions where 'convolutional' and
    'brute-force' DISAGREE + 120 where they agree; the convolutional lists of those too."""
    d = _load(golden_dir, "movegen.npz")
    idx = np.nonzero(d["has_brute"])[0]
    states = np.stack([_state_for_case(d["rows"][i], int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i])) for i in idx])
    e = trl.Engine(len(idx), seed=SEED)
    e.set_state(states)
    cnt, mv = e.legal_moves("brute-force")
    bc = np.concatenate([[0], np.cumsum(d["brute_counts"])])
    for j, i in enumerate(idx):
        assert np.array_equal(mv[j, :cnt[j]], d["brute_flat"][bc[i]:bc[i + 1]]), "movegen.npz case %d" % i
    assert not (e.get_state()[:, 313] & 1).any()                      # no cap overflowed
    t = _load(golden_dir, "movegen_t.npz")
    n = t["rows"].shape[0]
    assert int(t["differ"].sum()) >= 200
    states = np.stack([_state_for_case(t["rows"][i], int(t["piece"][i]), int(t["held"][i]), int(t["q0"][i])) for i in range(n)])
    e = trl.Engine(n, seed=SEED)
    e.set_state(states)
    cc = np.concatenate([[0], np.cumsum(t["conv_counts"])])
    bc = np.concatenate([[0], np.cumsum(t["brute_counts"])])
    cnt, mv = e.legal_moves("convolutional")
    for i in range(n):
        assert np.array_equal(mv[i, :cnt[i]], t["conv_flat"][cc[i]:cc[i + 1]]), "conv, T case %d" % i
    cnt, mv = e.legal_moves("brute-force")
    for i in range(n):
        assert np.array_equal(mv[i, :cnt[i]], t["brute_flat"][bc[i]:bc[i + 1]]), "brute-force, T case %d" % i
    assert not (e.get_state()[:, 313] & 1).any()
    with pytest.raises(ValueError):
        e.legal_moves("harddrop")                                      # not built: ValueError like an unknown name


def test_uniform_games_from_seed_only(golden_dir):
    """the reference's uniform-policy games are reproduced by the fused rollout kernel from
    (seed, stream) alone -- bags, garbage columns and move choices all come from the RNG."""
    d = _load(golden_dir, "env_traces.npz")
    meta = d["meta"]
    for ruleset, name in ((1, "s1"), (2, "s2")):
        games = [g for g in range(meta.shape[0]) if meta[g, 2] == 0 and meta[g, 0] == ruleset]
        streams = np.array([meta[g, 1] for g in games], dtype=np.uint32)
        e = trl.Engine(len(games), ruleset=name, seed=int(d["seed"]))
        e.reset(streams)
        shortest = int(min(meta[g, 3] for g in games))
        for k in range(1, shortest):
            e.rollout_random(1)
            st = e.get_state()
            for j, g in enumerate(games):
                assert np.array_equal(st[j], d["states"][meta[g, 4] + k]), "game %d ply %d" % (g, k)


@pytest.mark.parametrize("ruleset", ["s1", "s2"])
def test_rollout_matches_oracle(ruleset):
    """C2-shaped workload, seeded: 1024 games x 120 plies with auto-reset; final states, step
    count and the legal lists of the final positions equal the CPU oracle's."""
    n, plies = 1024, 120
    e = trl.Engine(n, ruleset=ruleset, seed=SEED)
    e.reset()
    ob = op.OracleBatch(n, ruleset=trl.engine.RULESETS[ruleset], seed=SEED)
    assert np.array_equal(e.get_state(), ob.export())
    steps = e.rollout_random(plies)
    osteps = ob.rollout_random(plies)
    assert steps == osteps
    got, want = e.get_state(), ob.export()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "%d games differ, first %d" % (bad.size, bad[0])
    assert (got[:, 308:312].view("<u4") > 0).any(), "no game was auto-reset: the test is too short"
    cnt, mv = e.legal_moves()
    for g in range(0, n, 7):
        assert np.array_equal(mv[g, :cnt[g]], ob.legal_moves(g))


def test_rollout_full_size_properties():
    """BASELINE config C2 (4096 games) at full length through size-independent properties:
    determinism, split invariance (1000 plies == 4 x 250), no cap overflow, board/stat
    invariants; games 0..63 are also checked against the oracle over all 1000 plies."""
    n, plies = 4096, 1000
    a = trl.Engine(n, seed=SEED); a.reset()
    b = trl.Engine(n, seed=SEED); b.reset()
    sa = a.rollout_random(plies)
    sb = sum(b.rollout_random(250) for _ in range(4))
    A, B = a.get_state(), b.get_state()
    assert sa == sb and np.array_equal(A, B)
    assert not (A[:, 313] & 3).any(), "overflow / bad-move flag"
    rows = A[:, :288].reshape(n, 2, 144)[:, :, :80].copy().view("<u2")
    assert (rows < 1024).all()
    assert (rows != 1023).all(), "a full row survived a lock"
    ply = A[:, 300:304].copy().view("<u4")[:, 0]
    pieces = A[:, :288].reshape(n, 2, 144)[:, :, 120:124].copy().view("<i4")[:, :, 0]
    assert np.array_equal(pieces.sum(axis=1).astype(np.uint32), ply)
    assert sa > 0.95 * n * plies
    ob = op.OracleBatch(64, ruleset=2, seed=SEED)
    ob.rollout_random(plies)
    assert np.array_equal(A[:64], ob.export())


def test_bad_move_is_rejected():
    e = trl.Engine(2, seed=SEED)
    e.reset()
    before = e.get_state()
    e.step(np.array([trl.encode_move(26, 0, 0), trl.MOVE_NONE], dtype=np.uint16), check_legal=True)
    after = e.get_state()
    assert after[0, 313] & 2 and np.array_equal(after[1], before[1])
    after[0, 313] &= ~np.uint8(2)
    assert np.array_equal(after[0], before[0])


def test_call_order_errors():
    e = trl.Engine(2, seed=SEED)
    with pytest.raises(trl.EngineError):
        e.legal_moves()
    with pytest.raises(trl.EngineError):
        e.rollout_random(1)


def test_device_pointer_entry_points():
    """trl_env_legal_dev / trl_env_step_dev / trl_env_state_dev: the same results with the buffers left in HBM
    (wrapped as torch tensors through __cuda_array_interface__) as with the host-copy variants"""
    import torch
    from tetris_rl_amd.gating import _DevView
    n = 64
    a = trl.Engine(n, seed=SEED); a.reset(); a.rollout_random(20)
    b = trl.Engine(n, seed=SEED); b.reset(); b.rollout_random(20)
    cnt, mv = a.legal_moves()
    L = trl.load_library()
    pc, pm, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.trl_env_legal_moves(b._h, None, None) == 0            # results stay on the device
    assert L.trl_env_legal_dev(b._h, C.byref(pc), C.byref(pm)) == 0 and L.trl_env_state_dev(b._h, C.byref(ps)) == 0
    torch.cuda.synchronize()                                       # device-wide: covers the engine's stream
    dev = torch.device("cuda", 0)
    tc = torch.as_tensor(_DevView(pc.value, (n,), "<i2"), device=dev)
    tm = torch.as_tensor(_DevView(pm.value, (n, 512), "<i2"), device=dev)
    assert np.array_equal(tc.cpu().numpy().view(np.uint16), cnt)
    got = tm.cpu().numpy().view(np.uint16)
    for g in range(n):
        assert np.array_equal(got[g, :cnt[g]], mv[g, :cnt[g]])
    pick = mv[np.arange(n), (np.arange(n) * 5) % np.maximum(cnt, 1)].copy()
    a.step(pick)
    dpick = torch.from_numpy(pick.view(np.int16)).to(dev)
    torch.cuda.synchronize()
    assert L.trl_env_step_dev(b._h, dpick.data_ptr(), 1, 1) == 0
    torch.cuda.synchronize()
    ts = torch.as_tensor(_DevView(ps.value, (n, 320), "|u1"), device=dev)
    assert np.array_equal(ts.cpu().numpy(), a.get_state())


def test_rollout_says_when_a_slot_runs_out_of_rng_streams():
    """rollout mode sets a finished game up again on RNG stream + 2^24 (uint32): the 256th re-setup of a slot wraps onto
    its first stream and the slot replays its games from there -- TRL_FLAG_STREAM_WRAP says so in the state (the search
    refuses the same condition up front, test_search_limits_are_refused); slots below the limit stay unflagged"""
    e = trl.Engine(4, ruleset="s2", seed=77)
    e.reset()
    st = e.get_state()
    st[:2, 308:312] = np.frombuffer(np.uint32(254).tobytes(), dtype=np.uint8)       # slots 0, 1: episode 254
    e.set_state(st)
    e.rollout_random(3 * 12, max_plies=12)                                           # three games per slot
    out = e.get_state()
    episodes = out[:, 308:312].copy().view(np.uint32).reshape(-1)
    assert (episodes[:2] >= 256).all() and (episodes[2:] < 10).all()
    assert ((out[:2, 313] & 4) == 4).all() and ((out[2:, 313] & 4) == 0).all()
    assert not (out[:, 313] & 3).any()


def test_legal_lists_on_synthetic_boards_match_the_oracle():
    """12 000 boards no game of the fixtures has seen -- random stacks of 3 .. 22 rows with holes, overhangs, wells and
    pockets, half of them with a T to place (the piece whose sub-planes depend on the walk's order: closure, mixed-cell
    rule, speculative and fallback walks all get work: of the T searches of this set 5 % are settled by the mixed-cell rule
    and 6 % are left to the ordered walk, scripts/bfs_closure_proto.py) -- through the device's legal-move search, against
    the oracle's statement of the reference's FIFO walk (pinned on the reference's own lists in test_oracle_golden.py)"""
    rng = np.random.default_rng(20260705)
    n = 12000
    states = []
    for i in range(n):
        rows = np.zeros(40, dtype=np.uint16)
        h = int(rng.integers(3, 23))
        p_fill = rng.uniform(0.35, 0.85)
        well = int(rng.integers(0, 10)) if rng.random() < 0.5 else -1
        for r in range(40 - h, 40):
            cells = rng.random(10) < (p_fill * (0.6 + 0.4 * (r - (40 - h)) / max(h - 1, 1)))
            if well >= 0 and rng.random() < 0.85:
                cells[well] = False
            if cells.all():
                cells[int(rng.integers(0, 10))] = False                # a full row would have been cleared
            rows[r] = int(sum(1 << c for c in range(10) if cells[c]))
        if rng.random() < 0.3:                                          # an overhang / pocket near the surface
            r = 40 - h - 1
            c0 = int(rng.integers(0, 8))
            rows[r] |= np.uint16(0b111 << c0) & np.uint16(0x3FF)
            rows[r] &= np.uint16(~(1 << int(rng.integers(0, 10))) & 0x3FF)
        piece = 6 if rng.random() < 0.5 else int(rng.integers(0, 7))   # T = 6 ("ZLOSIJT")
        held = int(rng.choice([7, 6, int(rng.integers(0, 7))]))
        q0 = int(rng.integers(0, 7))
        states.append(_state_for_case(rows, piece, held, q0))
    states = np.stack(states)
    e = trl.Engine(n, seed=SEED)
    e.set_state(states)
    cnt, mv = e.legal_moves()
    assert not (e.get_state()[:, 313] & 1).any()                       # no cap overflowed
    t_searches = nonempty = 0
    for i in range(n):
        want = op.OracleGame(state=states[i]).legal_moves()
        assert np.array_equal(mv[i, :cnt[i]], want), "synthetic board %d" % i
        nonempty += int(len(want) > 0)
        t_searches += int(states[i][137] == 6 or states[i][138] == 6)
    assert nonempty > 0.8 * n and t_searches > 0.5 * n


def test_moves_on_synthetic_boards_match_the_oracle():
    """a move on each of 6 000 synthetic positions -- on every other one the move that sends / clears most, else a random one -- (stacks with holes, wells and overhangs; pending garbage on
    both sides; combos and back-to-back runs under way; half of them with a T to place, so T-spin recognition, all-spin
    minis, line clears, attack, cancelling and garbage entry all fire) through k_step, against the oracle's make_move"""
    rng = np.random.default_rng(31337)
    n = 6000
    states = []
    for i in range(n):
        rows = np.zeros(40, dtype=np.uint16)
        h = int(rng.integers(3, 20))
        p_fill = rng.uniform(0.5, 0.93)
        well = int(rng.integers(0, 10))
        for r in range(40 - h, 40):
            cells = rng.random(10) < p_fill
            cells[well if rng.random() < 0.8 else int(rng.integers(0, 10))] = False
            rows[r] = int(sum(1 << c for c in range(10) if cells[c]))
        piece = 6 if rng.random() < 0.5 else int(rng.integers(0, 7))
        st = _state_for_case(rows, piece, int(rng.choice([7, 6, int(rng.integers(0, 7))])), int(rng.integers(0, 7)))
        st[144:144 + 80] = np.roll(st[0:80], 2 * int(rng.integers(0, 4)))   # the other player: a stack of its own
        st[144 + 137] = int(rng.integers(0, 7))
        for pl in (0, 144):                                              # pending garbage (columns as nibbles), stats
            glen = int(rng.integers(0, 9)) if rng.random() < 0.5 else 0
            g = rng.integers(0, 10, size=glen)
            for k, col in enumerate(g):
                st[pl + 80 + k // 2] |= np.uint8(int(col) << (4 * (k % 2)))
            st[pl + 140] = glen
            st[pl + 134:pl + 136] = np.array([int(rng.integers(0, 5))], dtype="<i2").view(np.uint8)     # combo under way
        st[288:292] = np.array([i], dtype="<u4").view(np.uint8)          # RNG stream of the game
        states.append(st)
    states = np.stack(states)
    e = trl.Engine(n, ruleset="s2", seed=SEED)
    e.set_state(states)
    canon = e.get_state()                                                # (the engine's canonical form of what was set)
    cnt, mv = e.legal_moves()
    # every other position plays the move that sends / clears most (tried on the oracle): random moves hardly ever clear a line
    def stat(st, mover, off):
        return int(st[144 * mover + off:144 * mover + off + 4].view("<i4")[0])
    pick = np.zeros(n, dtype=np.uint16)
    for i in range(n):
        if not cnt[i]:
            continue
        if i % 2:
            pick[i] = mv[i, int(rng.integers(0, cnt[i]))]
            continue
        mover, best, best_key = int(canon[i][312]), mv[i, 0], (-1, -1)
        for m in mv[i, :cnt[i]]:
            g = op.OracleGame(ruleset=2, seed=SEED, state=canon[i])
            g.make_move(int(m))
            out = g.export()
            key = (stat(out, mover, 124) - stat(canon[i], mover, 124), stat(out, mover, 128) - stat(canon[i], mover, 128))
            if key > best_key:
                best, best_key = m, key
        pick[i] = best
    live = cnt > 0
    e.step(pick, add_bag=True, check_legal=True)
    got = e.get_state()
    cleared = sent = 0
    for i in np.nonzero(live)[0]:
        g = op.OracleGame(ruleset=2, seed=SEED, state=canon[i])
        g.make_move(int(pick[i]))
        want = g.export()
        assert np.array_equal(got[i], want), "position %d, move %d: bytes %s" % (i, pick[i], np.nonzero(got[i] != want)[0][:12])
        mover = int(canon[i][312])
        lc = lambda st: int(st[144 * mover + 128:144 * mover + 132].view("<i4")[0])     # stats.lines_cleared of the mover
        ls = lambda st: int(st[144 * mover + 124:144 * mover + 128].view("<i4")[0])     # stats.lines_sent
        cleared += int(lc(want) > lc(canon[i]))
        sent += int(ls(want) > ls(canon[i]))
    assert live.sum() > 0.8 * n and cleared > 500 and sent > 300, (int(live.sum()), cleared, sent)
