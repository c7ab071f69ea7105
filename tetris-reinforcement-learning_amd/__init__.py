"""MI355X-native self-play rollout engine for turn-based TETR.IO (HIP, gfx950).

Import name: ``tetris_rl_amd`` (the directory name carries the reference's hyphenated
repository name, so the repo root ships ``tetris_rl_amd.py`` which loads this package).
"""
from .engine import (Engine, EngineError, encode_move, decode_move, load_library, mcts_config,  # noqa: F401
                     STATE_BYTES, MAX_LEGAL, MOVE_NONE, RECORD_DTYPE, REC_MOVE, REC_GAME_END)
from . import build as _build_mod


def build(force=False, verbose=False):
    return _build_mod.build(force=force, verbose=verbose)
from .selfplay import (Config, SelfPlay, SelfPlayGroups, make_self_play, GameCollector, play_game, make_training_set, self_play_loop,  # noqa: F401,E402
                       instantiate_network, load_best_model, highest_model_number, highest_data_number, search_options,
                       games_from_records)
from . import nets  # noqa: F401,E402
from .nets import (AlphaSame, AlphaSameConfig, BaseResNet, BaseResNetConfig, AuxBaseResNet,  # noqa: F401,E402
                   AuxBaseResNetConfig, port_keras_weights)
from . import compat  # noqa: F401,E402
from .compat import Game, get_move_matrix, get_move_list, evaluate, MCTS, search_statistics  # noqa: F401,E402
from . import gating  # noqa: F401,E402
from .gating import battle_networks  # noqa: F401,E402
from . import learner  # noqa: F401,E402
from .learner import train_network_pytorch, load_data_and_train_model  # noqa: F401,E402
