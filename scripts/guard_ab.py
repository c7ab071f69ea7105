"""A/B in one process: the split towers with and without their activation-range guard (two builds of the library,
interleaved launches, HIP events).  usage (GPU box): python scripts/guard_ab.py"""
import ctypes as C, os, subprocess, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
import importlib
B = importlib.import_module("tetris_rl_amd.build")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
alt = os.path.join(ROOT, "scripts", "_prof", "libtrl_noguard.so")          # git-ignored, travels with gpurun
os.makedirs(os.path.dirname(alt), exist_ok=True)
if not os.path.isfile(alt) or os.path.getmtime(alt) < os.path.getmtime(os.path.join(B.CSRC, "trl_net.hip")):
    subprocess.check_call([B._hipcc()] + B.HIPCC_FLAGS + ["-DTRL_RANGE_GUARD=0", "-o", alt] + [os.path.join(B.CSRC, s) for s in B.SOURCES])
if "--build-only" in sys.argv:
    sys.exit(0)
libs = {"guard": trl.load_library(), "noguard": C.CDLL(alt)}
for L in libs.values():
    L.trl_alphasame_trunk16_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_trunk_wide_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for blocks, filters, m in ((10, 16, 1024), (20, 64, 2048)):
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=filters)).cuda().prepare()
    w = net._folded["packed_split"]
    x = (torch.rand((m, 400), device="cuda") < 0.4).float()
    out = torch.empty((m, 400), device="cuda")

    def run(L):
        if filters == 16:
            return L.trl_alphasame_trunk16_split(x.data_ptr(), m, w.data_ptr(), blocks, out.data_ptr(), None, st)
        return L.trl_trunk_wide_split(x.data_ptr(), m, w.data_ptr(), filters, blocks, 0, 1, out.data_ptr(), None, st)

    times = {k: [] for k in libs}
    for rep in range(12):
        for k, L in libs.items():
            run(L); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                assert run(L) == 0
            e1.record(); e1.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    print("AlphaSame(%d,%d) %d boards: " % (blocks, filters, m) +
          ", ".join("%s %.4f ms (min %.4f)" % (k, statistics.median(v[2:]), min(v)) for k, v in times.items()), flush=True)
