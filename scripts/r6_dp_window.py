"""Price the data-parallel communication window on ONE GPU.

The default schedule ("conv", asr_amd/parallel.py) issues the 260 MB gradient all-reduce on a communication stream when the last recurrent
layer's gradients are final and lets it run beside the conv-stack backward (~2 ms of BatchNorm2d / conv2 dgrad + wgrad / conv1 wgrad kernels).
On a multi-GPU node that collective is RCCL's channel kernels: N resident workgroups that stream HBM and the xGMI links for the length of the
collective.  No multi-GPU box is available to this build, so the stand-in is scripts/probe_hog.hip: N workgroups x 512 threads streaming
HBM (load + load + add + store) until a deadline.  This script runs the metric configuration's train step through the real reducer
(one forced RCCL rank, DS2_FORCE_ALLREDUCE=1) with `dist.all_reduce` of the big bucket REPLACED by that kernel, for N in {0, 16, 32, 64}
workgroups and a duration of 1, 2 and 3 ms (the ideal 7-link time is 0.45 ms, a single ring 3 ms), and records per setting: the step time,
the span of the conv-stack backward on the compute stream, what of the "collective" outlasted it (= exposed communication), and whether any
persistent recurrence launch starved (the next step's forward recurrence follows the window).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/probe_hog.hip -o /tmp/libhog.so ; python scripts/r6_dp_window.py
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29733")
os.environ["DS2_FORCE_ALLREDUCE"] = "1"
os.environ["DS2_DP_MODE"] = os.environ.get("DS2_DP_MODE", "conv")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

import bench
from asr_amd.trainers import DeepSpeechTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
hog = C.CDLL(os.environ.get("HOG_LIB", "/tmp/libhog.so"))
hog.hog_launch.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
PER_WG = 16 << 20
src = torch.empty(64 * PER_WG // 4, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
moved = torch.zeros(1, dtype=torch.int64, device=dev)
setting = {"wgs": 0, "usec": 0.0}
real_all_reduce = dist.all_reduce
GRAD_ELEMS = 1


class _Done:
    def wait(self):
        return True


def all_reduce(t, *a, **k):
    """the big gradient bucket -> the stand-in kernel on the stream the reducer issues its collective on; everything else: RCCL (one rank)"""
    if t.numel() > (1 << 20) and setting["wgs"] > 0:
        # (the "serial" schedule reduces bucket by bucket: each bucket's stand-in runs for its share of the collective's duration)
        usec = setting["usec"] * min(1.0, t.numel() / float(GRAD_ELEMS))
        rc = hog.hog_launch(setting["wgs"], usec, src.data_ptr(), dst.data_ptr(), PER_WG, moved.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return _Done()
    return real_all_reduce(t, *a, **k)


dist.all_reduce = all_reduce
tr, batches, (rnn, H, L, Cc, B, tin) = bench._make_trainer("c3", "bf16", dev)
bx, bt, bp, bs = batches[0]
for _ in range(3):
    tr.step((bx, bt, bp.clone(), bs))
tr.synchronize()
red = tr._get_reducer()
GRAD_ELEMS = red.flat_grad.numel()
rows = []
print(f"# c3 bf16 (5 x 1024 BiGRU, B = 64, T_in = 1001), schedule {red.mode}, one forced RCCL rank; the big bucket's all-reduce replaced by N x 512-thread")
print("# HBM-streaming workgroups for D us (scripts/probe_hog.hip).  ms per step, conv-stack backward span, exposed communication, GB moved by the stand-in")
for usec in (1000.0, 2000.0, 3000.0):
    for wgs in (0, 16, 32, 64):
        if wgs == 0 and usec != 1000.0:
            continue
        setting.update(wgs=wgs, usec=usec)
        for _ in range(2):
            tr.step((bx, bt, bp.clone(), bs))
        tr.synchronize()
        starved0 = DeepSpeechTrainer.starved_steps
        moved.zero_()
        red.timing = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 8
        for _ in range(steps):
            tr.step((bx, bt, bp.clone(), bs))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        tr.synchronize()
        summ = red.timing_summary()
        red.timing = None
        row = {"wgs": wgs, "usec": usec if wgs else 0.0, "ms_per_step": ms, "conv_backward_ms": summ.get("conv_backward_ms"),
               "big_collective_ms": summ.get("big_collective_ms"), "outlasts_ms": summ.get("big_collective_outlasts_conv_backward_ms"),
               "exposed_comm_ms": summ.get("exposed_comm_ms"), "gb_per_step": float(moved.item()) / steps / 1e9,
               "starved_steps": DeepSpeechTrainer.starved_steps - starved0}
        rows.append(row)
        print(json.dumps(row), flush=True)
base = rows[0]["ms_per_step"]
print("# slowdown of the whole step against N = 0: " + ", ".join(f"N={r['wgs']} D={r['usec']:.0f}us: {r['ms_per_step'] - base:+.2f} ms" for r in rows[1:]))
dist.destroy_process_group()
