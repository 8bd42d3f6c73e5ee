"""-m gpu, round 3: the RCCL path of the data-parallel reducer with one rank (the driver's box has one GPU), the k-step bf16-vs-fp32 loss
curve at the metric configuration's size, and the bookkeeping around a starved persistent launch (BatchNorm statistics restored, epoch loss
taken back, inference returns NaN instead of synchronising)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RCCL_WORKER = r'''
import os, sys, json, hashlib
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))        # RCCL, one rank
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, FusedAdamW, engine, ops
from asr_amd.trainers import DeepSpeechTrainer
mode = os.environ["DS2_DP_MODE"]
B, tin, C = 64, 201, 29
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=1024, layers=2, classes=C))                          # c3's layer shape (256 persistent workgroups)
model.precision = "bf16"
opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
dev = torch.device("cuda", 0)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1)
x = x.cuda()
# what runs where: collectives (wrapped) record the stream they were issued on and an event pair; the conv-stack backward an event pair
log = {"collectives": [], "order": []}
orig_ar = dist.all_reduce
def ar(t, *a, **k):
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    w = orig_ar(t, *a, **k)
    e1.record(s)
    log["collectives"].append((int(t.numel()), s.cuda_stream, e0, e1))
    log["order"].append("allreduce:%d" % t.numel())
    return w
dist.all_reduce = ar
conv_ev = []
orig_c2, orig_c1 = ops.conv2_wgrad_nhwc_bf16, ops.conv1_wgrad_bf16
def c2(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); conv_ev.append(e); log["order"].append("conv2_wgrad")
    return orig_c2(*a, **k)
def c1(*a, **k):
    r = orig_c1(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); conv_ev.append(e); log["order"].append("conv1_wgrad")
    return r
ops.conv2_wgrad_nhwc_bf16, ops.conv1_wgrad_bf16 = c2, c1
losses, paths = [], []
for it in range(3):
    log["collectives"].clear(); log["order"].clear(); conv_ev.clear()
    valid, lv = tr.step((x, targets, pct.clone(), tsz))
    paths.append(ops.rnn_last_path())
    losses.append(lv)
    assert valid
tr.synchronize()
red = tr._get_reducer()
assert red.world == 1 and red.force and red.mode == mode
main = torch.cuda.current_stream().cuda_stream
big = max(log["collectives"], key=lambda c: c[0])
flat, _ = model.flat_parameters()
out = {"mode": mode, "losses": losses, "weights_sha": hashlib.sha256(flat.detach().cpu().numpy().tobytes()).hexdigest(),
       "weights_sample": flat.detach().cpu().numpy()[::9973].tolist(),
       "starved": DeepSpeechTrainer.starved_steps, "last_path": paths[-1], "order": list(log["order"]),
       "n_collectives": len(log["collectives"]), "big_elems": big[0], "big_on_comm_stream": big[1] != main,
       "gate_reduced": any(c[0] == 1 for c in log["collectives"]),
       # time from the start of the conv-stack backward to the end of the big collective, and to the end of the conv-stack backward (ms)
       "conv_start_to_big_start": conv_ev[0].elapsed_time(big[2]), "conv_start_to_conv_end": conv_ev[0].elapsed_time(conv_ev[-1])}
print("RCCL_JSON " + json.dumps(out))
dist.destroy_process_group()
'''


def _run_rccl(mode, tmp_path, port):
    script = str(tmp_path / f"rccl_{mode}.py")
    open(script, "w").write(RCCL_WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DS2_FORCE_ALLREDUCE="1", DS2_DP_MODE=mode,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_JSON ")][-1]
    return json.loads(line[len("RCCL_JSON "):])


def test_rccl_single_rank_all_schedules(tmp_path):
    """The `nccl` (= RCCL) branch of asr_amd/parallel.py, run by the driver's 1-GPU box: one rank, DS2_FORCE_ALLREDUCE=1, trainer.step at c3's
    layer width (2 x 1024 BiGRU, B = 64: 256-workgroup persistent recurrences) under the three schedules.
      * conv (default) and serial keep both persistent recurrences: identical losses and bit-identical weights after three steps;
        overlap switches the persistent BACKWARD recurrence off (other kernel family: same loss at step 0, weights within tolerance);
      * no launch starved; the validity flag goes through the MIN all-reduce with one rank too;
      * conv: the fc + RNN buckets travel as ONE collective, issued on the communication stream when the last recurrent layer is done —
        in front of the conv-stack backward in issue order, so that it runs beside it — and the compute stream joins only at the end."""
    res = {m: _run_rccl(m, tmp_path, 29611 + i) for i, m in enumerate(("conv", "serial", "overlap"))}
    for m, r in res.items():
        assert r["starved"] == 0 and r["gate_reduced"], (m, r)
        assert all(np.isfinite(r["losses"])), (m, r["losses"])
    assert res["conv"]["losses"] == res["serial"]["losses"] and res["conv"]["weights_sha"] == res["serial"]["weights_sha"]
    assert res["conv"]["last_path"] & 3 == 3 and res["serial"]["last_path"] & 3 == 3          # both recurrences persistent
    assert res["overlap"]["last_path"] & 2 == 0                                               # backward on the step kernels
    assert res["overlap"]["losses"][0] == res["conv"]["losses"][0]
    wa, wb = np.asarray(res["overlap"]["weights_sample"]), np.asarray(res["conv"]["weights_sample"])
    assert np.allclose(wa, wb, rtol=0, atol=3 * 3 * 1.5e-4 + 1e-6)           # three AdamW steps move a weight by at most ~3 lr each way
    c = res["conv"]
    assert c["big_on_comm_stream"] and c["n_collectives"] == 3               # fc+rnns bucket, conv bucket, validity flag
    order = c["order"]
    assert order.index("allreduce:%d" % c["big_elems"]) < order.index("conv2_wgrad") < order.index("conv1_wgrad"), order
    # the big collective starts no later than the conv-stack backward ends (it was released before it in issue order)
    assert c["conv_start_to_big_start"] <= c["conv_start_to_conv_end"], c
    s = res["serial"]
    assert not s["big_on_comm_stream"] and s["n_collectives"] == 2 + 2 + 1   # fc, rnns.1, rnns.0, conv, flag — all on the compute stream


def test_kstep_loss_curve_bf16_vs_fp32_at_the_metric_config():
    """SURVEY §8(d) "after k identical steps": ten fused train steps from identical weights on the identical batch at the metric
    configuration's own size (5 x 1024 BiGRU, B = 64, 10 s), once in the fp32 parity mode and once in the bf16 mode the headline number is
    measured in.  The bound that holds, and why: at step 0 (same weights) the losses agree to 3e-6; over the first four steps the gap stays
    within north_star's 1e-3; afterwards the two TRAJECTORIES separate slowly — this is training from random initialisation on one batch,
    the loss falls by 20-30 % per step (1307 -> 115 in ten steps) and AdamW's first updates move every weight by ~lr whatever the gradient's
    size, so the few small-gradient elements whose sign differs in bf16 become O(lr) weight differences.  Asserted: gap <= 1e-3 for steps
    0-3, <= 6e-3 for steps 4-9 (observed <= 3.2e-3), and never more than 5 % of that step's own loss decrease (observed <= 2.6 %).  The curve
    goes to gpurun_out/ (committed as profiles/r03_kstep_loss_curve_c3.txt)."""
    sys.path.insert(0, ROOT)
    import bench
    from test_gpu_model import make_model
    from asr_amd import CTCLoss, FusedAdamW
    from asr_amd.trainers import DeepSpeechTrainer
    rnn, H, L, C, B, tin = bench.WORKLOADS["c3"]
    x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1)
    x = x.cuda()
    dev = torch.device("cuda:0")
    curves = {}
    # third curve (round 4, the demonstration the round-3 review asked for): the fp32 path once more, all arithmetic fp32, but started from
    # weights that were rounded to bf16 ONCE — a perturbation of the size of a single bf16 operand rounding (2^-9 relative), nothing else.
    # If the late-step gap of the bf16 curve were an error of the bf16 kernels, this curve would stay on the fp32 one; it separates from it
    # as far as the bf16 curve does, i.e. the gap is the trajectory's own sensitivity to 2^-9-sized differences, not accumulated kernel error.
    for prec in ("fp32", "bf16", "fp32_from_bf16_rounded_weights"):
        torch.manual_seed(0)
        model = make_model(dict(rnn=rnn, hidden=H, layers=L, classes=C))
        model.precision = "bf16" if prec == "bf16" else "fp32"
        if prec == "fp32_from_bf16_rounded_weights":
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.copy_(p_.bfloat16().float())
        opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
        tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
        curves[prec] = [float(tr.step((x, targets, pct.clone(), tsz))[1]) for _ in range(10)]
        tr.synchronize()
        del tr, opt, model
        torch.cuda.empty_cache()
    rel = [abs(a - b) / abs(b) for a, b in zip(curves["bf16"], curves["fp32"])]
    relp = [abs(a - b) / abs(b) for a, b in zip(curves["fp32_from_bf16_rounded_weights"], curves["fp32"])]
    lines = ["step  loss_fp32      loss_bf16      |d|/loss    loss_fp32(bf16-rounded init)  |d|/loss"] + [
        f"{k:4d}  {b:12.6f}  {a:12.6f}  {r:.3e}   {c:12.6f}                  {rp:.3e}" for k, (a, b, r, c, rp) in
        enumerate(zip(curves["bf16"], curves["fp32"], rel, curves["fp32_from_bf16_rounded_weights"], relp))]
    text = "\n".join([f"10 fused train steps, {L}x{H} bi-{rnn}, B={B}, T_in={tin}, AdamW lr 1.5e-4, identical init and batch; last two columns: the fp32 "
                      "path started from weights rounded to bf16 once (a 2^-9 perturbation of the start, all arithmetic fp32)"] + lines)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "kstep_loss_curve_c3.txt"), "w").write(text + "\n")
    assert curves["fp32"][-1] < curves["fp32"][0]                                # it trains
    assert max(rel[:4]) <= 1e-3 and max(rel) <= 6e-3, rel
    # the demonstration: a one-off 2^-9 perturbation of the START, with fp32 arithmetic throughout, separates from the fp32 curve by the same
    # order as the bf16 path does over the late steps (within a factor of 4 either way) — the late gap is trajectory sensitivity
    late_b, late_p = max(rel[4:]), max(relp[4:])
    assert late_p >= 0.25 * late_b and late_p <= 4.0 * max(late_b, 1e-3), (late_b, late_p)
    for k in range(1, 10):
        assert abs(curves["bf16"][k] - curves["fp32"][k]) <= 0.05 * abs(curves["fp32"][k] - curves["fp32"][k - 1]), (k, curves)


STARVE_STEP_WORKER = r'''
import os, sys, json
import numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, FusedAdamW, ops, _lib
from asr_amd.trainers import DeepSpeechTrainer
B, tin, C = 16, 101, 29
dev = torch.device("cuda", 0)
batches = [bench.synthetic_batch(B, tin, C, 1 + 2 * k, ragged=False) for k in range(3)]
batches = [(x.cuda(), t, p, z) for x, t, p, z in batches]

def run(persistent_bwd):
    torch.manual_seed(0)
    model = make_model(dict(rnn="gru", hidden=256, layers=2, classes=C))
    model.precision = "bf16"
    opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
    model._ensure_flat(dev)
    tr._get_reducer()                                     # (creates the reducer: it arms both persistent kernels)
    ops.rnn_persistent_enable(False, persistent_bwd)      # forward on the step kernels (a valid loss); the persistent backward starves
    flat, _ = model.flat_parameters()
    w0 = flat.detach().clone()
    out = []
    for x, t, p, z in batches:
        out.append(tr.step((x, t, p.clone(), z)))
    tr.synchronize()
    return model, opt, tr, flat, w0, out

# reference run: every recurrence on the one-launch-per-step kernels, nothing can starve
mA, oA, tA, fA, w0, outA = run(False)
wA, stA, ctA = fA.detach().clone(), mA._flat.stats.detach().clone(), mA._flat.counters.detach().clone()
starved_before = DeepSpeechTrainer.starved_steps
# DS2_RNN_SPIN_LIMIT=0: the persistent backward launches of step 0 give up at their first failed poll -> the device gate rejects the step;
# step 1 discovers it (settle) and BOTH batches are computed again on the step kernels of the cooldown (DS2_RNN_REARM_CALLS=8 = two steps
# of this 2-layer model); step 2 runs persistent again, starves, and is re-run by synchronize()
model, opt, tr, flat, w0b, outB = run(True)
reported = sum(l for v, l in outB if v) - tr._take_back_rejected()
res = {"same_start": bool(torch.equal(w0, w0b)), "weights_moved": bool(not torch.equal(flat, w0b)),
       "weights_equal_step_kernel_run": bool(torch.equal(flat, wA)), "stats_equal": bool(torch.equal(model._flat.stats, stA)),
       "counters_equal": bool(torch.equal(model._flat.counters, ctA)), "counters_after": model._flat.counters.tolist(),
       "starved_steps": DeepSpeechTrainer.starved_steps - starved_before, "opt_steps": int(opt.state["step"]), "opt_steps_ref": int(oA.state["step"]),
       "epoch_loss": reported, "epoch_loss_ref": sum(l for v, l in outA if v), "valid": [bool(v) for v, _ in outB], "valid_ref": [bool(v) for v, _ in outA]}
x, targets, pct, tsz = batches[0]
# inference while the persistent kernels are armed again and starve: NaN logits without a host sync, then the check raises
ops.rnn_persistent_enable(True, True)
model.eval()
lib = _lib.load()
while ops.rnn_persistent_counters()[1] > 0:                 # burn the cooldown on the step kernels
    with torch.no_grad():
        out, _ = model.forward(x, (pct * x.size(3)).int())
    res["eval_finite_on_step_kernels"] = bool(torch.isfinite(out).all())
with torch.no_grad():
    out, _ = model.forward(x, (pct * x.size(3)).int())      # persistent again, starves (spin limit 0)
res["eval_path_persistent"] = bool(ops.rnn_last_path() & 1)
res["eval_nan"] = bool(torch.isnan(out).all())
try:
    ops.rnn_persistent_check()
    res["eval_check_raised"] = False
except _lib.DS2LibraryError:
    res["eval_check_raised"] = True
print("STARVE_JSON " + json.dumps(res))
'''


def test_starved_step_is_recomputed_not_dropped_and_inference_poison(tmp_path):
    """a train step whose persistent recurrence starved (forced: DS2_RNN_SPIN_LIMIT=0) is rejected by the device gate
    one step late — and then COMPUTED AGAIN on the one-launch-per-step kernels, together with the step that was launched before the starvation
    was known.  The reference never skips a valid batch (deepspeech_trainer.py:86-97): after three batches the weights, the BatchNorm running
    statistics / counters and the optimizer's step count are bit-identical to a run that used the step kernels throughout, the starved
    launches are counted, the epoch-loss bookkeeping equals the clean run's.  Inference never synchronises the device per forward: a starved
    launch turns the logits into NaN in stream order and the next check raises."""
    script = str(tmp_path / "starve_step.py")
    open(script, "w").write(STARVE_STEP_WORKER)
    env = dict(os.environ, DS2_RNN_SPIN_LIMIT="0", DS2_RNN_REARM_CALLS="8")
    r = subprocess.run([sys.executable, script, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("STARVE_JSON ")][-1][len("STARVE_JSON "):])
    assert res["same_start"] and res["weights_moved"], res
    assert res["weights_equal_step_kernel_run"] and res["stats_equal"] and res["counters_equal"], res      # no batch lost, none applied twice
    assert res["starved_steps"] >= 2 and res["opt_steps"] == res["opt_steps_ref"] == 3 and set(res["counters_after"]) == {3}, res
    assert all(res["valid_ref"]) and abs(res["epoch_loss"] - res["epoch_loss_ref"]) <= 1e-6 * abs(res["epoch_loss_ref"]), res
    assert res.get("eval_finite_on_step_kernels", True) and res["eval_path_persistent"] and res["eval_nan"] and res["eval_check_raised"], res


DP_STARVE_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)             # two ranks share the one GPU: gloo moves the CUDA buckets
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, FusedAdamW, ops
from asr_amd.trainers import DeepSpeechTrainer
B, tin, C = 16, 101, 29
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=128, layers=2, classes=C))      # H % 256 != 0: the all-gather backward kernel, bit-identical to the step kernels
model.precision = "bf16"
opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
out = []
for k in range(3):
    x, t, p, z = bench.synthetic_batch(B, tin, C, 100 * rank + 1 + 2 * k)  # each rank its own shard
    out.append(tr.step((x.cuda(), t, p.clone(), z)))
tr.synchronize()
reported = sum(l for v, l in out if v) - tr._take_back_rejected()
flat, _ = model.flat_parameters()
np.save(sys.argv[2] + f".rank{rank}.npy", flat.detach().cpu().numpy())
print("DPSTARVE_JSON " + json.dumps({"rank": rank, "starved_steps": DeepSpeechTrainer.starved_steps, "opt_steps": int(opt.state["step"]),
                                     "counters": model._flat.counters.tolist(), "epoch_loss": reported,
                                     "persistent_seen": bool(ops.rnn_persistent_counters()[0] > 0 or ops.rnn_last_path() & 3)}))
dist.barrier()
dist.destroy_process_group()
'''


def test_starved_rank_makes_every_rank_recompute_two_ranks_gloo(tmp_path):
    """The data-parallel form of the test above: rank 1's persistent launches starve (DS2_RNN_SPIN_LIMIT=0 in that process only); rank 0 runs the
    one-launch-per-step kernels, which cannot.  The device verdict is the MIN over ranks, so BOTH ranks compute the starved batch (and the one
    launched behind it) again, in step, with matching collectives: after three batches both replicas hold the same weights, bit-identical to
    those of a two-rank run in which nothing can starve (DS2_RNN_PERSISTENT=0 on both ranks); three optimizer updates, no batch lost, same
    epoch loss."""
    script = str(tmp_path / "w.py")
    open(script, "w").write(DP_STARVE_WORKER)
    results = {}
    # Rank 0 runs the one-launch-per-step kernels in BOTH runs: the two ranks share this box's one GPU, and a persistent launch of rank 0 (which
    # needs every CU) beside rank 1's kernels may or may not starve for real — that would make the comparison depend on timing.  Rank 0 still
    # takes the whole recovery path: the verdict it acts on is the MIN over ranks.
    for tag, envs in (("starve", ({"DS2_RNN_PERSISTENT": "0"}, {"DS2_RNN_SPIN_LIMIT": "0", "DS2_RNN_REARM_CALLS": "8"})),
                      ("steps", ({"DS2_RNN_PERSISTENT": "0"}, {"DS2_RNN_PERSISTENT": "0"}))):       # rank 1 on kernels that cannot starve
        out = str(tmp_path / tag)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + (tag == "steps")),
                       HSA_ENABLE_IPC_MODE_LEGACY="0", **envs[r])
            procs.append(subprocess.Popen([sys.executable, script, ROOT, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        recs = [json.loads([l for l in o.splitlines() if l.startswith("DPSTARVE_JSON ")][-1][len("DPSTARVE_JSON "):]) for o in outs]
        results[tag] = (sorted(recs, key=lambda d: d["rank"]), [np.load(out + f".rank{r}.npy") for r in range(2)])
    (rs, ws), (rc, wc) = results["starve"], results["steps"]
    assert np.array_equal(ws[0], ws[1]) and np.array_equal(wc[0], wc[1])                  # replicas stay bit-identical
    assert np.array_equal(ws[0], wc[0])                                                  # ... and equal the run that could not starve
    assert rs[1]["starved_steps"] >= 1 and rs[0]["starved_steps"] == 0 and rc[0]["starved_steps"] == rc[1]["starved_steps"] == 0, (rs, rc)
    for r in range(2):
        assert rs[r]["opt_steps"] == rc[r]["opt_steps"] == 3 and set(rs[r]["counters"]) == {3}, (rs, rc)
        assert abs(rs[r]["epoch_loss"] - rc[r]["epoch_loss"]) <= 1e-6 * abs(rc[r]["epoch_loss"]), (rs, rc)


AUTOGRAD_STARVE_WORKER = r'''
import os, sys, json
import numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, ops, _lib
from asr_amd.trainers import DeepSpeechTrainer
B, tin, C = 16, 101, 29
dev = torch.device("cuda", 0)
batches = [bench.synthetic_batch(B, tin, C, 1 + 2 * k) for k in range(3)]
res = {}
# ---- (a) a backward recurrence launched by loss.backward() runs on autograd's worker thread: it must use (and leave its starvation record
# in) the context of the thread that ran forward, where the enable switches were set and where the check is made
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=256, layers=2, classes=C))
model.precision = "bf16"
model._ensure_flat(dev)
ops.rnn_persistent_enable(False, True)                    # forward on the step kernels; the persistent backward starves (DS2_RNN_SPIN_LIMIT=0)
x, t, p, z = batches[0]
out, ol = model.forward(x.cuda(), (p * x.size(3)).int())
loss = CTCLoss(reduction="sum")(out.transpose(0, 1), t, ol, z) / B
torch.cuda.synchronize()
ops.rnn_persistent_check()                                # nothing starved so far
loss.backward()
torch.cuda.synchronize()
res["backward_path_persistent"] = bool(ops.rnn_last_path() & 2)       # the main thread's context saw the worker thread's launch
try:
    ops.rnn_persistent_check()
    res["check_raised"] = False
except _lib.DS2LibraryError:
    res["check_raised"] = True
ops.rnn_persistent_enable(False, False)
loss2 = CTCLoss(reduction="sum")(model.forward(x.cuda(), (p * x.size(3)).int())[0].transpose(0, 1), t, ol, z) / B
model.zero_grad(); loss2.backward(); torch.cuda.synchronize()
res["switch_reaches_autograd_backward"] = not bool(ops.rnn_last_path() & 2)
while ops.rnn_persistent_counters()[1] > 0:               # burn the cooldown
    model.zero_grad()
    l3 = CTCLoss(reduction="sum")(model.forward(x.cuda(), (p * x.size(3)).int())[0].transpose(0, 1), t, ol, z) / B
    l3.backward()
torch.cuda.synchronize()
del model

# ---- (b) the reference-API loop (fit + loss.backward() + torch AdamW): a starved backward is computed again, never applied, never dropped
def run(persistent_bwd):
    torch.manual_seed(0)
    model = make_model(dict(rnn="gru", hidden=256, layers=2, classes=C))
    model.precision = "bf16"
    opt = torch.optim.AdamW(model.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
    model._ensure_flat(dev)
    ops.rnn_persistent_enable(False, persistent_bwd)
    tr.train([(x, t, p.clone(), z) for x, t, p, z in batches])
    torch.cuda.synchronize()
    return model.flat_parameters()[0].detach().clone(), model._flat.counters.tolist(), tr._metrics.train.current.loss
wA, cA, lA = run(False)
s0 = DeepSpeechTrainer.starved_steps
wB, cB, lB = run(True)
res.update({"fit_weights_equal": bool(torch.equal(wA, wB)), "fit_counters": cB, "fit_counters_ref": cA, "fit_starved": DeepSpeechTrainer.starved_steps - s0,
            "fit_loss": lB, "fit_loss_ref": lA})
print("AG_JSON " + json.dumps(res))
'''


def test_autograd_backward_uses_the_forward_threads_recurrence_context(tmp_path):
    """`loss.backward()` runs `_DS2Function.backward` on PyTorch's autograd worker thread.  The recurrence context
    (enable switches, debug selectors, starvation record, cooldown) is captured in forward and bound around engine.backward, so (a) a backward
    launch that starves is seen by `rnn_persistent_check()` on the calling thread and the persistent switches set there reach it, and (b) the
    reference-API loop `fit` + `loss.backward()` + torch AdamW recomputes a batch whose backward starved: same weights, bit for bit, as a run on
    the step kernels, no batch dropped."""
    script = str(tmp_path / "ag.py")
    open(script, "w").write(AUTOGRAD_STARVE_WORKER)
    env = dict(os.environ, DS2_RNN_SPIN_LIMIT="0", DS2_RNN_REARM_CALLS="4")
    r = subprocess.run([sys.executable, script, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("AG_JSON ")][-1][len("AG_JSON "):])
    assert res["backward_path_persistent"] and res["check_raised"] and res["switch_reaches_autograd_backward"], res
    assert res["fit_weights_equal"] and res["fit_starved"] >= 1 and res["fit_counters"] == res["fit_counters_ref"], res
    assert abs(res["fit_loss"] - res["fit_loss_ref"]) <= 1e-6 * abs(res["fit_loss_ref"]), res


AUTO_WORKER = r'''
import os, sys, json, hashlib
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, FusedAdamW
from asr_amd.trainers import DeepSpeechTrainer
B, tin, C = 64, 201, 29
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=1024, layers=2, classes=C))
model.precision = "bf16"
opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
dev = torch.device("cuda", 0)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1)
x = x.cuda()
modes, losses = [], []
for it in range(7):
    valid, lv = tr.step((x, targets, pct.clone(), tsz))
    modes.append(tr._get_reducer().mode)
    losses.append(lv)
tr.synchronize()
red = tr._get_reducer()
flat, _ = model.flat_parameters()
print("AUTO_JSON " + json.dumps({"modes": modes, "losses": losses, "report": red.auto_report, "still_auto": red.auto, "starved": DeepSpeechTrainer.starved_steps,
                                 "weights_sha": hashlib.sha256(flat.detach().cpu().numpy().tobytes()).hexdigest()}))
dist.destroy_process_group()
'''


def test_dp_mode_auto_measures_both_schedules_and_settles(tmp_path):
    """DS2_DP_MODE=auto: steps 1-2 run the "conv" schedule, steps 3-4 "serial", both timed from backward's first
    bucket to the compute stream holding all reduced gradients; the faster one is kept and reported.  One forced RCCL rank: the two schedules
    give bit-identical weights, so the mixed run ends exactly where a pure "conv" run of the same seven steps ends."""
    res = {}
    for i, mode in enumerate(("auto", "conv")):
        script = str(tmp_path / f"auto_{mode}.py")
        open(script, "w").write(AUTO_WORKER)
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29641 + i), DS2_FORCE_ALLREDUCE="1", DS2_DP_MODE=mode,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, script, ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("AUTO_JSON ")][-1][len("AUTO_JSON "):])
    a = res["auto"]
    assert a["modes"][:3] == ["conv", "conv", "serial"] and a["modes"][3] == "serial" and not a["still_auto"], a
    rep = a["report"]
    assert rep["schedule_chosen"] in ("conv", "serial") and a["modes"][-1] == rep["schedule_chosen"], a
    assert rep["measured_over_steps"] == {"conv": 2, "serial": 2} and min(rep["backward_with_comm_ms"].values()) > 0, rep
    assert a["starved"] == 0 and a["losses"] == res["conv"]["losses"] and a["weights_sha"] == res["conv"]["weights_sha"]


DP_AUTO_WORKER = r'''
import os, sys, json, hashlib
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)             # two ranks share the one GPU: gloo moves the CUDA buckets
import bench
from test_gpu_model import make_model
from asr_amd import CTCLoss, FusedAdamW
from asr_amd.trainers import DeepSpeechTrainer
B, tin, C = 16, 101, 29
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=128, layers=2, classes=C))
model.precision = "bf16"
opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
modes = []
for k in range(7):
    x, t, p, z = bench.synthetic_batch(B, tin, C, 100 * rank + 1 + 2 * k)
    v, lv = tr.step((x.cuda(), t, p.clone(), z))
    assert v
    modes.append(tr._get_reducer().mode)
tr.synchronize()
red = tr._get_reducer()
flat, _ = model.flat_parameters()
print("DPAUTO_JSON " + json.dumps({"rank": rank, "modes": modes, "report": red.auto_report, "still_auto": red.auto,
                                   "sha": hashlib.sha256(flat.detach().cpu().numpy().tobytes()).hexdigest()}))
dist.barrier()
dist.destroy_process_group()
'''


def test_dp_mode_auto_two_ranks_agree(tmp_path):
    """DS2_DP_MODE=auto with two ranks (gloo, one GPU): the measured spans are MAX-reduced, so both ranks switch schedules at the same steps and
    settle on the same one; the replicas stay bit-identical through the schedule changes."""
    script = str(tmp_path / "w.py")
    open(script, "w").write(DP_AUTO_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", HSA_ENABLE_IPC_MODE_LEGACY="0", DS2_DP_MODE="auto")
        procs.append(subprocess.Popen([sys.executable, script, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    recs = sorted((json.loads([l for l in o.splitlines() if l.startswith("DPAUTO_JSON ")][-1][len("DPAUTO_JSON "):]) for o in outs), key=lambda d: d["rank"])
    assert recs[0]["modes"] == recs[1]["modes"] and recs[0]["modes"][:4] == ["conv", "conv", "serial", "serial"], recs
    assert recs[0]["report"] == recs[1]["report"] and recs[0]["report"]["schedule_chosen"] == recs[0]["modes"][-1] and not recs[0]["still_auto"], recs
    assert recs[0]["sha"] == recs[1]["sha"]
