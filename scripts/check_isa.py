"""Build-time checks on the generated gfx950 code of asr_amd/csrc/gemm_bf16.hip (no GPU needed: hipcc -S).

1. No hot GEMM kernel spills registers, and no instance of the persistent
   recurrence kernels of asr_amd/csrc/rnn.hip does.
2. TN kernel: the fragment reads (`ds_read_b64_tr_b16`, inline asm) and the `s_waitcnt lgkmcnt` that retires them are separate
   statements; between a read and the wait that covers it NO instruction may touch the destination registers - a register copy there
   would move data that has not landed (silently wrong results).  The C++ joins the two halves of an operand right behind the reads and
   relies on the register coalescer to make that a no-op; this walks the main loop and proves it for the binary actually built.

    python scripts/check_isa.py        -> prints a summary, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "asr_amd", "csrc", "gemm_bf16.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_asm(src: str = SRC, extra=()) -> str:
    out = os.path.join(tempfile.mkdtemp(prefix="ds2isa"), os.path.basename(src) + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), *extra, "-x", "hip", "--cuda-device-only",
           "-S", src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def spill_table(asm: str):
    names = re.findall(r"\.name:\s+(\S+)", asm)
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)
    return {n: int(s) for n, s in zip(names, spills)}


def _regs(tok: str):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_tn_loop(asm: str):
    """every kernel that gathers fragments with ds_read_b64_tr_b16 (both instances of the 256 x 256 TN kernel, the four-wave grouped one of
    gemm_tn_w4.h and the co-resident grouped one)"""
    found = list(re.finditer(r"^(_ZN\S*gemm_bf16_tn_(?:glds|group|w4)_kernel\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M))
    assert len(found) >= 4, "TN kernels not found"
    reads = violations = 0
    for m in found:
        r, v = _check_tn_body(m.group(1), m.group(2))
        reads += r
        violations += v
    return reads, violations


def _check_tn_body(name: str, body: str):
    lines = [l.strip() for l in body.splitlines()]
    lines = [l for l in lines if l and not l.startswith((";", "//")) and not (l.startswith(".") and not l.endswith(":"))]
    pending = []          # destination register sets of LDS reads not yet covered by a wait, oldest first
    reads = violations = 0
    for l in lines:
        op, _, rest = l.partition(" ")
        if l.endswith(":") or op.startswith(("s_branch", "s_cbranch")):
            pending = []                                      # basic-block boundary: the walk is per straight-line region (a copy the
            continue                                          # compiler inserts sits in the block of the read or of the wait)
        toks = [t.strip() for t in re.split(r"[ ,]+", rest) if t.strip()]
        if op == "ds_read_b64_tr_b16":
            reads += 1
            busy = set().union(*pending) if pending else set()
            if _regs(toks[1]) & busy:                       # address register still in flight
                violations += 1
            pending.append(_regs(toks[0]))
            continue
        if op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", rest)
            if mm:
                n = int(mm.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        if op.startswith("s_load"):                           # scalar loads share the counter and return out of order: the compiler waits
            continue                                          # lgkmcnt(0) behind them, which the branch above handles
        busy = set().union(*pending) if pending else set()
        if busy:
            used = set()
            for t in toks:
                used |= _regs(t)
            if used & busy:
                violations += 1
                print(f"  {name}: touches in-flight fragment registers:", l)
    return reads, violations


def check_ksplit_fetch(asm: str):
    """GRU training instances of the K-split backward recurrence (rnn_bwd_ksplit_kernel<3, NT, true>): the operand loads of a step are issued
    by inline asm and retired by a counted inline-asm wait (csrc/rnn_bwd_ksplit.h, ASM_FETCH); the compiler does not know that the result
    registers are in flight in between, so NO instruction between a load block and the wait that retires it may name one of them — a copy
    there would move data that has not landed.  Walks every such instance in layout order."""
    inst = viol = 0
    for m in re.finditer(r"^(_ZN\S*rnn_bwd_ksplit_kernelILi3ELi\d+ELb1E\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        inst += 1
        lines = [l.strip() for l in m.group(2).splitlines()]
        pending = None
        blocks = 0
        for i, l in enumerate(lines):
            if l.startswith("global_load_dwordx2") and " nt" in l and i > 0 and "#ASMSTART" in lines[i - 1]:
                regs = set()
                for k in range(4):
                    regs |= _regs(re.split(r"[ ,]+", lines[i + k].partition(" ")[2].strip())[0])
                pending, start = regs, i + 4
                blocks += 1
                continue
            if pending is not None and i >= start:
                if re.fullmatch(r"s_waitcnt vmcnt\(\d+\)", l) and "#ASMSTART" in lines[i - 1]:
                    pending = None
                    continue
                if l.startswith((";", ".")) or l.endswith(":"):
                    continue
                used = set()
                for t in re.split(r"[ ,]+", l.partition(" ")[2]):
                    used |= _regs(t.strip())
                if used & pending:
                    viol += 1
                    print(f"  {m.group(1)}: line {i} names an in-flight operand register: {l}")
        assert blocks == 2 and pending is None, (m.group(1), blocks)
    return inst, viol


def main() -> int:
    asm = compile_asm()
    bad = 0
    for name, n in spill_table(asm).items():
        if "gemm_bf16" not in name:
            continue
        allowed = 0
        flag = "" if n <= allowed else "   <-- SPILLS"
        print(f"{name}: vgpr spills {n}{flag}")
        bad += n > allowed
    # the persistent recurrence kernels (every template instance): a spill there sits inside the time loop
    rnn = compile_asm(os.path.join(ROOT, "asr_amd", "csrc", "rnn.hip"), ("-mllvm", "-amdgpu-kernarg-preload-count=9"))
    pers = {n: v for n, v in spill_table(rnn).items() if "persistent_kernel" in n or "ksplit_kernel" in n}
    # (the K-split backward kernel's widest instance - LSTM, H = 1280: 160 registers of W_hh fragments - is allowed its 12 dwords)
    spilled = {n: v for n, v in pers.items() if v > (12 if "ksplit_kernelILi4ELi10" in n else 0)}
    print(f"rnn.hip: {len(pers)} persistent-kernel instances (all-gather + K-split), {len(spilled)} with register spills")
    for n, v in spilled.items():
        print(f"  {n}: vgpr spills {v}   <-- SPILLS")
    bad += len(spilled)
    ki, kv = check_ksplit_fetch(rnn)
    print(f"K-split training instances: {ki} walked, {kv} instruction(s) naming in-flight operand registers")
    bad += kv + (ki == 0)
    reads, viol = check_tn_loop(asm)
    print(f"TN kernel: {reads} tr-reads walked, {viol} instruction(s) touching in-flight fragment registers")
    bad += viol
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
