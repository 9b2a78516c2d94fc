"""Static check of gfx950 machine code: no vector register is read or overwritten while a vector-memory load into it is outstanding.

Test infrastructure (used by tests/test_host_logic.py).  The compiler guarantees this for the loads it tracks; `k_rambo`'s
software-pipelined row loads (nf_amd/csrc/rambo.hip: inline-asm `global_load_dwordx4` + a hand-written `s_waitcnt vmcnt(n)`) are outside
its bookkeeping, so the built code object is checked instead, with the hardware's own rule:

  * every vector-memory instruction (global / buffer / scratch / flat load, store, atomic) increments VM_CNT; on gfx9-family parts
    they complete in issue order, so `s_waitcnt vmcnt(n)` guarantees completion of every operation that has at least n younger ones;
  * a load's destination registers are "in flight" from its issue until such a wait; any instruction that names one of them
    (source or destination) before that is a hazard -- a read sees stale data, a write is overwritten when the load lands.

The walk is a forward dataflow over the function's basic blocks (branch targets come from llvm-objdump's `<sym+0xoff>` annotations):
state = {register: fewest younger VM operations on any path}, merged with min / union at joins, iterated to the fixed point, so the
loop back-edge between the issue in one iteration and the wait at the top of the next is covered.
"""
import re
import subprocess

LLVM = "/opt/rocm/lib/llvm/bin"
_VM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load", "scratch_store",
       "flat_load", "flat_store", "flat_atomic")
_REG1 = re.compile(r"\b([va])(\d+)\b")
_REGN = re.compile(r"\b([va])\[(\d+):(\d+)\]")
CAP = 63


def disassemble(obj, td):
    """gfx950 code object of a host object file -> llvm-objdump text."""
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={td}/fat.bin",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={td}/dev.co"], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"{td}/dev.co"], check=True, capture_output=True, text=True).stdout


def functions(text):
    """-> {symbol: (start address, [(address, mnemonic, operands, target address or None)])}"""
    out, cur, start = {}, None, 0
    for ln in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
        if m:
            start, cur = int(m.group(1), 16), []
            out[m.group(2)] = (start, cur)
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        body, _, cmt = ln.partition("//")
        parts = body.split(None, 1)
        if not parts:
            continue
        ma = re.match(r"\s*([0-9A-Fa-f]+):", cmt)
        if not ma:
            continue
        tgt = None
        mt = re.search(r"<\S+\+0x([0-9a-f]+)>", cmt)
        if parts[0].startswith(("s_branch", "s_cbranch")):
            tgt = start + int(mt.group(1), 16) if mt else start
        cur.append((int(ma.group(1), 16), parts[0], parts[1].strip() if len(parts) > 1 else "", tgt))
    return out


def _regs(ops):
    s = set()
    for k, a, b in _REGN.findall(ops):
        s.update((k, i) for i in range(int(a), int(b) + 1))
    for k, a in _REG1.findall(ops):
        s.add((k, int(a)))
    return s


def _dest(ops):
    first = ops.split(",")[0]
    return _regs(first)


def check_function(insts, name="?"):
    """Returns (violations, stats): violations = [(address, text, registers)], stats = loads seen / waits seen / most registers in flight."""
    addr_ix = {a: i for i, (a, _, _, _) in enumerate(insts)}
    leaders = {0}
    for i, (_, mn, _, tgt) in enumerate(insts):
        if mn.startswith(("s_branch", "s_cbranch")):
            assert tgt in addr_ix, f"{name}: branch target {tgt:#x} is not an instruction"
            leaders.add(addr_ix[tgt])
            if i + 1 < len(insts):
                leaders.add(i + 1)
        elif mn == "s_endpgm" and i + 1 < len(insts):
            leaders.add(i + 1)
        assert not mn.startswith(("s_setpc", "s_swappc", "s_call")), f"{name}: indirect control flow ({mn}) is not modelled"
    order = sorted(leaders)
    blocks = {b: (b, (order[k + 1] if k + 1 < len(order) else len(insts))) for k, b in enumerate(order)}

    def succ(b):
        lo, hi = blocks[b]
        _, mn, _, tgt = insts[hi - 1]
        if mn == "s_endpgm":
            return []
        if mn.startswith("s_branch"):
            return [addr_ix[tgt]]
        out = [addr_ix[tgt]] if mn.startswith("s_cbranch") else []
        if hi < len(insts):
            out.append(hi)
        return out

    viol, seen = [], set()
    stats = {"loads": 0, "waits": 0, "max_in_flight": 0}

    def run(b, state, record):
        st = dict(state)
        lo, hi = blocks[b]
        for i in range(lo, hi):
            a, mn, ops, _ = insts[i]
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    n = int(m.group(1))
                    st = {r: c for r, c in st.items() if c < n}
                    if record:
                        stats["waits"] += 1
                elif re.fullmatch(r"(0x)?0+", ops.strip()):
                    st = {}
                continue
            if st:
                hit = _regs(ops) & st.keys()
                if hit and record and a not in seen:
                    seen.add(a)
                    viol.append((a, f"{mn} {ops}", sorted(hit)))
            if mn.startswith(_VM):
                st = {r: min(c + 1, CAP) for r, c in st.items()}
                returns = "load" in mn or (("atomic" in mn) and (" glc" in " " + ops or "sc0" in ops))
                if returns:
                    for r in _dest(ops):
                        st[r] = 0
                    if record:
                        stats["loads"] += 1
                        stats["max_in_flight"] = max(stats["max_in_flight"], len(st))
        return st

    entry = {b: None for b in blocks}
    entry[0] = {}
    work = [0]
    while work:
        b = work.pop()
        out = run(b, entry[b], False)
        for s in succ(b):
            old = entry[s]
            if old is None:
                new = dict(out)
            else:
                new = dict(old)
                for r, c in out.items():
                    new[r] = min(c, old.get(r, CAP + 1))
            if new != old:
                entry[s] = new
                work.append(s)
    for b in blocks:
        if entry[b] is not None:
            run(b, entry[b], True)
    return viol, stats
