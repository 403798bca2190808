"""Static check of gfx950 assembly for the hazard inline-asm prefetches can create (advisor, round 4): a vector-memory load
whose destination registers are touched by ANY later instruction before an `s_waitcnt vmcnt(n)` that covers the load.  The
compiler keeps this invariant for loads it emits itself; the `issue_load_*` helpers of csrc/mlp.hip hide their loads from it
(the registers look defined at the asm statement), so a register copy or re-use between an issue and its `settle_*` would read
or clobber a register the memory system still writes - silently, and only for some register allocations.  This walks every
kernel of an assembly file:

  * vector-memory operations retire in issue order (gfx9: loads and stores share vmcnt), so a load with y younger operations
    behind it is complete after `s_waitcnt vmcnt(n)` iff y >= n;
  * the state - per VGPR, the smallest such y over all paths - is propagated over the kernel's control-flow graph to a fixed
    point (prefetches that cross loop iterations included); states that differ in the compiler's own branch flags (SGPR pairs
    set to 0 / -1 and tested again behind merged tails) are kept apart, so that an arm that settled its prefetch is not
    confused with one that issued none;
  * an instruction naming a VGPR with a pending load is reported (the re-issue of a load into the same registers is allowed).

Further straight-line checks of hazards the compiler cannot see through inline asm: wide-store data (check_store_data), DPP
sources (check_dpp), VALU-written scalar bases (check_valu_sgpr), inline-asm VALU results read by matrix instructions
(check_asm_valu_mfma, round 6).

    python tools/check_inflight_loads.py file.s [kernel-name-substring]      exit status 1 if anything is reported
"""
import re
import sys

LOAD = re.compile(r"^(global|buffer|flat|scratch)_load_")
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)_")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"vmcnt\((\d+)\)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path, only=None):
    kernels, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
                if only is None or only in name:
                    kernels[name] = body
                name = None
                continue
            t = line.split(";")[0].strip()
            if t and not t.startswith(".") or LABEL.match(t or ""):
                body.append(t)
    return kernels


CAP = 63  # vmcnt is a 6-bit counter
SREG = re.compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")
MAX_KEYS = 24  # flag valuations kept apart per basic block


def sregs(tok):
    m = SREG.match(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return range(int(m.group(1)), int(m.group(1)) + 1)
    return range(int(m.group(2)), int(m.group(3)) + 1)


PK32 = re.compile(r"^v_pk_(fma|mul|add)_f32$|^v_pk_mov_b32$")
OPSEL = re.compile(r"\bop_sel:\[([01,]+)\]")
OPSELHI = re.compile(r"\bop_sel_hi:\[([01,]+)\]")


def pk32_regs(toks, ops):
    """Registers a packed-fp32 instruction actually reads / writes: a source pair v[a:a+1] whose op_sel and op_sel_hi bits both
    pick the LOW (or both the HIGH) half is a broadcast of ONE register - the compiler uses it for a scalar-in-VGPR multiplier
    (`v_pk_fma_f32 d, x, v[46:47], d op_sel_hi:[1,0,1]` reads v46 twice and never v47, which may then hold anything, a load
    in flight included: the half that is not selected does not reach the result)."""
    m, mh = OPSEL.search(ops), OPSELHI.search(ops)
    n_src = len([t for t in toks[1:] if not t.startswith(("op_sel", "neg_", "clamp"))])
    sel = [int(x) for x in m.group(1).split(",")] if m else [0] * n_src
    sel_hi = [int(x) for x in mh.group(1).split(",")] if mh else [1] * n_src
    out = regs_of(toks[0])
    k = 0
    for t in toks[1:]:
        t = t.split(" op_sel")[0].strip()
        if t.startswith(("op_sel", "neg_", "clamp")):
            continue
        rs = sorted(regs_of(t))
        if len(rs) == 2 and k < len(sel) and k < len(sel_hi) and sel[k] == sel_hi[k]:
            out.add(rs[sel[k]])
        else:
            out.update(rs)
        k += 1
    return out


class State:
    """pending: {vgpr: vector-memory operations issued after the pending load that writes it} (the MINIMUM over the paths
    merged into this state: the fewer younger operations, the later a `vmcnt(n)` retires the load).
    flags: {first SGPR of a pair: 0 or -1} for pairs last written by `s_mov_b64 s[a:b], 0 / -1` - the compiler's way of
    remembering which arm of a branch was taken, tested again after it merged the arms' tails; vcc: 'z' / 'nz' / None when
    it was formed from such a flag.  States with different flags are kept apart, so that "came through the arm that
    settled its prefetch" and "came through the arm that did not" are not confused."""

    def __init__(self, pending=None, flags=None, vcc=None):
        self.pending, self.flags, self.vcc = dict(pending or {}), dict(flags or {}), vcc

    def key(self):
        return (frozenset(self.flags.items()), self.vcc)

    def copy(self):
        return State(self.pending, self.flags, self.vcc)


def transfer(ins, st, report, where):
    mnem = ins.split()[0]
    ops = ins[len(mnem):]
    toks = [t.strip() for t in ops.split(",")] if ops.strip() else []
    if mnem == "s_waitcnt":
        m = WAIT.search(ins)
        if m:
            n = int(m.group(1))
            for r in [r for r, y in st.pending.items() if y >= n]:
                del st.pending[r]
        return
    # ---- scalar side: the flags
    if toks:
        d = toks[0]
        if d == "vcc":
            st.vcc = None
            if mnem in ("s_and_b64", "s_andn2_b64") and len(toks) == 3:
                other = [t for t in toks[1:] if t != "exec"]
                if "exec" in toks[1:] and len(other) == 1 and sregs(other[0]) is not None:
                    v = st.flags.get(sregs(other[0])[0]) if len(sregs(other[0])) == 2 else None
                    if v is not None and (mnem == "s_and_b64" or toks[1] == "exec"):
                        nz = (v == -1) if mnem == "s_and_b64" else (v == 0)
                        st.vcc = "nz" if nz else "z"
        else:
            rs = sregs(d)
            if rs is not None and not mnem.startswith(("s_cmp", "s_cbranch", "s_bitcmp", "s_store", "s_dcache")):
                for r in rs:  # a write to either half of a tracked pair ends what is known about it
                    st.flags.pop(r, None)
                    st.flags.pop(r - 1, None)
                if mnem == "s_mov_b64" and len(rs) == 2 and len(toks) == 2 and toks[1] in ("0", "-1"):
                    st.flags[rs[0]] = int(toks[1])
    if "vcc" in ops and toks and toks[0] != "vcc" and mnem.startswith(("v_cmp", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale", "v_mad_u64")):
        st.vcc = None  # (implicit or secondary vcc results)
    if mnem.startswith("s_"):
        return
    # ---- vector side: the pending loads
    used = regs_of(ops)
    if PK32.match(mnem):
        used = pk32_regs(toks, ops)
    if VMEM.match(mnem):
        dst = set()
        if LOAD.match(mnem) and "lds" not in mnem:
            dst = regs_of(toks[0])
        clash = (used - dst) & st.pending.keys()  # address / data registers of this operation that a pending load still writes
        if clash and report is not None:
            report.append((where, ins, sorted(clash)))
        for r in st.pending:
            st.pending[r] = min(CAP, st.pending[r] + 1)
        for r in dst:  # (a re-issue into the same registers is fine: loads retire in order)
            st.pending[r] = 0
        return
    clash = used & st.pending.keys()
    if clash and report is not None:
        report.append((where, ins, sorted(clash)))


def check(body):
    leaders, labels = {0}, {}
    for i, t in enumerate(body):
        m = LABEL.match(t)
        if m:
            labels[m.group(1)] = i
            leaders.add(i)
        elif re.match(r"^s_(c?branch|endpgm|setpc)", t) and i + 1 < len(body):
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of, blocks = {}, []
    for k, st in enumerate(starts):
        blocks.append((st, starts[k + 1] if k + 1 < len(starts) else len(body)))
        block_of[st] = k

    def successors(k, state):
        st, en = blocks[k]
        last = body[en - 1] if en > st else ""
        m = BRANCH.match(last)
        target = block_of[labels[m.group(1)]] if m and m.group(1) in labels else None
        fall = block_of[en] if en < len(body) and not last.startswith(("s_branch", "s_endpgm", "s_setpc")) else None
        if target is not None and state.vcc is not None and last.startswith(("s_cbranch_vccz", "s_cbranch_vccnz")):
            taken = (state.vcc == "z") == last.startswith("s_cbranch_vccz")
            return [target] if taken else ([fall] if fall is not None else [])
        return [x for x in (target, fall) if x is not None]

    entry = [dict() for _ in blocks]  # per block: flag valuation -> State
    entry[0][State().key()] = State()
    work = [(0, State().key())]
    while work:
        k, key = work.pop()
        if key not in entry[k]:
            continue
        state = entry[k][key].copy()
        st, en = blocks[k]
        for i in range(st, en):
            if not LABEL.match(body[i]):
                transfer(body[i], state, None, i)
        for j in successors(k, state):
            s2 = state.copy()
            if len(entry[j]) >= MAX_KEYS and s2.key() not in entry[j]:
                s2.flags, s2.vcc = {}, None  # too many valuations: fold into the one that knows nothing
            kj = s2.key()
            if kj not in entry[j]:
                entry[j][kj] = s2
                work.append((j, kj))
            else:
                cur, changed = entry[j][kj], False
                for r, y in s2.pending.items():
                    if r not in cur.pending or y < cur.pending[r]:
                        cur.pending[r] = y
                        changed = True
                if changed:
                    work.append((j, kj))
    report = []
    for k, (st, en) in enumerate(blocks):
        for state in entry[k].values():
            state = state.copy()
            for i in range(st, en):
                if not LABEL.match(body[i]):
                    transfer(body[i], state, report, i)
    seen, out = set(), []
    for r in report:
        if (r[0], r[1]) not in seen:
            seen.add((r[0], r[1]))
            out.append(r)
    return sorted(out)


WIDE_STORE = re.compile(r"^(global|buffer|flat|scratch)_store_dwordx[34]\b")


def check_store_data(body, wait_states=2):
    """Second hazard of hand-written memory instructions (the one behind round 5's wrong output-layer dW): a vector store of
    more than 64 bits reads its data registers AFTER issue, so an instruction that overwrites them within the next
    `wait_states` issue slots races the read (the compiler inserts s_nop for its own stores; an `asm volatile` store of a
    temporary has to carry its own).  Straight-line scan: every dwordx3 / dwordx4 store, the slots behind it up to the next
    label or branch."""
    out = []
    for i, t in enumerate(body):
        mnem = t.split()[0]
        if not WIDE_STORE.match(mnem):
            continue
        toks = [x.strip() for x in t[len(mnem):].split(",")]
        data = regs_of(toks[1]) if len(toks) > 1 else set()
        slots, j = 0, i + 1
        while j < len(body) and slots < wait_states:
            u = body[j]
            if LABEL.match(u) or re.match(r"^s_(c?branch|endpgm|setpc)", u):
                break
            m2 = u.split()[0]
            if m2 == "s_nop":
                slots += int(u.split()[1]) + 1
            else:
                if not m2.startswith("s_") and not VMEM.match(m2) and not m2.startswith("ds_write") and not m2.startswith("ds_add"):
                    dst = regs_of(u[len(m2):].split(",")[0])
                    if dst & data:
                        out.append((j, u, sorted(dst & data)))
                elif LOAD.match(m2) and regs_of(u[len(m2):].split(",")[0]) & data:
                    pass  # (a load's write-back is many cycles away)
                slots += 1
            j += 1
    return out


DPP = re.compile(r"\b(row_shr|row_shl|row_ror|quad_perm|row_bcast|row_mirror|row_half_mirror|wave_shr|wave_shl|wave_ror|wave_rol|row_newbcast)\b")


def check_dpp(body, wait_states=2):
    """Third hazard the hazard recogniser cannot see through inline asm (csrc/hashgrid.hip's segmented scan is written as
    `v_fmac_f32_dpp` in asm): a DPP instruction reads its source VGPR from the neighbouring lanes' register file ports, and
    a VALU write of that VGPR needs `wait_states` issue slots before it.  Straight-line scan inside basic blocks."""
    out = []
    recent = []  # (slots ago is implied by position) destination registers of the last VALU instructions, newest last
    for i, t in enumerate(body):
        if LABEL.match(t) or re.match(r"^s_(c?branch|endpgm|setpc)", t):
            recent = []
            continue
        mnem = t.split()[0]
        if mnem == "s_nop":
            recent += [set()] * (int(t.split()[1]) + 1)
            recent = recent[-wait_states:]
            continue
        toks = [x.strip() for x in t[len(mnem):].split(",")]
        if mnem.startswith("v_") and DPP.search(t) and len(toks) > 1:
            src = regs_of(toks[1].split()[0])
            clash = set().union(*recent[-wait_states:]) & src if recent else set()
            if clash:
                out.append((i, t, sorted(clash)))
        recent.append(regs_of(toks[0]) if mnem.startswith("v_") and not mnem.startswith("v_cmp") and toks else set())
        recent = recent[-wait_states:]
    return out


def check_valu_sgpr(body, wait_states=5):
    """Fourth: an SGPR written by the VALU (v_readfirstlane / v_readlane / a compare into an SGPR pair) and read as the scalar
    base of a vector-memory instruction needs `wait_states` issue slots in between (the asm loads of csrc/mlp.hip take their
    bases in SGPRs).  A later SALU write of the register ends the hazard.  Straight-line scan inside basic blocks."""
    out, recent = [], []  # recent: per issue slot, the SGPRs the VALU wrote (newest last)
    for i, t in enumerate(body):
        if LABEL.match(t) or re.match(r"^s_(c?branch|endpgm|setpc)", t):
            recent = []
            continue
        mnem = t.split()[0]
        if mnem == "s_nop":
            recent = (recent + [set()] * (int(t.split()[1]) + 1))[-wait_states:]
            continue
        toks = [x.strip() for x in t[len(mnem):].split(",")]
        if VMEM.match(mnem):
            used = set()
            for tok in toks:
                for w in tok.split():
                    r = sregs(w)
                    if r is not None:
                        used |= set(r)
            clash = (set().union(*recent) & used) if recent else set()
            if clash:
                out.append((i, t, sorted(clash)))
        written = set()
        r = sregs(toks[0]) if toks else None
        if r is not None:
            if mnem.startswith("v_"):
                written = set(r)
            elif mnem.startswith("s_") and not mnem.startswith(("s_cmp", "s_bitcmp", "s_waitcnt")):
                recent = [w - set(r) for w in recent]  # overwritten by the SALU: no longer the VALU's value
        recent = (recent + [written])[-wait_states:]
    return out


ASM_VALU = ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_bfe_i32")  # VALU mnemonics this code base only writes in inline asm


def check_asm_valu_mfma(body, wait_states=2):
    """Fifth (round 6): a VGPR written by the VALU and read by a matrix instruction needs `wait_states` issue slots in between.
    The hazard recogniser provides them for the instructions the compiler emits; a write made by inline asm (csrc/mlp.hip's
    split2(): v_fma_mix*; its gates: v_bfe_i32) is invisible to it.  Found on the device as run-to-run differences of one
    weight-gradient block (tools/diag_fp16s_repro.py).  s_waitcnt is not counted as a slot (it may not stall).  Straight-line
    scan inside basic blocks."""
    out, recent = [], []  # per issue slot: registers written by an asm-only VALU mnemonic (newest last)
    for i, t in enumerate(body):
        if LABEL.match(t) or re.match(r"^s_(c?branch|endpgm|setpc)", t):
            recent = []
            continue
        mnem = t.split()[0]
        if mnem == "s_nop":
            recent = (recent + [set()] * (int(t.split()[1]) + 1))[-wait_states:]
            continue
        if mnem == "s_waitcnt":
            continue
        toks = [x.strip() for x in t[len(mnem):].split(",")]
        if mnem.startswith("v_mfma") and len(toks) > 1:
            src = set()
            for tok in toks[1:]:
                src |= regs_of(tok)
            clash = (set().union(*recent) & src) if recent else set()
            if clash:
                out.append((i, t, sorted(clash)))
        recent = (recent + [regs_of(toks[0]) if mnem in ASM_VALU and toks else set()])[-wait_states:]
    return out


def main():
    only = sys.argv[2] if len(sys.argv) > 2 else None
    bad = 0
    kernels = parse(sys.argv[1], only)
    for name, body in kernels.items():
        rep = check(body) + [(w, i + "   [data registers of the wide store in front of it]", r) for w, i, r in check_store_data(body)]
        rep += [(w, i + "   [DPP source written by the VALU less than two slots before]", r) for w, i, r in check_dpp(body)]
        rep += [(w, i + "   [scalar base written by the VALU less than five slots before]", [f"s{x}" for x in r]) for w, i, r in check_valu_sgpr(body)]
        rep += [(w, i + "   [operand written by an inline-asm VALU instruction less than two slots before]", r) for w, i, r in check_asm_valu_mfma(body)]
        n_loads = sum(1 for t in body if LOAD.match(t.split()[0]))
        print(f"{name[:100]}: {len(body)} instructions, {n_loads} vector loads, {len(rep)} findings")
        for where, ins, regs in rep[:20]:
            print(f"    instruction {where}: `{ins}` touches v{regs} while a load into them is in flight")
        bad += len(rep)
    print(f"{len(kernels)} kernels, {bad} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
