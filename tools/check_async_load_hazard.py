"""ISA guard for kernels that issue loads from inline asm and wait for them by hand (gemm_d8.hip): the compiler treats an asm load's destination
as defined at the asm statement and may read or copy it (v_mov on a loop edge, a spill) BEFORE the data has landed - silently wrong results
(cdna_hip_programming.md 5.7 item 1; round 6 hit exactly this in one instantiation of the refactored d8 kernel).  Linear scan of `hipcc -S`
output per kernel: a register written by a global_load_dwordx4 stays "in flight" until an s_waitcnt vmcnt(n) retires it (vmcnt counts every
vector-memory instruction in issue order, LDS-DMA and stores included); any instruction that READS an in-flight register is reported.  The scan
follows the listing order, not the control flow (a loop's back edge is not replayed): a clean report is necessary, not sufficient.
usage: python tools/check_async_load_hazard.py file.s [substring-of-kernel-name ...]"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(name, lines):
    q, bad = [], []                      # q: outstanding vector-memory instructions, oldest first: set of destination VGPRs (empty for DMA / stores)
    for ln, t in lines:
        op, _, rest = t.partition(" ")
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            q = []                       # what follows is reached from elsewhere: unknown state, assumed drained (fewer false positives
            continue                     # on if / else arms; a hazard on such an arm's own straight-line code is still seen)
        toks = [x.strip() for x in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                while len(q) > n:
                    q.pop(0)
            continue
        inflight = set().union(*q) if q else set()
        if op.startswith(("global_load_lds", "buffer_load")) and "lds" in t:
            q.append(set()); continue
        if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
            srcs = set().union(*[regs(x.split()[0]) for x in toks[1:]]) if len(toks) > 1 else set()
            if srcs & inflight:
                bad.append((ln, t))
            q.append(regs(toks[0].split()[0]) if toks else set()); continue
        if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic")):
            srcs = set().union(*[regs(x.split()[0]) for x in toks]) if toks else set()
            if srcs & inflight:
                bad.append((ln, t))
            q.append(set()); continue
        if not inflight or not toks:
            continue
        first_is_dst = not op.startswith(("ds_write", "ds_store", "s_", "v_cmp", "v_cmpx"))
        srcs = set()
        for x in (toks[1:] if first_is_dst else toks):
            srcs |= regs(x.split()[0])
        if op.startswith("v_mfma") or op.startswith("v_fma") or op.startswith("v_mac"):
            pass                          # the accumulator is listed among the sources already
        if srcs & inflight:
            bad.append((ln, t))
    return bad


def main():
    path, want = sys.argv[1], sys.argv[2:]
    cur, body, n_bad = None, {}, 0
    for i, l in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1); body[cur] = []; continue
        if l.startswith(".Lfunc_end"):
            cur = None; continue
        t = re.sub(r"\s*;.*$", "", l.strip())
        if cur and t and not t.startswith((".", ";")):
            body[cur].append((i, t))
    for k, lines in body.items():
        if want and not any(w in k for w in want):
            continue
        if not any("global_load_dwordx4" in t for _, t in lines):
            continue
        bad = scan(k, lines)
        n_bad += len(bad)
        print(f"{k[:70]:70s} {'CLEAN' if not bad else 'HAZARD x%d' % len(bad)}")
        for ln, t in bad[:6]:
            print(f"      line {ln}: {t}")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
