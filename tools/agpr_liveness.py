"""Dev tool: registers a kernel can read before writing them, from a `hipcc -S` listing: liveness of the accumulation registers
(AGPRs) over the kernel's control-flow graph -- an AGPR that is live into the entry block is read on SOME path before any write
(the signature of the round-1 fault: results that depend on what the previous kernel left in the register file).
A static over-approximation: writes under a partial EXEC mask count as writes, paths are not checked for feasibility; the dynamic
check is tests/test_gpu_poison.py.
Usage: python tools/agpr_liveness.py <listing.s> [kernel-name-substring]"""
import re
import sys


def regs(tok):
    """AGPR indices named by one operand token: a5, a[6:9]"""
    m = re.fullmatch(r"a(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def analyse(path, key="plan_step_kernel"):
    lines = open(path).read().split("\n")
    out = []
    for st in [i for i, l in enumerate(lines) if re.match(r"_Z\w+:", l) and key in l]:
        en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        name = lines[st].rstrip(":")
        blocks, order, cur = {}, [], "entry"
        blocks[cur] = []
        order.append(cur)
        for l in lines[st + 1:en + 1]:
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                cur = m.group(1)
                blocks[cur] = []
                order.append(cur)
                continue
            t = l.split(";")[0].strip()
            if t and not t.startswith("."):
                blocks[cur].append(t)
        succ, use, dfn = {}, {}, {}
        for bi, b in enumerate(order):
            u, d, s, fall = set(), set(), [], True
            for ins in blocks[b]:
                op, _, rest = ins.partition(" ")
                toks = [x.strip() for x in rest.split(",")]
                # stores and compares have no register destination; everything else writes its first operand
                no_dst = op.startswith(("global_store", "scratch_store", "ds_write", "flat_store", "buffer_store", "s_cbranch", "s_branch",
                                        "s_waitcnt", "s_nop", "s_barrier", "v_cmp", "ds_or", "ds_add", "global_atomic", "s_endpgm"))
                srcs = toks if no_dst else toks[1:]
                for tk in srcs:
                    for r in regs(tk.split(" ")[0]):
                        if r not in d:
                            u.add(r)
                if not no_dst and toks:
                    for r in regs(toks[0].split(" ")[0]):
                        d.add(r)
                if op == "s_branch":
                    s.append(toks[0]); fall = False
                elif op.startswith("s_cbranch"):
                    s.append(toks[0])
                elif op == "s_endpgm":
                    fall = False
                elif op == "s_trap":          # __builtin_trap() (wrong block size): the wave does not continue
                    s, fall = [], False
                    break
            if fall and bi + 1 < len(order):
                s.append(order[bi + 1])
            succ[b], use[b], dfn[b] = s, u, d
        live_in = {b: set(use[b]) for b in order}
        changed = True
        while changed:
            changed = False
            for b in reversed(order):
                lo = set()
                for s in succ[b]:
                    lo |= live_in.get(s, set())
                li = use[b] | (lo - dfn[b])
                if li != live_in[b]:
                    live_in[b] = li
                    changed = True
        out.append((name, sorted(live_in["entry"]), len(order)))
    return out


if __name__ == "__main__":
    for name, live, nb in analyse(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "plan_step_kernel"):
        print(f"{name[:90]}: {nb} blocks; AGPRs live into the kernel (read before written on some path): {live if live else 'none'}")
