#!/usr/bin/env python3
"""Per-phase instruction budget of the headline EMS kernel from its ISA.

  python tools/isa_budget.py [--nm 32] [--nc 3] [--stats profiles/r03_stamps_counts.json] [--out profiles/r03_isa_budget]

The kernel source carries section marks at the points where the diagnostic build takes its cycle stamps (STAMP(i) in
nbl_cn_ems256.hip).  Built with -DNBL_EMS_MARKS every mark is a comment line in the ISA and changes nothing else (the
instruction counts per class of the marked and the plain build are identical; checked below).  This tool

 1. compiles the translation unit to gfx950 assembly with and without the marks,
 2. cuts the kernel into basic blocks, classifies every instruction (VALU / SALU / LDS / VMEM / SMEM / branch / wait),
 3. STATIC budget: instructions per section as they stand in the code,
 4. DYNAMIC budget: expected executions per check-wave.  Block frequencies are the solution of the flow equations of the
    control-flow graph (f = e_entry + P^T f) with one taken-probability per conditional branch.  The probabilities come from
    the rules in `branch_rules()` -- every rule is printed with the branches it matched, so the model can be checked line by
    line -- and from measured run statistics of the same workload (tools/stamps.py: quickselect trips, inexact cuts, gather
    trips), not from guesses, where the branch depends on data.
 5. prints the totals beside the PMC counters of the same kernel (SQ_INSTS_VALU / _SALU / _LDS per wave) as the check of
    the model.
"""
import argparse
import json
import math
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nbldpc_amd", "csrc", "nbl_cn_ems256.hip")
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", "-mllvm", "-enable-pre=false", "--offload-arch=gfx950", "-x", "hip", "--cuda-device-only", "-S"]

# sections: the mark that ENDS them (STAMP(i) sits at the end of section i in the source)
SECTION_OF_MARK = {0: "load + variable-node pass", 1: "rank 0 (maximum, its symbol, lower bound)", 5: "staging into the check domain + conf(q,1)",
                   2: "histogram", 4: "cut location + quickselect + list compaction", 6: "three pair convolutions (LDS fetch_max scatter)",
                   7: "four gather convolutions", 8: "emit (shape, un-permute, store)"}
CLASSES = ("valu", "salu", "lds", "vmem", "smem", "branch", "wait")


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def compile_asm(extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc"] + FLAGS + extra + ["-o", out, SRC], stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def kernel_lines(lines, nm, fused, nc):
    tag = f"cn_ems_q256_dc4_kernelILi{nm}ELb{1 if fused else 0}ELi{nc}E"
    a = [i for i, l in enumerate(lines) if l.startswith("_Z") and tag in l.split(":")[0] and ":" in l][0]
    b = [i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    return lines[a + 1:b]


class Block:
    def __init__(self, label):
        self.label = label
        self.ins = []       # (op, text)
        self.section = None
        self.term = None    # (kind, target) kind in {"cond", "jump", "end"}
        self.cond_op = None


def build_blocks(K):
    """Basic blocks in layout order; a block ends at a branch or before a label."""
    blocks = [Block("entry")]
    mark_after = []  # (block index, position in block, mark)
    for l in K:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(Block(m.group(1)))
            continue
        if "NBLMARK" in t:
            mark_after.append((len(blocks) - 1, len(blocks[-1].ins), int(t.split()[-1])))
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        blocks[-1].ins.append((op, t))
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            tgt = t.split()[1] if op != "s_endpgm" else None
            blocks[-1].term = ("cond" if op.startswith("s_cbranch") else "jump" if op == "s_branch" else "end", tgt)
            blocks[-1].cond_op = op
            blocks.append(Block(None))
    return blocks, mark_after


def count_ins(ins):
    c = dict.fromkeys(CLASSES, 0)
    for op, _ in ins:
        k = classify(op)
        c[k] = c.get(k, 0) + 1
    return c


def binom_stats(n=32, p=0.25, un=4):
    """run length n_r ~ Binomial(32, 1/4) (the nm list members fall into the four (bit 0, bit 7) classes of the check-domain
    symbol with equal probability): P(run >= un), E[floor(run / un)], E[run mod un], P(run mod un > 0)"""
    pm = [math.comb(n, k) * p ** k * (1 - p) ** (n - k) for k in range(n + 1)]
    return dict(p_ge=sum(pm[un:]), e_trips=sum(pm[k] * (k // un) for k in range(n + 1)), e_rem=sum(pm[k] * (k % un) for k in range(n + 1)),
                p_rem=sum(pm[k] for k in range(n + 1) if k % un))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nm", type=int, default=32)
    ap.add_argument("--nc", type=int, default=3)
    ap.add_argument("--stats", default=os.path.join(ROOT, "profiles", "r03_stamps_counts.json"))
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "r03_summary.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_isa_budget"))
    a = ap.parse_args()

    plain = kernel_lines(compile_asm([]), a.nm, True, a.nc)
    marked = kernel_lines(compile_asm(["-DNBL_EMS_MARKS"]), a.nm, True, a.nc)
    bp, _ = build_blocks(plain)
    bm, marks = build_blocks(marked)
    tot_plain = count_ins([i for b in bp for i in b.ins])
    tot_marked = count_ins([i for b in bm for i in b.ins])
    same = tot_plain == tot_marked
    blocks = bm

    # section of every instruction: the next mark in layout order closes the section
    order = [m for _, _, m in marks]
    bounds = [(bi, pos) for bi, pos, _ in marks]
    sec_ins = {m: [] for m in order}
    tail = []
    ins_section = {}
    for bi, b in enumerate(blocks):
        for pi, ins in enumerate(b.ins):
            k = 0
            while k < len(bounds) and (bi, pi) >= bounds[k]:
                k += 1
            sec = order[k] if k < len(order) else None
            ins_section[(bi, pi)] = sec
            (sec_ins[sec] if sec is not None else tail).append(ins)
    for bi, b in enumerate(blocks):
        b.section = ins_section.get((bi, 0), ins_section.get((bi - 1, 0)))

    # ---- measured run statistics of the bench workload (tools/stamps.py on the GPU) ----------------------------------------
    stats = dict(qs_trips_per_check=2.2, inexact_edges_per_check=None, un4_trips_per_check=None, rem_trips_per_check=None, ties_per_check=0.0)
    stats_src = "defaults (DESIGN.md: 2.2 quickselect trips per check; the rest from the binomial model)"
    if os.path.exists(a.stats):
        stats.update(json.load(open(a.stats)))
        stats_src = os.path.relpath(a.stats, ROOT)
    bs = binom_stats(a.nm, 0.25, 4)
    p_inexact = (stats["inexact_edges_per_check"] / 4.0) if stats.get("inexact_edges_per_check") is not None else min(1.0, stats["qs_trips_per_check"] / 4.0 / 2.75)
    qs_trips_per_inexact = stats["qs_trips_per_check"] / 4.0 / max(p_inexact, 1e-9)

    label_index = {b.label: i for i, b in enumerate(blocks) if b.label}

    def region(bi):
        """instruction counts of the blocks a forward branch at the end of block bi skips"""
        t = label_index[blocks[bi].term[1]]
        return count_ins([i for b in blocks[bi + 1:t] for i in b.ins]), t - bi - 1

    # ---- one taken-probability per conditional branch -------------------------------------------------------------------------
    rules_log = []

    # the prologue (slot -> codeword, early exits): its conditional branches in layout order, for a fixed-iteration run without
    # an active list -- [slot >= B: no] [no active list: yes] [(inside the list look-up)] [no codeword: no] [early-exit mode: no]
    # [always-true guard in front of the exit block: yes] [codeword already done: no]
    cond_blocks = [bi for bi, b in enumerate(blocks) if b.term and b.term[0] == "cond"]
    PROLOGUE = dict(zip(cond_blocks[:7], [0.0, 1.0, 0.5, 0.0, 0.0, 1.0, 0.0]))
    exit_blocks = {bi for bi, b in enumerate(blocks) if b.ins and b.ins[0][0] == "s_endpgm"}
    assert label_index[blocks[cond_blocks[3]].term[1]] in exit_blocks and label_index[blocks[cond_blocks[6]].term[1]] in exit_blocks, "prologue shape changed"

    def prob(bi):
        b = blocks[bi]
        tgt = label_index[b.term[1]]
        sec = b.section
        if bi in PROLOGUE:
            return PROLOGUE[bi], "prologue: slot -> codeword, early-exit tests (fixed iterations, no active list: straight through)", "prologue"
        if tgt <= bi:  # back edge: a loop
            body = count_ins([i for x in blocks[tgt:bi + 1] for i in x.ins])
            if sec == 7:  # gather convolutions: un4 loops and remainder loops, recognised by their size
                gathers = body["lds"]
                if gathers >= 8:  # four entries per trip
                    # entered with P(run >= 4), E[trips] = E[floor(run / 4)]  ->  geometric back-edge probability with that mean
                    e_in = bs["e_trips"] / bs["p_ge"]
                    return 1.0 - 1.0 / e_in, f"gather loop, four entries per trip ({body['valu']} VALU, {body['lds']} LDS): E[trips | entered] = {e_in:.3f}", "exit" if "scc0" in b.cond_op else "stay"
                e_in = bs["e_rem"] / bs["p_rem"]
                return 1.0 - 1.0 / e_in, f"gather remainder loop ({body['valu']} VALU, {body['lds']} LDS): E[trips | entered] = {e_in:.3f}", "stay"
            if sec == 4:
                if body["valu"] >= 20 and body["lds"] == 0 and body["salu"] > 40:  # quickselect: exits through its forward branches
                    return None, "quickselect loop (unconditional back edge)", "stay"
                return 0.0, "tie loop of finish_ties (ties straddling the cut: not seen on real-valued LLRs)", "stay"
            return 0.0, "other loop: one trip", "stay"
        cnt, nblk = region(bi)
        n = sum(cnt[c] for c in CLASSES)
        if sec == 0 or sec is None and bi < 40:
            pass
        # wave_max_exact's FP64 fallback (several lanes hold the maximal 32-bit key): 6 x (2 DPP moves + v_max_f64) + readlanes
        if 28 <= cnt["valu"] <= 34 and cnt["salu"] <= 2 and cnt["lds"] == 0 and cnt["vmem"] == 0:
            return 1.0, "FP64 fallback of the keyed wave maximum (several lanes with the maximal key): not taken", "skip"
        if sec == 0:
            if cnt["valu"] >= 50 and cnt["vmem"] >= 1:
                return 0.5, "variable-node pass: hard decision by the check that holds the variable's FIRST edge (dv = 2: every second edge)", "skip"
            if cnt["vmem"] >= 1 and cnt["valu"] <= 4 and "exec" in b.cond_op:
                return 0.0, "lane 0 stores the decision (EXEC is never empty)", "skip"
            if cnt["smem"] >= 1 and n <= 30:
                return 1.0, "active-list look-up of nbl_codeword (fixed iterations: no list)", "skip"
        if sec == 5 or sec == 1:
            return 0.5, "wave-uniform in-pair swap of conf(q,1) / rank-0 symbol half (both sides cost the same)", "skip"
        if sec == 4:
            if cnt["salu"] >= 70 and cnt["valu"] >= 15 and cnt["vmem"] == 0:
                # the branch around the quickselect: taken when the cut bucket ends exactly at the nm-th entry
                taken_is_exact = True
                return 1.0 - p_inexact, f"cut bucket ends exactly at the nm-th entry (measured: inexact on {p_inexact:.3f} of the edges)", "skip"
            if cnt["vmem"] >= 1 and cnt["valu"] >= 40:
                return 1.0, "finish_ties (ties straddling the cut): not taken", "skip"
        if sec == 7:
            # guards of the gather loops: region = the loop they skip
            if cnt["lds"] >= 8 and cnt["branch"] <= 2:
                return 1.0 - bs["p_ge"], f"run shorter than four entries: skip the four-entries-per-trip loop (binomial model: P = {1 - bs['p_ge']:.3f})", "skip"
            if 2 <= cnt["lds"] <= 6 and cnt["branch"] <= 2:
                return 1.0 - bs["p_rem"], f"run length a multiple of four: skip the remainder loop (binomial model: P = {1 - bs['p_rem']:.3f})", "skip"
        if sec == 8:
            if 15 <= cnt["valu"] <= 22 and cnt["lds"] == 0:
                return 1.0, "shape_llr general path (factor != 1 or offset != 0): the bench profile is unshaped", "skip"
        return 0.5, f"default ({n} instructions skipped)", "skip"

    nb = len(blocks)
    P = np.zeros((nb, nb))
    qs_loops = []
    table = []
    for bi, b in enumerate(blocks):
        if b.term is None:
            if bi + 1 < nb:
                P[bi, bi + 1] = 1.0
            continue
        kind, tgt = b.term
        if kind == "end":
            continue
        t = label_index[tgt]
        if kind == "jump":
            P[bi, t] = 1.0
            if t <= bi and b.section == 4:
                qs_loops.append((t, bi))
            continue
        p, why, _ = prob(bi)
        if p is None:
            p = 0.5
        P[bi, t] += p
        if bi + 1 < nb:
            P[bi, bi + 1] += 1.0 - p
        cnt, nblk = (region(bi) if t > bi else (count_ins([i for x in blocks[t:bi + 1] for i in x.ins]), bi - t + 1))
        table.append(dict(block=bi, op=b.cond_op, target_block=t, section=b.section, p_taken=round(p, 4), rule=why,
                          region={k: v for k, v in cnt.items() if v}))
    # quickselect loops: unconditional back edge, the exits are the forward branches inside.  The expected number of trips per
    # entry is measured; scale the exit probabilities of the loop's forward exits so that the chain reproduces it.
    for (h, l) in qs_loops:
        exits = [bi for bi in range(h, l + 1) if blocks[bi].term and blocks[bi].term[0] == "cond" and label_index[blocks[bi].term[1]] > l]
        # single-parameter fit: every exit branch gets the same probability pe with 1 - (1 - pe)^len(exits) = 1 / trips
        if exits:
            per_trip = 1.0 / max(qs_trips_per_inexact, 1.0)
            pe = 1.0 - (1.0 - per_trip) ** (1.0 / len(exits))
            for bi in exits:
                t = label_index[blocks[bi].term[1]]
                P[bi, :] = 0
                P[bi, t] = pe
                P[bi, bi + 1] = 1.0 - pe
                for row in table:
                    if row["block"] == bi:
                        row["p_taken"] = round(pe, 4)
                        row["rule"] = f"exit of the quickselect loop (measured {qs_trips_per_inexact:.2f} trips per inexact cut)"
    e = np.zeros(nb)
    e[0] = 1.0
    f = np.linalg.solve(np.eye(nb) - P.T, e)

    # ---- budgets --------------------------------------------------------------------------------------------------------------
    static = {}
    dynamic = {}
    for bi, b in enumerate(blocks):
        for pi, (op, _) in enumerate(b.ins):
            sec = ins_section[(bi, pi)]
            k = classify(op)
            static.setdefault(sec, dict.fromkeys(CLASSES, 0))
            dynamic.setdefault(sec, dict.fromkeys(CLASSES, 0.0))
            if k in static[sec]:
                static[sec][k] += 1
                dynamic[sec][k] += f[bi]
    seq = order + [None]
    rows = []
    for sec in seq:
        name = SECTION_OF_MARK.get(sec, "tail (stamp write-back: nothing in the product build)")
        rows.append(dict(section=name, mark=sec, static=static.get(sec, {}), dynamic={k: round(v, 1) for k, v in dynamic.get(sec, {}).items()}))
    tot_dyn = {k: round(sum(r["dynamic"].get(k, 0) for r in rows), 1) for k in CLASSES}
    tot_sta = {k: sum(r["static"].get(k, 0) for r in rows) for k in CLASSES}
    pmc = {}
    if os.path.exists(a.pmc):
        s = json.load(open(a.pmc))
        for k, v in s.get("sq_pmc", {}).items():
            if f"cn_ems_q256_dc4_kernel<{a.nm}, true, {a.nc}>" in k and "per_wave" in v:
                pw = v["per_wave"]
                pmc = dict(SQ_INSTS_VALU=pw["valu_insts"], SQ_INSTS_SALU=pw["salu_insts"], SQ_INSTS_LDS=pw["lds_insts"], SQ_INSTS_VMEM=pw["vmem_insts"])
    out = dict(kernel=f"cn_ems_q256_dc4_kernel<{a.nm}, true, {a.nc}>", marks_change_nothing=same, static_total=tot_sta, dynamic_total_per_check_wave=tot_dyn,
               pmc_per_wave=pmc, run_statistics=dict(source=stats_src, **stats, p_inexact=p_inexact, qs_trips_per_inexact_cut=qs_trips_per_inexact, run_length_model=bs),
               sections=rows, branches=table)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".txt", "w") as fo:
        def w(s=""):
            fo.write(s + "\n")
            print(s)
        w(f"instruction budget of {out['kernel']} per check-wave (tools/isa_budget.py; marked build == plain build: {same})")
        w(f"run statistics: {stats_src}: quickselect trips per check {stats['qs_trips_per_check']}, inexact cuts per edge {p_inexact:.3f}")
        w()
        w(f"{'section':62s} | {'static VALU SALU  LDS VMEM':>27s} | {'dynamic VALU   SALU    LDS   VMEM':>34s} | share of VALU")
        for r in rows:
            s, d = r["static"], r["dynamic"]
            if not s:
                continue
            w(f"{r['section']:62s} | {s['valu']:11d} {s['salu']:4d} {s['lds']:4d} {s['vmem']:4d} | {d['valu']:12.1f} {d['salu']:6.1f} {d['lds']:6.1f} {d['vmem']:6.1f} | {100 * d['valu'] / max(tot_dyn['valu'], 1):5.1f} %")
        w(f"{'total':62s} | {tot_sta['valu']:11d} {tot_sta['salu']:4d} {tot_sta['lds']:4d} {tot_sta['vmem']:4d} | {tot_dyn['valu']:12.1f} {tot_dyn['salu']:6.1f} {tot_dyn['lds']:6.1f} {tot_dyn['vmem']:6.1f} |")
        if pmc:
            w(f"{'PMC per wave (' + os.path.relpath(a.pmc, ROOT) + ')':62s} | {'':27s} | {pmc.get('SQ_INSTS_VALU', 0):12.1f} {pmc.get('SQ_INSTS_SALU', 0):6.1f} {pmc.get('SQ_INSTS_LDS', 0):6.1f} {pmc.get('SQ_INSTS_VMEM', 0):6.1f} |")
        w()
        w("conditional branches and the probability the model gives them (block numbers in layout order):")
        agg = {}
        for t in table:
            key = (t["section"], t["rule"], t["p_taken"])
            agg.setdefault(key, []).append(t["block"])
        for (sec, rule, p), bl in sorted(agg.items(), key=lambda x: (order.index(x[0][0]) if x[0][0] in order else 99, x[1][0])):
            w(f"  [{SECTION_OF_MARK.get(sec, 'tail')[:28]:28s}] p_taken={p:<6} x{len(bl):3d}  {rule}")


if __name__ == "__main__":
    main()
