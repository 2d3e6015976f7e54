#!/usr/bin/env python3
"""Executed-instruction counts of a gfx950 kernel, derived from its ISA.

  python tools/isa_count.py [--kernel k_merkle4] [--out profiles/r02_isa_counts.json]

Compiles poseidon252_amd/csrc/kernels.hip to assembly with the flags of poseidon252_amd/build.py (or reads --asm),
then walks ONE wave through the kernel: the scalar unit's control flow (s_mov / s_add / s_cmp / s_cselect /
s_cbranch ... on registers whose values are compile-time constants: loop counters, table offsets) is interpreted
exactly, vector instructions are only counted.  Lane guards (`s_and_saveexec` + `s_cbranch_execz`) are taken as
"some lane active".  A branch on a value the interpreter does not know is an error, never a guess — unless the
kernel argument it depends on is given with --arg (e.g. the sponge's in_len / out_len).

The result is the number of times each instruction mnemonic is issued by one wave for ONE pass of the kernel
(= per lane: per digest / permutation chain), which is what `bench.py` reports as `roofline.executed` and what
`SQ_INSTS_VALU / SQ_WAVES` of a --pmc pass must reproduce (profiles/r02_*pmc*.txt).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK32 = 0xFFFFFFFF

# issue cost classes on gfx950, wave64 (measured: bench_tools/valu_rates.hip, profiles/r02_valu_rates_gfx950.txt)
FOUR_CYCLE_PREFIXES = ("v_mad_i64_i32", "v_mad_u64_u32", "v_lshl_add_u64", "v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64",
                       "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_alignbit_b32", "v_add3_u32", "v_add_co_u32",
                       "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_bfe_i32",
                       "v_cmp_", "v_readlane", "v_writelane", "v_mad_u32_u24", "v_lshl_add_u32", "v_add_lshl_u32", "v_xad_u32",
                       "v_perm_b32", "v_bfi_b32", "v_mov_b64", "v_pk_mov_b32", "v_lshlrev_b32", "v_mul_i32_i24", "v_mul_u32_u24",
                       "v_fma_f64", "v_mov_b32_dpp")


def compile_asm(extra_flags=(), with_remarks=False):
    """kernels.hip -> gfx950 ISA text (hipcc -S, the library's flags).  The compile takes ~55 s and two CPU tests want its output
    (tests/test_isa_counts.py, tests/test_kernel_resources.py): the result is cached under the temp directory, keyed by the digest of
    the sources and flags; the resource-usage remarks (-Rpass-analysis=kernel-resource-usage, stderr) are kept beside it.
    with_remarks=True returns (asm_path, remarks_text)."""
    import hashlib
    sys.path.insert(0, ROOT)
    from poseidon252_amd import build as b
    b._gen_assets()
    flags = [f for f in b.HIPCC_FLAGS if f != "-fPIC"] + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage"]
    h = hashlib.sha256(" ".join([b._hipcc()] + flags).encode())
    for f in sorted(os.listdir(b.CSRC)) + [os.path.join("_gen", "assets.inc")]:
        path = os.path.join(b.CSRC, f)
        if os.path.isfile(path) and f.endswith((".hip", ".hpp", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(path, "rb").read())
    d = os.path.join(tempfile.gettempdir(), "p252_isa_cache_" + h.hexdigest()[:20])
    out, rem = os.path.join(d, "kernels.s"), os.path.join(d, "remarks.txt")
    if not (os.path.exists(out) and os.path.exists(rem)):
        os.makedirs(d, exist_ok=True)
        tmp = out + ".%d.tmp" % os.getpid()
        cmd = [b._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", tmp, os.path.join(b.CSRC, "kernels.hip")]
        proc = subprocess.run(cmd, capture_output=True, text=True, cwd=b.CSRC)
        if proc.returncode != 0:
            raise SystemExit("hipcc -S kernels.hip failed:\n" + proc.stderr[-3000:])
        open(rem + ".tmp", "w").write(proc.stderr)
        os.replace(rem + ".tmp", rem)
        os.replace(tmp, out)
    return (out, open(rem).read()) if with_remarks else out


def kernel_body(asm_text, name):
    """lines of the function whose mangled name contains `name` (label .. s_endpgm)"""
    lines = asm_text.splitlines()
    start = None
    t = re.fullmatch(r"(\w+)<(\d+)>", name)  # a template instance: k_merkle4_coop<8> -> ...k_merkle4_coopILi8EE...
    pat = r"\d+%sILi%sEE" % (t.group(1), t.group(2)) if t else r"\d+%s[EI]" % re.escape(name)
    name = t.group(1) if t else name
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and name in m.group(1):
            # exact kernel: "9k_merkle4E" must not match "14k_merkle4_pathE"
            if re.search(pat, m.group(1)):
                start = i
                break
    if start is None:
        raise SystemExit("kernel %s not found" % name)
    body = []
    for l in lines[start + 1:]:
        body.append(l)
        if l.strip().startswith("s_endpgm"):
            break
    return body


def parse(body):
    """-> list of (mnemonic, operands) and label -> index"""
    prog, labels = [], {}
    for l in body:
        l = l.split(";")[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r"^(\.?\w+):", l)
        if m:
            labels[m.group(1)] = len(prog)
            continue
        t = l.strip()
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        prog.append((parts[0], ops))
    return prog, labels


class Unknown(Exception):
    pass


def simulate(prog, labels, args, max_steps=50_000_000):
    s = {}          # sgpr index -> int (32-bit) when known
    scc = None
    vccz = None     # True: vcc == 0
    counts = {}
    lanes = {}      # (vgpr, lane) -> scalar value parked there by v_writelane_b32

    def sval(op):
        op = op.strip()
        if re.fullmatch(r"s\d+", op):
            return s.get(int(op[1:]))
        if op in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0", "scc"):
            return None
        try:
            return int(op, 0) & MASK32
        except ValueError:
            return None

    def s64(op):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", op.strip())
        if m:
            lo, hi = s.get(int(m.group(1))), s.get(int(m.group(1)) + 1)
            return None if lo is None or hi is None else (hi << 32) | lo
        if op.strip() == "exec":
            return "exec"
        try:
            return int(op, 0) & 0xFFFFFFFFFFFFFFFF
        except ValueError:
            return None

    def set64(op, v):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", op.strip())
        if not m:
            return
        a = int(m.group(1))
        if v is None or v == "exec":
            s[a], s[a + 1] = None, None
        else:
            s[a], s[a + 1] = v & MASK32, (v >> 32) & MASK32

    def setd(op, v):
        m = re.fullmatch(r"s(\d+)", op.strip())
        if m:
            s[int(m.group(1))] = None if v is None else v & MASK32
        else:
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", op.strip())
            if m:
                for k in range(int(m.group(1)), int(m.group(2)) + 1):
                    s[k] = None

    def signed(v):
        return v - (1 << 32) if v & 0x80000000 else v

    vconst = {}     # vgpr index -> wave-uniform constant it holds (v_mov_b32 from an immediate / a known SGPR)

    def vval(op):
        op = op.strip()
        m = re.fullmatch(r"v(\d+)", op)
        if m:
            return vconst.get(int(m.group(1)))
        return sval(op)

    def vkill(op):
        op = op.strip()
        m = re.fullmatch(r"v(\d+)", op)
        if m:
            vconst.pop(int(m.group(1)), None)
        else:
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
            if m:
                for k in range(int(m.group(1)), int(m.group(2)) + 1):
                    vconst.pop(k, None)

    pc, steps = 0, 0
    while True:
        steps += 1
        if steps > max_steps:
            raise SystemExit("runaway simulation")
        mn, ops = prog[pc]
        counts[mn] = counts.get(mn, 0) + 1
        nxt = pc + 1
        if mn == "s_endpgm":
            break
        if ops and (mn.startswith(("v_", "global_load", "buffer_load", "flat_load", "ds_read", "scratch_load"))):
            src = None
            if mn == "v_mov_b32_e32" and len(ops) == 2:
                src = vval(ops[1])
            elif mn in ("v_add_u32_e32", "v_add_u32") and len(ops) == 3:  # loop bound advanced on the vector unit
                a, b = vval(ops[1]), vval(ops[2])
                src = None if a is None or b is None else (a + b) & MASK32
            vkill(ops[0])
            if src is not None:
                vconst[int(ops[0].strip()[1:])] = src
        if mn in ("s_mov_b32", "s_movk_i32"):
            v = sval(ops[1])
            if mn == "s_movk_i32" and v is not None:
                v = (v - 0x10000 if v & 0x8000 else v) & MASK32
            setd(ops[0], v)
        elif mn == "s_mov_b64":
            set64(ops[0], s64(ops[1]))
        elif mn in ("s_add_i32", "s_add_u32", "s_sub_i32", "s_sub_u32", "s_mul_i32", "s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32", "s_ashr_i32"):
            a, b = sval(ops[1]), sval(ops[2])
            if a is None or b is None:
                setd(ops[0], None)
                scc = None
            else:
                if mn.startswith("s_add"):
                    r = a + b
                    scc = (r >> 32) & 1 if mn == "s_add_u32" else None
                elif mn.startswith("s_sub"):
                    r = a - b
                    scc = None
                elif mn == "s_mul_i32":
                    r = a * b
                elif mn == "s_lshl_b32":
                    r = a << (b & 31)
                    scc = int((r & MASK32) != 0)
                elif mn == "s_lshr_b32":
                    r = a >> (b & 31)
                    scc = int(r != 0)
                elif mn == "s_ashr_i32":
                    r = signed(a) >> (b & 31)
                    scc = int((r & MASK32) != 0)
                elif mn == "s_and_b32":
                    r = a & b
                    scc = int(r != 0)
                else:
                    r = a | b
                    scc = int(r != 0)
                setd(ops[0], r)
        elif mn == "s_addc_u32":
            setd(ops[0], None)
        elif mn.startswith("s_cmp_") or mn.startswith("s_cmpk_"):
            a, b = sval(ops[0]), sval(ops[1])
            if a is None or b is None:
                scc = None
            else:
                kind = mn.split("_")[2]
                sg = mn.endswith("i32")
                x, y = (signed(a), signed(b)) if sg else (a, b)
                if mn.startswith("s_cmpk_") and sg:
                    y = signed(b) if b & 0x80000000 else (b - 0x10000 if b & 0x8000 else b)
                scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[kind])
        elif mn == "s_cselect_b64":
            if scc is None:
                set64(ops[0], None)
            else:
                set64(ops[0], s64(ops[1]) if scc else s64(ops[2]))
        elif mn == "s_cselect_b32":
            setd(ops[0], None if scc is None else (sval(ops[1]) if scc else sval(ops[2])))
        elif mn in ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_xor_b64"):
            a, b = s64(ops[1]), s64(ops[2])
            r = None
            if mn == "s_and_b64":
                if a == "exec" and isinstance(b, int):
                    r = "exec" if b == 0xFFFFFFFFFFFFFFFF else (0 if b == 0 else None)
                elif b == "exec" and isinstance(a, int):
                    r = "exec" if a == 0xFFFFFFFFFFFFFFFF else (0 if a == 0 else None)
                elif isinstance(a, int) and isinstance(b, int):
                    r = a & b
            elif mn == "s_andn2_b64":
                if a == "exec" and isinstance(b, int):
                    r = "exec" if b == 0 else (0 if b == 0xFFFFFFFFFFFFFFFF else None)
                elif isinstance(a, int) and isinstance(b, int):
                    r = a & ~b & 0xFFFFFFFFFFFFFFFF
            elif mn == "s_or_b64" and isinstance(a, int) and isinstance(b, int):
                r = a | b
            elif mn == "s_xor_b64" and isinstance(a, int) and isinstance(b, int):
                r = a ^ b
            if ops[0].strip() == "vcc":
                vccz = None if r is None else (r == 0)  # ("exec" stands for a non-empty mask)
                scc = None if r is None else int(r != 0)
            elif ops[0].strip() == "exec":
                pass
            else:
                set64(ops[0], r if isinstance(r, int) else None)
                scc = None if not isinstance(r, int) else int(r != 0)
        elif mn == "s_and_saveexec_b64" or mn == "s_or_saveexec_b64":
            set64(ops[0], None)
            scc = 1  # some lane stays active
        elif mn == "s_branch":
            nxt = labels[ops[0]]
        elif mn in ("s_cbranch_scc0", "s_cbranch_scc1"):
            if scc is None:
                raise Unknown("branch on unknown scc at instruction %d (%s %s)" % (pc, mn, ops))
            if scc == (1 if mn.endswith("1") else 0):
                nxt = labels[ops[0]]
        elif mn in ("s_cbranch_vccz", "s_cbranch_vccnz"):
            if vccz is None:
                raise Unknown("branch on unknown vcc at instruction %d" % pc)
            if vccz == mn.endswith("vccz"):
                nxt = labels[ops[0]]
        elif mn == "s_cbranch_execz":
            pass  # lane guard: some lane is active
        elif mn == "s_cbranch_execnz":
            nxt = labels[ops[0]]
        elif mn.startswith("s_load_dword"):
            # kernel arguments the caller pinned (--arg OFFSET=VALUE, offsets into the kernarg segment via s[0:1] ... )
            base = ops[1].strip()
            off = int(ops[2], 0) if len(ops) > 2 and re.fullmatch(r"(0x)?[0-9a-fA-F]+", ops[2].strip()) else None
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", ops[0].strip())
            first, last = (int(m.group(1)), int(m.group(2))) if m else (int(ops[0].strip()[1:]), int(ops[0].strip()[1:]))
            for k in range(first, last + 1):
                s[k] = None
            if base in args.get("__kernarg_regs__", ()) and off is not None:
                for k in range(first, last + 1):
                    key = off + 4 * (k - first)
                    if key in args:
                        s[k] = args[key] & MASK32
        elif mn.startswith("s_") and ops and re.fullmatch(r"s\d+|s\[\d+:\d+\]", ops[0].strip()) and not mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep")):
            setd(ops[0], None)  # any other scalar op: result unknown
            if mn not in ("s_load_dword",):
                scc = None if mn.startswith(("s_lshl", "s_lshr", "s_bfe", "s_not", "s_xor", "s_andn2", "s_orn2", "s_min", "s_max", "s_abs")) else scc
        elif mn == "v_writelane_b32":  # SGPR spill slot: (vgpr, lane) <- scalar value
            lanes[(ops[0].strip(), ops[2].strip())] = sval(ops[1])
        elif mn == "v_readlane_b32":
            setd(ops[0], lanes.get((ops[1].strip(), ops[2].strip())))
        elif mn == "v_readfirstlane_b32":
            setd(ops[0], None)
        elif mn.startswith("v_cmp") or mn.startswith("v_cmpx"):
            # a loop bound the compiler parked in a VGPR (wave-uniform constant) compared on the vector unit
            vccz = None
            m = re.fullmatch(r"v_cmp_(eq|ne|lg|gt|ge|lt|le)_(u32|i32)_e32", mn)
            if m and ops[0].strip() == "vcc":
                a, b = vval(ops[1]), vval(ops[2])
                if a is not None and b is not None:
                    x, y = (signed(a), signed(b)) if m.group(2) == "i32" else (a, b)
                    res = {"eq": x == y, "ne": x != y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[m.group(1)]
                    vccz = not res
        pc = nxt
    return counts


def summarise(counts):
    valu = {k: v for k, v in counts.items() if k.startswith("v_")}
    four = sum(v for k, v in valu.items() if k.startswith(FOUR_CYCLE_PREFIXES))
    total = sum(valu.values())
    return {
        "valu_total": total,
        "v_mad_i64_i32": counts.get("v_mad_i64_i32", 0) + counts.get("v_mad_u64_u32", 0),
        "valu_4cycle_class": four,
        "valu_2cycle_class": total - four,
        "valu_issue_cycles": 4 * four + 2 * (total - four),
        "salu": sum(v for k, v in counts.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop", "s_endpgm"))),
        "smem": sum(v for k, v in counts.items() if k.startswith("s_load")),
        "vmem": sum(v for k, v in counts.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
        "lds": sum(v for k, v in counts.items() if k.startswith("ds_")),  # (ds_bpermute_b32: the cooperative kernel's cross-lane moves)
        "by_mnemonic": dict(sorted(valu.items(), key=lambda kv: -kv[1])),
    }


def count_kernel(asm_text, kernel, args=None):
    prog, labels = parse(kernel_body(asm_text, kernel))
    return summarise(simulate(prog, labels, args or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", help="assembly file (default: compile kernels.hip now)")
    ap.add_argument("--kernel", action="append", help="kernel name (default: k_merkle4, k_permute, k_merkle4_coop<8> and k_merkle4_coop<4>)")
    ap.add_argument("--out")
    a = ap.parse_args()
    path = a.asm or compile_asm()
    text = open(path).read()
    res = {}
    for k in a.kernel or ["k_merkle4", "k_permute", "k_merkle4_coop<8>", "k_merkle4_coop<4>"]:
        res[k] = count_kernel(text, k)
    res["_note"] = ("instructions issued by ONE wave for one pass of the kernel (per lane: one Merkle4 digest / one permutation), from the "
                    "ISA of kernels.hip at this commit: scalar control flow interpreted, vector instructions counted (tools/isa_count.py)")
    out = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(out + "\n")
    for k, v in res.items():
        if k.startswith("_"):
            continue
        print("%-12s VALU %6d  MAD %6d  4-cycle class %6d  2-cycle class %6d  issue cycles %7d  SALU %5d  SMEM %4d  VMEM %3d" % (
            k, v["valu_total"], v["v_mad_i64_i32"], v["valu_4cycle_class"], v["valu_2cycle_class"], v["valu_issue_cycles"], v["salu"], v["smem"], v["vmem"]))


if __name__ == "__main__":
    main()
