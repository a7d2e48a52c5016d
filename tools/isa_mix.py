"""Static instruction mix of gfx950 kernels: how many VALU instructions of each issue class a wave executes.

    python tools/isa_mix.py [--src boundless_amd/csrc/ntt.hip] [--kernel SUBSTR ...] [--json]

Compiles the translation unit to gfx950 assembly with the library's flags (hipcc -S --cuda-device-only; works without a GPU) and
counts, per kernel symbol, the instructions of a straight-line pass over its text.  For kernels without data-dependent loops (the
compile-time-specialised NTT passes: every stage loop is unrolled) the static count IS the per-wave dynamic count, which
bench.py cross-checks against PMC SQ_INSTS_VALU (profiles/*_kernel_valu_counts.json).  Loops (the multi-column pass A runs its
body `cpw` times) are reported separately: `loop_body` = the instructions between the back-edge target and the back edge.

Issue classes (profiles/r01_microbench2_instr_cost.jsonl, measured ns per wave-instruction per SIMD on gfx950):
  cheap (~1.0 ns = 2 cycles): v_add_u32 v_sub_u32 v_subrev_u32 v_and_b32 v_or_b32 v_xor_b32 v_mov_b32 v_ashrrev_i32 and fp32
  mul   (~1.9 ns = 4 cycles): v_mul_lo/hi_u32 v_mad_u64_u32 v_mad_i64_i32 v_mul/mad_u32_u24 v_min/max_* v_lshl* v_lshr*
                               v_add3 v_lshl_add v_cmp_* carry-out adds v_add_co/v_addc v_bfe v_perm v_cndmask (VCC-dependent)
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHEAP = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_ashrrev_i32", "v_not_b32",
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
}
CYCLES = {"cheap": 2, "mul": 4}


def classify(op):
    op = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if not op.startswith("v_"):
        if op.startswith(("ds_",)):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        if op.startswith("s_waitcnt"):
            return "waitcnt"
        if op.startswith("s_barrier"):
            return "barrier"
        if op.startswith("s_nop"):
            return "nop"
        if op.startswith("s_"):
            return "salu"
        return "other"
    return "cheap" if op in CHEAP else "mul"


def compile_asm(src):
    from boundless_amd import build as b

    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    flags = [f for f in b.FLAGS if f != "-fPIC"]
    cmd = ["hipcc", "-x", "hip"] + flags + [b.cuid_flag(src), "-S", "--cuda-device-only", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
INSN = re.compile(r"^\s+([a-z][a-z0-9_]+)\b(.*)$")


def kernels(asm_path):
    """yield (symbol, [(label | None, opcode, operands)]) for every kernel body in the file"""
    cur, body = None, []
    for line in open(asm_path):
        m = LABEL.match(line)
        if m:
            name = m.group(1)
            if name.startswith("_Z") or (cur is None and not name.startswith(".")):
                if cur:
                    yield cur, body
                cur, body = name, []
                continue
            if cur and name.startswith(".LBB"):
                body.append((name, None, None))
            continue
        if cur is None:
            continue
        if line.lstrip().startswith(".") or line.lstrip().startswith(";"):
            if ".end_amdhsa_kernel" in line or line.strip().startswith(".section"):
                pass
            continue
        m = INSN.match(line)
        if m:
            op = m.group(1)
            body.append((None, op, m.group(2)))
            if op == "s_endpgm":
                yield cur, body
                cur, body = None, []
    if cur:
        yield cur, body


def mix(body):
    cls = collections.Counter()
    ops = collections.Counter()
    for _, op, _ in body:
        if op is None:
            continue
        c = classify(op)
        cls[c] += 1
        if c in ("mul", "cheap"):
            ops[op.replace("_e32", "").replace("_e64", "")] += 1
    return cls, ops


def loops(body):
    """innermost backward branches: (target label, index range)"""
    pos = {}
    out = []
    for i, (lab, op, arg) in enumerate(body):
        if lab:
            pos[lab] = i
        elif op and op.startswith(("s_cbranch", "s_branch")):
            tgt = arg.strip().split()[0] if arg.strip() else ""
            if tgt in pos:
                out.append((tgt, pos[tgt], i))
    return out


def summarise(body):
    cls, ops = mix(body)
    d = {
        "mul_class_insts": cls["mul"], "cheap_insts": cls["cheap"], "valu_insts": cls["mul"] + cls["cheap"],
        "weighted_issue_cycles": CYCLES["mul"] * cls["mul"] + CYCLES["cheap"] * cls["cheap"],
        "lds_insts": cls["lds"], "vmem_insts": cls["vmem"], "waitcnt": cls["waitcnt"], "barriers": cls["barrier"], "s_nop": cls["nop"],
        "salu_insts": cls["salu"],
        "top_ops": dict(ops.most_common(14)),
    }
    lp = []
    for tgt, a, b in loops(body):
        c2, _ = mix(body[a:b + 1])
        lp.append({"label": tgt, "mul_class_insts": c2["mul"], "cheap_insts": c2["cheap"], "lds_insts": c2["lds"], "vmem_insts": c2["vmem"]})
    if lp:
        d["loops"] = lp
    return d


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.strip().split("\n")))
    except OSError:
        return {n: n for n in names}


def analyse(src, filters=()):
    asm = compile_asm(src)
    ks = list(kernels(asm))
    dm = demangle([k for k, _ in ks])
    out = {}
    for sym, body in ks:
        name = dm.get(sym, sym)
        if filters and not any(f in name for f in filters):
            continue
        out[name] = summarise(body)
    os.unlink(asm)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=os.path.join(ROOT, "boundless_amd", "csrc", "ntt.hip"))
    ap.add_argument("--asm", default=None, help="an already compiled .s file instead of --src")
    ap.add_argument("--kernel", action="append", default=[])
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    if a.asm:
        ks = list(kernels(a.asm))
        dm = demangle([k for k, _ in ks])
        res = {dm[s]: summarise(b) for s, b in ks if not a.kernel or any(f in dm[s] for f in a.kernel)}
    else:
        res = analyse(a.src, a.kernel)
    if a.json:
        from boundless_amd.build import device_code_hash

        print(json.dumps({"device_code_sha": device_code_hash(), "note": "static per-wave instruction counts of a straight-line pass over each "
                          "kernel's gfx950 text (tools/isa_mix.py); `loops` = the body of each backward branch (the multi-column pass A runs "
                          "its column loop `cpw` times); classes as in profiles/r01_microbench2_instr_cost.jsonl", "kernels": res}, indent=1))
        return
    for name, d in res.items():
        print(f"{name}\n   mul {d['mul_class_insts']}  cheap {d['cheap_insts']}  weighted cycles {d['weighted_issue_cycles']}  "
              f"lds {d['lds_insts']} vmem {d['vmem_insts']} waitcnt {d['waitcnt']} barrier {d['barriers']} nop {d['s_nop']} salu {d['salu_insts']}")
        print("   ", d["top_ops"])
        for l in d.get("loops", []):
            print("    loop", l)


if __name__ == "__main__":
    main()
