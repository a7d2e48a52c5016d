"""How the three conventions of compute_image_id that recollection could not settle were settled (round 3).

The reference ships one input/output pair for risc0_zkvm::compute_image_id (crates/povw/elfs/boundless-povw-log-updater.{bin,iid},
asserted equal by crates/povw/src/log_updater.rs:383-388).  The ID is SHA-256 over the Poseidon2 Merkle root, so a wrong guess
gives no partial signal; this script enumerates the combinations of the open conventions and reports the ones that reproduce
the 32 bytes.  Exactly one does:  user ELF wins overlapping addresses / (lo, hi) half-words interleaved / one permutation per
node / right child first / canonical digest words / pc = 0.   Run:  python tools/image_id_search.py     (CPU, ~1 min; uses the
C oracle's Poseidon2 as the permutation, which is what the match then pins.)

Fixed by other evidence and not searched: USER_START_ADDR = 0x0001_0000 (symbol table and first instructions of the kernel ELF
inside the .bin: `lui a0,0x10; lw a2,0(a0); addi a2,a2,-4; sw a2,MEPC`), SUSPEND_PC/MODE at 0xffff_0210/0214 (risc0 rv32im-v2
platform map; MEPC 0xffff_0200, GLOBAL_OUTPUT 0xffff_0240 and GLOBAL_INPUT 0xffff_0260 of the same map are in that symbol table),
the tagged-struct layout of SystemState (contracts/test/Blake3Groth16Verifier.t.sol:40-52 spells it out).
"""
import hashlib
import itertools
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib as ol  # noqa: E402

L = ol.lib()
REF = os.path.join(ROOT, "tests", "golden", "reference")
b = open(os.path.join(REF, "boundless-povw-log-updater.bin"), "rb").read()
target = open(os.path.join(REF, "boundless-povw-log-updater.iid"), "rb").read()
hl = struct.unpack("<I", b[8:12])[0]
ul = struct.unpack("<I", b[12 + hl : 16 + hl])[0]
user, kern = b[16 + hl : 16 + hl + ul], b[16 + hl + ul :]


def load_elf(e):
    entry, phoff = struct.unpack("<II", e[24:32])
    phentsize, phnum = struct.unpack("<HH", e[42:46])
    img = {}
    for i in range(phnum):
        o = phoff + i * phentsize
        p_type, p_off, p_vaddr, _, p_filesz, p_memsz, _, _ = struct.unpack("<8I", e[o : o + 32])
        if p_type != 1:
            continue
        for k in range(0, p_memsz, 4):
            img[p_vaddr + k] = 0 if k >= p_filesz else int.from_bytes(e[p_off + k : p_off + min(k + 4, p_filesz)], "little")
    return entry, img


ue, ui = load_elf(user)
ke, ki = load_elf(kern)


def sponge(cells_canon):
    m = ol.encode(np.asarray(cells_canon, dtype=np.uint64))
    d = np.zeros(8, np.uint32)
    L.bxo_hash_elem_slice(d, m, len(m), 1)
    return d


def page_cells(words, pv):
    lo, hi = words & 0xFFFF, words >> 16
    c = np.empty(512, np.uint64)
    if pv == "lo,hi interleaved":
        c[0::2], c[1::2] = lo, hi
    elif pv == "hi,lo interleaved":
        c[0::2], c[1::2] = hi, lo
    else:  # 8 low halves then 8 high halves per block
        c = c.reshape(32, 16)
        c[:, :8], c[:, 8:] = lo.reshape(32, 8), hi.reshape(32, 8)
        c = c.reshape(512)
    return c


def node(l, r, nv, order, form):
    a, c = (r, l) if order == "right first" else (l, r)
    if nv == "one permutation":
        d = np.zeros(8, np.uint32)
        L.bxo_hash_pair(d, np.ascontiguousarray(a), np.ascontiguousarray(c))
        return d
    w = np.concatenate([a, c]).astype(np.uint64)  # digests re-split into half-words: two blocks
    if form == "canonical":
        w = ol.decode(w).astype(np.uint64)
    cc = np.empty(32, np.uint64)
    cc[0::2], cc[1::2] = w & 0xFFFF, w >> 16
    return sponge(cc)


def sysstate(root_words, pc):
    tag = hashlib.sha256(b"risc0.SystemState").digest()
    return hashlib.sha256(tag + np.asarray(root_words, dtype="<u4").tobytes() + struct.pack("<I", pc) + struct.pack("<H", 1)).digest()


def run(prec, pv, nv, order, form):
    base = dict(ui) if prec == "kernel wins" else dict(ki)
    base.update(ki if prec == "kernel wins" else ui)
    base[0x10000], base[0xFFFF0210], base[0xFFFF0214] = ue, ke, 1
    pages = {}
    for a, w in base.items():
        pages.setdefault(a >> 10, np.zeros(256, np.uint64))[(a & 1023) >> 2] = w
    level = {pg: sponge(page_cells(ws, pv)) for pg, ws in pages.items()}
    z = sponge(page_cells(np.zeros(256, np.uint64), pv))
    for _ in range(22):
        nxt = {}
        for idx in {k >> 1 for k in level}:
            nxt[idx] = node(level.get(2 * idx, z), level.get(2 * idx + 1, z), nv, order, form)
        z = node(z, z, nv, order, form)
        level = nxt
    return level[0]


hits = []
space = itertools.product(("kernel wins", "user wins"), ("lo,hi interleaved", "8 lo then 8 hi", "hi,lo interleaved"),
                          ("one permutation", "two blocks of half-words"), ("right first", "left first"), ("montgomery", "canonical"))
for combo in space:
    r = run(*combo)
    v = r if combo[4] == "montgomery" else ol.decode(r)
    for pc in (0, ke, ue):
        ok = sysstate(v, pc) == target
        if ok:
            hits.append(combo + (f"pc={pc:#x}",))
    print(("MATCH " if any(h[:5] == combo for h in hits) else "      ") + " | ".join(combo))
    if r.astype("<u4").tobytes() == target or v.astype("<u4").tobytes() == target:
        hits.append(combo + ("raw root",))
print(f"{len(hits)} combination(s) reproduce the reference's .iid:")
for h in hits:
    print("  ", " | ".join(h))
