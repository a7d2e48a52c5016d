"""Builds boundless_amd/lib/libbx_hip_hal.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m boundless_amd.build [--force]

One object per translation unit (rebuilt only when its sources change), linked into a single shared library.  Everything is
compiled with -fvisibility=hidden and the public headers wrap their declarations in `#pragma GCC visibility push(default)`, so the
dynamic symbol table holds exactly the `extern "C"` entry points of include/*.h and no C++ internals (tests/test_abi_cpu.py).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libbx_hip_hal.so")
CMD = os.path.join(HERE, "cmd")
AGENT_BIN = os.path.join(HERE, "bin", "bx-agent")
ARCH = "gfx950"
# hipcc derives a "compilation unit id" from the source file's PATH and puts it into a device symbol (__hip_cuid_<id>), so the same
# sources built in another directory gave another code object — and another device_code_hash() stamp.  _compile() therefore passes
# -cuid=<file name>: unique per translation unit, independent of where the tree lives (tests/test_profile_stamp_cpu.py).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
         "-ffp-contract=off", f"-I{INC}", f"-I{CSRC}"]


def _device_sources():
    """The *.hip translation units and every local header / table they include, transitively (paths, sorted by name)."""
    import re

    inc = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
    seen, todo = {}, [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")]
    while todo:
        path = os.path.normpath(todo.pop())
        if path in seen or not os.path.isfile(path):
            continue
        text = open(path, "r", errors="replace").read()
        seen[path] = True
        for name in inc.findall(text):
            for base in (os.path.dirname(path), INC, CSRC):
                cand = os.path.normpath(os.path.join(base, name))
                if os.path.isfile(cand):
                    todo.append(cand)
                    break
    return sorted(seen, key=lambda q: os.path.basename(q))


def csrc_hash(device_only=True):
    """SHA-256 (first 16 hex digits) over the CODE of the library's sources, in name order: comments are stripped and runs of white
    space collapsed first, so that editing the prose of a header does not make every committed profile look stale.
    device_only (the stamp of the profile summaries): the translation units that contain device code (*.hip) and the headers and
    tables they include, transitively — what a replayed PMC figure (bytes, VALU instructions per kernel) depends on; the host-only
    sources (agent, REST client, planner, verifier, control IDs, ELF loader and their headers) cannot change a kernel.
    device_only=False: every source of the library = the identity of the binary.  Profile summaries under profiles/ are stamped with
    the former by the tools that write them, and bench.py reports `profile_stale` when the library it runs was built from different
    device sources than the ones a replayed figure was collected on.  (A content hash rather than a git tree hash: the GPU box
    receives a snapshot without .git.)"""
    import hashlib
    import re

    strip = re.compile(rb"//[^\n]*|/\*.*?\*/", re.S)
    h = hashlib.sha256()
    if device_only:
        paths = _device_sources()
    else:
        paths = [os.path.join(d, f) for d in (CSRC, INC) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".cpp", ".hpp", ".h", ".inc"))]
    for path in paths:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(b" ".join(strip.sub(b" ", open(path, "rb").read()).split()))
    return h.hexdigest()[:16]


def elf_section(path, name):
    """Bytes of section `name` of an ELF64 little-endian file (None if absent); plain Python, no binutils needed on the GPU box."""
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError(f"{path}: not an ELF64 little-endian file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def hdr(i):
        nm, _ty, _fl, _ad, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return nm, off, size

    _, stroff, strsize = hdr(shstrndx)
    strtab = data[stroff:stroff + strsize]
    for i in range(shnum):
        nm, off, size = hdr(i)
        end = strtab.index(b"\0", nm)
        if strtab[nm:end].decode() == name:
            return data[off:off + size]
    return None


def device_code_hash(lib=None):
    """SHA-256 (first 16 hex digits) of the DEVICE CODE the library carries: its `.hip_fatbin` section, i.e. the gfx950 code objects
    hipcc embedded (one per translation unit, -fno-gpu-rdc).  This is the stamp of the PMC summaries under profiles/ and what
    bench.py's `profile_stale` compares: a host-only edit (a declaration in bx_prover.h, the agent, the verifier) recompiles
    prover.hip but leaves its code object — and this hash — unchanged; an edit of any kernel changes it
    (tests/test_profile_stamp_cpu.py).  csrc_hash() above hashed source TEXT of the include closure, so one host-only declaration in a
    header staled every profile."""
    import hashlib

    sec = elf_section(lib or LIB, ".hip_fatbin")
    if sec is None:
        raise RuntimeError(f"{lib or LIB}: no .hip_fatbin section")
    return hashlib.sha256(sec).hexdigest()[:16]


def cuid_flag(src):
    """-cuid for one translation unit: its file name (letters and digits), not its path"""
    return "-cuid=bx" + "".join(ch for ch in os.path.basename(src) if ch.isalnum())


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    m = 0.0
    for d in (CSRC, INC):
        for f in os.listdir(d):
            if f.endswith((".hpp", ".h", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


EXPORTS = os.path.join(CSRC, "exports.map")
FLAGS_STAMP = os.path.join(OBJ, "flags.stamp")


def _flags_fingerprint():
    """What every object depends on besides its sources: the compile flags, the per-TU -cuid scheme and the compiler.  Stored next to
    the objects; a tree whose objects were built with other flags (e.g. before -fvisibility=hidden or the path-independent -cuid) is
    rebuilt from scratch instead of keeping objects whose device_code_hash() differs from a clean build's."""
    import hashlib

    flags = [f for f in FLAGS if not f.startswith("-I")]  # include paths name where the tree lives (it moves: the GPU box gets a copy)
    return hashlib.sha256("\n".join(flags + [cuid_flag(s) for s in sources()] + [ARCH]).encode()).hexdigest()


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_mtime):
        return obj, False
    cmd = ["hipcc", "-x", "hip"] + FLAGS + [cuid_flag(src), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _deps_mtime()
    srcs = sources()
    fp = _flags_fingerprint()
    if not (os.path.exists(FLAGS_STAMP) and open(FLAGS_STAMP).read().strip() == fp):
        force = True  # objects of unknown or different flags
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    objs = [o for o, _ in results]
    with open(FLAGS_STAMP, "w") as f:
        f.write(fp + "\n")
    # the link also depends on the version script (which symbols are exported)
    if any(changed for _, changed in results) or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(EXPORTS):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", f"-Wl,--version-script={EXPORTS}",
               "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"{LIB} up to date")
    build_agent_binary(verbose)
    return LIB


def build_agent_binary(verbose=True):
    """boundless_amd/bin/bx-agent: the worker process (cmd/bx_agent_main.cpp) — plain C++ on the library's C ABI, no device code."""
    src = os.path.join(CMD, "bx_agent_main.cpp")
    os.makedirs(os.path.dirname(AGENT_BIN), exist_ok=True)
    newest = max(os.path.getmtime(src), os.path.getmtime(LIB), _deps_mtime())
    if os.path.exists(AGENT_BIN) and os.path.getmtime(AGENT_BIN) >= newest:
        return AGENT_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", f"-I{INC}", src, "-o", AGENT_BIN, f"-L{os.path.dirname(LIB)}", "-lbx_hip_hal",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building bx-agent failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {AGENT_BIN}")
    return AGENT_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
