"""In-tree build of the HIP library (hipcc, gfx950 only).

`python apd-mvs_amd/build.py` or `build_library()` compiles csrc/*.hip into
apd-mvs_amd/_build/libapd_mi355x.so.  -ffp-contract=off and no fast-math are part of the
arithmetic contract: the kernels must round exactly like the CPU oracle.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB_PATH = os.path.join(OUT_DIR, "libapd_mi355x.so")
SOURCES = ["apd_kernels.hip", "apd_kernels_k67w.hip", "apd_kernels_k1415w.hip", "apd_kernels_weak.hip", "apd_fusion.hip", "apd_exchange.hip", "apd_capi.hip"]
HEADERS = ["apd_device.h", "apd_sweep.h", "apd_window.h", "apd_tuning.h", "apd_lab.h", "apd_fusion_math.h", os.path.join("..", "..", "include", "apd_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


# Per-file flags.  K6/K7: let LLVM keep the kernel's small per-lane arrays (arm positions, refinement hypotheses, view
# priors / probabilities -- indexed by wave-uniform loop counters) in registers instead of scratch memory; measured on
# configs[1]: 303.5 -> 311.4 Mpix*iter/s (profiles/r02/tune_scratch.txt).  The 9 x N cost table stays in scratch: 72
# registers more do not fit four waves per SIMD, and three waves per SIMD are 8 % slower (same file).
FILE_FLAGS = {"apd_kernels_k67w.hip": ["-mllvm", "-amdgpu-promote-alloca-to-vector-limit=2048"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=()):
    """sha256 over file names + contents + `extra` strings, 16 hex digits."""
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()[:16]


def _header_paths():
    return [os.path.join(CSRC, h) for h in HEADERS]


def expected_build_id(extra_flags=()):
    """What apd_build_id() of a library built from THIS tree returns: a digest of every HIP source, every header and the compiler
    flags.  A stale libapd_mi355x.so (sources edited after the build, or a push that kept old binaries) answers something else:
    __graft_entry__.build() / smoke() and apd_mvs_amd.lib() compare the two (the mtime test could not tell)."""
    extra_flags = list(extra_flags) + os.environ.get("APD_EXTRA_FLAGS", "").split()
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _header_paths(),
                   FLAGS + [k + "=" + " ".join(v) for k, v in sorted(FILE_FLAGS.items())] + extra_flags)


def _stale(target, key):
    """Content-keyed rebuild test: `target` is current iff target + '.key' holds `key`."""
    try:
        with open(target + ".key") as f:
            return not os.path.exists(target) or f.read().strip() != key
    except OSError:
        return True


def _stamp(target, key):
    with open(target + ".key", "w") as f:
        f.write(key + "\n")


def build_library(force=False, verbose=False, extra_flags=()):
    os.makedirs(OUT_DIR, exist_ok=True)
    extra_flags = list(extra_flags) + os.environ.get("APD_EXTRA_FLAGS", "").split()
    hdrs = _header_paths()
    build_id = expected_build_id(extra_flags)
    inc = os.path.join(OUT_DIR, "apd_build_id.inc")   # included by apd_capi.hip: the string apd_build_id() returns
    text = '"%s"\n' % build_id
    if not os.path.exists(inc) or open(inc).read() != text:
        with open(inc, "w") as f:
            f.write(text)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        flags = FLAGS + FILE_FLAGS.get(src, []) + list(extra_flags)
        key = _digest([s] + hdrs, flags + ([build_id] if src == "apd_capi.hip" else []))
        if force or _stale(o, key):
            jobs.append((o, key, [HIPCC] + flags + ["-I" + OUT_DIR, "-c", s, "-o", o]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    def compile_one(job):
        o, key, cmd = job
        if os.path.exists(o + ".key"):
            os.remove(o + ".key")
        out = run(cmd)
        _stamp(o, key)
        return out

    if jobs:
        with ThreadPoolExecutor(max_workers=6) as ex:
            for out in ex.map(compile_one, jobs):
                if verbose and out.strip():
                    print(out)
    if jobs or _stale(LIB_PATH, build_id):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"])
        _stamp(LIB_PATH, build_id)
    return LIB_PATH


HOST_DIR = os.path.join(HERE, "host")
HOST_BIN = os.path.join(OUT_DIR, "APD")
HOST_LIB = os.path.join(OUT_DIR, "libapd_host.so")
HOST_SOURCES = ["APD.cpp", "jpeg_gray.cpp", "fusion.cpp", "multi_device.cpp"]


def build_host(force=False, verbose=False):
    """The C++ drop-in: `_build/APD` (reference CLI) and `_build/libapd_host.so` (file-format helpers)."""
    build_library()
    cxx = os.environ.get("CXX", "g++")
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(CSRC, "apd_fusion_math.h"), os.path.join(HOST_DIR, "APD.h"), os.path.join(HOST_DIR, "schedule.h"), os.path.join(HOST_DIR, "main.cpp"), os.path.join(HOST_DIR, "host_capi.cpp"),
                   os.path.join(HERE, "..", "include", "apd_mi355x.h"), LIB_PATH]
    common = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result"]  # contract C9: no FMA contraction in the fusion arithmetic
    link = ["-L" + OUT_DIR, "-lapd_mi355x", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + OUT_DIR, "-pthread"]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build failed:\n" + " ".join(cmd) + "\n" + r.stdout)

    key = _digest([d for d in deps if d != LIB_PATH], common + [expected_build_id()])
    if force or _stale(HOST_BIN, key):
        run([cxx] + common + srcs + [os.path.join(HOST_DIR, "main.cpp"), "-o", HOST_BIN] + link)
        _stamp(HOST_BIN, key)
    if force or _stale(HOST_LIB, key):
        run([cxx] + common + ["-shared"] + srcs + [os.path.join(HOST_DIR, "host_capi.cpp"), "-o", HOST_LIB] + link)
        _stamp(HOST_LIB, key)
    return HOST_BIN, HOST_LIB


TOOLS_DIR = os.path.join(HERE, "..", "tools")


def build_tools(force=False):
    """tools/_build/: valu_rates (exhaustive ISA checks of the arithmetic contract), valu_issue (issue cost per VALU instruction
    class, the basis of bench.py's roofline peak), unaligned_gather (cost of dword gathers at 2-byte alignment), lds_addr_bits (ds_read does not ignore high address bits), tcp_patterns (L1 tag accesses per gather by lane address pattern), tcp_mix (cost of gathers whose lanes partly miss the L1), rccl_init_time (what RCCL's set-up consists of)."""
    out_dir = os.path.join(TOOLS_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    outs = []
    for name, flags in (("valu_rates", ["-ffp-contract=off", "-fno-slp-vectorize"]), ("valu_issue", []), ("unaligned_gather", []),
                        ("lds_addr_bits", ["-Wno-unused-value"]), ("tcp_patterns", []), ("tcp_mix", []),
                        ("rccl_init_time", ["-Wno-unused-result", "-Wno-unused-value", "-ldl", "-pthread"])):
        src = os.path.join(TOOLS_DIR, name + ".hip")
        out = os.path.join(out_dir, name)
        if force or _newer(out, [src]):
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17"] + flags + [src, "-o", out],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on tools/%s.hip:\n%s" % (name, r.stdout))
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    print(build_tools(force="--force" in sys.argv))
