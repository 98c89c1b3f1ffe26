"""In-tree native builds (no setup.py, no JIT cache): every shared object lands
next to its sources under jsmpeg_amd/ or oracle/ so that it travels to the GPU
box with the gpurun snapshot.  `python -m jsmpeg_amd.build [target...]`."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
ORACLE = os.path.join(ROOT, "oracle")
REFERENCE = os.environ.get("JSMPEG_REFERENCE", "/root/reference")
NODE_INCLUDE = "/usr/include/node"

LIB_SYNTH = os.path.join(PKG, "libjsmpeg_synth.so")
LIB_HIP = os.environ.get("JSMPEG_HIP_LIB") or os.path.join(PKG, "libjsmpeg_hip.so")   # the override: tuning experiments only (tools/variants.sh)
ADDON_NODE = os.path.join(PKG, "js", "jsmpeg_hip.node")
LIB_ORACLE = os.path.join(ORACLE, "libmpeg1_oracle.so")
REF_DIR = os.path.join(ORACLE, "_ref")
LIB_REF = os.path.join(REF_DIR, "libjsmpeg_ref.so")
WASM_REF = os.path.join(REF_DIR, "jsmpeg_ref.wasm")
JS_REF = os.path.join(REF_DIR, "jsmpeg_ref.min.js")


_hip_runtime = None


def load_hip_library(path=None):
    """ctypes.CDLL of the decode library, bound to ONE HIP runtime per process.  The library needs libamdhip64.so.7; so
    does PyTorch, which ships its own copy under torch/lib.  Whichever is loaded first wins for the whole process (same
    SONAME), and torch does not find a GPU through the system's copy ("No HIP GPUs are available").  So: when PyTorch is
    installed and has not been imported yet, ITS runtime is loaded first (by path, without importing torch); a torch
    imported earlier has done that already.  Without PyTorch (the Node host, a C++ host) the system's runtime is used."""
    global _hip_runtime
    import ctypes
    import importlib.util
    if _hip_runtime is None:
        _hip_runtime = False
        if "torch" not in sys.modules:
            try:
                spec = importlib.util.find_spec("torch")
            except (ImportError, ValueError):
                spec = None
            if spec is not None and spec.origin:
                cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
                if os.path.exists(cand):
                    _hip_runtime = ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL(path or LIB_HIP)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _csrc_headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
           [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]


def build_synth(force=False):
    src = [os.path.join(CSRC, "synth_es.c"), os.path.join(CSRC, "synth_mp2.c")]
    if force or _newer(LIB_SYNTH, src + _csrc_headers()):
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-o", LIB_SYNTH] + src)
    return LIB_SYNTH


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_hip(force=False):
    """The product: HIP kernels + C-ABI runtime for gfx950 (cross-compiles
    without a GPU)."""
    src = hip_sources()
    defs = os.environ.get("JSMPEG_HIP_DEFS", "").split()   # tuning experiments only (-DJM_...=...)
    if force or defs or os.environ.get("JSMPEG_HIP_FORCE") or _newer(LIB_HIP, src + _csrc_headers()):
        _run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
              "-o", LIB_HIP] + defs + src)
        if not os.environ.get("JSMPEG_HIP_SKIP_ISA_CHECK"):      # (tuning experiments with other -D sets may skip it; the product build never does)
            try:
                check_parse_isa(defs)
            except Exception:
                os.unlink(LIB_HIP)                               # a library whose parse kernels may read registers in flight is not left lying around
                raise
    return LIB_HIP


def check_parse_isa(defs=()):
    """The slice parse issues loads in one asm statement and waits for them in another (slice_parse.h: the carried bit window,
    the refill in two halves); correct only while the compiler puts nothing that touches those registers in between.
    tools/check_parse_isa.py reads the gfx950 assembly of both parse kernels for exactly that -- part of every build of the
    library (round 5 advisor): another hipcc must fail HERE, not decode garbage on the GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_parse_isa.py")] + list(defs), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("tools/check_parse_isa.py: the parse kernels touch registers that are in flight:\n" + r.stdout[-2000:])
    return r.stdout


def check_kernel_resources():
    """The hot kernels must not touch scratch memory (private-segment spills
    measured wrong results AND cost bandwidth on this path): recompile
    kernels.hip device-only with resource remarks and fail on any scratch."""
    import re
    usage, name = {}, None
    out = ""
    for f in ("kernels.hip", "ts_kernels.hip", "mp2_stage.hip"):
        src = os.path.join(CSRC, f)
        out += subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "--cuda-device-only", "-c", src, "-o", os.devnull,
                               "-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True).stderr
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    bad = {k: v for k, v in usage.items() if v.get("ScratchSize", 0) != 0}
    if bad or not usage:
        raise RuntimeError("kernels using scratch memory (or no kernels found): %r" % (bad or out[-400:],))
    return usage


def build_addon(force=False):
    """N-API addon: thin glue from the Node host side to the C ABI."""
    src = [os.path.join(CSRC, "napi_addon.c")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("napi_") and f.endswith(".c") and f != "napi_addon.c")
    if not os.path.exists(src[0]) or not os.path.isdir(NODE_INCLUDE):
        return None
    build_hip(force)
    if force or _newer(ADDON_NODE, src + _csrc_headers() + [LIB_HIP]):
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-DNODE_GYP_MODULE_NAME=jsmpeg_hip", "-I", NODE_INCLUDE,
              "-I", os.path.join(ROOT, "include"), "-o", ADDON_NODE] + src +
             ["-L", PKG, "-ljsmpeg_hip", "-Wl,-rpath,$ORIGIN/.."])
    return ADDON_NODE


def build_oracle(force=False):
    """CPU restatement (test infrastructure only)."""
    src = [os.path.join(ORACLE, f) for f in ("mpeg1_oracle.c", "ycbcr_oracle.c", "ts_oracle.c", "mp2_oracle.c")]
    deps = [os.path.join(ORACLE, "mpeg1_oracle.h"), os.path.join(CSRC, "mp2_window.h")]
    if force or _newer(LIB_ORACLE, src + deps):
        # -ffp-contract=off: the MP2 restatement's float products must round as written (oracle/mp2_oracle.c header)
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-o", LIB_ORACLE] + src + ["-lm"])
    return LIB_ORACLE


def build_ref(force=False):
    """oracle/_ref: the reference's own C decoder compiled from where it lies
    (container only; the GPU box uses the prebuilt files).  Delegates to the
    committed recipe oracle/Makefile."""
    if not os.path.isdir(REFERENCE):
        return LIB_REF if os.path.exists(LIB_REF) else None
    _run(["make", "-s", "-C", ORACLE, "ref", "REFERENCE=" + REFERENCE] + (["-B"] if force else []))
    return LIB_REF


def build_all(force=False):
    out = {"synth": build_synth(force), "oracle": build_oracle(force), "ref": build_ref(force)}
    if shutil.which("hipcc"):
        out["hip"] = build_hip(force)
        out["addon"] = build_addon(force)
    return out


if __name__ == "__main__":
    targets = sys.argv[1:] or ["all"]
    for t in targets:
        print(t, "->", globals()["build_" + t]())
