"""Build the product's native artefacts in-tree (no JIT cache): libfighip.so (gfx950 HIP engine + C ABI),
figfill (C++ host drop-in for FillGaps.cpp) and libfighost.so (the host code as a library for the Python
binding).  Test infrastructure (oracle, oracle/_ref, the emulation build) is built by tools/build_test_infra.py."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
BINDIR = os.path.join(PKG, "bin")
LIB = os.path.join(LIBDIR, "libfighip.so")
FIGFILL = os.path.join(BINDIR, "figfill")
FIGTOOL = os.path.join(BINDIR, "figtool")
HOSTLIB = os.path.join(LIBDIR, "libfighost.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the reference multiplies and adds separately (no FMA on its x86-64 build);
# bit-identical pile-up weights need the same two roundings.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-value"]


def _csrc_files():
    """Every source the device library / emulation depends on (all of csrc/ + the ABI header)."""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".cpp"))]
    out.append(os.path.join(ROOT, "include", "figbird_hip.h"))
    return out


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]}")
    return r


def build_lib(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _csrc_files()
    if force or not _newer(LIB, srcs):
        _run([HIPCC] + HIP_FLAGS + ["-o", LIB, os.path.join(CSRC, "fig_abi.hip")])
    return LIB


def build_prof_lib() -> str:
    """Diagnostic build with the per-wave phase timers (-DFIG_PROF, fig_engine.h): libfighip_prof.so, never shipped or loaded by
    default (tools/gpu_probe.py takes it through FIG_LIB)."""
    out = os.path.join(LIBDIR, "libfighip_prof.so")
    if not _newer(out, _csrc_files()):
        _run([HIPCC] + HIP_FLAGS + ["-DFIG_PROF", "-o", out, os.path.join(CSRC, "fig_abi.hip")])
    return out


def _host_sources():
    host = os.path.join(CSRC, "host")
    hdrs = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h"))
    libs = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".cpp") and f.startswith("fig_"))
    return host, hdrs, libs


def build_figfill(force: bool = False) -> str:
    """figfill (the FillGaps.cpp drop-in) and libfighost.so: every host/fig_*.cpp + figfill_main.cpp."""
    os.makedirs(BINDIR, exist_ok=True)
    host, hdrs, libs = _host_sources()
    main_cpp = os.path.join(host, "figfill_main.cpp")
    srcs = hdrs + libs + [main_cpp, os.path.join(ROOT, "include", "figbird_hip.h")]      # (figtool_main.cpp is built below)
    common = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread"]
    if force or not _newer(FIGFILL, srcs + [LIB]):
        _run(common + ["-o", FIGFILL, main_cpp] + libs + ["-L" + LIBDIR, "-lfighip", "-Wl,-rpath,$ORIGIN/../lib"])
    if force or not _newer(HOSTLIB, srcs):
        _run(common + ["-fPIC", "-shared", "-o", HOSTLIB] + libs)
    tool_cpp = os.path.join(host, "figtool_main.cpp")
    if force or not _newer(FIGTOOL, hdrs + libs + [tool_cpp]):
        _run(common + ["-o", FIGTOOL, tool_cpp] + libs)
    return FIGFILL


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_figfill(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", LIB, FIGFILL, HOSTLIB)
